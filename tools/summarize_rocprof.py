#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary for profiles/.
usage: summarize_rocprof.py <dir with stats/ pmc_*/ subdirs> <out.txt> [note]"""
import csv
import glob
import os
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    lines = [f"# rocprofv3 summary of {root}", note, ""]
    for f in sorted(glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True)):
        lines.append(f"## kernel stats ({os.path.relpath(f, root)})")
        lines.append("calls, total_ms, avg_ms, pct, name")
        for r in csv.DictReader(open(f)):
            lines.append(f"{r['Calls']}, {float(r['TotalDurationNs']) / 1e6:.3f}, {float(r['AverageNs']) / 1e6:.4f}, {float(r['Percentage']):.2f}, {r['Name'][:110]}")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        agg = {}
        meta = {}
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:90], r["Counter_Name"])
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            meta[r["Kernel_Name"][:90]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size"])
        lines.append(f"## PMC ({os.path.relpath(f, root)})  [FETCH_SIZE/WRITE_SIZE are in KB; see MI355X_MICROARCH.md HBM section for the gfx950 x2 read correction]")
        lines.append("dispatches, sum, per_dispatch, counter, kernel")
        for (k, c), (n, s) in sorted(agg.items()):
            lines.append(f"{n}, {s:.6g}, {s / n:.6g}, {c}, {k}")
        lines.append("kernel resources (VGPR, AGPR, SGPR, LDS, scratch, workgroup):")
        for k, m in meta.items():
            lines.append(f"  {m}  {k}")
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
