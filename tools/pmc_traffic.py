#!/usr/bin/env python3
"""Condense two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit in one pass on gfx950) of
`bench.py --steps 1 --spp 4 --no-cpu-baseline` into profiles/traffic_latest.json: memory-side bytes per launch of the
dominant kernel, for bench.py's roofline.traffic.

usage: pmc_traffic.py <dir with the counter_collection CSVs> <workload name> <out.json> [kernel name, default pt_persistent_kernel]
       [launches per frame at full spp / launches at the profiled spp: scales the per-launch bytes of a kernel whose one launch renders
        the whole frame, default 1]

FETCH_SIZE / WRITE_SIZE are reported in KB and count the L2's fabric-side requests, Infinity-Cache hits included
(MI355X_MICROARCH.md, HBM section).  The guide's gfx950 correction — FETCH_SIZE tallies 128-byte requests at 64 B, so
double it — is calibrated for wide coalesced reads; this kernel's 16-byte-per-lane loads are divergent (one line per
lane), which the guide calls uncalibrated, so both the raw and the doubled figure are kept and `bytes_per_launch` uses
the doubled one (the upper bound)."""
import csv
import glob
import json
import os
import sys


def main():
    root, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "pt_persistent_kernel"
    scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    spp_note = sys.argv[6] if len(sys.argv) > 6 else "4"
    tot = {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]}
    for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel + "<false" in r["Kernel_Name"] and r["Counter_Name"] in tot:
                tot[r["Counter_Name"]][0] += 1
                tot[r["Counter_Name"]][1] += float(r["Counter_Value"])
    if not tot["FETCH_SIZE"][0] or not tot["WRITE_SIZE"][0]:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for " + kernel)
    # the persistent pipeline renders a frame in two launches (phases): per FRAME = sum over the launches of one render
    per = 2 if kernel == "pt_persistent_kernel" and tot["FETCH_SIZE"][0] % 2 == 0 else 1
    fetch_kb = tot["FETCH_SIZE"][1] / (tot["FETCH_SIZE"][0] / per) * scale
    write_kb = tot["WRITE_SIZE"][1] / (tot["WRITE_SIZE"][0] / per) * scale
    j = {"workload": workload, "kernel": kernel, "launches_sampled": tot["FETCH_SIZE"][0],
         "fetch_size_kb_per_frame_raw": round(fetch_kb, 1), "write_size_kb_per_frame": round(write_kb, 1),
         "launches_per_frame": per,
         "bytes_per_frame": int((2 * fetch_kb + write_kb) * 1024),
         "bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024 / per),
         "bytes_per_launch_uncorrected": int((fetch_kb + write_kb) * 1024 / per),
         "frame_scale": scale,
         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --spp {spp_note} --no-cpu-baseline`"
                   + (f", scaled x{scale:g} to the frame's full sample count (the traffic of this kernel is proportional to the samples it renders)" if scale != 1 else "") + "; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); fabric-side requests, Infinity-Cache hits included. "
                   "The doubling is calibrated for wide coalesced reads; this kernel's reads are divergent 16-byte-per-lane loads of 64-B nodes and 48-B "
                   "triangle records (one 64-B half line per lane), for which a request is more likely a true 64-B request: bytes_per_launch_uncorrected "
                   "(raw) is the better estimate here and bytes_per_launch (doubled) an upper bound"}
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
