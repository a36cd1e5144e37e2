set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; rc=$?
tail -3 gpurun_out/r3_bench1.err; cat gpurun_out/r3_bench1.json | cut -c1-1500
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" "RTAMD_PT_PHASES=3" > gpurun_out/r3_probe3.log 2>&1; rc=$?
grep -v "in-flight\|finished by" gpurun_out/r3_probe3.log | tail -12
exit $rc
