set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_edge_cases.py -x -q > gpurun_out/r3_t2.log 2>&1; rc=$?
tail -5 gpurun_out/r3_t2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" "RTAMD_PT_PHASE0=8" "RTAMD_PT_PHASE0=32" "RTAMD_PT_PHASES=3" > gpurun_out/r3_probe4.log 2>&1; rc=$?
grep -v "in-flight\|finished by" gpurun_out/r3_probe4.log | tail -14
exit $rc
