set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity_hw6.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_hw8.py -x -q > gpurun_out/r3_t7.log 2>&1; rc=$?
tail -5 gpurun_out/r3_t7.log
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_GROUP_SHIFT=6" "RTAMD_PT_GROUP_SHIFT=5" > gpurun_out/r3_p6b.log 2>&1; rc=$?
grep "exit times\|Msamples" gpurun_out/r3_p6b.log | sed 's/, queries.*//'
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_GROUP_SHIFT=4" > gpurun_out/r3_p8b.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8b.log | sed 's/, queries.*//'
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_GROUP_SHIFT=6" >> gpurun_out/r3_p8b.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8b.log | sed 's/, queries.*//' | tail -2
exit $rc
