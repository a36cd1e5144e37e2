set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 RTAMD_DUMP_WG=$PWD/gpurun_out/r3_wg.txt timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" > gpurun_out/r3_probe41.log 2>&1 || exit $?
grep "Msamples\|exit times\|re-deal" gpurun_out/r3_probe41.log | sed 's/, pipeline 2//; s/, queries.*//'
timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "" >> gpurun_out/r3_probe41.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_probe41.log | sed 's/, pipeline 2//; s/, queries.*//' | tail -2
timeout -k 10 300 python -m pytest tests/test_gpu_device_math.py tests/test_gpu_scenes.py -x -q > gpurun_out/r3_t34.log 2>&1; rc=$?
tail -2 gpurun_out/r3_t34.log
exit $rc
