set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_device_bvh.py tests/test_gpu_hw5.py -x -q -s > gpurun_out/r3_t11.log 2>&1; rc=$?
grep "coincident\|passed\|failed\|Error\|error" gpurun_out/r3_t11.log | tail -8
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r3_t11.log; exit $rc; fi
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" > gpurun_out/r3_p8j.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8j.log | sed 's/, queries.*//'
exit $rc
