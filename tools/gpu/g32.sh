set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_throughput_mode.py -x -q -s > gpurun_out/r3_t14.log 2>&1; rc=$?
grep "roulette\|rmse vs\|passed\|failed\|assert" gpurun_out/r3_t14.log | tail -8
exit $rc
