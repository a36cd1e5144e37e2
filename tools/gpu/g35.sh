set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_throughput_mode.py tests/test_gpu_parity_hw6.py -x -q -s > gpurun_out/r3_t15.log 2>&1; rc=$?
grep "hw6 .*K=\|passed\|failed\|Error" gpurun_out/r3_t15.log | tail -8
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r3_t15.log; fi
exit $rc
