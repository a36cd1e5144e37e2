set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t39.log 2>&1; rc=$?
grep "pixels differ between" gpurun_out/r3_t39.log | head -2
tail -3 gpurun_out/r3_t39.log
exit $rc
