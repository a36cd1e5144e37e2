set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3_t21.log 2>&1; rc=$?
tail -15 gpurun_out/r3_t21.log
exit $rc
