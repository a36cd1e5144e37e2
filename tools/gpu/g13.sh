set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/r3_t5.log 2>&1; rc=$?
tail -6 gpurun_out/r3_t5.log
grep -c "bit_exact True" gpurun_out/r3_t5.log; grep "bit_exact False" gpurun_out/r3_t5.log | head -40
exit $rc
