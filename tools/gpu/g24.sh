set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t10.log 2>&1; rc=$?
tail -6 gpurun_out/r3_t10.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/profiling/profile_bench.sh r03b > gpurun_out/r03b_run.log 2>&1; rc=$?
tail -5 gpurun_out/r03b_run.log
exit $rc
