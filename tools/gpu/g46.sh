set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t23.log 2>&1; rc=$?
tail -4 gpurun_out/r3_t23.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r3_t23.log; exit $rc; fi
bash tools/profiling/profile_bench.sh r03d > gpurun_out/r03d_run.log 2>&1; rc=$?
tail -3 gpurun_out/r03d_run.log
exit $rc
