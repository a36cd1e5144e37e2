set -o pipefail
mkdir -p gpurun_out
bash tools/profiling/profile_bench.sh r03g > gpurun_out/r03g_run.log 2>&1; rc=$?
tail -2 gpurun_out/r03g_run.log
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" 2>&1 | grep Msamples | sed 's/, pipeline 2//; s/, queries.*//'
exit $rc
