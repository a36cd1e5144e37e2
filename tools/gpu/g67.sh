set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t35.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t35.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r3_t35.log; exit $rc; fi
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" > gpurun_out/r3_shard8b.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 2 "" >> gpurun_out/r3_shard8b.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 4 "" >> gpurun_out/r3_shard8b.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 3 --width 64 --height 64 "" >> gpurun_out/r3_shard8b.log 2>&1 || exit $?
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 32 --reps 1 --counters "" > gpurun_out/r3_counters_final.log 2>&1 || exit $?
grep Msamples gpurun_out/r3_shard8b.log | sed 's/, pipeline 2//; s/, queries.*//'
bash tools/profiling/profile_bench.sh r03f > gpurun_out/r03f_run.log 2>&1; rc=$?
tail -2 gpurun_out/r03f_run.log
exit $rc
