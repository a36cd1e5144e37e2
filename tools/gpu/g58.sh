set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" > gpurun_out/r3_shard8.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 2 "" >> gpurun_out/r3_shard8.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 4 "" >> gpurun_out/r3_shard8.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 3 --width 64 --height 64 "" >> gpurun_out/r3_shard8.log 2>&1 || exit $?
grep Msamples gpurun_out/r3_shard8.log | sed 's/, pipeline 2//; s/, queries.*//'
bash tools/profiling/profile_bench.sh r03e > gpurun_out/r03e_run.log 2>&1; rc=$?
tail -3 gpurun_out/r03e_run.log
exit $rc
