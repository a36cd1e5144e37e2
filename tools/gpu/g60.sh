set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe37.log
timeout -k 10 900 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_WALK_MIN=8" "RTAMD_PT_WALK_MIN=16" "RTAMD_PT_WALK_MIN=24" "RTAMD_PT_WALK_MIN=32" "RTAMD_PT_WALK_MIN=16 RTAMD_PT_WALK_MIN_SHADE=1" "RTAMD_PT_WALK_MIN=32 RTAMD_PT_WALK_MIN_SHADE=1" "RTAMD_PT_WALK_MIN=48" "" > $L 2>&1 || exit $?
grep "Msamples" $L | sed 's/, pipeline 2//; s/, queries.*//'
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_WALK_MIN=16" "RTAMD_PT_WALK_MIN=32" > gpurun_out/r3_probe38.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_probe38.log | sed 's/, pipeline 2//; s/, queries.*//'
