set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe49.log
timeout -k 10 900 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_TRACE_REFILL=8" "RTAMD_TRACE_REFILL=12" "RTAMD_TRACE_REFILL=20" "RTAMD_LIGHT_REFILL=6" "RTAMD_LIGHT_REFILL=10" "RTAMD_LIGHT_REFILL=24" "RTAMD_TRACE_REFILL=12 RTAMD_LIGHT_REFILL=10" "RTAMD_TRACE_LEAF_BATCH=32 RTAMD_WF_LEAF_SHARE_256=128" "RTAMD_PT_SHADE_MIN=48" "RTAMD_PT_SHADE_MIN=24" "" > $L 2>&1 || exit $?
grep "Msamples" $L | sed 's/, pipeline 2//; s/, queries.*//'
