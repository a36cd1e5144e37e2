set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_SPEED_GAMMA=1.7" "RTAMD_PT_SPEED_GAMMA=2.0" "RTAMD_PT_PHASE0=8" "RTAMD_PT_PHASE0=24" "RTAMD_PT_PHASE0=32" "RTAMD_TRACE_LEAF_BATCH=28" "RTAMD_TRACE_LEAF_BATCH=28 RTAMD_WF_LEAF_SHARE_256=144" "RTAMD_PT_SHADE_THR0=192 RTAMD_PT_SHADE_STEP=512" "RTAMD_PT_SHADE_THR0=256 RTAMD_PT_SHADE_STEP=256" "RTAMD_TRACE_REFILL=20" "RTAMD_WF_SPLIT=1:1" "RTAMD_WF_SPLIT=8:7" "" > gpurun_out/r3_probe10.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_probe10.log | sed 's/, pipeline.*//'
exit $rc
