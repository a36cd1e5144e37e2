set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_SHADE_MIN=8" "RTAMD_TRACE_REFILL=8" "RTAMD_TRACE_REFILL=4" "RTAMD_TRACE_REFILL=8 RTAMD_PT_SHADE_MIN=8" "RTAMD_TRACE_LEAF_BATCH=12" "RTAMD_TRACE_LEAF_BATCH=8 RTAMD_TRACE_REFILL=8" "RTAMD_WF_LEAF_SHARE_256=64" "RTAMD_PT_SHADE_THR0=32 RTAMD_PT_SHADE_STEP=64" "RTAMD_PT_SHADE_THR0=64 RTAMD_PT_SHADE_STEP=64" "RTAMD_PT_STOPS=2,8" "RTAMD_PT_STOPS=1,8,32" > gpurun_out/r3_p8g.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8g.log | sed 's/, pipeline.*//'
timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_SHADE_MIN=8" "RTAMD_TRACE_LEAF_BATCH=12" "RTAMD_WF_LEAF_SHARE_256=64" "RTAMD_PT_SHADE_THR0=64 RTAMD_PT_SHADE_STEP=256" "RTAMD_PT_SHADE_THR0=256 RTAMD_PT_SHADE_STEP=512" "RTAMD_WF_SPLIT=2:1" "RTAMD_WF_SPLIT=1:2" > gpurun_out/r3_p6g.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p6g.log | sed 's/, pipeline.*//'
exit $rc
