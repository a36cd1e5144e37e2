set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_SHADE_MIN=32" "RTAMD_PT_SHADE_MIN=16" "RTAMD_TRACE_REFILL=8" "RTAMD_TRACE_REFILL=8 RTAMD_PT_SHADE_MIN=32" "RTAMD_TRACE_REFILL=4 RTAMD_PT_SHADE_MIN=16" "RTAMD_PT_SHADE_THR0=64 RTAMD_PT_SHADE_STEP=128" "RTAMD_PT_PHASE0=8" "RTAMD_PT_PHASE0=32" "RTAMD_PT_PHASES=3" "RTAMD_PT_GROUP_SHIFT=5" > gpurun_out/r3_p6e.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p6e.log | sed 's/, pipeline.*//'
exit $rc
