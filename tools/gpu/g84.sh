set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_SPEED_GAMMA_OWN=0.2" "RTAMD_PT_SPEED_GAMMA_OWN=0.6" "RTAMD_PT_STOPS=2,8" "RTAMD_PT_STOPS=2,32" "RTAMD_PT_STOPS=4,16" "RTAMD_PT_SHADE_MIN=8" "RTAMD_PT_SHADE_MIN=24" "RTAMD_TRACE_LEAF_BATCH=20" "RTAMD_TRACE_REFILL=12" "RTAMD_LIGHT_REFILL=8" "" > gpurun_out/r3_p6f.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_p6f.log | sed 's/, pipeline 2//; s/, queries.*//'
