set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/tuning/p6_probe.py --spp 256 "RTAMD_PT_SHADE_MIN=16" "RTAMD_PT_SHADE_MIN=16 RTAMD_PT_STOPS=1,16" "RTAMD_PT_SHADE_MIN=16 RTAMD_PT_STOPS=2,16" "RTAMD_PT_SHADE_MIN=16 RTAMD_PT_STOPS=2,24" "RTAMD_PT_SHADE_MIN=16 RTAMD_PT_STOPS=4,32" "RTAMD_PT_SHADE_MIN=16 RTAMD_PT_STOPS=8" "RTAMD_PT_SHADE_MIN=8" "RTAMD_PT_SHADE_MIN=16 RTAMD_PT_STOPS=2,16,96" > gpurun_out/r3_p6f.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p6f.log | sed 's/, pipeline.*//'
timeout -k 10 500 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_SHADE_MIN=32" "RTAMD_PT_SHADE_MIN=16" "RTAMD_PT_STOPS=2,16" "RTAMD_PT_STOPS=8" > gpurun_out/r3_p8f.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8f.log | sed 's/, pipeline.*//'
timeout -k 10 500 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_SHADE_MIN=32" "RTAMD_PT_SHADE_MIN=16" "RTAMD_PT_STOPS=2,16" "RTAMD_PT_STOPS=2,16 RTAMD_PT_SHADE_MIN=16" >> gpurun_out/r3_p8f.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8f.log | sed 's/, pipeline.*//' | tail -5
exit $rc
