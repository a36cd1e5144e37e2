set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe44.log
timeout -k 10 900 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_SPEED_GAMMA=1.75" "RTAMD_PT_SPEED_GAMMA=2.0" "RTAMD_PT_SPEED_GAMMA=2.25" "RTAMD_PT_SPEED_GAMMA=2.0 RTAMD_PT_SPEED_GAMMA_OWN=0.6" "RTAMD_PT_SPEED_GAMMA=2.0 RTAMD_PT_SPEED_GAMMA_OWN=0.2" "RTAMD_PT_SPEED_GAMMA=2.0" "" > $L 2>&1 || exit $?
grep "Msamples" $L | sed 's/, pipeline 2//; s/, queries.*//'
timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_SPEED_GAMMA=1.75" "RTAMD_PT_SPEED_GAMMA=2.0" "RTAMD_PT_SPEED_GAMMA=1.25" > gpurun_out/r3_p6d.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_p6d.log | sed 's/, pipeline 2//; s/, queries.*//'
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_SPEED_GAMMA=2.0" > gpurun_out/r3_probe45.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_probe45.log | sed 's/, pipeline 2//; s/, queries.*//'
