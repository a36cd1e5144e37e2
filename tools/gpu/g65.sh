set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe42.log
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 900 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_SPEED_GAMMA=2.0" "RTAMD_PT_SPEED_GAMMA=2.5" "RTAMD_PT_SPEED_GAMMA=3.0" "RTAMD_PT_SPEED_GAMMA=4.0" "RTAMD_PT_SPEED_GAMMA=2.5 RTAMD_PT_SPEED_GAMMA_OWN=0.6" "RTAMD_PT_SPEED_GAMMA=2.5 RTAMD_PT_SPEED_GAMMA_OWN=0.2" > $L 2>&1 || exit $?
grep "Msamples\|exit times" $L | sed 's/, pipeline 2//; s/, queries.*//; s/.rtamd. persistent kernel .last launch.: 1280 workgroups, //; s/ after the first start.*//' | awk 'NR%3!=1'
timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_SPEED_GAMMA=2.5" "RTAMD_PT_SPEED_GAMMA=3.5" > gpurun_out/r3_p6c.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_p6c.log | sed 's/, pipeline 2//; s/, queries.*//'
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_SPEED_GAMMA=2.5" "RTAMD_PT_SPEED_GAMMA=3.5" > gpurun_out/r3_probe43.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_probe43.log | sed 's/, pipeline 2//; s/, queries.*//'
