set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe48.log
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 900 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_PHASES=3" "RTAMD_PT_STOPS=16,128" "RTAMD_PT_STOPS=16,192" "RTAMD_PT_STOPS=8,64" "RTAMD_PT_STOPS=32" "RTAMD_PT_STOPS=8" "RTAMD_PT_STOPS=16,64,160" "" > $L 2>&1 || exit $?
grep "Msamples\|exit times" $L | sed 's/, pipeline 2//; s/, queries.*//; s/.rtamd. persistent kernel .last launch.: 1280 workgroups, //; s/ after the first start.*//' | awk 'NR%3!=1'
