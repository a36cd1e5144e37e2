set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe46.log
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 900 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_PRIO=4" "RTAMD_PT_PRIO=5" "RTAMD_PT_PRIO=6" "RTAMD_PT_PRIO=7" "RTAMD_PT_PRIO=8" "RTAMD_PT_PRIO=6 RTAMD_PT_SPEED_GAMMA=1.0" "RTAMD_PT_PRIO=6 RTAMD_PT_SPEED_GAMMA=1.5" "" > $L 2>&1 || exit $?
grep "Msamples\|exit times" $L | sed 's/, pipeline 2//; s/, queries.*//; s/.rtamd. persistent kernel .last launch.: 1280 workgroups, //; s/ after the first start.*//' | awk 'NR%3!=1'
for e in "X=1" "RTAMD_NO_EXACT_BOXES=1" "X=1" "RTAMD_NO_EXACT_BOXES=1"; do
env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" > gpurun_out/r3_gate.log 2>&1 || exit $?
echo "$e: $(grep Msamples gpurun_out/r3_gate.log | sed 's/, pipeline 2//; s/, queries.*//')"
done
