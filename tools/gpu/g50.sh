set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe27.log
: > $L
for e in "X=1" "RTAMD_LIGHT_SAH_LEAF=1" "RTAMD_LIGHT_SAH_LEAF=2" "RTAMD_LIGHT_SAH_LEAF=4" "RTAMD_WALK_SAH_LEAF=4" "RTAMD_WALK_SAH_LEAF=2" "RTAMD_WALK_SAH_LEAF=1" "RTAMD_LIGHT_SAH_LEAF=1 RTAMD_WALK_SAH_LEAF=2"; do
echo "== $e" >> $L
env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
echo "== counters RTAMD_LIGHT_SAH_LEAF=1" >> $L
RTAMD_LIGHT_SAH_LEAF=1 RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 32 --reps 1 --counters "" >> $L 2>&1 || exit $?
grep "==\|Msamples\|walker's\|light tests" $L | sed 's/, pipeline 2//; s/, queries.*//'
