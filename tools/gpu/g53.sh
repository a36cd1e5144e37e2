set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe30.log
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 32 --reps 1 --counters "" > $L 2>&1 || exit $?
grep "rtamd" $L | grep -v "exit times\|in-flight\|finished by" 
timeout -k 10 600 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_LIGHT_REFILL=8" "RTAMD_LIGHT_REFILL=12" "RTAMD_LIGHT_REFILL=24" "RTAMD_LIGHT_REFILL=4" "" > gpurun_out/r3_probe31.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_probe31.log | sed 's/, pipeline 2//; s/, queries.*//'
