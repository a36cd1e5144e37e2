set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe26.log
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 32 --reps 1 --counters "" > $L 2>&1 || exit $?
grep "rtamd\|Msamples" $L | grep -v "exit times"
