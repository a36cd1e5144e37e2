set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 64 --counters "" > gpurun_out/r3_p6e.log 2>&1 || exit $?
grep "rtamd\|Msamples" gpurun_out/r3_p6e.log | grep -v "in-flight\|finished by" | cut -c1-500
