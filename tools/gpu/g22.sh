set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 32 --counters "" > gpurun_out/r3_p6j.log 2>&1; rc=$?
grep "by number of hits" gpurun_out/r3_p6j.log | tail -1
exit $rc
