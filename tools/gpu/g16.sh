set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 RTAMD_DUMP_DEAL=gpurun_out/r3_deal6.txt RTAMD_DUMP_WG=gpurun_out/r3_wg6.txt timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 --counters "" > gpurun_out/r3_p6c.log 2>&1; rc=$?
grep "exit times\|Msamples\|wave time" gpurun_out/r3_p6c.log | sed 's/, queries.*//' | tail -4
exit $rc
