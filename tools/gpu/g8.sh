set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 RTAMD_DUMP_DEAL=gpurun_out/r3_deal.txt RTAMD_DUMP_WG=gpurun_out/r3_wg.txt timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" > gpurun_out/r3_probe8.log 2>&1; rc=$?
grep -v "in-flight\|finished by" gpurun_out/r3_probe8.log | tail -2
exit $rc
