set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_NO_FRONT_FIRST=1" > gpurun_out/r3_p6d.log 2>&1; rc=$?
grep "exit times\|Msamples" gpurun_out/r3_p6d.log | sed 's/, queries.*//'
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_NO_FRONT_FIRST=1" > gpurun_out/r3_p8d.log 2>&1; rc=$?
grep "exit times\|Msamples" gpurun_out/r3_p8d.log | sed 's/, queries.*//; s/; exact closest.*//'
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_NO_FRONT_FIRST=1" "RTAMD_PT_BLOCKS=1024" "RTAMD_PT_BLOCKS=768" "RTAMD_PT_BLOCKS=512" > gpurun_out/r3_p8e.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8e.log | sed 's/, queries.*//'
exit $rc
