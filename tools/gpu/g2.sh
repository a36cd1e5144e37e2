set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/pt_probe.py --spp 64 --counters --reps 1 "" > gpurun_out/r3_probe2a.log 2>&1; rc=$?
tail -8 gpurun_out/r3_probe2a.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python tools/tuning/pt_probe.py --spp 64 "" "RTAMD_PT_BLOCKS=1024" "RTAMD_PT_BLOCKS=768" "RTAMD_TRACE_REFILL=8" "RTAMD_TRACE_REFILL=24" "RTAMD_TRACE_REFILL=32" "RTAMD_TRACE_LEAF_BATCH=12" "RTAMD_TRACE_LEAF_BATCH=28" "RTAMD_WF_LEAF_SHARE_256=80" "RTAMD_WF_LEAF_SHARE_256=144" "RTAMD_PT_NO_REBALANCE=1" "RTAMD_WF_SPLIT=1:1" "RTAMD_KERNEL=wavefront" > gpurun_out/r3_probe2.log 2>&1; rc=$?
tail -14 gpurun_out/r3_probe2.log
exit $rc
