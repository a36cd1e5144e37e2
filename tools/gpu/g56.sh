set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe34.log
: > $L
for rep in 1 2; do
for v in "" _prev; do
echo "== lib$v" >> $L
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd$v.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
done
echo "== counters" >> $L
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 32 --reps 1 --counters "" >> $L 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "RTAMD_TRACE_LEAF_BATCH=28" "RTAMD_TRACE_LEAF_BATCH=28 RTAMD_WF_LEAF_SHARE_256=144" "RTAMD_TRACE_LEAF_BATCH=14 RTAMD_WF_LEAF_SHARE_256=80" >> $L 2>&1 || exit $?
grep "==\|Msamples\|walker's\|role" $L | sed 's/, pipeline 2//; s/; exact closest.*//; s/; walker lane.*//'
timeout -k 10 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_device_bvh.py -x -q > gpurun_out/r3_t28.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t28.log
exit $rc
