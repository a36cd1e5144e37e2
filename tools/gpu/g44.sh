set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe22.log
: > $L
for rep in 1 2; do
echo "== four-wide grid nodes" >> $L
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
echo "== two-box grid nodes" >> $L
RTAMD_DEBUG_COUNTERS=1 RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd_bin.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
grep "==\|Msamples\|exit times\|nodes" $L | sed 's/, pipeline 2//; s/; exact closest.*//'
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py -x -q > gpurun_out/r3_t22.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t22.log
exit $rc
