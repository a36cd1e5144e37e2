set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe19.log
: > $L
for rep in 1 2; do
echo "== packed" >> $L
timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
echo "== scalar" >> $L
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd_scalar.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
grep "==\|Msamples" $L | sed 's/, pipeline 2//; s/; exact closest.*//'
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py -x -q > gpurun_out/r3_t18.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t18.log
exit $rc
