set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_device_math.py -x -q -s > gpurun_out/r3_t19.log 2>&1; rc=$?
grep "grid over\|passed\|failed\|Error" gpurun_out/r3_t19.log | tail -8
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r3_t19.log; exit $rc; fi
L=gpurun_out/r3_probe21.log
: > $L
for rep in 1 2; do
echo "== 32-byte grid nodes" >> $L
timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
echo "== 64-byte float nodes" >> $L
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd_n64.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
grep "==\|Msamples" $L | sed 's/, pipeline 2//; s/; exact closest.*//'
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py -x -q > gpurun_out/r3_t20.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t20.log
exit $rc
