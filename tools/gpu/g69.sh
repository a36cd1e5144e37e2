set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe47.log
: > $L
for rep in 1 2; do
for v in "" _prev; do
echo "== lib$v" >> $L
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd$v.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
done
grep "==\|Msamples" $L | sed 's/, pipeline 2//; s/, queries.*//'
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t36.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t36.log
exit $rc
