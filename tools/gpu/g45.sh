set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe23.log
: > $L
for v in "" _c22 _c23 _c33 _c34 ""; do
echo "== cost weights$v" >> $L
RTAMD_DEBUG_COUNTERS=1 RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd$v.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
grep "==\|Msamples\|exit times" $L | sed 's/, pipeline 2//; s/; exact closest.*//; s/.rtamd. persistent kernel .last launch.: 1280 workgroups, //'
