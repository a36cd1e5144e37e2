set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_probe20.log
: > $L
for rep in 1 2; do
echo "== default" >> $L
timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
echo "== extra load per trace node visit" >> $L
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd_xload.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> $L 2>&1 || exit $?
done
grep "==\|Msamples" $L | sed 's/, pipeline 2//; s/; exact closest.*//'
