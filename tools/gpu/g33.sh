set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3_probe18.log
echo "== five waves" >> gpurun_out/r3_probe18.log
timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> gpurun_out/r3_probe18.log 2>&1 || exit $?
echo "== six waves" >> gpurun_out/r3_probe18.log
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd_6w.so RTAMD_PT_BLOCKS=1536 RTAMD_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "RTAMD_PT_GROUP_SHIFT=4" >> gpurun_out/r3_probe18.log 2>&1 || exit $?
grep "==\|Msamples\|exit times" gpurun_out/r3_probe18.log | sed 's/, pipeline 2//; s/; exact closest.*//'
