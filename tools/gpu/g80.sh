set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in "" _prev; do
echo "== lib$v"
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd$v.so timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" 2>&1 | grep Msamples | sed 's/, pipeline 2//'
done
done
timeout -k 10 300 python tests/diagnostics/find_bad_pixels.py 1600 960 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_parity_hw8.py tests/test_gpu_parity_hw7.py -x -q 2>&1 | tail -2
