set -o pipefail
mkdir -p gpurun_out
for v in _n64; do
echo "== lib$v"
RTAMD_LIB=$PWD/raytracing-course-hw_amd/librtamd$v.so timeout -k 10 300 python tests/diagnostics/find_bad_pixels.py 1600 960 2>&1 | tail -2
done
echo "== host tree (RTAMD_HOST_BVH=1)"
RTAMD_HOST_BVH=1 timeout -k 10 300 python tests/diagnostics/find_bad_pixels.py 1600 960 2>&1 | tail -2
echo "== trace with counters build, default"
timeout -k 10 300 python tests/diagnostics/trace_pixel.py 1618 967 --spp 256 2>&1 | grep "query 189" -A3
