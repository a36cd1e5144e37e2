set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity_hw6.py tests/test_gpu_device_bvh.py tests/test_gpu_cli.py -x -q > gpurun_out/r3_t6.log 2>&1; rc=$?
tail -5 gpurun_out/r3_t6.log
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" "RTAMD_PT_NO_SPEEDS=1" "RTAMD_PT_BLOCKS=1024" > gpurun_out/r3_p6a.log 2>&1; rc=$?
grep -v "in-flight\|amdgpu.ids" gpurun_out/r3_p6a.log | tail -12
exit $rc
