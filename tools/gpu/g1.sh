set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_parity_hw8.py tests/test_gpu_parity_hw7.py tests/test_gpu_edge_cases.py -x -q > gpurun_out/r3_t1.log 2>&1; rc=$?
tail -15 gpurun_out/r3_t1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python tools/tuning/pt_probe.py --spp 64 "" "RTAMD_KERNEL=wavefront" "RTAMD_PT_SHADE_THR0=64 RTAMD_PT_SHADE_STEP=256" "RTAMD_PT_SHADE_THR0=64 RTAMD_PT_SHADE_STEP=128" "RTAMD_PT_SHADE_THR0=128 RTAMD_PT_SHADE_STEP=1024" "RTAMD_NO_EXACT_BOXES=1" > gpurun_out/r3_probe1.log 2>&1; rc=$?
cat gpurun_out/r3_probe1.log | tail -12
exit $rc
