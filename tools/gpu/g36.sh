set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_parity_hw8.py tests/test_gpu_parity_hw7.py tests/test_gpu_edge_cases.py tests/test_gpu_throughput_mode.py tests/test_gpu_device_bvh.py -x -q > gpurun_out/r3_t16.log 2>&1; rc=$?
tail -4 gpurun_out/r3_t16.log
if [ $rc -ne 0 ]; then exit $rc; fi
: > gpurun_out/r3_probe19.log
for e in "X=1" "RTAMD_HOST_LIGHT_BVH=1"; do
  echo "== $e" >> gpurun_out/r3_probe19.log
  env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> gpurun_out/r3_probe19.log 2>&1 || exit $?
  env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 16 --reps 1 --counters "" >> gpurun_out/r3_probe19.log 2>&1 || exit $?
done
grep "==\|Msamples" gpurun_out/r3_probe19.log | sed 's/, pipeline 2//'
