set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_parity_hw8.py tests/test_gpu_parity_hw7.py tests/test_gpu_edge_cases.py tests/test_gpu_throughput_mode.py -x -q > gpurun_out/r3_t3.log 2>&1; rc=$?
tail -4 gpurun_out/r3_t3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 RTAMD_DUMP_DEAL=gpurun_out/r3_deal.txt RTAMD_DUMP_WG=gpurun_out/r3_wg.txt timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" > gpurun_out/r3_probe5.log 2>&1; rc=$?
grep -v "in-flight\|finished by" gpurun_out/r3_probe5.log | tail -4
python tools/tuning/wg_balance.py gpurun_out/r3_deal.txt gpurun_out/r3_wg.txt
exit $rc
