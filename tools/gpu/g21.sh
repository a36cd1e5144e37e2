set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity_hw6.py tests/test_gpu_edge_cases.py tests/test_gpu_scenes.py -x -q > gpurun_out/r3_t8.log 2>&1; rc=$?
tail -4 gpurun_out/r3_t8.log
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 --counters "" > gpurun_out/r3_p6h.log 2>&1; rc=$?
grep "exit times\|Msamples\|wave time\|slow role" gpurun_out/r3_p6h.log | sed 's/, queries.*//' | tail -5
timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" > gpurun_out/r3_p6i.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p6i.log | sed 's/, queries.*//'
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" > gpurun_out/r3_p8i.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p8i.log | sed 's/, queries.*//'
exit $rc
