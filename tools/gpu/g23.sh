set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity_hw6.py -x -q > gpurun_out/r3_t9.log 2>&1; rc=$?
tail -4 gpurun_out/r3_t9.log
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 --counters "" > gpurun_out/r3_p6k.log 2>&1; rc=$?
grep "exit times\|Msamples\|wave time\|slow role" gpurun_out/r3_p6k.log | sed 's/, queries.*//' | tail -4
timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" > gpurun_out/r3_p6l.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p6l.log | sed 's/, queries.*//'
exit $rc
