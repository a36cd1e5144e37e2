set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/tuning/p6_probe.py --spp 256 "" "" > gpurun_out/r3_p6b.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_p6b.log | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t32.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t32.log
exit $rc
