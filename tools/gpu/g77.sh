set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tests/diagnostics/find_bad_pixels.py 1600 960 2>&1 | tail -2
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" "" 2>&1 | grep Msamples | sed 's/, pipeline 2//'
RTAMD_NO_TRIPWIRES=1 timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" 2>&1 | grep Msamples | sed 's/, pipeline 2//'
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t38.log 2>&1; rc=$?
tail -3 gpurun_out/r3_t38.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 700 python tests/diagnostics/validate_headline.py --tiles 300 --out gpurun_out/r03_headline_parity_300.json > gpurun_out/r3_audit_hw8b.log 2>&1; rc=$?
tail -2 gpurun_out/r3_audit_hw8b.log
exit $rc
