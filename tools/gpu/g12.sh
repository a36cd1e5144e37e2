set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t4.log 2>&1; rc=$?
tail -6 gpurun_out/r3_t4.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
: > gpurun_out/r3_probe12.log
for e in "X=1" "RTAMD_CULL_K=0.000244" "RTAMD_NO_EXACT_BOXES=1"; do
  echo "== $e" >> gpurun_out/r3_probe12.log
  env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> gpurun_out/r3_probe12.log 2>&1 || exit $?
done
grep "==\|Msamples" gpurun_out/r3_probe12.log | sed 's/, pipeline 2//'
