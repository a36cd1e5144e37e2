set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t12.log 2>&1; rc=$?
tail -4 gpurun_out/r3_t12.log
if [ $rc -ne 0 ]; then exit $rc; fi
: > gpurun_out/r3_probe13.log
for e in "X=1" "RTAMD_NO_EXACT_BOXES=1"; do
  echo "== $e" >> gpurun_out/r3_probe13.log
  env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> gpurun_out/r3_probe13.log 2>&1 || exit $?
done
grep "==\|Msamples" gpurun_out/r3_probe13.log | sed 's/, pipeline 2//'
timeout -k 10 300 python tools/tuning/p6_probe.py --spp 256 "" > gpurun_out/r3_p6m.log 2>&1; rc=$?
grep "Msamples" gpurun_out/r3_p6m.log | sed 's/, queries.*//'
exit $rc
