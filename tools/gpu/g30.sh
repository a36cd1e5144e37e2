set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3_probe16.log
for e in "X=1" "RTAMD_C2X_SCALE=0.25" "RTAMD_C2X_SCALE=0.05" "RTAMD_C2X_SCALE=0" "RTAMD_C2X_SCALE=0 RTAMD_CULL_K=0.001953125"; do
  echo "== $e" >> gpurun_out/r3_probe16.log
  env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> gpurun_out/r3_probe16.log 2>&1 || exit $?
done
grep "==\|Msamples" gpurun_out/r3_probe16.log | sed 's/, pipeline 2//'
