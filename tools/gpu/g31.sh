set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3_probe17.log
for e in "X=1" "RTAMD_DIAG_LOOKBEHIND_ONLY=1" "RTAMD_DIAG_LOOKBEHIND_ONLY=1 RTAMD_CULL_K=0.00001" "RTAMD_DIAG_LOOKBEHIND_ONLY=1 RTAMD_CULL_K=0.00001 RTAMD_C2X_SCALE=0" "RTAMD_NO_EXACT_BOXES=1"; do
  echo "== $e" >> gpurun_out/r3_probe17.log
  env $e timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 "" >> gpurun_out/r3_probe17.log 2>&1 || exit $?
done
grep "==\|Msamples" gpurun_out/r3_probe17.log | sed 's/, pipeline 2//'
