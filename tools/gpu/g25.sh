set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3_p8lat.log
for wh in "64 64" "128 128" "256 256" "512 288" "960 540"; do
  set -- $wh
  timeout -k 10 200 python tools/tuning/pt_probe.py --spp 256 --reps 2 --width $1 --height $2 "" "RTAMD_PT_NO_REBALANCE=1" >> gpurun_out/r3_p8lat.log 2>&1 || exit $?
done
grep "Msamples" gpurun_out/r3_p8lat.log | sed 's/, pipeline.*//'
