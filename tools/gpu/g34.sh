set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 8 "" "RTAMD_PT_SHADE_MIN=8" > gpurun_out/r3_p8k.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 4 "" >> gpurun_out/r3_p8k.log 2>&1 || exit $?
timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 2 --shards 2 "" >> gpurun_out/r3_p8k.log 2>&1 || exit $?
grep "Msamples" gpurun_out/r3_p8k.log | sed 's/, pipeline 2.*//'
