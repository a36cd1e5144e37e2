set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 400 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" "RTAMD_PT_SPEED_GAMMA_OWN=0.3" "RTAMD_PT_SPEED_GAMMA_OWN=0.7" "RTAMD_PT_SPEED_GAMMA=1.5" "RTAMD_PT_SPEED_GAMMA=1.5 RTAMD_PT_SPEED_GAMMA_OWN=0.3" "RTAMD_PT_PHASES=3" > gpurun_out/r3_probe9.log 2>&1; rc=$?
grep -v "in-flight\|finished by\|amdgpu.ids" gpurun_out/r3_probe9.log | grep "exit times\|Msamples" | sed 's/; exact closest.*//'
exit $rc
