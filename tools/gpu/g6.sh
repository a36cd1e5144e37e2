set -o pipefail
mkdir -p gpurun_out
RTAMD_DEBUG_COUNTERS=1 RTAMD_DUMP_DEAL=gpurun_out/r3_deal.txt RTAMD_DUMP_WG=gpurun_out/r3_wg.txt timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 1 "" > gpurun_out/r3_probe6.log 2>&1; rc=$?
grep -v "in-flight\|finished by" gpurun_out/r3_probe6.log | tail -4
python tools/tuning/wg_balance.py gpurun_out/r3_deal.txt gpurun_out/r3_wg.txt | head -12
if [ $rc -ne 0 ]; then exit $rc; fi
RTAMD_DEBUG_COUNTERS=1 timeout -k 10 300 python tools/tuning/pt_probe.py --spp 256 --reps 1 "RTAMD_PT_PHASES=3" "RTAMD_PT_NO_SPEEDS=1" "RTAMD_PT_PHASE0=8" "RTAMD_PT_PHASE0=32" > gpurun_out/r3_probe6b.log 2>&1; rc=$?
grep -v "in-flight\|finished by\|amdgpu.ids" gpurun_out/r3_probe6b.log | grep "exit times\|Msamples"
exit $rc
