# PMC passes (rocprofv3, counters only with --kernel-trace) of the persistent pipeline and, for comparison under the same counters,
# the round pipeline, on the headline scene at reduced spp (full-size queues).  Usage on the GPU box: bash tools/profiling/pmc_persistent.sh <tag> [spp]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
SPP=${2:-16}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() { # name, env assignment (or x=1), counters...
  local name=$1; shift; local envs=$1; shift
  env $envs true
  ( export $envs; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- python3 tools/tuning/pt_probe.py --spp $SPP --reps 1 "" > $O/pmc_$name.log 2>&1 ) || echo "pass $name failed" >> $O/errors.log
}
run sq1_persistent RTAMD_X=1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2_persistent RTAMD_X=1 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run sq3_persistent RTAMD_X=1 SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH
run tcc_persistent RTAMD_X=1 TCC_HIT_sum TCC_MISS_sum
run tcp_persistent RTAMD_X=1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run fetch_persistent RTAMD_X=1 FETCH_SIZE
run write_persistent RTAMD_X=1 WRITE_SIZE
run sq1_rounds RTAMD_KERNEL=wavefront SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2_rounds RTAMD_KERNEL=wavefront SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run tcp_rounds RTAMD_KERNEL=wavefront TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
mkdir -p $O/pmc && mv $O/pmc_* $O/pmc/ 2>/dev/null || true
python3 tools/summarize_rocprof.py $O/pmc $O/${TAG}_pmc.txt "pt_probe.py --spp $SPP (headline scene, 1920x1080, full-size path population): persistent pipeline vs round pipeline under the same counters" > /dev/null
grep -h "Msamples" $O/pmc/*.log > $O/${TAG}_pmc_rates.txt || true
rm -rf $O/pmc
ls -la $O
