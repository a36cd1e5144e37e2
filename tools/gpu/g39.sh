set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_tcp
mkdir -p $O
cd $R
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_READ_sum TCP_TD_TCP_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TD_LOAD_WAVEFRONT_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 bench.py --steps 1 --warmup 0 --spp 16 --no-cpu-baseline > $O/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3_tcp'
tot=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(O+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pt_persistent_kernel' in r.get('Kernel_Name','') and '<false' in r.get('Kernel_Name',''):
            tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
with open(O+'/summary.txt','w') as out:
    for k in sorted(tot): out.write(f"{k} {tot[k]:.6g} over {n[k]} dispatch rows\n")
print(open(O+'/summary.txt').read())
PY
rm -rf $O/p[0-9]*/
