run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1:],d['value'],d['roofline']['kernel_avg_launch_ms'],flush=True)" "$@"; }
run A=0
run RTAMD_WF_SPLIT=7:8
run RTAMD_WF_SPLIT=8:7
run RTAMD_WF_SPLIT=4:5
run RTAMD_WF_DYNAMIC_256=32
run RTAMD_WF_DYNAMIC_256=96
run RTAMD_TRACE_REFILL=24 RTAMD_LIGHT_REFILL=24
run RTAMD_TRACE_REFILL=12 RTAMD_LIGHT_REFILL=12
run RTAMD_TRACE_LEAF_BATCH=24 RTAMD_LIGHT_LEAF_BATCH=16
run RTAMD_TRACE_LEAF_BATCH=16 RTAMD_LIGHT_LEAF_BATCH=24
run A=1
