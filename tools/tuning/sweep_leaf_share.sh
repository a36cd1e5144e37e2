# Share of a wave's active lanes that must wait at a leaf before the leaf phase starts (in 1/256), capped by the leaf batch.
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1:],d['value'],d['roofline']['kernel_avg_launch_ms'],flush=True)" "$@"; }
run RTAMD_WF_LEAF_SHARE_256=32
run RTAMD_WF_LEAF_SHARE_256=64
run RTAMD_WF_LEAF_SHARE_256=85
run RTAMD_WF_LEAF_SHARE_256=110
run RTAMD_WF_LEAF_SHARE_256=128
run RTAMD_WF_LEAF_SHARE_256=170
run RTAMD_WF_LEAF_SHARE_256=85 RTAMD_TRACE_LEAF_BATCH=28 RTAMD_LIGHT_LEAF_BATCH=28
run RTAMD_WF_LEAF_SHARE_256=128 RTAMD_TRACE_LEAF_BATCH=32 RTAMD_LIGHT_LEAF_BATCH=32
run RTAMD_WF_LEAF_SHARE_256=85 RTAMD_TRACE_REFILL=8 RTAMD_LIGHT_REFILL=8
