# Tail balancing of wf_traverse_kernel: size of the dynamically handed-out chunks and the share of each queue handed out dynamically.
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1:],d['value'],d['roofline']['kernel_avg_launch_ms'],flush=True)" "$@"; }
run RTAMD_WF_STEAL_CHUNK=16
run RTAMD_WF_STEAL_CHUNK=32
run RTAMD_WF_STEAL_CHUNK=128
run RTAMD_WF_DYNAMIC_256=128
run RTAMD_WF_DYNAMIC_256=192
run RTAMD_WF_DYNAMIC_256=255
run RTAMD_WF_DYNAMIC_256=128 RTAMD_WF_STEAL_CHUNK=32
run RTAMD_WF_DYNAMIC_256=255 RTAMD_WF_STEAL_CHUNK=32
