set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tests/diagnostics/walk_tree_diff.py --spp 256 --max-pixels 24 "grid:" "rounds:RTAMD_KERNEL=wavefront RTAMD_ROUNDS_EXACT=1" > gpurun_out/r3_diff1.log 2>&1; rc=$?
tail -32 gpurun_out/r3_diff1.log
exit $rc
