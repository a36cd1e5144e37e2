set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python tests/diagnostics/validate_headline.py --tiles 300 --out gpurun_out/r03_headline_parity_300.json > gpurun_out/r3_audit_hw8.log 2>&1; rc=$?
tail -4 gpurun_out/r3_audit_hw8.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tests/diagnostics/walk_tree_diff.py --spp 32 --max-pixels 40 > gpurun_out/r3_audit_trees.log 2>&1; rc=$?
tail -4 gpurun_out/r3_audit_trees.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python tests/diagnostics/hw6_fullframe_check.py --spp 16 --crop 384 > gpurun_out/r3_audit_hw6.log 2>&1; rc=$?
tail -6 gpurun_out/r3_audit_hw6.log
exit $rc
