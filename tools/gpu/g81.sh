set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python tests/diagnostics/validate_headline.py --tiles 300 --seed 7 --out gpurun_out/r03_headline_parity_300_seed7.json > gpurun_out/r3_audit_hw8c.log 2>&1; rc=$?
tail -2 gpurun_out/r3_audit_hw8c.log
exit $rc
