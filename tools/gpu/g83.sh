set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t40.log 2>&1; rc=$?
tail -2 gpurun_out/r3_t40.log
exit $rc
