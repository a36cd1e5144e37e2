set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tests/diagnostics/find_bad_pixels.py 1600 960 > gpurun_out/r3_bad.log 2>&1; rc=$?
cat gpurun_out/r3_bad.log | tail -5
if [ $rc -ne 0 ]; then exit $rc; fi
px=$(grep "  pixel" gpurun_out/r3_bad.log | head -1 | sed 's/.*pixel (\([0-9]*\),\([0-9]*\)).*/\1 \2/')
if [ -n "$px" ]; then
  timeout -k 10 400 python tests/diagnostics/trace_pixel.py $px --spp 256 > gpurun_out/r3_trace.log 2>&1; rc=$?
  tail -30 gpurun_out/r3_trace.log
fi
exit $rc
