set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py -x -q -k "full_config" 2>&1 | tail -3
RTAMD_NO_TRIPWIRES=1 timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py -x -q -k "full_config" 2>&1 | grep -E "tripwires|passed|failed" | tail -3
