"""bench.py prints ONE JSON line with the driver's contract fields plus `roofline` and `cpu_baseline` (small workload here; the
headline workload is what the driver runs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "synth_room_small_320x180x16", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "n/a" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "NOT the headline config" in d["metric"] and d["config"]["workload"].startswith("synth_room_small")
    assert d["value"] > 0 and abs(d["value"] - 320 * 180 * 16 / (d["ms_per_step"] * 1e3)) < 0.02 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_launch_ms"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["kernel"] == "pt_persistent_kernel" and r["traffic"] is None and r["traffic_measured_in_this_run"] is False  # PMC traffic is recorded for the headline workload only
    assert r["kernel_launches_per_step"] == 1 and "timed renders" in r["s_bar_p_bar_from"] and 1.0 < r["s_bar"] < 6.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= c["cores_physical"] >= 1 and c["cpu_model"]


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` from a plain shell on the one-GPU box: RTAMD_BENCH_REHEARSAL=1 puts both ranks on cuda:0 and gathers
    over gloo (RCCL refuses two ranks on one device); everything else is the real N>1 path: self-launch, strong scaling of the named
    frame, shard renders, preallocated gather, barrier / max-over-ranks timing."""
    env = dict(os.environ, RTAMD_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "synth_room_small_320x180x16", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["width"] == 320 and d["config"]["height"] == 180 and d["value"] > 0
