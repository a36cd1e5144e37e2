"""`python bench.py --gpus N` from a plain shell: the un-launched parent starts the ranks (torch.distributed.run as a CHILD
process), relays rank 0's single JSON line and returns the child's exit code.  Here on CPU the ranks run with
RTAMD_BENCH_FAKE_RENDER=1 (gloo, a position pattern instead of a render), which drives the real launcher, the preallocated
FrameGatherer, the barrier / max-over-ranks timing and the JSON contract; the render itself is covered by the -m gpu tests."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, n):
    env = dict(os.environ, RTAMD_BENCH_FAKE_RENDER="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout  # stdout is exactly ONE JSON line
    return json.loads(lines[0])


def test_plain_python_invocation_launches_two_ranks_strong_scaling_by_default():
    d = _run(["--workload", "synth_room_small_320x180x16"], 2)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["width"] == 320 and d["config"]["height"] == 180      # the NAMED frame split two ways, not a grown one
    assert d["config"]["gathered_frame_ok"] is True                          # tiles of both ranks landed where they belong
    assert "NOT the headline config" in d["metric"] and "FAKE RENDER" in d["metric"]
    assert abs(d["value"] - 320 * 180 * 16 / (d["ms_per_step"] * 1e3)) < 0.02 * d["value"]


def test_weak_scaling_is_labelled_with_the_grown_frame():
    d = _run(["--workload", "synth_room_small_320x180x16", "--scaling", "weak"], 2)
    w, h = d["config"]["width"], d["config"]["height"]
    assert d["scaling"] == "weak" and (w, h) != (320, 180) and abs(w * h - 2 * 320 * 180) < 0.05 * 2 * 320 * 180
    assert f"{w}x{h}" in d["metric"] and "weak scaling" in d["metric"] and d["config"]["gathered_frame_ok"] is True


def test_headline_workload_strong_keeps_the_1080p_frame():
    d = _run([], 2)  # default workload = the metric's own 1920x1080x256 frame
    assert d["config"]["width"] == 1920 and d["config"]["height"] == 1080 and d["scaling"] == "strong" and d["config"]["gathered_frame_ok"] is True


def test_eight_ranks_gather_the_3840x2160_frame_of_config_4():
    """BASELINE.json configs[4]'s layout — 3840x2160 cut into 32x32 tiles dealt round-robin to EIGHT ranks (1,013 or 1,012 tiles each, the
    last tile row cut by the frame edge), one gather to rank 0 — with fake renders over gloo: every tile of every rank must land where
    it belongs.  (The real render of one such shard is a -m gpu test; eight physical GPUs have never been available to this repo.)"""
    d = _run(["--workload", "synth_room_v1_3840x2160x1024"], 8)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["width"] == 3840 and d["config"]["height"] == 2160
    assert d["config"]["gathered_frame_ok"] is True and "FAKE RENDER" in d["metric"]


def test_failure_of_the_ranks_is_the_parents_exit_code():
    env = dict(os.environ, RTAMD_BENCH_FAKE_RENDER="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "no_such_workload"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
