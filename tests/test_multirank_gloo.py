"""N>1 plumbing on CPU: two gloo ranks shard a frame by tiles, gather to rank 0, and the assembled frame must be
bit-identical to the unsharded render.  The per-rank "renderer" here is the CPU oracle writing the compact shard
layout — the collective / layout logic under test is exactly what bench.py uses with RCCL on GPUs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["RT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RT_ROOT"], "tests"))
rt = importlib.import_module("raytracing-course-hw_amd")
rtd = importlib.import_module("raytracing-course-hw_amd.distributed")
import oracle_lib, pin_cases
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, SPP, TILE = 70, 45, 2, 16
sd = pin_cases.load_sphere()
orc = oracle_lib.Hw8Oracle(sd)
p = rtd.shard_params(W, H, SPP, rank, world, TILE)
n = rt.lib.rt_output_elems(p)
buf = np.zeros(n, np.float32)
tiles_x = (W + TILE - 1) // TILE
tiles_y = (H + TILE - 1) // TILE
st = 0
for t in range(rank, tiles_x * tiles_y, world):            # this rank's tiles, compact layout
    x0, y0 = (t % tiles_x) * TILE, (t // tiles_x) * TILE
    w, h = min(TILE, W - x0), min(TILE, H - y0)
    rgb, _, _ = orc.render(W, H, SPP, rect=(x0, y0, w, h), threads=2)
    tile = np.zeros((TILE, TILE, 3), np.float32); tile[:h, :w] = rgb
    buf[st * TILE * TILE * 3:(st + 1) * TILE * TILE * 3] = tile.reshape(-1)
    st += 1
frame = rtd.gather_frame(dist, torch.from_numpy(buf), W, H, SPP, rank, world, TILE)
if rank == 0:
    full, _, _ = orc.render(W, H, SPP, threads=2)
    assert frame.shape == full.shape
    assert np.array_equal(frame, full), "sharded frame differs from the unsharded one"
    print("GLOO_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_shard_gather_is_bit_identical(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "GLOO_OK" in out.stdout
