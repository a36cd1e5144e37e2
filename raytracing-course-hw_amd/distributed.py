"""Multi-GPU frame assembly: one process per GPU, pixel tiles dealt round-robin, ONE exchange step.

Every rank renders the tiles  t % world == rank  into a compact shard buffer (layout in include/rtamd.h);
rank 0 gathers the buffers (torch.distributed gather: RCCL over xGMI on GPUs, gloo in the CPU tests) and
scatters them into the full frame.  Rendering needs no collective: the per-pixel seed y*W+x is global, so any
partition produces the same pixels (reference: hw8/src/sceneio.cpp:389-391)."""
import importlib

import numpy as np

rt = importlib.import_module(__name__.rsplit(".", 1)[0])


def shard_params(width, height, samples, rank, world, tile=32, **kw):
    return rt.make_params(width, height, samples, shard_index=rank, shard_count=world, tile=tile, **kw)


def shard_elems(width, height, samples, world, tile=32):
    """Output elements of every rank's shard buffer."""
    return [int(rt.lib.rt_output_elems(shard_params(width, height, samples, r, world, tile))) for r in range(world)]


def gather_frame(dist, local, width, height, samples, rank, world, tile=32, as_numpy=True):
    """local: 1-D torch tensor holding this rank's shard buffer (u8 or f32).  Rank 0 gathers the buffers and assembles the
    (H, W, 3) frame WHERE THE BUFFERS ARE (on its GPU with RCCL, on the host with gloo): shard r holds the tiles r, r+world,
    ... in order, so each buffer is one strided assignment into the frame's tile grid.  Returns the frame on rank 0 (numpy
    if as_numpy, else a torch tensor on the gather's device — no device-to-host copy) and None elsewhere."""
    import torch
    sizes = shard_elems(width, height, samples, world, tile)
    pad = max(sizes)
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()  # gloo moves host memory; RCCL (backend "nccl") keeps the tiles on the GPUs
    send = local
    if local.numel() != pad:
        send = torch.zeros(pad, dtype=local.dtype, device=local.device)
        send[:local.numel()] = local
    bufs = [torch.empty(pad, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == 0 else None
    dist.gather(send, bufs, dst=0)
    if rank != 0:
        return None
    tiles_x, tiles_y = (width + tile - 1) // tile, (height + tile - 1) // tile
    grid = torch.zeros((tiles_y * tiles_x, tile, tile, 3), dtype=local.dtype, device=local.device)
    for r in range(world):
        grid[r::world] = bufs[r][:sizes[r]].view(-1, tile, tile, 3)
    frame = grid.view(tiles_y, tiles_x, tile, tile, 3).permute(0, 2, 1, 3, 4).reshape(tiles_y * tile, tiles_x * tile, 3)[:height, :width]
    return frame.cpu().numpy() if as_numpy else frame
