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


class FrameGatherer:
    """The exchange step with everything allocated ONCE: padded send buffer, rank 0's receive buffers and the tile grid.
    gather(local) moves this rank's shard buffer to rank 0 and returns the assembled (H, W, 3) frame there (a view into
    the preallocated grid, on `device`), None on the other ranks."""

    def __init__(self, dist, width, height, samples, rank, world, tile, dtype, device):
        import torch
        self.dist, self.rank, self.world, self.tile, self.width, self.height = dist, rank, world, tile, width, height
        self.sizes = shard_elems(width, height, samples, world, tile)
        self.pad = max(self.sizes)
        self.device = torch.device(device)
        self.send = torch.zeros(self.pad, dtype=dtype, device=self.device)
        self.bufs = [torch.empty(self.pad, dtype=dtype, device=self.device) for _ in range(world)] if rank == 0 else None
        self.tiles_x, self.tiles_y = (width + tile - 1) // tile, (height + tile - 1) // tile
        self.grid = torch.zeros((self.tiles_y * self.tiles_x, tile, tile, 3), dtype=dtype, device=self.device) if rank == 0 else None

    def gather(self, local):
        n = self.sizes[self.rank]
        self.send[:n].copy_(local[:n])  # device-to-device on the GPUs; device-to-host in a gloo rehearsal
        self.dist.gather(self.send, self.bufs, dst=0)
        if self.rank != 0:
            return None
        for r in range(self.world):
            self.grid[r::self.world] = self.bufs[r][:self.sizes[r]].view(-1, self.tile, self.tile, 3)
        t = self.tile
        return self.grid.view(self.tiles_y, self.tiles_x, t, t, 3).permute(0, 2, 1, 3, 4).reshape(self.tiles_y * t, self.tiles_x * t, 3)[:self.height, :self.width]


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
