"""tiles.py — host logic of the pixel-tile shard (north_star): which rows a rank owns,
and how rank 0 puts gathered per-rank accumulators back into frame order. Pure numpy + optional
torch.distributed; used by bench.py (RCCL) and by the gloo tests."""
import numpy as np

import ptss


def rank_rows(height, band_rows, world):
    """[rows of rank 0, rows of rank 1, ...] — interleaved bands of band_rows rows."""
    return [ptss.tile_rows(height, band_rows, r, world) for r in range(world)]


def untile(per_rank, width, height, band_rows):
    """per_rank[r]: array (local_pixels_r, C) in the rank's local pixel order -> (height*width, C) frame order."""
    world = len(per_rank)
    rows = rank_rows(height, band_rows, world)
    c = per_rank[0].shape[1]
    out = np.zeros((height, width, c), dtype=per_rank[0].dtype)
    for r in range(world):
        n = len(rows[r])
        out[rows[r]] = np.asarray(per_rank[r])[: n * width].reshape(n, width, c)
    return out.reshape(height * width, c)


def extract(frame, width, height, band_rows, rank, world):
    """Inverse of untile for one rank: frame (height*width, C) -> that rank's local pixels."""
    rows = ptss.tile_rows(height, band_rows, rank, world)
    f = np.asarray(frame).reshape(height, width, -1)
    return f[rows].reshape(len(rows) * width, f.shape[2])


def gather_to_rank0(local, width, height, band_rows, dist, device=None):
    """One gather of the per-rank accumulators (padded to the largest tile) and the un-tiling on rank 0.
    `local`: torch tensor (local_pixels, C). Returns the frame-ordered numpy array on rank 0, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [len(ptss.tile_rows(height, band_rows, r, world)) * width for r in range(world)]
    maxn = max(sizes)
    send = torch.zeros((maxn, local.shape[1]), dtype=local.dtype, device=local.device)
    send[: local.shape[0]].copy_(local)
    bufs = [torch.zeros_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, bufs, dst=0)
    if rank != 0:
        return None
    return untile([b[: sizes[r]].cpu().numpy() for r, b in enumerate(bufs)], width, height, band_rows)
