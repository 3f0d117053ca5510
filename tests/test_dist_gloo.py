"""The N > 1 host path on CPU: world_size-2 (and 3) gloo process groups run the same tile ownership,
gather and un-tile code bench.py runs over RCCL. Each rank stands in for its GPU with the oracle's
rows of the frame — legitimate because the RNG stream is bound to the GLOBAL pixel, so a rank's tile
is, bit for bit, those rows of the full frame (checked on real hardware in test_gpu_tiles.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, BAND, BOUNCES, TICKS = 40, 22, 4, 3, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (os.path.join(ROOT, "cuda-path-tracer-ss_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle
    import ptss
    import tiles
    oracle.set_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene = ptss.Scene("cornell")
    o = oracle.Oracle(scene.desc, W, H, max_iterations=BOUNCES)
    for _ in range(TICKS):
        o.generate_frame()
    full = o.accumulator().astype(np.int32)
    mine = tiles.extract(full, W, H, BAND, rank, world)
    local = torch.from_numpy(mine.copy())
    frame = tiles.gather_to_rank0(local, W, H, BAND, dist)
    # bench.py's reductions: MAX of the elapsed times, SUM of the ray counts
    stats = torch.tensor([1.0 + rank, float(len(mine))], dtype=torch.float64)
    mx, sm = stats.clone(), stats.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(outdir, "frame.npy"), frame)
        np.save(os.path.join(outdir, "full.npy"), full)
        np.save(os.path.join(outdir, "stats.npy"), np.array([float(mx[0]), float(sm[1])]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_untile_over_gloo(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    frame, full = np.load(tmp_path / "frame.npy"), np.load(tmp_path / "full.npy")
    assert frame.shape == full.shape and np.array_equal(frame, full)
    mx, total = np.load(tmp_path / "stats.npy")
    assert mx == float(world) and total == W * H
