#!/usr/bin/env python3
"""bench.py — Mrays/s of the wavefront path-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path = one call of the generateFrame drop-in (reference
CudaTracer.cu:587-647): eye rays -> up to 8 x [intersect + NEE + scatter + compaction] -> accumulate,
for S = --samples-per-pass independent samples per pixel (cfg.samplesPerPass; S = 1 is the reference's one
sample per tick; the default S = 40 x 50 passes is the config's 2000 spp, and keeps every launch — and every
shard of an 8-GPU run — wide enough to fill the chip; the image does not depend on N). Workload at every
N: BASELINE.json configs[2]/[3] — 1920x1080, scene preset "mixed" (22 spheres + 16 triangles, Lambert /
Phong / Cook-Torrance / glass / mirror), 8 bounces. For N > 1 the
SAME frame is sharded by interleaved 8-row bands across the ranks (north_star: pixel-tile shard),
so total work is fixed ("strong" scaling); the integer accumulators are gathered to rank 0 with one
RCCL gather inside the timed region.

`value` counts rays the way BASELINE.md §2 defines them: one live ray processed in one bounce,
summed over bounces and passes (device-side counter), over the wall time of the K timed steps.
The `roofline` object prices the bounce kernel: algorithmic bytes = 152 B per ray-bounce (76 B SoA
state read + 76 B written) / HIP-event time of the kernel, against 8 TB/s HBM.
`cpu_baseline` times oracle/ (the CPU restatement, OpenMP) on a bounded slice of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))

WIDTH, HEIGHT, BOUNCES, PRESET, SEED = 1920, 1080, 8, "mixed", 0x5EED
BYTES_PER_RAY_BOUNCE = 152          # BASELINE.md §3 / SURVEY.md §8(d)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
BAND_ROWS = 8


def cpu_baseline(budget_s=12.0):
    """oracle/ (kind "port") on the same scene/resolution/bounces; a few passes, bounded by time."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import ptss
    scene = ptss.Scene(PRESET)
    cores = oracle.cpu_share()   # affinity capped by the cgroup quota (16 on a 1-GPU box)
    oracle.set_threads(cores)
    o = oracle.Oracle(scene.desc, WIDTH, HEIGHT, max_iterations=BOUNCES, seed=SEED)  # RNG init not timed
    o.generate_frame()  # warm-up pass (page-in), not timed
    r0 = o.total_ray_bounces()
    t0 = time.perf_counter()
    passes = 0
    while passes < 2 or (time.perf_counter() - t0 < budget_s and passes < 256):
        o.generate_frame()
        passes += 1
    dt = time.perf_counter() - t0
    rays = o.total_ray_bounces() - r0
    o.close()
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{passes} passes (1 spp each) of {WIDTH}x{HEIGHT} '{PRESET}', {BOUNCES} bounces, "
                      f"{rays} ray-bounces in {dt:.1f} s, OpenMP over rays"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--samples-per-pass", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    import torch
    import ptss

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # Rehearsal knobs (tests / one-GPU boxes only; the driver never sets them): PTSS_BENCH_ONE_GPU=1 puts every rank on
    # device 0, PTSS_BENCH_BACKEND=gloo swaps RCCL for gloo (collectives staged through host memory).
    backend = os.environ.get("PTSS_BENCH_BACKEND", "nccl")
    if os.environ.get("PTSS_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import tiles
    scene = ptss.Scene(PRESET)
    r = ptss.Renderer(scene, WIDTH, HEIGHT, max_iterations=BOUNCES, seed=SEED, device=local_rank,
                      tile_rank=rank, tile_world=world, band_rows=BAND_ROWS, sync_each_frame=False,
                      time_kernels=not args.no_kernel_timing, samples_per_pass=args.samples_per_pass)
    stream = torch.cuda.current_stream()
    r.set_stream(stream.cuda_stream)
    # torch owns the buffers that leave the renderer: accumulator (gathered) and display pixels
    acc = torch.zeros((r.local_pixels, 3), dtype=torch.int32, device="cuda")
    pix = torch.zeros((r.local_pixels, 4), dtype=torch.uint8, device="cuda")
    r.bind_accumulator(acc.data_ptr())
    gather_list = None
    if world > 1:
        sizes = [len(ptss.tile_rows(HEIGHT, BAND_ROWS, k, world)) * WIDTH for k in range(world)]
        maxn = max(sizes)
        send = torch.zeros((maxn, 3), dtype=torch.int32, device=coll_dev)
        if rank == 0:
            gather_list = [torch.zeros((maxn, 3), dtype=torch.int32, device=coll_dev) for _ in range(world)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        r.generate_frame(pix.data_ptr())

    for _ in range(args.warmup):
        step()
    barrier()
    r.bounce_kernel_time()            # reset the event accumulators
    rays0 = r.total_ray_bounces()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:              # the frame's one collective: accumulator tiles -> rank 0
        send[:r.local_pixels].copy_(acc)
        dist.gather(send, gather_list, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0

    if rank == 0 and dist is not None:   # untimed sanity: the gathered tiles tile the frame
        sizes = [len(ptss.tile_rows(HEIGHT, BAND_ROWS, k, world)) * WIDTH for k in range(world)]
        frame = tiles.untile([g[:sizes[k]].cpu().numpy() for k, g in enumerate(gather_list)], WIDTH, HEIGHT, BAND_ROWS)
        assert frame.shape == (WIDTH * HEIGHT, 3) and int(frame.max()) <= 255 * (args.steps + args.warmup) * args.samples_per_pass
    rays = r.total_ray_bounces() - rays0
    kms, klaunches = (0.0, 0) if args.no_kernel_timing else r.bounce_kernel_time()
    stats = torch.tensor([elapsed, float(rays), kms, float(klaunches)], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, rays, kms, klaunches = float(mx[0]), float(sm[1]), float(sm[2]), float(sm[3])
        my_rays = float(stats[1])
    else:
        my_rays = float(rays)

    if rank == 0:
        out = {
            "metric": "Mrays/sec at 1920x1080, 8 bounces, 2000 spp (ray = one live ray processed in one bounce)",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[2]: {WIDTH}x{HEIGHT} '{PRESET}' preset (22 spheres, 16 triangles, "
                                   f"Lambert/Phong/Cook-Torrance/glass/mirror), {BOUNCES} bounces, "
                                   f"{args.steps * args.samples_per_pass} spp = {args.steps} passes x {args.samples_per_pass} "
                                   f"sample lanes per pixel",
                       "samples_per_pass": args.samples_per_pass,
                       "sharding": f"{world} rank(s), interleaved {BAND_ROWS}-row pixel bands"
                                   + (", one RCCL gather of the uint3 accumulator" if world > 1 else ""),
                       "seed": SEED},
            "mpaths_per_s": round(WIDTH * HEIGHT * args.steps * args.samples_per_pass / elapsed / 1e6, 2),
            "ray_bounces": int(rays),
        }
        if kms > 0:
            # per launch: algorithmic bytes of the rays one launch processes / that launch's duration;
            # averaged over every bounce-kernel launch of the timed region (all ranks)
            gbs = rays * BYTES_PER_RAY_BOUNCE / (kms * 1e-3) / 1e9
            pmc = None
            pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc_file):
                try:  # measured per launch on ONE GPU for one lane count: reported only for that very configuration
                    j = json.load(open(pmc_file))
                    if world == 1 and j.get("samples_per_pass") == args.samples_per_pass:
                        pmc = j.get("hbm_bytes_per_launch")
                except Exception:
                    pmc = None
            out["roofline"] = {
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc,
                "kernel": "ptss::bounceKernel", "launches": int(klaunches),
                "avg_launch_us": round(kms * 1e3 / max(klaunches, 1), 2),
                "algorithmic_bytes_per_launch": round(rays * BYTES_PER_RAY_BOUNCE / max(klaunches, 1)),
                "kernel_grays_per_s": round(rays / (kms * 1e-3) / 1e9, 3),
                "note": "brute-force intersection puts this kernel on the FP32-VALU side of the ridge "
                        "(BASELINE.md §3, DESIGN.md); HBM fraction is reported as the contract asks",
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
