#!/usr/bin/env python3
"""bench.py — Mrays/s of the wavefront path-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py [--config c3] --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is P passes of the hot path; a pass = one call of the generateFrame drop-in (reference
CudaTracer.cu:587-647): eye rays -> up to B x [intersect + NEE + scatter + compaction] -> accumulate,
for S independent samples per pixel (cfg.samplesPerPass; S = 1 is the reference's one sample per tick). The K timed steps
together render the configuration's spp: K x P x S = spp (`plan_steps`; with the defaults P = 1). Workloads (`--config`,
BASELINE.json `configs`):

  c3 (default)  configs[2]/[3]: 1920x1080, preset "mixed" (22 spheres + 16 triangles, Lambert / Phong / Cook-Torrance /
                glass / mirror), 8 bounces, 2000 spp: default 50 steps x 1 pass x S = 40; the driver's --steps 20 gives
                20 steps x 2 passes x S = 50 = the same 2000 spp
  c2            configs[1]: 1280x720, preset "lambert" (the default scene's 36 primitives, every non-emissive material
                Lambert), 8 bounces, 512 spp: default 16 steps x 1 pass x S = 32
  c5            configs[4]: 3840x2160, preset "stress" (1,024 spheres + open Cornell box), 12 bounces; the config asks for
                4096 spp on 8 GPUs — one GPU runs 128 spp of that very frame here (8 steps x 1 pass x S = 16; a rate metric).
                S = 16 since the end of round 3: 133 M rays per pass keep the late bounces' launches wide (S = 4: 5,900,
                8: 6,210, 16: 6,450 Mrays/s; the library takes < 226 M rays per pass)

For N > 1 the SAME frame is sharded by interleaved 8-row bands across the ranks (north_star: pixel-tile shard),
so total work is fixed ("strong" scaling); the integer accumulators are gathered to rank 0 with one
RCCL gather inside the timed region.

`value` counts rays the way BASELINE.md §2 defines them: one live ray processed in one bounce,
summed over bounces and passes (device-side counter), over the wall time of the K timed steps.
`roofline` prices the bounce kernel two ways and names the binding one: `frac` = algorithmic bytes (152 B per
ray-bounce: 76 B SoA state read + 76 B written) / HIP-event time of the kernel against 8 TB/s HBM;
`valu.issue_frac` = wave-level VALU instructions per launch (SQ_INSTS_VALU, from the committed PMC summary
profiles/pmc_counters.json, taken with this configuration) / (1,024 SIMDs x 2.4 GHz / 2 cycles x the launch time
measured in THIS run). `roofline.moved_bytes_frac` prices the same launches by the bytes the fused kernels really move. `cpu_baseline` times oracle/
(the CPU restatement, OpenMP) on a bounded slice of the same workload. `s1_mrays_per_s` is a short leg of the same frame at
S = 1, the reference's own mode, with the default configuration (one lane, ordered on the caller's stream);
`s1_free_running_lanes_mrays_per_s` the same for a caller that opted into free-running frame lanes (cfg.lanesFreeRun: two ray
populations on two streams at 1080p, image and loop guard identical to one); `s1_two_shards_two_streams_mrays_per_s` the same
frame as two pixel-band shard CONTEXTS on two streams (the multi-GPU sharding on one GPU: the loop-guard caveat applies).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))

SEED = 0x5EED
BYTES_PER_RAY_BOUNCE = 152          # BASELINE.md §3 / SURVEY.md §8(d)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2   # 1,024 SIMDs, one wave64 VALU instruction per 2 cycles, 2.4 GHz
BAND_ROWS = 8

CONFIGS = {
    # spp: what the BASELINE config asks; run_spp: what the K timed steps of a run render together (c5: a stated share of it on
    # one GPU); samples: S of the default run; s_range: the S a run may choose so that K x P x S = run_spp for another K
    "c2": dict(index=1, width=1280, height=720, bounces=8, preset="lambert", spp=512, run_spp=512, samples=32, steps=16, s_range=(16, 64),
               what="36 primitives (20 spheres, 16 triangles), Lambert only"),
    "c3": dict(index=2, width=1920, height=1080, bounces=8, preset="mixed", spp=2000, run_spp=2000, samples=40, steps=50, s_range=(16, 64),
               what="22 spheres, 16 triangles, Lambert/Phong/Cook-Torrance/glass/mirror"),
    "c5": dict(index=4, width=3840, height=2160, bounces=12, preset="stress", spp=4096, run_spp=128, samples=16, steps=8, s_range=(16, 16),
               what="1,024 random spheres + 12 triangles, all material classes (stream-compaction stress)"),
}


def plan_steps(cfg, steps=None, samples=None):
    """(K, P, S): K timed steps of P passes of S sample lanes per pixel, K x P x S >= run_spp with equality whenever run_spp / K
    has a divisor in the configuration's S range (c3, the driver's --steps 20: 2000 / 20 = 100 = 2 passes x S = 50; the default
    K = 50: 1 pass x S = 40). An explicit --samples-per-pass is kept and P rounds up."""
    K = steps if steps is not None else cfg["steps"]
    per_step = -(-cfg["run_spp"] // K)
    if samples is None:
        lo, hi = cfg["s_range"]
        fits = [d for d in range(lo, hi + 1) if per_step % d == 0]
        samples = max(fits) if fits else (cfg["samples"] if per_step >= cfg["samples"] else max(lo, min(hi, per_step)))
    return K, -(-per_step // samples), samples


def moved_bytes(live, first_in=32, later_in=76, survivor_out=76, ended_out=36):
    """HBM bytes the fused kernels actually move for one pass with `live[b]` rays entering bounce b: bounce 0
    reads a 32-B home record instead of a 76-B pool slot; a surviving ray writes its 76-B slot; a path that ends writes 36 B
    (its 8-bit sample word + the 32-B home record) instead."""
    total = 0
    for b, n in enumerate(live):
        nxt = live[b + 1] if b + 1 < len(live) else 0
        total += n * (first_in if b == 0 else later_in) + nxt * survivor_out + (n - nxt) * ended_out
    return total


def cpu_baseline(cfg, budget_s=12.0):
    """oracle/ (kind "port") on the same scene and bounce count; a few passes, bounded by time. The 4K / 1,024-sphere
    frame costs minutes per pass on the host, so c5 is sampled at a quarter of the resolution in each direction."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import ptss
    scene = ptss.Scene(cfg["preset"])
    w, h = cfg["width"], cfg["height"]
    scaled = ""
    if cfg["preset"] == "stress":
        w, h = w // 4, h // 4
        scaled = f" (the {cfg['width']}x{cfg['height']} frame sampled at 1/4 resolution per axis: same scene, camera and bounces)"
    cores = oracle.cpu_share()   # affinity capped by the cgroup quota (16 on a 1-GPU box)
    oracle.set_threads(cores)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=cfg["bounces"], seed=SEED)  # RNG init not timed
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
            "sample": f"{passes} passes (1 spp each) of {w}x{h} '{cfg['preset']}'{scaled}, {cfg['bounces']} bounces, "
                      f"{rays} ray-bounces in {dt:.1f} s, OpenMP over rays (bounce kernel, compaction by per-thread counts + "
                      f"prefix, accumulate)"}


def pmc_counters(config, samples):
    """Committed PMC summary (tools/pmc_counters.py, separate rocprofv3 --pmc passes) for exactly this configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        j = json.load(open(path))
        e = j.get(f"{config}_s{samples}")
        if e:
            return e, "profiles/pmc_counters.json[%s_s%d] (%s)" % (config, samples, j.get("source", "?"))
    except (OSError, ValueError):
        pass
    return None, None


def s1_leg(ptss, torch, scene, cfg, shards=1, passes=300, warmup=40, frame_lanes=0, lanes_free_run=False):
    """The reference's own mode on the same frame: one sample per pixel per generateFrame call (CudaTracer.cu:587-647).
    shards = 1: one context, the reference's semantics to the letter. shards = K > 1: the same frame as K interleaved
    pixel-band shards (cfg.tileWorld = K, the multi-GPU sharding) in THIS process on this one GPU, each on its own stream,
    so that the tail of one shard's small launches overlaps the other shards' kernels; same image whenever more than 128
    rays stay alive frame-wide (the sharding caveat)."""
    rs, pix = [], []
    for k in range(shards):
        r = ptss.Renderer(scene, cfg["width"], cfg["height"], max_iterations=cfg["bounces"], seed=SEED, device=torch.cuda.current_device(),
                          tile_rank=k, tile_world=shards, band_rows=BAND_ROWS, sync_each_frame=False, samples_per_pass=1,
                          frame_lanes=frame_lanes if shards == 1 else 1,   # shard contexts: one lane each (the experiment is the two contexts)
                          lanes_free_run=lanes_free_run and shards == 1)
        r.set_stream(torch.cuda.Stream().cuda_stream if shards > 1 else torch.cuda.current_stream().cuda_stream)
        rs.append(r)
        pix.append(torch.zeros((r.local_pixels, 4), dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()

    def frame():
        for r, p in zip(rs, pix):
            r.generate_frame(p.data_ptr())

    for _ in range(warmup):
        frame()
    torch.cuda.synchronize()
    r0 = sum(r.total_ray_bounces() for r in rs)
    t0 = time.perf_counter()
    for _ in range(passes):
        frame()
    torch.cuda.synchronize()
    for r in rs:
        r.synchronize()
    dt = time.perf_counter() - t0
    rays = sum(r.total_ray_bounces() for r in rs) - r0
    for r in rs:
        r.close()
    return round(rays / dt / 1e6, 2), round(dt / passes * 1e3, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--samples-per-pass", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-s1-leg", action="store_true")
    # Rehearsal switches for tests on a one-GPU box (the driver never passes them): every rank on device 0, and gloo
    # (collectives staged through host memory) in place of RCCL.
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    # ... and the N > 1 code path (process group, barrier, gather, all-reduce) taken with ONE rank: the only way RCCL itself
    # — its communicator, its kernels, next to libptss.so's HIP runtime in one process — can be executed on a one-GPU box.
    ap.add_argument("--collectives-at-one-rank", action="store_true")
    ap.add_argument("--dump-frame", default=None, help="tests: rank 0 saves the whole-frame integer accumulator (.npy)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    steps, passes_per_step, samples = plan_steps(cfg, args.steps, args.samples_per_pass)
    W, H, B, preset = cfg["width"], cfg["height"], cfg["bounces"], cfg["preset"]

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
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    dist = None
    if world > 1 or args.collectives_at_one_rank:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --collectives-at-one-rank without a launcher
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import tiles
    scene = ptss.Scene(preset)
    r = ptss.Renderer(scene, W, H, max_iterations=B, seed=SEED, device=local_rank,
                      tile_rank=rank, tile_world=world, band_rows=BAND_ROWS, sync_each_frame=False,
                      time_kernels=not args.no_kernel_timing, samples_per_pass=samples,
                      # Nothing is enqueued on this stream between two frames (the one reader of the accumulator, the gather, comes
                      # after the last frame and behind its join), so the context may run its lanes free (include/ptss.h
                      # lanesFreeRun): the library then gives a small shard of a multi-GPU frame two lanes, a wide pass one.
                      lanes_free_run=True)
    stream = torch.cuda.current_stream()
    r.set_stream(stream.cuda_stream)
    # torch owns the buffers that leave the renderer: accumulator (gathered) and display pixels
    acc = torch.zeros((r.local_pixels, 3), dtype=torch.int32, device="cuda")
    pix = torch.zeros((r.local_pixels, 4), dtype=torch.uint8, device="cuda")
    r.bind_accumulator(acc.data_ptr())
    gather_list = None
    if dist is not None:
        sizes = [len(ptss.tile_rows(H, BAND_ROWS, k, world)) * W for k in range(world)]
        maxn = max(sizes)
        send = torch.zeros((maxn, 3), dtype=torch.int32, device=coll_dev)
        if rank == 0:
            gather_list = [torch.zeros((maxn, 3), dtype=torch.int32, device=coll_dev) for _ in range(world)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for _ in range(passes_per_step):
            r.generate_frame(pix.data_ptr())

    for _ in range(args.warmup):
        step()
    barrier()
    r.bounce_kernel_time()            # reset the event accumulators
    rays0 = r.total_ray_bounces()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if dist is not None:              # the frame's one collective: accumulator tiles -> rank 0
        send[:r.local_pixels].copy_(acc)
        dist.gather(send, gather_list, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0

    if rank == 0 and dist is not None:   # untimed sanity: the gathered tiles tile the frame
        sizes = [len(ptss.tile_rows(H, BAND_ROWS, k, world)) * W for k in range(world)]
        frame = tiles.untile([g[:sizes[k]].cpu().numpy() for k, g in enumerate(gather_list)], W, H, BAND_ROWS)
        assert frame.shape == (W * H, 3) and int(frame.max()) <= 255 * (steps + args.warmup) * passes_per_step * samples
        if args.dump_frame:
            import numpy as np
            np.save(args.dump_frame, frame)
    if rank == 0 and dist is None and args.dump_frame:
        import numpy as np
        np.save(args.dump_frame, acc.cpu().numpy())
    rays = r.total_ray_bounces() - rays0
    # bounce-kernel BUSY time of this rank: the union of its launches' event intervals (one lane: the sum of the launch
    # durations; lanes overlap in time, and a sum of their durations would not be a time)
    kms, klaunches = (0.0, 0) if args.no_kernel_timing else r.bounce_kernel_time()
    lanes = r.frame_lanes
    live = [int(x) for x in r.live_counts()]      # this rank's rays entering each bounce of the last pass
    moved = float(moved_bytes(live)) * (float(rays) / max(sum(live), 1))   # scaled from the last pass to the timed region
    stats = torch.tensor([elapsed, float(rays), kms, float(klaunches), moved], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, rays, kms, klaunches, moved = float(mx[0]), float(sm[1]), float(sm[2]), float(sm[3]), float(sm[4])

    if rank == 0:
        spp_run = steps * passes_per_step * samples
        asks = (f"= BASELINE configs[{cfg['index']}]'s {cfg['spp']} spp" if spp_run == cfg["spp"] else
                f"run (BASELINE configs[{cfg['index']}] asks {cfg['spp']} spp; a rate metric)")
        out = {
            "metric": f"Mrays/sec at {W}x{H}, {B} bounces, {spp_run} spp {asks}; ray = one live ray processed in one bounce",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "passes_per_step": passes_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[{cfg['index']}]: {W}x{H} '{preset}' preset ({cfg['what']}), {B} bounces, "
                                   f"{spp_run} spp = {steps} steps x {passes_per_step} pass(es) x {samples} sample lanes per pixel",
                       "name": args.config,
                       "samples_per_pass": samples,
                       "frame_lanes": lanes,
                       "lanes_free_run": True,
                       "sharding": f"{world} rank(s), interleaved {BAND_ROWS}-row pixel bands"
                                   + (", one RCCL gather of the uint3 accumulator" if world > 1 else ""),
                       "seed": SEED},
            "parity": "bit-exact against this repo's CPU oracle (tests/); unpinned by the reference, which holds no fixtures "
                      "and cannot be built here (DESIGN.md §4)",
            "mpaths_per_s": round(W * H * spp_run / elapsed / 1e6, 2),
            "ray_bounces": int(rays),
            "live_counts": live if world == 1 else None,  # rays entering each bounce, last pass
        }
        if kms > 0:
            # per launch: algorithmic bytes of the rays one launch processes / that launch's duration;
            # averaged over every bounce-kernel launch of the timed region (all ranks)
            # (N ranks: kms is the SUM of the ranks' busy times and rays their total, so rays / (kms / N) would be the
            # aggregate rate; per GPU, which is what a per-device peak prices, it is rays / kms)
            gbs = rays * BYTES_PER_RAY_BOUNCE / (kms * 1e-3) / 1e9
            hbm_frac = gbs / HBM_PEAK_GBS
            avg_launch_s = kms * 1e-3 / max(klaunches, 1)
            roof = {
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 4), "traffic": None,
                # the same launches priced by the bytes the fused kernels really move (bounce 0 reads a 32-B home record, an
                # ended path writes 36 B): `frac` is SURVEY §8(d)'s 152 B per ray-bounce, this is the honest lower figure
                "moved_bytes_frac": round(moved / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "moved_bytes_per_ray_bounce": round(moved / max(rays, 1), 1),
                "kernel": "ptss::bounceKernel", "launches": int(klaunches),
                "avg_launch_us": round(avg_launch_s * 1e6, 2),
                "algorithmic_bytes_per_launch": round(rays * BYTES_PER_RAY_BOUNCE / max(klaunches, 1)),
                "kernel_grays_per_s": round(rays / (kms * 1e-3) / 1e9, 3),
                "per": "GPU" if world > 1 else "launch",
            }
            if lanes > 1:
                roof["lanes_note"] = (f"{lanes} frame lanes: their kernels overlap in time, so the launch time is the union of the "
                                      f"launches' event intervals divided by the launch count — an effective, not an isolated, duration")
            # the PMC summary is per launch of the N = 1 run at the default S; a rank of N renders 1 / N of the frame, and another S
            # scales a launch's rays by S / S_default: counters per launch scale with the rays of a launch
            pmc, src = pmc_counters(args.config, cfg["samples"])
            if pmc:
                scale = (samples / cfg["samples"]) / world
                how = "measured per launch in separate --pmc passes, not in this run"
                if scale != 1.0:
                    how += f"; scaled by {scale:.4g} = rays per launch of this run / of the profiled one ({world} rank(s), S = {samples})"
                roof["traffic"] = round(pmc.get("hbm_bytes_per_launch") * scale) if pmc.get("hbm_bytes_per_launch") else None
                roof["traffic_source"] = src + " — " + how
                insts = pmc.get("valu_insts_per_launch")
                if insts:
                    insts = insts * scale
                    issue = insts / avg_launch_s / VALU_PEAK_WAVE_INSTR_PER_S
                    roof["valu"] = {
                        "insts_per_launch": round(insts), "issue_frac": round(issue, 4),
                        "lanes_active": pmc.get("valu_lanes_active"),
                        "insts_per_64_ray_tile": pmc.get("valu_insts_per_tile_mid_bounce"),
                        "peak": "1,024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction",
                        "source": src + "; launch time from this run",
                        "measured_cost_per_instruction": "plain FP32/int 2.3, compare / select / SGPR-operand / 3-input integer "
                                                         "4.3, transcendental 8.4 SIMD-cycles at 7 waves per SIMD "
                                                         "(tools/microbench/vgpr_banks.hip): the mix this kernel issues cannot reach 2",
                    }
                    if issue > hbm_frac:
                        roof["bound"] = "valu"
            roof["note"] = ("brute-force intersection puts this kernel on the FP32-VALU side of the ridge (BASELINE.md §3, DESIGN.md §3): "
                            "`frac`/`achieved`/`peak` are the HBM figures the contract asks for, `bound` names the higher of "
                            "the HBM and VALU-issue fractions")
            out["roofline"] = roof
        if world == 1 and not args.no_s1_leg:
            v, ms = s1_leg(ptss, torch, scene, cfg)               # the default configuration: ordered on the caller's stream, one lane
            out["s1_mrays_per_s"] = v
            out["s1_ms_per_pass"] = ms
            vf, msf = s1_leg(ptss, torch, scene, cfg, lanes_free_run=True)  # opt-in (cfg.lanesFreeRun): the library's lane choice, 2 at 1080p
            out["s1_free_running_lanes_mrays_per_s"] = vf
            out["s1_free_running_lanes_ms_per_pass"] = msf
            v2, ms2 = s1_leg(ptss, torch, scene, cfg, shards=2, frame_lanes=1)
            out["s1_two_shards_two_streams_mrays_per_s"] = v2
            out["s1_two_shards_two_streams_ms_per_pass"] = ms2
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)

    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
