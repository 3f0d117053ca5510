"""bench.py's N>1 path rehearsed on ONE GPU: ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N), with the two rehearsal switches
(--rehearse-on-one-gpu --backend gloo) that let them share device 0 and talk over gloo instead of RCCL.
Checks the contract line (one JSON line from rank 0, whole-job value, strong scaling), that sharding does not change
the number of rays traced, and that the gathered frame IS the unsharded accumulator, element for element.

The driver's own shape is 8 ranks; a GPU box admits at most 6 processes on its card at once (the test runner itself
is one of them), so the widest rehearsal here is 4 ranks (135 row bands over 4 ranks: uneven shards, so the gather's padding
to the largest shard is exercised as it is at 8). RCCL refuses two ranks on one device, so the sharded runs talk over gloo;
RCCL itself is executed with ONE rank through the same code path (test_rccl_itself_runs_the_collective_path_with_one_rank)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(world, port, tmp_path, steps=2, warmup=1, extra=(), rehearsal=("--rehearse-on-one-gpu", "--backend", "gloo")):
    dump = os.path.join(str(tmp_path), f"frame_{world}_{port}.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup),
           "--no-s1-leg", "--dump-frame", dump] + list(rehearsal) + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), np.load(dump)


def test_ranks_print_one_contract_line_and_gather_the_unsharded_frame(tmp_path):
    # (S = 20 everywhere — the sample lanes own the random streams, so only runs of one S trace the same rays —: a 4-rank shard
    # is then 10.4 M rays per pass, which the library gives two frame lanes, as it does an 8-rank shard at the default S)
    s20 = ["--samples-per-pass", "20"]
    one, frame1 = run_bench(1, 29541, tmp_path, extra=["--no-cpu-baseline"] + s20)
    two, frame2 = run_bench(2, 29542, tmp_path, extra=s20)
    six, frame6 = run_bench(4, 29546, tmp_path, extra=s20)
    for j, n in ((one, 1), (two, 2), (six, 4)):
        assert j["n_gpus"] == n and j["steps"] == 2 and j["warmup"] == 1
        assert j["unit"] == "Mrays/s" and j["higher_is_better"] is True and j["vs_baseline"] is None
        assert j["value"] > 0 and j["ms_per_step"] > 0
        # the roofline object is there at every N — also where a rank's shard is small enough for the library to give it two
        # frame lanes (the 4-rank shards: the launch time is then the union of the lanes' launch intervals) — with the
        # moved-bytes figure beside SURVEY §8(d)'s 152 B per ray-bounce and the PMC counters scaled to the shard
        roof = j["roofline"]
        assert roof["bound"] in ("hbm", "valu") and 0 < roof["frac"] < 1
        assert 0 < roof["moved_bytes_frac"] < roof["frac"] and 100 < roof["moved_bytes_per_ray_bounce"] < 152
        assert roof["traffic"] > 0 and roof["valu"]["issue_frac"] > 0 and roof["avg_launch_us"] > 0
        assert j["scaling"] == "strong" and "cpu_baseline" not in j
        assert "2000 spp = BASELINE configs[2]'s 2000 spp" in j["metric"] and j["passes_per_step"] == 50   # 2 x 50 x S = 20
    assert six["config"]["frame_lanes"] == 2 and "lanes_note" in six["roofline"] and one["config"]["frame_lanes"] == 1
    # Same seed, same passes: the shards together trace exactly the rays the single context traces ...
    assert two["ray_bounces"] == one["ray_bounces"] == six["ray_bounces"]
    # ... and the gathered, un-tiled accumulator is the single context's, element for element (1920x1080: far more than
    # 128 rays stay alive frame-wide at every bounce, so the sharded loop guard never differs — DESIGN.md §5)
    assert frame1.shape == frame2.shape == frame6.shape == (1920 * 1080, 3)
    assert np.array_equal(frame1.astype(np.int64), frame2.astype(np.int64))
    assert np.array_equal(frame1.astype(np.int64), frame6.astype(np.int64))


def test_the_drivers_twenty_steps_render_the_configs_2000_spp(tmp_path):
    """`bench.py --gpus 1 --steps 20 --warmup 5` is what the driver runs: 20 steps x 2 passes x S = 50 = BASELINE configs[2]'s
    2000 spp, and the line says so; the roofline object is complete."""
    j, _ = run_bench(1, 29571, tmp_path, steps=20, warmup=5, extra=["--no-cpu-baseline"], rehearsal=())
    assert j["steps"] == 20 and j["warmup"] == 5 and j["passes_per_step"] == 2 and j["config"]["samples_per_pass"] == 50
    assert "2000 spp = BASELINE configs[2]'s 2000 spp" in j["metric"]
    roof = j["roofline"]
    assert roof["launches"] == 20 * 2 * 8 and 0 < roof["moved_bytes_frac"] < roof["frac"] < 1
    assert roof["traffic"] > 0 and "scaled by 1.25" in roof["traffic_source"]
    assert j["ray_bounces"] > 2_000_000_000 * 5


def test_rccl_itself_runs_the_collective_path_with_one_rank(tmp_path):
    """What a one-GPU box can execute of the real thing: bench.py's N > 1 code path — init_process_group("nccl"), barriers,
    the gather of the int32 accumulator tile, the all-reduces of the timing record — with world size 1 over RCCL, in the
    process that also holds libptss.so and its HIP runtime. The line and the frame are those of the plain one-GPU run."""
    plain, frame_plain = run_bench(1, 29561, tmp_path, extra=["--no-cpu-baseline"], rehearsal=())
    rccl, frame_rccl = run_bench(1, 29562, tmp_path, extra=["--no-cpu-baseline", "--collectives-at-one-rank", "--backend", "nccl"],
                                 rehearsal=())
    assert rccl["n_gpus"] == 1 and rccl["ray_bounces"] == plain["ray_bounces"]
    assert frame_rccl.shape == (1920 * 1080, 3)
    assert np.array_equal(frame_plain.astype(np.int64), frame_rccl.astype(np.int64))


def test_config5_runs_sharded(tmp_path):
    """bench.py --config c5 --gpus N works (the config is an 8-GPU one): 4 ranks on this one GPU, one pass."""
    one, f1 = run_bench(1, 29551, tmp_path, steps=1, warmup=0, extra=["--config", "c5", "--no-cpu-baseline"])
    four, f4 = run_bench(4, 29554, tmp_path, steps=1, warmup=0, extra=["--config", "c5"])
    assert four["config"]["name"] == "c5" and four["n_gpus"] == 4
    assert four["ray_bounces"] == one["ray_bounces"]
    assert np.array_equal(f1.astype(np.int64), f4.astype(np.int64))


def test_config2_line_is_complete(tmp_path):
    """`bench.py --config c2` as the driver would run it on one GPU (short): the contract keys, the roofline object with its
    HBM figures and — from the committed PMC summary of exactly this configuration — traffic and the VALU object, the S = 1
    legs and the CPU baseline with its sample description."""
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["config"]["name"] == "c2" and "1280x720" in j["config"]["workload"] and "513 spp run" in j["metric"]
    roof = j["roofline"]
    assert roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and roof["bound"] in ("hbm", "valu")
    assert roof["traffic"] > 0 and "pmc_counters.json" in roof["traffic_source"]
    assert 0 < roof["valu"]["issue_frac"] < 1 and 0 < roof["valu"]["lanes_active"] <= 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1 and "1280x720" in j["cpu_baseline"]["sample"]
    assert j["s1_mrays_per_s"] > 0 and j["s1_free_running_lanes_mrays_per_s"] > 0
