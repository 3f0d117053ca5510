"""bench.py's N>1 path rehearsed on ONE GPU: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N), with the two rehearsal switches
(--rehearse-on-one-gpu --backend gloo) that let them share device 0 and talk over gloo instead of RCCL. Checks the contract line (one JSON line from
rank 0, whole-job value, strong scaling) and that sharding does not change the number of rays traced."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(world, port, steps=2, warmup=1):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup),
           "--rehearse-on-one-gpu", "--backend", "gloo", "--no-s1-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_print_one_contract_line():
    one = run_bench(1, 29541)
    two = run_bench(2, 29542)
    for j, n in ((one, 1), (two, 2)):
        assert j["n_gpus"] == n and j["steps"] == 2 and j["warmup"] == 1
        assert j["unit"] == "Mrays/s" and j["higher_is_better"] is True and j["vs_baseline"] is None
        assert j["value"] > 0 and j["ms_per_step"] > 0
        assert j["roofline"]["bound"] in ("hbm", "valu") and 0 < j["roofline"]["frac"] < 1
    assert two["scaling"] == "strong"
    assert "cpu_baseline" in one and "cpu_baseline" not in two  # rank 0 at N = 1 only
    # Same seed, same passes: the shards together trace exactly the rays the single context traces.
    assert two["ray_bounces"] == one["ray_bounces"]
