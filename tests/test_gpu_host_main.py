"""The C++ host mirror end to end: ptss_main is the reference's main() + GPUAnimBitmap loop +
generateFrame + Key + saveScreenshot over the C-ABI. Its screenshot must be the oracle's display
buffer, byte for byte; key presses must act like the reference's Key()/moveCamera()."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import ptss

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "cuda-path-tracer-ss_amd", "lib", "ptss_main")


def read_tga(path):
    b = open(path, "rb").read()
    w, h = b[12] | (b[13] << 8), b[14] | (b[15] << 8)
    assert b[2] == 2 and b[16] == 24
    return np.frombuffer(b, np.uint8, w * h * 3, 18).reshape(h * w, 3)[:, ::-1]  # BGR -> RGB, bottom-up order kept


def run_main(args, tmp_path):
    out = str(tmp_path / "shot.tga")
    r = subprocess.run([MAIN] + args + ["--out", out, "--quiet"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    return read_tga(out), r.stdout


def test_main_screenshot_equals_oracle_display(tmp_path):
    rgb, stdout = run_main(["--preset", "cornell", "--size", "96x64", "--ticks", "6", "--bounces", "4"], tmp_path)
    o = oracle.Oracle(ptss.Scene("cornell").desc, 96, 64, max_iterations=4)
    for _ in range(6):
        o.generate_frame()
    assert np.array_equal(rgb, o.pixels()[:, :3])
    assert f"{o.total_ray_bounces()} ray-bounces" in stdout


def test_main_keys_move_camera_and_toggle_mode(tmp_path):
    # 'w','d','f' before the first tick: camera moved; then the oracle must agree when given the same camera
    rgb, _ = run_main(["--preset", "cornell", "--size", "64x64", "--ticks", "4", "--bounces", "3", "--keys", "wdf"], tmp_path)
    cam = ptss.default_camera()
    for k in "wdf":
        ptss.move_camera(cam, k)
    o = oracle.Oracle(ptss.Scene("cornell").desc, 64, 64, max_iterations=3)
    o.set_camera(cam)
    for _ in range(4):
        o.generate_frame()
    assert np.array_equal(rgb, o.pixels()[:, :3])


def test_main_default_arguments_are_the_reference_defaults(tmp_path):
    # no arguments but a small tick count: 512x512 (DIM), default scene, 15 bounces
    rgb, stdout = run_main(["--ticks", "1"], tmp_path)
    assert rgb.shape[0] == 512 * 512
    o = oracle.Oracle(ptss.Scene("default").desc, 512, 512, max_iterations=15)
    o.generate_frame()
    assert np.array_equal(rgb, o.pixels()[:, :3])


def test_main_over_rccl_with_one_gpu_equals_the_single_gpu_path(tmp_path):
    """ptss_main --gpus N (host/MultiGpu.cpp): one process, a context + stream per device, ncclCommInitAll, ONE ncclGather of the
    accumulator tiles to device 0, un-tile + display scaling on the host. A one-GPU box can execute N = 1 — communicator,
    gather, un-tile all run — and the screenshot must be the plain path's, byte for byte. (N > 1 on N devices: unexecuted.)"""
    args = ["--preset", "mixed", "--size", "160x90", "--ticks", "5", "--bounces", "6"]
    plain, out_plain = run_main(args, tmp_path)
    rccl, out_rccl = run_main(args + ["--gpus", "1"], tmp_path)
    assert np.array_equal(plain, rccl)
    assert "1 shard(s) on as many GPUs, gathered by ncclGather" in out_rccl
    assert out_plain.split(",")[1] == out_rccl.split(",")[1]      # the same ray-bounce total


@pytest.mark.parametrize("shards,S", [(2, 1), (3, 1), (4, 2)])
def test_main_sharded_on_one_gpu_reassembles_the_frame(tmp_path, shards, S):
    """The same host path with N shards EMULATED on device 0 (--emulate-gpus: RCCL refuses two ranks on one device, so the
    gather is done by device copies): N contexts on N streams, uneven shards (90 rows in bands of 8 over 3 or 4 shards), the
    padded tiles, the un-tile and the display scaling. The frame equals the unsharded one (more than 128 rays stay alive
    frame-wide at every bounce here, so the sharded loop guard never differs)."""
    args = ["--preset", "mixed", "--size", "160x90", "--ticks", "4", "--bounces", "6", "--samples-per-pass", str(S)]
    plain, out_plain = run_main(args, tmp_path)
    sharded, out = run_main(args + ["--emulate-gpus", str(shards)], tmp_path)
    assert np.array_equal(plain, sharded)
    assert f"{shards} shard(s) on device 0, gathered by device copies" in out
    assert out_plain.split(",")[1] == out.split(",")[1]
