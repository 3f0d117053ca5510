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
