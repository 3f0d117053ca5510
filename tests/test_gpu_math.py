"""The device fast paths of ptm::rcp / ptm::sqrt / ptm::div (hardware approximation + one fma correction) are only
legitimate because they equal the IEEE results bit for bit. That is proven here by exhaustion on the GPU under test:
every one of the 2^32 float32 patterns for rcp and sqrt; every one of the 2^46 mantissa pairs for div (~45 s), plus
the guarded function across an exponent grid and the special values. Likewise the table form of the 8-bit tone map
(csrc/ptquant.h) against the literal clamp / pow / scale sequence, for every float32 pattern."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "cuda-path-tracer-ss_amd", "lib", "ptss_mathcheck")


def test_fast_paths_are_ieee_by_exhaustion():
    chunks = os.environ.get("PTSS_DIV_CHUNKS", "32")  # 32 = all 2^46 mantissa pairs
    r = subprocess.run([CHECK, chunks], capture_output=True, text=True, timeout=900)
    m = re.search(r"rcp_mismatch=(\d+) sqrt_mismatch=(\d+) checked=(\d+)", r.stdout)
    assert m, r.stdout + r.stderr
    assert int(m.group(3)) == 2 ** 32
    assert (int(m.group(1)), int(m.group(2))) == (0, 0), r.stdout
    d = re.search(r"div_mantissa_mismatch=(\d+) div_pairs_checked=(\d+) control_uncorrected_mismatch=(\d+) "
                  r"div_exponent_mismatch=(\d+) div_special_mismatch=(\d+)", r.stdout)
    assert d, r.stdout
    assert int(d.group(1)) == 0 and int(d.group(4)) == 0 and int(d.group(5)) == 0, r.stdout
    assert int(d.group(2)) == int(chunks) * (2 ** 18) * (2 ** 23)
    assert int(d.group(3)) > 0, "control: the uncorrected quotient must differ somewhere, or the check is vacuous"
    q = re.search(r"quant_mismatch=(\d+) quant_checked=(\d+) quant_guess_alone_mismatch=(\d+) quant_table_monotone=(\d)", r.stdout)
    assert q, r.stdout
    assert int(q.group(1)) == 0 and int(q.group(2)) == 2 ** 32 and q.group(4) == "1", r.stdout
    assert int(q.group(3)) > 0, "control: the hardware guess alone must be off somewhere, or the check is vacuous"
    assert r.returncode == 0
