import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cuda-path-tracer-ss_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the native artefacts once if they are missing (hipcc cross-compiles without a GPU)."""
    import ptss
    import oracle as oracle_mod
    missing = [p for p in (ptss.HOST_LIB, ptss.DEVICE_LIB, oracle_mod.LIB) if not os.path.exists(p)]
    if missing:
        import __graft_entry__
        __graft_entry__.build()
    # PyTorch ships its own copy of the HIP runtime and libptss.so links ROCm's: when both live in one process, torch's
    # must initialise first (bench.py's order) — a torch.cuda call that comes after several libptss contexts have come
    # and gone can fail with "No HIP GPUs are available". Harmless without a GPU.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
