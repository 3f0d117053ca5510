"""oracle/build.py — builds the CPU oracle (TEST INFRASTRUCTURE; nothing under cuda-path-tracer-ss_amd/ refers to it).

  _build/liboracle.so        oracle.cpp against the shared math header csrc/ptmath.h, -ffp-contract=off -fno-fast-math:
                             the bit-exact checker of the parity tests, and bench.py's cpu_baseline ("port")
  _build/liboracle_libm.so   the same source with -DORACLE_LIBM_MATH: libm float functions and plain vector arithmetic
                             (oracle/libm_math.h) — shares no arithmetic with the product; tests/test_oracle_libm.py

oracle/_ref/ (the reference itself compiled from /root/reference) does not exist: the reference needs nvcc, cuRAND,
Thrust, glm and GLUT/GLEW, none of which this image holds."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "cuda-path-tracer-ss_amd", "csrc")   # ptmath.h only: the arithmetic both sides pin
OUT = os.path.join(HERE, "_build")
BASE = ["-O2", "-std=c++17", "-fPIC", "-mfma", "-mavx2", "-Wall", "-Wno-unused-function", "-fopenmp", "-shared"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources():
    hs = [os.path.join(d, f) for d in (INC, CSRC, HERE) if os.path.isdir(d) for f in os.listdir(d) if f.endswith(".h")]
    return [os.path.join(HERE, "oracle.cpp")] + hs


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_oracle(force=False):
    os.makedirs(OUT, exist_ok=True)
    out = os.path.join(OUT, "liboracle.so")
    if force or _newer(out, _sources()):
        _run(["g++"] + BASE + ["-ffp-contract=off", "-fno-fast-math", "-I", INC, "-I", CSRC, os.path.join(HERE, "oracle.cpp"), "-o", out])
    return out


def build_oracle_libm(force=False):
    os.makedirs(OUT, exist_ok=True)
    out = os.path.join(OUT, "liboracle_libm.so")
    if force or _newer(out, _sources()):
        _run(["g++"] + BASE + ["-DORACLE_LIBM_MATH", "-I", INC, "-I", HERE, os.path.join(HERE, "oracle.cpp"), "-o", out])
    return out


def build_all(force=False):
    return [build_oracle(force), build_oracle_libm(force)]


if __name__ == "__main__":
    build_all("--force" in sys.argv)
