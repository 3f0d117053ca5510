"""Builds every native artefact of the repo, in-tree.

  libptss.so        hipcc --offload-arch=gfx950   csrc/*.hip        the product (kernels + C-ABI)
  libptss_host.so   g++                            host/*.cpp        host mirror (Scene, camera, TGA, probes)
  ptss_main         g++ (host only) + libptss      host/main.cpp     headless drop-in of the reference's main(); --gpus N: one
                    + librccl                                       context per GPU, one ncclGather (host/MultiGpu.cpp)
(The CPU oracle — test infrastructure — has its own recipe, oracle/build.py; nothing here refers to it.)

Both sides of the parity contract are compiled with -ffp-contract=off and without fast-math
(DESIGN.md "Parity"); the CPU objects use -mfma -mavx2 so that ptm::fma is one correctly rounded
instruction (any x86-64 server since 2013; the GPU box's host qualifies).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIBDIR = os.path.join(HERE, "lib")

FP_FLAGS = ["-ffp-contract=off", "-fno-fast-math"]
CPU_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-mfma", "-mavx2", "-Wall", "-Wno-unused-function"] + FP_FLAGS
# -fno-slp-vectorize -fno-vectorize: left alone, hipcc pairs independent float operations into v_pk_{mul,add,fma}_f32
# (about 190 of them in one bounce kernel, plus the v_mov shuffles that build the register pairs). On gfx950 a packed
# op issues in ~5.3 SIMD cycles against ~3.3 for each of the two VOP2 ops it replaces (tools/microbench/pk_rate.hip),
# so the pairing is at best neutral and the shuffles make it a loss: -3 % on the bounce kernel (profiles/README.md).
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-fno-vectorize", "-Wall",
             "-Wno-unused-function"] + FP_FLAGS


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _headers():
    hs = []
    for d in (INC, CSRC, HOST):
        if os.path.isdir(d):
            hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    return hs


def build_host(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "libptss_host.so")
    srcs = [os.path.join(HOST, f) for f in ("Scene.cpp", "HostOps.cpp", "host_capi.cpp")]
    if force or _newer(out, srcs + _headers()):
        _run(["g++"] + CPU_FLAGS + ["-shared", "-I", INC, "-I", CSRC, "-I", HOST] + srcs + ["-o", out])
    return out


def build_device(force=False, defines=(), name="libptss.so"):
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, name)
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if force or _newer(out, srcs + _headers()):
        extra = os.environ.get("PTSS_EXTRA_HIPFLAGS", "").split()  # A/B measurements only (tools/build_variants.py)
        _run([hipcc] + HIP_FLAGS + extra + ["-D" + d for d in defines] + ["-shared", "-I", INC, "-I", CSRC] + srcs + ["-o", out])
    return out


def build_main(force=False):
    """ptss_main: the headless twin of the reference's executable (host C++ only; links the C-ABI + HIP runtime)."""
    out = os.path.join(LIBDIR, "ptss_main")
    srcs = [os.path.join(HOST, f) for f in ("main.cpp", "CudaTracer.cpp", "MultiGpu.cpp", "Scene.cpp", "HostOps.cpp")]
    if force or _newer(out, srcs + _headers() + [os.path.join(LIBDIR, "libptss.so")]):
        _run(["g++"] + CPU_FLAGS + ["-D__HIP_PLATFORM_AMD__", "-I", INC, "-I", CSRC, "-I", HOST, "-I", "/opt/rocm/include"] + srcs +
             ["-L", LIBDIR, "-lptss", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib",
              "-o", out])
    return out


def build_mathcheck(force=False):
    """ptss_mathcheck: exhaustive float32 check of the device fast paths of ptm::rcp / ptm::sqrt (tests/test_gpu_math.py)."""
    out = os.path.join(LIBDIR, "ptss_mathcheck")
    src = os.path.join(ROOT, "tests", "csrc", "math_exhaustive.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if force or _newer(out, [src] + _headers()):
        _run([hipcc] + [f for f in HIP_FLAGS if f != "-fPIC"] + ["-I", INC, "-I", CSRC, src, "-o", out])
    return out


def build_all(force=False):
    return [build_host(force), build_device(force), build_main(force), build_mathcheck(force)]


if __name__ == "__main__":
    force = "--force" in sys.argv
    what = [a for a in sys.argv[1:] if not a.startswith("-")]
    table = {"host": build_host, "device": build_device, "main": build_main,
             "mathcheck": build_mathcheck}
    if not what:
        build_all(force)
    for w in what:
        table[w](force)
