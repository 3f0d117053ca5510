"""tools/asm_build.py <tag> [--edit none|cnd64] [-DNAME=V ...] — builds lib/libptss_<tag>.so with the bounce kernels' gfx950
assembly passed through a text edit between the compiler and the assembler (measurement builds: what would a different
instruction choice be worth?).  edit=cnd64: VOP2 `v_cndmask_b32_e32 d, a, b, vcc` -> VOP3 `v_cndmask_b32_e64 d, a, b, vcc`.
Pipeline = hipcc's own (hipcc -###): device cc1 -> .s -> [edit] -> assembler -> lld -> clang-offload-bundler -> host cc1
with -fcuda-include-gpubinary -> link with the ordinary object of ptss_api.hip."""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ptss_build", os.path.join(ROOT, "cuda-path-tracer-ss_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
LLVM = "/opt/rocm/lib/llvm/bin"


def run(cmd):
    print("+", " ".join(cmd)[:200], flush=True)
    subprocess.check_call(cmd)


def edit_cnd64(text):
    n = 0
    out = []
    pat = re.compile(r"^(\s*)v_cndmask_b32_e32 (v\d+), ([^,]+), (v\d+), vcc\s*$")
    for line in text.split("\n"):
        m = pat.match(line)
        if m:
            out.append(f"{m.group(1)}v_cndmask_b32_e64 {m.group(2)}, {m.group(3)}, {m.group(4)}, vcc")
            n += 1
        else:
            out.append(line)
    print(f"cnd64: {n} instructions re-encoded", flush=True)
    return "\n".join(out)


def main():
    tag = sys.argv[1]
    edit = "none"
    defs = []
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--edit":
            edit = args.pop(0)
        else:
            defs.append(a)
    tmp = os.path.join("/tmp", "asm_build_" + tag)
    os.makedirs(tmp, exist_ok=True)
    inc = ["-I", b.INC, "-I", b.CSRC]
    kern = os.path.join(b.CSRC, "ptss_kernels.hip")
    api = os.path.join(b.CSRC, "ptss_api.hip")
    s = os.path.join(tmp, "k.s")
    run(["hipcc"] + b.HIP_FLAGS + defs + inc + ["--cuda-device-only", "-S", kern, "-o", s])
    text = open(s).read()
    if edit == "cnd64":
        text = edit_cnd64(text)
    open(s, "w").write(text)
    run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", os.path.join(tmp, "k.o")])
    run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", os.path.join(tmp, "k.out"), os.path.join(tmp, "k.o")])
    run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", "-input=" + os.path.join(tmp, "k.out"), "-output=" + os.path.join(tmp, "k.hipfb")])
    run(["hipcc"] + b.HIP_FLAGS + defs + inc + ["--cuda-host-only", "-c", kern, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", os.path.join(tmp, "k.hipfb"),
                                               "-o", os.path.join(tmp, "k_host.o")])
    run(["hipcc"] + b.HIP_FLAGS + defs + inc + ["-c", api, "-o", os.path.join(tmp, "api.o")])
    out = os.path.join(b.LIBDIR, f"libptss_{tag}.so")
    run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", os.path.join(tmp, "k_host.o"), os.path.join(tmp, "api.o"), "-o", out])
    print(out)


if __name__ == "__main__":
    main()
