"""Builds A/B variants of libptss.so (tuning macros of csrc/ptss_device.h) as lib/libptss_<tag>.so.
Select one at run time with PTSS_LIBNAME=libptss_<tag>.so. Used only for measurements (profiles/)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ptss_build", os.path.join(ROOT, "cuda-path-tracer-ss_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

VARIANTS = {
    "w1": ["PTSS_MINWAVES=1"],
    "w4": ["PTSS_MINWAVES=4"],
    "w5": ["PTSS_MINWAVES=5"],
    "w6": ["PTSS_MINWAVES=6"],
    "w7": ["PTSS_MINWAVES=7"],
    "w8": ["PTSS_MINWAVES=8"],
    "w6d": ["PTSS_MINWAVES=6", "PTSS_DEFER_LOADS=1"],
    "w7d": ["PTSS_MINWAVES=7", "PTSS_DEFER_LOADS=1"],
    "w8d": ["PTSS_MINWAVES=8", "PTSS_DEFER_LOADS=1"],
    "b128": ["PTSS_BLOCK=128", "PTSS_SHARDS=32"],
    "b512": ["PTSS_BLOCK=512"],
    "s8": ["PTSS_SHARDS=8"],
    "s32": ["PTSS_SHARDS=32"],
    "stamps": ["PTSS_STAMPS=1"],
    "qhist": ["PTSS_QHIST=1"],
    "norgs": ["PTSS_REGROUP_SHADOW=0"],  # many-sphere scenes: shadow rays walk all their chunks lane by lane
    "warm2": ["PTSS_WARM=2"],
    "warm16": ["PTSS_WARM=16"],
    "norg": ["PTSS_REGROUP=0"],  # many-sphere scenes: every lane walks its own chunks in the closest hit too
    "ck4": ["PTSS_CHUNK=4"],
    "ck16": ["PTSS_CHUNK=16"],
    "chist": ["PTSS_CHIST=1"],
    "nosplit": ["PTSS_SPLIT_SPARSE=0"],
    "powq": ["PTSS_QUANT_TABLE=0"],  # literal clamp/pow/scale tone map
    "blockc": ["PTSS_WAVE_COMPACT=0"],
    "wc_s32": ["PTSS_SHARDS=32"],
    "wc_s64": ["PTSS_SHARDS=64"],
    "wc_b128": ["PTSS_BLOCK=128", "PTSS_SHARDS=32"],
    "wc_b64": ["PTSS_BLOCK=64", "PTSS_SHARDS=64"],
    # ablations (results are WRONG by construction; timing only)
    "a1": ["PTSS_ABLATE=1"],
    "a2": ["PTSS_ABLATE=2"],
    "a3": ["PTSS_ABLATE=3"],
    "a7": ["PTSS_ABLATE=7"],
    "a8": ["PTSS_ABLATE=8"],   # no finishPath (tone map, accumulate, park RNG)
    "a64": ["PTSS_ABLATE=64"],    # finishPath without the accumulator atomics (S > 1)
    "a128": ["PTSS_ABLATE=128"],  # finishPath without parking the RNG state
    "a15": ["PTSS_ABLATE=15"],
    "g2": ["PTSS_TRI_GUARD2=1"],  # triangle reciprocal with both range compares
}

if __name__ == "__main__":
    for tag in (sys.argv[1:] or VARIANTS):
        b.build_device(force=True, defines=VARIANTS[tag], name=f"libptss_{tag}.so")
