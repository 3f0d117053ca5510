"""Builds A/B variants of libptss.so (the eight switches of csrc/ptss_device.h; extra ad-hoc -D flags: `python tools/build_variants.py tag=NAME=V,NAME2=V2`) as lib/libptss_<tag>.so.
Select one at run time with PTSS_LIBNAME=libptss_<tag>.so. Used only for measurements (profiles/)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ptss_build", os.path.join(ROOT, "cuda-path-tracer-ss_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

VARIANTS = {
    "knobs": ["PTSS_TUNING_KNOBS=1"],   # reads PTSS_GRID_CAP / PTSS_SCENE_PATH from the environment (tools/sweep_env.sh)
    # diagnostic counters (csrc/ptss_diag.h)
    "chist": ["PTSS_DIAG=1"],      # tools/candidate_hist.py
    "shist": ["PTSS_DIAG=2"],      # tools/scatter_hist.py
    "cullstat": ["PTSS_DIAG=4"],   # tools/cull_stat.py
    "pairstat": ["PTSS_DIAG=8"],   # tools/pair_stat.py
    "qhist": ["PTSS_DIAG=16"],     # tools/queue_hist.py
    # register budgets
    "wb5": ["PTSS_MINWAVES_BOUNDED=5"],
    "wb7": ["PTSS_MINWAVES_BOUNDED=7"],
    "f7": ["PTSS_MINWAVES_FIRST=7"],
    "f5": ["PTSS_MINWAVES_FIRST=5"],
    "w6": ["PTSS_MINWAVES=6"],
    "w8": ["PTSS_MINWAVES=8"],
    # geometry of tiles, shards, chunks
    "b128": ["PTSS_BLOCK=128", "PTSS_SHARDS=32"],
    "s8": ["PTSS_SHARDS=8"],
    "s32": ["PTSS_SHARDS=32"],
    "ck8": ["PTSS_CHUNK=8"],
    "ck32": ["PTSS_CHUNK=32"],
    # ablations (results are WRONG by construction; timing only)
    "a1": ["PTSS_ABLATE=1"],   # no NEE
    "a2": ["PTSS_ABLATE=2"],   # no closest-hit loops
    "a3": ["PTSS_ABLATE=3"],
    "a4": ["PTSS_ABLATE=4"],   # no scatter
    "a7": ["PTSS_ABLATE=7"],
    "a8": ["PTSS_ABLATE=8"],   # no finishPath (tone map, accumulate, park RNG)
    "a15": ["PTSS_ABLATE=15"],  # copy only
}

if __name__ == "__main__":
    for tag in (sys.argv[1:] or VARIANTS):
        if "=" in tag:   # ad-hoc: tag=NAME=V,NAME2=V2 (switches of an experiment in progress)
            tag, _, defs = tag.partition("=")
            defines = defs.split(",")
        else:
            defines = VARIANTS[tag]
        b.build_device(force=True, defines=defines, name=f"libptss_{tag}.so")
