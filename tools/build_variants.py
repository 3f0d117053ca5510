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
    "knobs": ["PTSS_TUNING_KNOBS=1"],   # reads PTSS_GRID_CAP / PTSS_SCENE_PATH from the environment (tools/sweep_env.sh)
    "lfork": ["PTSS_LANE_ALWAYS_FORK=1"],
    "fp": ["PTSS_FORCE_PAIRS=1"],   # the paired any-hit in every scene with two lights (what it costs configs[2]'s)
    "nopairs": ["PTSS_NEE_PAIRS=0"],   # one queue entry per shadow segment (before the paired any-hit)
    "cullstat": ["PTSS_CULLSTAT=1"],   # diagnostic: tools/cull_stat.py
    "pairstat": ["PTSS_CULLSTAT=1", "PTSS_PAIRSTAT=1"],   # diagnostic: tools/pair_stat.py
    "wb5": ["PTSS_MINWAVES_BOUNDED=5"],   # the bounded-geometry instantiations at 5 / 7 waves per SIMD
    "wb7": ["PTSS_MINWAVES_BOUNDED=7"],
    "wb5su7": ["PTSS_MINWAVES_BOUNDED=5", "PTSS_SPHERE_UNROLL=7"],
    "wb6su7": ["PTSS_SPHERE_UNROLL=7"],
    "cr1": ["PTSS_CLASS_RANK=1"],   # survivors ranked by material class inside the wave
    "cr2": ["PTSS_CLASS_RANK=2"],   # ... by direction octant
    "f7": ["PTSS_MINWAVES_FIRST=7"],   # bounce 0 at 6 waves per SIMD (80 VGPRs)
    "f5": ["PTSS_MINWAVES_FIRST=5"],
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
    "warm0": ["PTSS_WARM=0"],
    "warm1": ["PTSS_WARM=1"],
    "warm2": ["PTSS_WARM=2"],
    "warm3": ["PTSS_WARM=3"],
    "warm4": ["PTSS_WARM=4"],
    "warm16": ["PTSS_WARM=16"],
    "norg": ["PTSS_REGROUP=0"],  # many-sphere scenes: every lane walks its own chunks in the closest hit too
    "ck4": ["PTSS_CHUNK=4"],
    "ck8": ["PTSS_CHUNK=8"],
    "ck32": ["PTSS_CHUNK=32"],
    "chist": ["PTSS_CHIST=1"],
    "shist": ["PTSS_SHIST=1"],   # scatter(): waves and lanes per block (tools/scatter_hist.py)
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
    "a4": ["PTSS_ABLATE=4"],   # no scatter
    "a7": ["PTSS_ABLATE=7"],
    "a8": ["PTSS_ABLATE=8"],   # no finishPath (tone map, accumulate, park RNG)
    "a64": ["PTSS_ABLATE=64"],    # finishPath without the accumulator atomics (S > 1)
    "a128": ["PTSS_ABLATE=128"],  # finishPath without parking the RNG state
    "a15": ["PTSS_ABLATE=15"],
    "a16": ["PTSS_ABLATE=16"],   # at most two sphere candidates per lane resolved (closest hit, dense any-hit)
    "g2": ["PTSS_TRI_GUARD2=1"],
    "r1": ["PTSS_SPHERE_UNROLL=0", "PTSS_TRI_STRAIGHT=0"],   # the round-1 loops
    "ts2": ["PTSS_TRI_STRAIGHT=2"],                      # one exit kept, bare reciprocal, min3, selects
    "nofs": ["PTSS_FRESNEL_SKIP=0"],
    "ts2nofs": ["PTSS_TRI_STRAIGHT=2", "PTSS_FRESNEL_SKIP=0"],
    "nots": ["PTSS_TRI_STRAIGHT=0"],                     # closest-hit triangle loop with wave-uniform exits
    "row128": ["PTSS_ROW128=1"],                         # scene rows as 16-byte fetches (ds_read_b128)
    "su25r": ["PTSS_SPHERE_UNROLL=25"],
    "nosu": ["PTSS_SPHERE_UNROLL=0"],
    "su1": ["PTSS_SPHERE_UNROLL=1"],    # closest hit: four spheres per trip
    "su9": ["PTSS_SPHERE_UNROLL=9"],    # + dense shadow passes: two per trip
    "su25": ["PTSS_SPHERE_UNROLL=25"],  # + lane-split shadow passes: two per trip
    "su7": ["PTSS_SPHERE_UNROLL=7"],    # four per trip everywhere (spills)  # sphere candidate masks one sphere per trip (the round-1 loop)  # triangle reciprocal with both range compares
}

if __name__ == "__main__":
    for tag in (sys.argv[1:] or VARIANTS):
        b.build_device(force=True, defines=VARIANTS[tag], name=f"libptss_{tag}.so")
