"""tools/pmc_counters.py <gpurun_out/tag> <config> <S> — per-launch averages of the bounce kernels from the separate
rocprofv3 --pmc passes of tools/profile_round.sh, merged into profiles/pmc_counters.json under the key "<config>_s<S>"
(read by bench.py into roofline.traffic and roofline.valu):
  hbm_bytes_per_launch   FETCH_SIZE x 2 (gfx950 reports half of wide reads, MI355X_MICROARCH.md; the copy-only
                         calibration build of this very kernel confirmed 1.998 in round 1) + WRITE_SIZE, KiB -> bytes
  valu_insts_per_launch  SQ_INSTS_VALU (wave-level instructions)
  valu_lanes_active      SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)
  valu_insts_per_tile_mid_bounce   SQ_INSTS_VALU of the mid-bounce instantiation / (its rays / 64), rays from bench.json's live counts
"per launch" = averaged over every bounce-kernel dispatch of the profiled run (all bounces alike, as bench.py's HIP-event
average is)."""
import collections
import csv
import json
import os
import sys

src, config, S = sys.argv[1], sys.argv[2], int(sys.argv[3])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KiB = 1024


def per_dispatch(path, counter, sub="bounceKernel"):
    per = collections.defaultdict(float)
    name = {}
    if not os.path.exists(path):
        return per, name
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    return per, name


def mean(d):
    return sum(d.values()) / len(d) if d else None


fetch, _ = per_dispatch(os.path.join(src, "pmc_FETCH_SIZE/p_counter_collection.csv"), "FETCH_SIZE")
write, _ = per_dispatch(os.path.join(src, "pmc_WRITE_SIZE/p_counter_collection.csv"), "WRITE_SIZE")
insts, names = per_dispatch(os.path.join(src, "pmc_sq/p_counter_collection.csv"), "SQ_INSTS_VALU")
thr, _ = per_dispatch(os.path.join(src, "pmc_sq/p_counter_collection.csv"), "SQ_THREAD_CYCLES_VALU")
entry = {"dispatches": len(insts)}
if fetch and write:
    entry["hbm_read_bytes_per_launch"] = round(2 * mean(fetch) * KiB)
    entry["hbm_write_bytes_per_launch"] = round(mean(write) * KiB)
    entry["hbm_bytes_per_launch"] = entry["hbm_read_bytes_per_launch"] + entry["hbm_write_bytes_per_launch"]
if insts:
    entry["valu_insts_per_launch"] = round(mean(insts))
    entry["valu_lanes_active"] = round(sum(thr.values()) / (64.0 * sum(insts.values())), 4)
    # mid-bounce instantiation: <kLast = false, ..., kFirst = false, ...>
    import re

    def targs(n):  # <kLast, kSceneInLds, kFirst, kAccel, kBounded>
        m = re.search(r"bounceKernel<(\w+), (\w+), (\w+), (\w+)(?:, (\w+))?(?:, (\w+))?>", n)
        return m.groups() if m else None
    mid = {k: v for k, v in insts.items() if targs(names[k]) and targs(names[k])[0] == "false" and targs(names[k])[2] == "false"}
    try:
        b = json.load(open(os.path.join(src, "bench.json")))
        live = b["live_counts"]
        rays_mid = sum(live[1:-1]) / max(len(live) - 2, 1)
        if mid and rays_mid:
            entry["valu_insts_per_tile_mid_bounce"] = round(mean(mid) / (rays_mid / 64.0))
            midthr = {k: thr[k] for k in mid}
            entry["valu_lanes_active_mid_bounce"] = round(sum(midthr.values()) / (64.0 * sum(mid.values())), 4)
    except (OSError, KeyError, ValueError):
        pass
path = os.path.join(ROOT, "profiles", "pmc_counters.json")
try:
    allc = json.load(open(path))
except (OSError, ValueError):
    allc = {}
allc[f"{config}_s{S}"] = entry
allc["source"] = "profiles/%s/<config>/ (tools/profile_round.sh, separate rocprofv3 --pmc passes)" % os.path.basename(os.path.dirname(src.rstrip("/")))
json.dump(allc, open(path, "w"), indent=1)
print(json.dumps({f"{config}_s{S}": entry}, indent=1))
