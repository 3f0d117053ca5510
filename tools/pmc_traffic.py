"""tools/pmc_traffic.py <gpurun_out/rNN> — HBM bytes per bounce-kernel launch from the FETCH_SIZE / WRITE_SIZE passes of
tools/profile_round.sh, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, confirmed by the copy-only
calibration build); writes profiles/pmc_traffic.json (read by bench.py into roofline.traffic)."""
import collections
import csv
import json
import os
import sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KiB = 1024
W, H = 1920, 1080
try:  # the PMC passes run bench.py's default workload; bench.json (written after this script) only confirms S
    SPP = json.load(open(os.path.join(src, "bench.json")))["config"]["samples_per_pass"]
except (OSError, KeyError, ValueError):
    SPP = 40


def avg(path, sub, counter):
    rows = [r for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    per = collections.defaultdict(float)
    for r in rows:
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return (sum(per.values()) / len(per), len(per)) if per else (0.0, 0)


# <kLast, kSceneInLds, kFirst, kAccel>: the default bench runs the LDS path without the many-sphere structure
kinds = {"first": "bounceKernel<false, true, true, false>", "mid": "bounceKernel<false, true, false, false>",
         "last": "bounceKernel<true, true, false, false>"}
out = {}
for tag, sub in kinds.items():
    f, nf = avg(os.path.join(src, "pmc_FETCH_SIZE/p_counter_collection.csv"), sub, "FETCH_SIZE")
    w, nw = avg(os.path.join(src, "pmc_WRITE_SIZE/p_counter_collection.csv"), sub, "WRITE_SIZE")
    out[tag] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "dispatches": nf, "hbm_read_bytes": 2 * f * KiB, "hbm_write_bytes": w * KiB}
cf, _ = avg(os.path.join(src, "cal_FETCH_SIZE/p_counter_collection.csv"), kinds["mid"], "FETCH_SIZE")
cw, _ = avg(os.path.join(src, "cal_WRITE_SIZE/p_counter_collection.csv"), kinds["mid"], "WRITE_SIZE")
true_bytes = W * H * SPP * 76
out["calibration_copy_only_build"] = {"known_bytes_each_way": true_bytes, "FETCH_SIZE_KiB": cf, "WRITE_SIZE_KiB": cw,
                                      "fetch_factor_needed": true_bytes / (cf * KiB) if cf else None,
                                      "write_factor_needed": true_bytes / (cw * KiB) if cw else None}
tot = lambda d: d["hbm_read_bytes"] + d["hbm_write_bytes"]
out["hbm_bytes_per_launch"] = round((tot(out["first"]) + 6 * tot(out["mid"]) + tot(out["last"])) / 8)
out["samples_per_pass"] = SPP
out["note"] = ("HBM bytes per bounce-kernel launch averaged over the 8 launches of a pass (bounce 0 + 6 mid + last), 1920x1080 'mixed', "
               "8 bounces, %d sample lanes. FETCH_SIZE doubled (gfx950 reports 1/2; the copy-only calibration build of this very kernel "
               "needs a factor %.3f), WRITE_SIZE as is (factor %.3f). Separate --pmc passes; source: profiles/%s/pmc_and_trace_summary.txt"
               % (SPP, out["calibration_copy_only_build"]["fetch_factor_needed"] or 0, out["calibration_copy_only_build"]["write_factor_needed"] or 0,
                  os.path.basename(src.rstrip("/"))))
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
