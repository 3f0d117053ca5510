"""pmc_summary.py — per-kernel averages from rocprofv3 --pmc / --kernel-trace CSV output directories.
usage: python tools/pmc_summary.py <dir-with-*_counter_collection.csv-or-*_kernel_trace.csv> [...]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("ptss::", "")
    return n[-48:]


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
            rows = list(csv.DictReader(open(f)))
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            disp = collections.defaultdict(set)
            for r in rows:
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
            print(f"== {f}")
            for k, v in agg.items():
                n = len(disp[k])
                print(f"  {k}  dispatches={n}")
                for c, x in sorted(v.items()):
                    print(f"      {c:28s} {x / n:16.1f} per dispatch")
        for f in sorted(glob.glob(os.path.join(d, "*_kernel_trace.csv"))):
            rows = list(csv.DictReader(open(f)))
            if not rows:
                continue
            per = collections.defaultdict(list)
            for r in rows:
                per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                     r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size")))
            print(f"== {f}")
            for k, v in per.items():
                durs = [x[1] for x in v]
                print(f"  {k:48s} n={len(v):5d} avg={sum(durs) / len(durs) / 1e3:9.2f} us  min={min(durs) / 1e3:8.2f} max={max(durs) / 1e3:8.2f}"
                      f"  vgpr={v[0][2]} lds={v[0][3]}")


if __name__ == "__main__":
    main()
