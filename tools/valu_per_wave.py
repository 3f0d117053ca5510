"""tools/valu_per_wave.py <rocprofv3 --pmc output dir> — VALU / SALU / LDS instructions per wave of every kernel in a counter-collection run
(SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_WAVES, SQ_WAIT_INST_ANY, SQ_WAVE_CYCLES): what a code change did to the instruction
count of a kernel, e.g. between two builds (PTSS_LIBNAME) on the same workload."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][-70:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = max(c.get("SQ_WAVES", 0), 1)
    print("%-72s launches %5d waves %10d | per wave: VALU %8.1f SALU %7.1f LDS %7.1f | wait-inst / wave-cycles %.3f" % (
        k, n[k], w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w,
        c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 0), 1)))
