#!/bin/bash
# tools/split_trace.sh — per-bounce kernel durations of one pass with the class split off / on (knobs build; rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sp in ${MODES:-0 1}; do
  out=gpurun_out/split/trace_$sp; rm -rf $out
  PTSS_LIBNAME=${LIB:-libptss_knobs.so} PTSS_CLASS_SPLIT=$sp rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py ${BENCH_ARGS:-} --steps 6 --warmup 2 --no-cpu-baseline --no-s1-leg --no-kernel-timing > $out.log 2>&1
  echo "== split $sp"
  python3 - <<PY
import csv, glob
f = glob.glob("$out/**/t_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bounceKernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nb = ${NB:-8}
reps = 4
last = rows[-nb * reps:]
tot = 0.0
for i in range(nb):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last[i::nb])
    med = (d[1] + d[2]) / 2
    tot += med
    print("  bounce %2d  %9.1f us (median of %d passes)" % (i, med, reps))
print("  sum        %9.1f us" % tot)
PY
done
