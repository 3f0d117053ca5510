#!/bin/bash
# tools/kernel_times.sh <tag> [<tag> ...] — per-instantiation bounce-kernel times (rocprofv3 --kernel-trace --stats) of
# lib/libptss_<tag>.so ("base" = libptss.so), 30 bench steps each. Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  lib=libptss_${tag}.so; [ "$tag" = base ] && lib=libptss.so
  out=gpurun_out/kt_$tag; rm -rf $out
  PTSS_LIBNAME=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py ${BENCH_ARGS:-} --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-s1-leg --no-kernel-timing > $out.log 2>&1
  echo "== $tag"
  python3 - <<PY
import csv
for r in csv.DictReader(open("$out/t_kernel_stats.csv")):
    if "bounceKernel" in r["Name"]:
        n = r["Name"]; k = n[n.index("<"):n.index(">")+1]
        print("  %-22s calls %5s  avg %9.1f us" % (k, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
