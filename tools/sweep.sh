#!/bin/bash
# tools/sweep.sh <tag> [<tag> ...] — parity smoke + short bench for lib/libptss_<tag>.so variants ("base" = libptss.so)
mkdir -p gpurun_out/sweep
for tag in "$@"; do
  lib=libptss_${tag}.so; [ "$tag" = base ] && lib=libptss.so
  if PTSS_LIBNAME=$lib python __graft_entry__.py smoke > gpurun_out/sweep/$tag.smoke 2>&1; then ok=parity-ok; else ok=PARITY-FAIL; fi
  PTSS_LIBNAME=$lib python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-s1-leg > gpurun_out/sweep/$tag.json 2> gpurun_out/sweep/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/sweep/$tag.json"))
    print("%-8s %s  %8.1f Mrays/s  %.4f ms/step  bounce %.1f us/launch" % ("$tag", "$ok", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$tag", "$ok", "bench failed", e)
PY
done
