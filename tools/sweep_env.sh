#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... — short bench of the "knobs" build (tools/build_variants.py knobs: the shipped libptss.so
# reads no environment) with VAR=v for each v
var=$1; shift
for v in "$@"; do
  env PTSS_LIBNAME=libptss_knobs.so $var=$v python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-s1-leg > gpurun_out/sweep_env.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sweep_env.json")); r = d["roofline"]
print("$var=$v  %8.1f Mrays/s  %.4f ms/step  bounce %.1f us/launch" % (d["value"], d["ms_per_step"], r["avg_launch_us"]))
PY
done
