#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... — short bench of libptss.so with VAR=v for each v
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline > gpurun_out/sweep_env.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/sweep_env.json")); r = d["roofline"]
print("$var=$v  %8.1f Mrays/s  %.4f ms/step  bounce %.1f us/launch" % (d["value"], d["ms_per_step"], r["avg_launch_us"]))
PY
done
