#!/bin/bash
# timing-only sweep (ablation builds are wrong by construction): kernel Grays/s and us/launch
mkdir -p gpurun_out/sweep
for tag in "$@"; do
  lib=libptss_${tag}.so; [ "$tag" = base ] && lib=libptss.so
  PTSS_LIBNAME=$lib python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-s1-leg > gpurun_out/sweep/$tag.json 2> gpurun_out/sweep/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/sweep/$tag.json")); r = d["roofline"]
    print("%-8s %8.1f Mrays/s  %.4f ms/step  bounce %.1f us/launch  rays/launch %.0f  kernel %.2f Grays/s  ps/ray %.1f" % ("$tag", d["value"], d["ms_per_step"], r["avg_launch_us"], d["ray_bounces"]/r["launches"], r["kernel_grays_per_s"], 1e3/r["kernel_grays_per_s"]))
except Exception as e:
    print("$tag", "bench failed", e)
PY
done
