#!/bin/bash
# tools/ab_split.sh — interleaved A/B of the class split (knobs build: PTSS_CLASS_SPLIT = 0 off, 1 from bounce 0, 16 from bounce 1)
out=gpurun_out/split; mkdir -p $out
run() { # tag mode benchargs...
  tag=$1; sp=$2; shift 2
  PTSS_LIBNAME=libptss_knobs.so PTSS_CLASS_SPLIT=$sp python bench.py "$@" --no-cpu-baseline > $out/$tag.json 2> $out/$tag.err
  python -c "import json;j=json.load(open('$out/$tag.json'));print('$tag',j['value'],j['roofline']['avg_launch_us'],j.get('s1_mrays_per_s'),j.get('s1_free_running_lanes_mrays_per_s'))"
}
for rep in 1 2; do for m in ${MODES:-0 1 16}; do run c3_$m.$rep $m; done; done
for rep in 1 2; do for m in ${MODES:-0 1 16}; do run c5_$m.$rep $m --config c5; done; done
