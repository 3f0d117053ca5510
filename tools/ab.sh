#!/bin/bash
# tools/ab.sh <out-dir> <bench args...> -- <tag> [<tag> ...] — interleaved A/B of lib/libptss_<tag>.so builds ("base" = libptss.so) on one box:
# every tag twice, round-robin, one bench.py line each (no CPU baseline, no S = 1 legs); prints Mrays/s and the bounce kernel's launch time.
out=$1; shift; args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
mkdir -p $out
for rep in 1 2; do for t in "$@"; do
  lib=libptss_$t.so; [ $t = base ] && lib=libptss.so
  PTSS_LIBNAME=$lib python bench.py "${args[@]}" --no-cpu-baseline --no-s1-leg > $out/$t.$rep.json 2> $out/$t.$rep.err
  python -c "import json;j=json.load(open('$out/$t.$rep.json'));print('$t',j['value'],j['roofline']['avg_launch_us'])"
done; done
