mkdir -p gpurun_out/ab
for i in 1 2 3; do
 for tag in slean base; do
  lib=libptss_${tag}.so; [ "$tag" = base ] && lib=libptss.so
  for cfg in c3 c2 c5; do
  PTSS_LIBNAME=$lib timeout -k 10 120 python bench.py --config $cfg --steps 60 --warmup 6 --no-cpu-baseline --no-s1-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag $cfg %.1f' % d['value'])" || exit 1
  done
 done
done
