#!/bin/bash
# VALU/SALU wave-instruction counts per bounce launch for ablation builds (which phases cost what)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  lib=libptss_${tag}.so; [ "$tag" = base ] && lib=libptss.so
  rm -rf gpurun_out/pv_$tag
  PTSS_LIBNAME=$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 --output-format csv -d gpurun_out/pv_$tag -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-s1-leg --no-kernel-timing > gpurun_out/pv_$tag.log 2>&1
  echo "== $tag"; python3 tools/pmc_summary.py gpurun_out/pv_$tag | grep -A9 "bounceKernel<false, true, false, false, true, false>" | grep -v "^--"
done
