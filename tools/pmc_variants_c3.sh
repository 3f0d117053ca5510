#!/bin/bash
# tools/pmc_variants_c3.sh — VALU / SALU / LDS instructions per wave (= per 64-ray tile... a workgroup of four waves traces 256 rays) of the
# c3 workload under the shipped build and the ablation builds a1 (no NEE), a2 (no closest-hit loops), a4 (no scatter), a8 (no finishPath):
# where the mid-bounce kernel's instructions go. Needs: python tools/build_variants.py a1 a2 a4 a8
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc3
for t in base a1 a2 a4 a8; do
  lib=libptss_$t.so; [ $t = base ] && lib=libptss.so
  export PTSS_LIBNAME=$lib
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc3/$t -o p -- python3 bench.py --config c3 --samples-per-pass 40 --no-cpu-baseline --no-s1-leg --steps 4 --warmup 1 --no-kernel-timing > gpurun_out/pmc3/$t.log 2>&1
  echo == $t; python3 tools/valu_per_wave.py gpurun_out/pmc3/$t > gpurun_out/pmc3/$t.txt; sed -n 1,4p gpurun_out/pmc3/$t.txt
done
