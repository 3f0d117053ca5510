set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc5
for t in base a2 a1; do
  lib=libptss_$t.so; [ $t = base ] && lib=libptss.so
  export PTSS_LIBNAME=$lib
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc5/$t -o p -- python3 bench.py --config c5 --samples-per-pass 16 --no-cpu-baseline --no-s1-leg --steps 4 --warmup 1 --no-kernel-timing > gpurun_out/pmc5/$t.log 2>&1
  echo == $t; python3 tools/valu_per_wave.py gpurun_out/pmc5/$t | head -4
done
