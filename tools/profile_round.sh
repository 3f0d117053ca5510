#!/bin/bash
# tools/profile_round.sh <tag> — the measurements a round commits under profiles/:
#   1. bench.py, default flags (the BASELINE workload), run LAST so that roofline.traffic is this run's PMC figure -> bench.json
#   2. rocprofv3 --kernel-trace --stats of a 200-step bench                     -> kernel_stats.csv
#   3. PMC passes (separate runs: FETCH_SIZE / WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md)
#      of the shipped build and of the copy-only calibration build (libptss_a15.so: reads and writes
#      every ray once, nothing else — a known byte count in this kernel's own access pattern)
# Run on the GPU box from the repo root:  tools/profile_round.sh r01
set -e
tag=${1:-r01}
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/pmc_$c.log 2>&1
  if [ -f cuda-path-tracer-ss_amd/lib/libptss_a15.so ]; then
    PTSS_LIBNAME=libptss_a15.so rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/cal_$c -o p -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/cal_$c.log 2>&1
  fi
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -o p -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_mem -o p -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/pmc_mem.log 2>&1
python3 tools/pmc_summary.py $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/cal_FETCH_SIZE $out/cal_WRITE_SIZE $out/pmc_sq $out/pmc_mem > $out/summary.txt 2>&1
python3 tools/pmc_traffic.py $out > $out/pmc_traffic.log 2>&1   # refreshes profiles/pmc_traffic.json, which bench.py reads
cp profiles/pmc_traffic.json $out/pmc_traffic.json
python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
