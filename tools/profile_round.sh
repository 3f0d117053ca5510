#!/bin/bash
# tools/profile_round.sh <tag> [config] [S] — the measurements a round commits under profiles/<tag>/<config>/:
#   1. rocprofv3 --kernel-trace --stats of a bench run                                    -> trace/ (kernel_stats.csv)
#   2. PMC passes, each its own run with --kernel-trace only (FETCH_SIZE / WRITE_SIZE cannot share a pass;
#      MI355X_MICROARCH.md): HBM bytes, SQ instruction / wave counters, LDS + cache counters
#   3. tools/pmc_counters.py -> profiles/pmc_counters.json["<config>_s<S>"] (bench.py reads it)
#   4. bench.py, default flags for that config, LAST, so that its roofline carries this run's PMC figures -> bench.json
# Run on the GPU box from the repo root:  tools/profile_round.sh r02 c3
set -e
tag=${1:-r02}; cfg=${2:-c3}; S=${3:-}
# the PMC passes and the trace run a handful of steps: pin S to the configuration's default (or the one asked for), or bench.py
# would pick another S for that step count and the counters would describe another launch size
[ -z "$S" ] && S=$(python3 -c "import bench; print(bench.CONFIGS['$cfg']['samples'])")
sarg="--samples-per-pass $S"
out=gpurun_out/$tag/$cfg
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
steps=${STEPS:-60}; psteps=${PMC_STEPS:-10}
quick="--config $cfg $sarg --no-cpu-baseline --no-s1-leg"
python3 bench.py $quick --steps 4 --warmup 1 > $out/bench.json 2> $out/bench.err   # live counts for pmc_counters.py (replaced below)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py $quick --steps $steps --warmup 5 > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py $quick --steps $psteps --warmup 2 --no-kernel-timing > $out/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -o p -- python3 bench.py $quick --steps $psteps --warmup 2 --no-kernel-timing > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_mem -o p -- python3 bench.py $quick --steps $psteps --warmup 2 --no-kernel-timing > $out/pmc_mem.log 2>&1
S_used=$(python3 -c "import json;print(json.load(open('$out/bench.json'))['config']['samples_per_pass'])")
python3 tools/pmc_summary.py $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq $out/pmc_mem > $out/summary.txt 2>&1
python3 tools/pmc_counters.py $out $cfg $S_used > $out/pmc_counters.log 2>&1
cp profiles/pmc_counters.json $out/pmc_counters.json
python3 bench.py --config $cfg > $out/bench.json 2> $out/bench.err   # the default run of that configuration (its default S is the one profiled)
cat $out/bench.json
