#!/bin/bash
# tools/stress_counters.sh — SQ / LDS counters of the bounce kernels on the configs[5] scene (tools/stress_bench.py),
# with the chunked sphere traversal and without it. Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in 1 0; do
  out=gpurun_out/stress_pmc_$mode; rm -rf $out ${out}b
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $out -o p -- python3 tools/stress_bench.py 1 3 $mode > $out.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d ${out}b -o p -- python3 tools/stress_bench.py 1 3 $mode > ${out}b.log 2>&1
  echo "== chunked sphere traversal: $mode"
  python3 tools/pmc_summary.py $out ${out}b 2>/dev/null | grep -A9 "bounceKernel<false" | grep -v "^--" | head -40
done
