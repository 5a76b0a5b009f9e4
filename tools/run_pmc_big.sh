#!/bin/bash
# SQ counters of the GEMM / attention kernels stand-alone (tools/bench_kernels.py nt tn attn): separate --pmc passes, --kernel-trace only
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04}_pmc_big.txt
: > $OUT
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  rm -rf /tmp/pmc_x
  rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_x -o run -- python tools/bench_kernels.py nt tn attn > /dev/null 2> gpurun_out/pmc_big.err
  DB=$(find /tmp/pmc_x -name "*.db" | head -1)
  echo "## pass: --pmc $CTRS" >> $OUT
  python tools/pmc_summary.py "$DB" | grep -E "gemm_nt_dp|gemm_tn_dp|attn_fwd_kernel<8, false, false, true>|attn_bwd_dq_kernel<4, false, false, true>|attn_bwd_dkv_kernel<4, false, false, true>|^\| kernel|^\|---" >> $OUT
done
wc -l $OUT
