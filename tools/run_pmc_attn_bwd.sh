#!/bin/bash
# SQ counters of the attention backward kernels (two-kernel form and the merged one) on tools/attn_bwd_bench.py: separate --pmc passes, --kernel-trace only
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r05}_pmc_attn_bwd.txt
: > $OUT
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_x
  rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_x -o run -- python tools/attn_bwd_bench.py > /dev/null 2> gpurun_out/pmc_attn_bwd.err
  DB=$(find /tmp/pmc_x -name "*.db" | head -1)
  echo "## pass: --pmc $CTRS" >> $OUT
  python tools/pmc_summary.py "$DB" | grep -E "attn_bwd|^\| kernel|^\|---" | sed 's/AttnArgs.*` |/` |/' >> $OUT
done
cat $OUT
