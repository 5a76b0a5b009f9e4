#!/bin/bash
# SQ counters of the row kernels (ln_bwd / add_ln_fwd) stand-alone: separate --pmc passes, --kernel-trace only
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04}_pmc_rows.txt
: > $OUT
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pmc_x
  rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_x -o run -- python tools/bench_kernels.py rows > /dev/null 2> gpurun_out/pmc_rows.err
  DB=$(find /tmp/pmc_x -name "*.db" | head -1)
  echo "## pass: --pmc $CTRS" >> $OUT
  python tools/pmc_summary.py "$DB" | grep -E "ln_bwd|add_ln_fwd|^\| kernel|^\|---" >> $OUT
done
cat $OUT
