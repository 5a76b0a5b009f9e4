#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the attention + GEMM kernels for a secondary model (two separate --pmc passes, --kernel-trace only)
MODEL=${1:-longformer}; TAG=${2:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_pmc_${MODEL}.txt
: > $OUT
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${CTR}
  rocprofv3 --kernel-trace --pmc $CTR -d /tmp/pmc_${CTR} -o run -- python bench.py --model $MODEL --no-cpu-baseline --no-via-trainer --no-roofline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/${TAG}_pmc_${MODEL}_${CTR}.err
  DB=$(find /tmp/pmc_${CTR} -name "*.db" | head -1)
  echo "## pass: --pmc $CTR (python bench.py --model $MODEL --steps 2 --warmup 1)" >> $OUT
  python tools/pmc_summary.py "$DB" | head -${PMC_ROWS:-40} >> $OUT
done
cat $OUT
