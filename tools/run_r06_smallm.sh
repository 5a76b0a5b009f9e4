#!/bin/bash
# round 6: where the small-M launch shapes lose: rocprofv3 kernel tables of bert-base 8 x 512 and longformer 4 x 2048
TAG=${1:-r06_smallm}
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "bert8:--seqs-per-gpu 8 --steps 40 --warmup 10:60" "lf2048:--model longformer --seq-len 2048 --seqs-per-gpu 4 --steps 20 --warmup 5:35"; do
  name=${cfg%%:*}; rest=${cfg#*:}; flags=${rest%:*}; steps=${rest##*:}
  rm -rf /tmp/prof_${TAG}_${name}
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_${name} -o run -- python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs $flags > gpurun_out/${TAG}_${name}_stdout.json 2> /dev/null
  DB=$(find /tmp/prof_${TAG}_${name} -name "*.db" | head -1)
  python tools/prof_summary.py "$DB" gpurun_out/${TAG}_${name} $steps > /dev/null
  python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs $flags 2>/dev/null | tail -1 | cut -c1-300
  head -30 gpurun_out/${TAG}_${name}_kernel_stats.md | cut -c1-150
done
