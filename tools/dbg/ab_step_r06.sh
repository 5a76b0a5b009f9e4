#!/bin/bash
# round 6: same-box A/B of two library builds in the training step: bench ms/step interleaved, then one rocprofv3 kernel table each
# usage: tools/dbg/ab_step_r06.sh <tag> [base.so]
TAG=${1:-ab}; BASE=${2:-/root/repo/_ab/libamdseg_base.so}
mkdir -p gpurun_out; export TMPDIR=/tmp
F="--no-cpu-baseline --no-via-trainer --no-extra-legs --steps 60 --warmup 10"
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export AMDSEG_LIB=$BASE; else unset AMDSEG_LIB; fi
    python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$which $i', d['value'], d['ms_per_step'], 'nt', r['avg_launch_us'], r['frac'], 'enc', r.get('encoder_gemms',{}).get('frac'), r.get('encoder_gemms_frac_executed'))"
  done
done
for which in base new; do
  if [ $which = base ]; then export AMDSEG_LIB=$BASE; else unset AMDSEG_LIB; fi
  rm -rf /tmp/prof_${TAG}_$which
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$which -o run -- python bench.py $F > /dev/null 2>&1
  DB=$(find /tmp/prof_${TAG}_$which -name "*.db" | head -1)
  python tools/prof_summary.py "$DB" gpurun_out/${TAG}_$which 80 > /dev/null
  echo "== $which"; head -20 gpurun_out/${TAG}_${which}_kernel_stats.md | tail -16 | cut -c1-120
done
