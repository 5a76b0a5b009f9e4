#!/bin/bash
# same-box A/B of two library builds on the 8 x 4096 models (and the 4 x 2048 launch shape): bench ms/step, interleaved
# usage: tools/dbg/ab_models_r06.sh <base.so>
BASE=${1:-/root/repo/_ab/libamdseg_base14.so}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for CFG in "--model longformer" "--model ponet" "--model bigbird" "--model longformer --seq-len 2048 --seqs-per-gpu 4"; do
  for i in 1 2; do
    for which in base new; do
      if [ $which = base ]; then export AMDSEG_LIB=$BASE; else unset AMDSEG_LIB; fi
      python bench.py $CFG --steps 20 --warmup 5 --no-extra-legs --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG | $which $i |', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
    done
  done
done
