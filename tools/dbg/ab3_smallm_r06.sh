#!/bin/bash
# three library builds at the small launch shapes, interleaved: never-small variant, the previous dispatch, the shipped one
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for CFG in ${CFGS:-"--seqs-per-gpu 12" "--seqs-per-gpu 20" "--seqs-per-gpu 24"}; do
  for i in 1 2; do
    for which in nosmall prev shipped; do
      unset AMDSEG_LIB
      [ $which = nosmall ] && export AMDSEG_LIB=/root/repo/_ab/libamdseg_nosmall.so
      [ $which = prev ] && export AMDSEG_LIB=/root/repo/_ab/libamdseg_final.so
      python bench.py $CFG --steps 30 --warmup 8 --no-extra-legs --no-roofline --no-cpu-baseline --no-via-trainer 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG | $which $i |', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
    done
  done
done
