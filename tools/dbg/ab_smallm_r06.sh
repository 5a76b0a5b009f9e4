#!/bin/bash
# same-box A/B of two library builds at the small launch shapes
BASE=${1:-/root/repo/_ab/libamdseg_nosmall.so}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== kernel level, variant"; AMDSEG_LIB=$BASE python tools/dbg/nt_addres_shapes.py 2>/dev/null
echo "== kernel level, shipped"; python tools/dbg/nt_addres_shapes.py 2>/dev/null
for CFG in "--seqs-per-gpu 8" "--model longformer --seq-len 2048 --seqs-per-gpu 4" "--seqs-per-gpu 16"; do
  for i in 1 2; do
    for which in variant shipped; do
      if [ $which = variant ]; then export AMDSEG_LIB=$BASE; else unset AMDSEG_LIB; fi
      python bench.py $CFG --steps 30 --warmup 8 --no-extra-legs --no-roofline --no-cpu-baseline --no-via-trainer 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG | $which $i |', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
    done
  done
done
