python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -4
F="--no-cpu-baseline --no-via-trainer --no-extra-legs"
for i in 1 2; do
 for which in base new; do
  if [ $which = base ]; then export AMDSEG_LIB=/root/repo/_ab/libamdseg_base13.so; else unset AMDSEG_LIB; fi
  for cfg in "--seqs-per-gpu 8 --steps 40 --warmup 10" "--model longformer --seq-len 2048 --seqs-per-gpu 4 --steps 20 --warmup 5" "--seqs-per-gpu 16 --steps 40 --warmup 10"; do
    python bench.py $F $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which $i [$cfg]', d['value'], d['ms_per_step'])"
  done
 done
done
