for rep in 1 2; do
for v in "" "AMDSEG_OVERLAP_WGRAD=1"; do
  echo "== bert 8x512 $v"; env $v python bench.py --seqs-per-gpu 8 --steps 60 --warmup 15 --no-cpu-baseline --no-via-trainer --no-extra-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== longformer 4x2048 $v"; env $v python bench.py --model longformer --seq-len 2048 --seqs-per-gpu 4 --steps 30 --warmup 8 --no-cpu-baseline --no-via-trainer --no-extra-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
