#!/bin/bash
# secondary configurations of the bench in one GPU-box session -> gpurun_out/<tag>_models.txt (one JSON line each, trimmed)
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_models.txt
: > $OUT
run() { echo "## bench.py $*" >> $OUT; python bench.py --no-cpu-baseline --no-via-trainer "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print(json.dumps({k:d.get(k) for k in ('metric','value','ms_per_step','dtype','mfma_frac_whole_step')} | {'roofline_frac': r.get('frac'), 'pool_roofline': (d.get('pool_roofline') or {}).get('frac')}))" >> $OUT; }
run --mode infer --steps 40 --warmup 10
run --mode infer --precision fp32 --steps 10 --warmup 3
run --precision parity --steps 10 --warmup 3
run --model longformer --steps 10 --warmup 3
run --model ponet --steps 10 --warmup 3
run --model bigbird --steps 10 --warmup 3
run --workload plain --steps 40 --warmup 10
run --model longformer --mode infer --steps 20 --warmup 5
run --model ponet --mode infer --steps 20 --warmup 5
run --model bigbird --mode infer --steps 20 --warmup 5
run --model longformer --seq-len 2048 --seqs-per-gpu 4 --steps 20 --warmup 5
run --model bigbird --seq-len 2048 --seqs-per-gpu 4 --steps 20 --warmup 5
run --model longformer --precision parity --steps 8 --warmup 3
run --seqs-per-gpu 8 --steps 40 --warmup 10
run --mode infer --precision parity --steps 30 --warmup 8
run --model longformer --mode infer --precision parity --steps 15 --warmup 5
cat $OUT
