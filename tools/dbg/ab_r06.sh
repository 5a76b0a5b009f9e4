F="--no-cpu-baseline --no-via-trainer --no-extra-legs --steps 60 --warmup 10"
for i in 1 2 3; do
  for x in "" "--prof-in-timed"; do
    python bench.py $F $x 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$x]', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r['method'][:90])"
  done
done
