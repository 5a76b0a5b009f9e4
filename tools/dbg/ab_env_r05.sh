for rep in 1 2 3; do
for v in "" "AMDSEG_DP_BN=192" "AMDSEG_OVERLAP_WGRAD=1" "AMDSEG_DP_BN=192 AMDSEG_OVERLAP_WGRAD=1"; do
  echo "== $v"; env $v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-via-trainer --no-extra-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['final_loss'])"
done; done
