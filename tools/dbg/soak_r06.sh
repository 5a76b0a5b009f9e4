# 1500-step soak of the default training step + two more default-shape samples (final sources of the round)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs --no-roofline --steps 1500 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('soak 1500 steps:', d['value'], 'seq/s', d['ms_per_step'], 'ms, final loss', d.get('final_loss'))" > gpurun_out/r06_soak.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[]', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('frac_executed'), r['encoder_gemms']['frac_executed'])" >> gpurun_out/r06_soak.txt; done
cat gpurun_out/r06_soak.txt
