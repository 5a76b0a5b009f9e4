#!/bin/bash
# round 4 A/B legs on one box (same box, back to back): `tools/run_r04_ab.sh "tag:ENV=.. ENV=.." ...`
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs --steps 60 --warmup 15 $BENCH_ARGS"
for spec in "$@"; do
  tag=${spec%%:*}; env=${spec#*:}; [ "$env" = "$spec" ] && env=
  env $env $B 2> gpurun_out/r04_ab_$tag.err | tail -1 > gpurun_out/r04_ab_$tag.json
  python - <<P
import json
d=json.load(open("gpurun_out/r04_ab_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["ms_per_step_median"], {k[:12]:v["avg_launch_us"] for k,v in d.get("roofline",{}).get("kernels",{}).items()})
P
done
