#!/bin/bash
# the training step beside H occupied CUs (another process's spinning workgroups): what an overlapped RCCL all-reduce's channels would take away
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/dbg/cu_hog.hip -o gpurun_out/cu_hog || exit 1
B="python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs --steps 40 --warmup 10"
for H in 0 16; do for BUD in 0 240; do export AMDSEG_CU_BUDGET=$BUD; echo "budget $BUD";
  if [ "$H" != "0" ]; then timeout 60 gpurun_out/cu_hog $H 25 > gpurun_out/cu_hog_$H.log 2>&1 & HP=$!; sleep 3; fi
  $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('hog $H', d['value'], d['ms_per_step'], {n[:10]: v['avg_launch_us'] for n,v in k.items() if n.startswith('gemm') or n.startswith('attn_bwd')})"
  if [ "$H" != "0" ]; then wait $HP; fi; done
done
