#!/bin/bash
# rocprofv3 kernel table + bench line of the reference's shipped launch shape (run_finetune.sh: Longformer, L = 2048, 4 sequences per GPU)
TAG=${1:-r04_lf2048}
export TMPDIR=/tmp
rm -rf /tmp/prof_l2
rocprofv3 --kernel-trace --stats -d /tmp/prof_l2 -o run -- python bench.py --model longformer --seq-len 2048 --seqs-per-gpu 4 --steps 40 --warmup 10 --no-cpu-baseline --no-via-trainer > gpurun_out/${TAG}_stdout.json 2>/dev/null
python tools/prof_summary.py $(find /tmp/prof_l2 -name "*.db" | head -1) gpurun_out/${TAG} 60 > /dev/null
head -44 gpurun_out/${TAG}_kernel_stats.md | cut -c1-150
python bench.py --model longformer --seq-len 2048 --seqs-per-gpu 4 --steps 40 --warmup 10 --no-cpu-baseline --no-via-trainer --no-roofline 2>/dev/null | tail -1 | cut -c1-200
