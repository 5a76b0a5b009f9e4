cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for M in longformer ponet bigbird; do
  rm -rf /tmp/prof_$M
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$M -o run -- python bench.py --model $M --steps 10 --warmup 3 --no-extra-legs --no-roofline > gpurun_out/r06_${M}_bench_under_rocprof.json 2>/dev/null
  DB=$(find /tmp/prof_$M -name "*.db" | head -1)
  python tools/prof_summary.py "$DB" gpurun_out/r06_${M} 13 > /dev/null
  head -40 gpurun_out/r06_${M}_kernel_stats.md | cut -c1-150
done
