#!/bin/bash
# rocprofv3 --kernel-trace summaries of the secondary bench configurations -> gpurun_out/<tag>_<name>_kernel_stats.md (+ the bench line of the same run)
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
prof() {
  NAME=$1; STEPS=$2; shift 2
  rm -rf /tmp/prof_$NAME
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -o run -- python bench.py "$@" --no-cpu-baseline --no-via-trainer --steps $STEPS --warmup 5 > gpurun_out/${TAG}_${NAME}_stdout.json 2> /dev/null
  DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
  python tools/prof_summary.py "$DB" gpurun_out/${TAG}_${NAME} $((STEPS + 15)) > /dev/null
  echo "" >> gpurun_out/${TAG}_${NAME}_kernel_stats.md
  echo "bench line of the same run (under rocprofv3; steps counted: $STEPS timed + 5 warm-up + 10 armed for the in-step roofline):" >> gpurun_out/${TAG}_${NAME}_kernel_stats.md
  tail -1 gpurun_out/${TAG}_${NAME}_stdout.json | cut -c1-420 >> gpurun_out/${TAG}_${NAME}_kernel_stats.md
  head -12 gpurun_out/${TAG}_${NAME}_kernel_stats.md | cut -c1-140
}
prof longformer 15 --model longformer
prof ponet 15 --model ponet
prof bigbird 15 --model bigbird
prof parity 10 --precision parity
prof infer 40 --mode infer
