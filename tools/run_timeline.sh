#!/bin/bash
# kernel timeline of one training step (two-stream models): tools/run_timeline.sh <tag> <bench.py arguments...>
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o run -- python bench.py "$@" --no-cpu-baseline --no-via-trainer --steps 12 --warmup 6 > gpurun_out/${TAG}_tl_stdout.json 2> gpurun_out/${TAG}_tl_stderr.log
DB=$(find /tmp/prof_tl -name "*.db" | head -1)
python tools/trace_dump.py "$DB" gpurun_out/${TAG}_timeline.txt
python tools/trace_streams.py "$DB" 8 > gpurun_out/${TAG}_streams.txt 2>&1
tail -30 gpurun_out/${TAG}_streams.txt
