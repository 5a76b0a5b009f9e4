#!/bin/bash
# one GPU-box session: tests, default bench, rocprof trace of the same bench command; outputs under gpurun_out/
set -x
TAG=${1:-r03}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log
python bench.py > gpurun_out/${TAG}_bench_stdout.json 2> gpurun_out/${TAG}_bench_stderr.log
export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG} -o run -- python bench.py --no-cpu-baseline --no-via-trainer > gpurun_out/${TAG}_prof_bench_stdout.json 2> gpurun_out/${TAG}_prof_stderr.log
DB=$(find /tmp/prof_${TAG} -name "*.db" | head -1)
python tools/prof_summary.py "$DB" gpurun_out/${TAG}_bench 130 > /dev/null
tail -3 gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_bench_stdout.json
head -20 gpurun_out/${TAG}_bench_kernel_stats.md
cat gpurun_out/${TAG}_prof_bench_stdout.json
