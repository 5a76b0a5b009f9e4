#!/bin/bash
# one GPU-box session of round 6: the GPU suite, the default bench, the rocprofv3 table of the same bench command, the secondary legs, the PMC passes
TAG=${1:-r06_final}
mkdir -p gpurun_out
grep -m1 "model name" /proc/cpuinfo > gpurun_out/${TAG}_host.txt; uptime >> gpurun_out/${TAG}_host.txt
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
python bench.py > gpurun_out/${TAG}_bench_stdout.json 2> gpurun_out/${TAG}_bench_stderr.log
export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG} -o run -- python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs > gpurun_out/${TAG}_bench_stdout_under_rocprof.json 2> /dev/null
DB=$(find /tmp/prof_${TAG} -name "*.db" | head -1)
python tools/prof_summary.py "$DB" gpurun_out/${TAG}_bench 130 > /dev/null
tools/run_all_models.sh ${TAG} > /dev/null 2>&1
tools/run_pmc_instep.sh ${TAG} > /dev/null 2>&1
cp gpurun_out/parity_values.json gpurun_out/${TAG}_parity_values.json 2>/dev/null
tail -4 gpurun_out/${TAG}_pytest_gpu.txt
tail -1 gpurun_out/${TAG}_bench_stdout.json | cut -c1-700
head -22 gpurun_out/${TAG}_bench_kernel_stats.md | cut -c1-140
cat gpurun_out/${TAG}_models.txt
head -3 gpurun_out/${TAG}_pmc_instep.txt
