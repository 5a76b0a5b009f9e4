#!/bin/bash
# GPU-box session of round 5 (final state): the GPU suite, the default bench, the rocprofv3 table of the same bench command, the N = 2 / N = 8 launcher
# paths (ranks sharing this box's one GPU over gloo), the secondary legs, the two PMC passes + profiles/pmc_traffic.json stamped with the sources' hash
TAG=${1:-r05_final}
GIT=${2:-unknown}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
cp gpurun_out/parity_values.json gpurun_out/${TAG}_parity_values.json 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_stdout.json 2> gpurun_out/${TAG}_bench_stderr.log
export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG} -o run -- python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs > gpurun_out/${TAG}_bench_stdout_under_rocprof.json 2> /dev/null
DB=$(find /tmp/prof_${TAG} -name "*.db" | head -1)
python tools/prof_summary.py "$DB" gpurun_out/${TAG}_bench 130 > /dev/null
for N in 2 8; do
  AMDSEG_DIST_BACKEND=gloo timeout 900 python bench.py --gpus $N --seqs-per-gpu 4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_gpus${N}_gloo.json 2> gpurun_out/${TAG}_bench_gpus${N}_gloo.err
done
tools/run_all_models.sh ${TAG} > /dev/null 2>&1
tools/run_pmc_instep.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/${TAG}_pmc_instep.txt ${GIT} profiles/${TAG}_pmc_instep.md > /dev/null 2>&1 && cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
tail -4 gpurun_out/${TAG}_pytest_gpu.txt
tail -1 gpurun_out/${TAG}_bench_stdout.json | cut -c1-600
head -22 gpurun_out/${TAG}_bench_kernel_stats.md | cut -c1-140
for N in 2 8; do tail -1 gpurun_out/${TAG}_bench_gpus${N}_gloo.json | cut -c1-300; tail -2 gpurun_out/${TAG}_bench_gpus${N}_gloo.err | cut -c1-300; done
cat gpurun_out/${TAG}_models.txt
cat gpurun_out/${TAG}_pmc_instep.txt | cut -c1-160
