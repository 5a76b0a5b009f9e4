#!/bin/bash
# in-step HBM traffic of the dominant kernels: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as
# MI355X_MICROARCH.md's HBM section prescribes) over the real bench command; per-kernel averages -> gpurun_out/<tag>_pmc_instep.txt
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_pmc_instep.txt
# the hash of the kernel sources the passes RUN on goes into the output (tools/pmc_to_json.py refuses a file whose hash is not the shipped sources')
echo "## csrc_sha: $(python -c 'from spokennlp_amd.build import sources_sha; print(sources_sha())')" > $OUT
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${CTR}
  rocprofv3 --kernel-trace --pmc $CTR -d /tmp/pmc_${CTR} -o run -- python bench.py --no-cpu-baseline --no-via-trainer --no-roofline --no-extra-legs --steps 4 --warmup 2 > /dev/null 2> gpurun_out/${TAG}_pmc_${CTR}.err
  DB=$(find /tmp/pmc_${CTR} -name "*.db" | head -1)
  echo "## pass: --pmc $CTR (python bench.py --steps 4 --warmup 2: 6 real training steps)" >> $OUT
  python tools/pmc_summary.py "$DB" | grep -E "gemm_nt_dp|gemm_tn_dp|attn_|ln_bwd|add_ln_fwd|adamw|^\| kernel|^\|---" >> $OUT
done
cat $OUT
