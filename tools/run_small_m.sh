#!/bin/bash
# small-M regime (run_finetune.sh's own launch shape: Longformer, 4 x 2048 tokens per GPU): GEMM tile choices + stream timeline
mkdir -p gpurun_out
O=gpurun_out/small_m.log
: > $O
for cfg in "" "AMDSEG_DP_BN=192" "AMDSEG_NT_ADAPTIVE=0" "AMDSEG_NT_ADAPTIVE=0 AMDSEG_DP_BN=192"; do
  echo "=== M=8192 $cfg" >> $O
  env BK_M=8192 $cfg python tools/bench_kernels.py nt tn >> $O 2>&1
done
for cfg in "" "AMDSEG_DP_BN=192" "AMDSEG_LF_OVERLAP=0"; do
  echo "=== longformer 4x2048 $cfg" >> $O
  env $cfg python bench.py --model longformer --seq-len 2048 --seqs-per-gpu 4 --no-cpu-baseline --no-via-trainer --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))" >> $O 2>&1
done
export TMPDIR=/tmp
rm -rf /tmp/prof_sm
rocprofv3 --kernel-trace --stats -d /tmp/prof_sm -o run -- python bench.py --model longformer --seq-len 2048 --seqs-per-gpu 4 --no-cpu-baseline --no-via-trainer --steps 40 --warmup 10 > gpurun_out/small_m_prof_stdout.json 2> gpurun_out/small_m_prof_stderr.log
DB=$(find /tmp/prof_sm -name "*.db" | head -1)
python tools/prof_summary.py "$DB" gpurun_out/small_m_overlap 50 > /dev/null
python tools/trace_gaps.py "$DB" 50 20 >> $O 2>&1
python tools/trace_streams.py "$DB" 20 >> $O 2>&1
cat $O
