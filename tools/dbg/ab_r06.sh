uptime
for k in 3072 2304 768; do BK_M=4096 BK_N=768 BK_K=$k python tools/dbg/nt_u8_ab2.py _ab/libamdseg_base13.so spokennlp_amd/libamdseg.so 2>&1 | grep -v "u8"; done
BK_M=8192 BK_N=768 BK_K=3072 python tools/dbg/nt_u8_ab2.py _ab/libamdseg_base13.so spokennlp_amd/libamdseg.so 2>&1 | grep -v "u8"
F="--no-cpu-baseline --no-via-trainer --no-extra-legs"
for i in 1 2 3; do
 for which in base new; do
  if [ $which = base ]; then export AMDSEG_LIB=/root/repo/_ab/libamdseg_base13.so; else unset AMDSEG_LIB; fi
  python bench.py $F --seqs-per-gpu 8 --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which $i', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
 done
done
