for rep in 1 2; do
for v in "" "AMDSEG_FUSE_DROP_RES=1" "AMDSEG_ATTN_NW4=1" "AMDSEG_LNP_WGS_PER_CU=4" "AMDSEG_DP_MIN_K=1024"; do
  echo "== $v"; env $v python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-via-trainer --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernels']
print(d['value'], d['ms_per_step'], d['ms_per_step_median'], 'nt', k['gemm_nt_dp_kernel']['avg_launch_us'], 'tn', k['gemm_tn_dp_kernel']['avg_launch_us'], 'enc', r['encoder_gemms']['frac'], r['encoder_gemms']['frac_executed'])"
done; done
