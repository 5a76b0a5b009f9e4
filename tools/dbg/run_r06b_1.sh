set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_keepmask.py -x -q -m gpu > gpurun_out/r06b_tests1.txt 2>&1; tail -5 gpurun_out/r06b_tests1.txt
timeout 600 python tools/dbg/ln_km_fused_probe.py --step > gpurun_out/r06b_probe.txt 2>&1; tail -30 gpurun_out/r06b_probe.txt
