python -m pytest tests/test_gpu_kernels.py tests/test_gpu_keepmask.py tests/test_gpu_longformer.py tests/test_gpu_bigbird.py -q -x 2>&1 | tail -3
tools/dbg/ab_step_r06.sh r06_dq /root/repo/_ab/libamdseg_base13.so 2>&1 | grep -E "^base|^new|attn_bwd_dq|attn_bwd_dkv|attn_fwd|== "
