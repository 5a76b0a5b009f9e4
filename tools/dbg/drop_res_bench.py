#!/usr/bin/env python
"""stand-alone cost of the fused dense + dropout + residual epilogue (amdseg_gemm_nt_bias_drop_res) against gemm_nt(BIAS) / gemm_nt(ADD_RES)
and the row kernel it would save half of (add_ln_fwd 4 passes -> 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spokennlp_amd import ops
dev = torch.device("cuda:0")
M, N = 16384, 768


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for K in (768, 3072):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).bfloat16(); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    g = torch.ones(N, device=dev); b = torch.zeros(N, device=dev)
    t_bias = timeit(lambda: ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias, out=out))
    t_res = timeit(lambda: ops.gemm_nt(A, B, ops.EPI_ADD_RES, R=R, out=out))
    t_f0 = timeit(lambda: ops.gemm_nt_bias_drop_res(A, B, bias, R, p=0.0))
    t_f1 = timeit(lambda: ops.gemm_nt_bias_drop_res(A, B, bias, R, p=0.1, seed=3, want_bits=False))
    t_f2 = timeit(lambda: ops.gemm_nt_bias_drop_res(A, B, bias, R, p=0.1, seed=3, want_bits=True))
    y = out.clone()
    t_ln4 = timeit(lambda: ops.add_ln_fwd(y, R, g, b, 1e-12, p=0.1, seed=3))
    t_ln2 = timeit(lambda: ops.add_ln_fwd(y, None, g, b, 1e-12))
    print(f"K={K}: gemm BIAS {t_bias:.1f}  ADD_RES {t_res:.1f}  fused p=0 {t_f0:.1f}  fused p=0.1 no bits {t_f1:.1f}  with bits {t_f2:.1f} us | add_ln_fwd 4-pass {t_ln4:.1f}  LN-only {t_ln2:.1f}")
