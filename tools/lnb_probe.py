#!/usr/bin/env python
"""where ln_bwd's time goes: row loop alone (no partials), with the per-block partials, with the second-stage reduce"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spokennlp_amd import ops
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
for M, H in [(16384, 768), (32768, 768)]:
    y = torch.randn(M, H, device=dev).bfloat16(); x = torch.randn(M, H, device=dev).bfloat16()
    g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
    for p in (0.0, 0.1):
        out, mean, rstd = ops.add_ln_fwd(y.clone(), x, g, b, 1e-12, p=p, seed=3)
        dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dbias = torch.zeros(H, device=dev)
        part = torch.empty(ops.ln_partials_numel(M, H), device=dev)
        t0 = timeit(lambda: ops.ln_bwd(x, y, mean, rstd, g, p=p, seed=3))
        t1 = timeit(lambda: ops.ln_bwd(x, y, mean, rstd, g, p=p, seed=3, partials=part))
        t2 = timeit(lambda: ops.ln_bwd(x, y, mean, rstd, g, p=p, seed=3, dgamma=dg, dbeta=db, dbias=dbias, partials=part))
        print(f"M={M} p={p}: rows only {t0*1e6:.1f} us, + block partials {t1*1e6:.1f} us, + second-stage reduce {t2*1e6:.1f} us")
