"""kernel-level timing of the N = 768 / 2304 / 3072 GEMMs of a layer at small M; run once per library build (AMDSEG_LIB)"""
import sys
import torch
sys.path.insert(0, ".")
from spokennlp_amd import ops

dev = torch.device("cuda:0")
Ms = [int(x) for x in sys.argv[1:]] or [2048, 4096, 6144, 8192, 10240]
for M in Ms:
    for N, K, epi, name in [(768, 768, ops.EPI_BIAS, "bias"), (768, 768, ops.EPI_NONE, "none"), (768, 3072, ops.EPI_BIAS, "bias"),
                            (768, 3072, ops.EPI_ADD_RES, "add_res"), (768, 2304, ops.EPI_ADD_RES, "add_res"), (2304, 768, ops.EPI_BIAS, "bias"),
                            (3072, 768, ops.EPI_NONE, "none")]:
        A = torch.randn(M, K, device=dev).bfloat16()
        B = torch.randn(N, K, device=dev).bfloat16()
        R = torch.randn(M, N, device=dev).bfloat16() if epi == ops.EPI_ADD_RES else None
        bias = torch.randn(N, device=dev) if epi == ops.EPI_BIAS else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm_nt(A, B, epi, bias=bias, R=R, out=out)
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        print(f"M={M} N={N} K={K} {name}: {us:.1f} us", flush=True)
