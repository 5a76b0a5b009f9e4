"""kernel-level timing of the residual-adding input-gradient GEMMs (EPI_ADD_RES, N = 768) over M; run once per library build (AMDSEG_LIB)"""
import sys
import torch
sys.path.insert(0, ".")
from spokennlp_amd import ops

dev = torch.device("cuda:0")
for M in (4096, 8192, 12288, 16384):
    for K in (3072, 2304):
        A = torch.randn(M, K, device=dev).bfloat16()
        B = torch.randn(768, K, device=dev).bfloat16()
        R = torch.randn(M, 768, device=dev).bfloat16()
        out = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm_nt(A, B, ops.EPI_ADD_RES, R=R, out=out)
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        print(f"M={M} K={K}: {us:.1f} us  {2 * M * 768 * K / us / 1e6:.0f} TFLOP/s", flush=True)
