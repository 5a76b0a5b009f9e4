#!/usr/bin/env python
"""probe: one M = 16384 GELU GEMM (768 tiles = three lock-step rounds, epilogues = HBM time) against the same work as two M = 8192 launches on two
streams at once (workgroups of both kernels interleave on the CUs: their epilogues fall into each other's K loops) and against the two launches
back to back on one stream"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spokennlp_amd import ops
dev = torch.device("cuda:0")
M, N, K = 16384, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); out2 = torch.empty_like(out)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = M // 2

def full():
    ops.gemm_nt(A, B, ops.EPI_BIAS_GELU, bias=bias, out=out, out2=out2)

def halves_serial():
    ops.gemm_nt(A[:h], B, ops.EPI_BIAS_GELU, bias=bias, out=out[:h], out2=out2[:h])
    ops.gemm_nt(A[h:], B, ops.EPI_BIAS_GELU, bias=bias, out=out[h:], out2=out2[h:])

def halves_parallel():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ops.gemm_nt(A[:h], B, ops.EPI_BIAS_GELU, bias=bias, out=out[:h], out2=out2[:h])
    with torch.cuda.stream(s2):
        ops.gemm_nt(A[h:], B, ops.EPI_BIAS_GELU, bias=bias, out=out[h:], out2=out2[h:])
    cur.wait_stream(s1); cur.wait_stream(s2)

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for r in range(3):
    print("full %.1f us   two halves, one stream %.1f   two halves, two streams %.1f" % (timeit(full), timeit(halves_serial), timeit(halves_parallel)))
