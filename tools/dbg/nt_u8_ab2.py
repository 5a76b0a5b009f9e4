#!/usr/bin/env python
"""Same-process, interleaved A/B of two builds of libamdseg on the two GELU GEMMs (one-byte derivative form), M = 16384, N = 3072, K = 768:
python tools/dbg/nt_u8_ab2.py <base.so> <new.so>"""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spokennlp_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("BK_M", 16384)), int(os.environ.get("BK_N", 3072)), int(os.environ.get("BK_K", 768))
libs = []
for path in sys.argv[1:3]:
    h = C.CDLL(path)
    h.amdseg_gemm_nt.argtypes = L._PROTOS["amdseg_gemm_nt"]
    h.amdseg_gemm_nt.restype = C.c_int
    libs.append((os.path.basename(path), h))
g = torch.Generator().manual_seed(0)
A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
B = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
bias = torch.randn(N, generator=g).to(dev)
R = torch.randn(M, N, generator=g).to(dev).bfloat16()
Q = torch.zeros(M, N, dtype=torch.uint8, device=dev)
H = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dU = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
fl = L.EPI_KEEP_DERIV | L.EPI_DERIV_U8
legs = {
    "bias+GELU+u8": lambda h: h.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, H.data_ptr(), N, M, N, K, ops.EPI_BIAS_GELU | fl, bias.data_ptr(), None, 0, Q.data_ptr(), N, 0, st),
    "x u8 deriv": lambda h: h.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, dU.data_ptr(), N, M, N, K, ops.EPI_GELU_BWD | fl, None, Q.data_ptr(), N, None, 0, 0, st),
    "+ residual": lambda h: h.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, dU.data_ptr(), N, M, N, K, ops.EPI_ADD_RES, None, R.data_ptr(), N, None, 0, 0, st),
    "plain": lambda h: h.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, dU.data_ptr(), N, M, N, K, ops.EPI_NONE, None, None, 0, None, 0, 0, st),
}


def timeit(fn, reps=40, warm=4):
    for _ in range(warm):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, leg in legs.items():
    acc = {n: [] for n, _ in libs}
    for rnd in range(8):
        for n, h in (libs if rnd % 2 == 0 else libs[::-1]):
            acc[n].append(timeit(lambda: leg(h)))
    print(f"M={M} N={N} K={K} {name:14s} " + " | ".join(f"{n}: mean {sum(v)/len(v):.1f} min {min(v):.1f} us" for n, v in acc.items()), flush=True)
