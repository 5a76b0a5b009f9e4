#!/usr/bin/env python
"""A/B of the two GELU GEMMs in their shipped (one-byte derivative) form at M = 16384, N = 3072, K = 768, against another build of the library:
AMDSEG_LIB=<other .so> python tools/dbg/nt_u8_ab.py   (same box, run the two alternately)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spokennlp_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("BK_M", 16384)), 3072, 768
lib = L.load()
g = torch.Generator().manual_seed(0)
A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
B = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
bias = torch.randn(N, generator=g).to(dev)
Q = torch.zeros(M, N, dtype=torch.uint8, device=dev)
H = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dU = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
fl = L.EPI_KEEP_DERIV | L.EPI_DERIV_U8


def fwd():
    L.check(lib.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, H.data_ptr(), N, M, N, K, ops.EPI_BIAS_GELU | fl, bias.data_ptr(), None, 0, Q.data_ptr(), N, 0, st), "fwd")


def bwd():
    L.check(lib.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, dU.data_ptr(), N, M, N, K, ops.EPI_GELU_BWD | fl, None, Q.data_ptr(), N, None, 0, 0, st), "bwd")


def plain():
    L.check(lib.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, dU.data_ptr(), N, M, N, K, ops.EPI_NONE, None, None, 0, None, 0, 0, st), "plain")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fwd(); bwd()
torch.cuda.synchronize()
chk = int(dU.view(torch.int16).to(torch.int64).sum().item())
print(f"{os.environ.get('AMDSEG_LIB', 'in-tree')}: bias+GELU+u8 {timeit(fwd):.1f} us | x u8 derivative {timeit(bwd):.1f} us | plain {timeit(plain):.1f} us | dU checksum {chk}")
