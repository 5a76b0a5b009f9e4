"""is the dual-output GELU epilogue of the 256 x 256 NT kernel bound by HBM contention between the workgroups of a round?  One round of T tiles
(T = 60 / 120 / 252), N = 3072, K = 768, plain vs bias + GELU (two outputs) vs GELU' x R epilogue."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spokennlp_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, reps=40, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N, K = 3072, 768
for M in (1280, 2560, 5376, 16384):
    A = torch.randn(M, K, device=dev).bfloat16(); Bm = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); out2 = torch.empty_like(out)
    t0 = timeit(lambda: ops.gemm_nt(A, Bm, ops.EPI_NONE, out=out))
    t1 = timeit(lambda: ops.gemm_nt(A, Bm, ops.EPI_BIAS_GELU, bias=bias, out=out, out2=out2))
    t2 = timeit(lambda: ops.gemm_nt(A, Bm, ops.EPI_GELU_BWD, R=R, out=out))
    tiles = (M // 256) * (N // 256)
    print(f"M = {M:6d} ({tiles:4d} tiles, {tiles / 256:.2f} rounds): plain {t0:6.1f} us, bias + GELU (2 outputs) {t1:6.1f} us (+{t1 - t0:5.1f}), GELU' x R {t2:6.1f} us (+{t2 - t0:5.1f})")
