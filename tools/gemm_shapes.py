import sys, torch
sys.path.insert(0, "/root/repo")
from spokennlp_amd import ops
dev = torch.device("cuda")
def t(f, n=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for M, N, K in [(4096, 3840, 4096), (8192, 7680, 8192), (16384, 3072, 3072), (16384, 768, 3072), (32768, 768, 3072), (16384, 3072, 768), (16384, 2304, 768), (65536, 768, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dt = t(lambda: ops.gemm_nt(A, B, ops.EPI_NONE, out=out))
    dv = t(lambda: torch.matmul(A, B.t(), out=out))
    print(f"M={M} N={N} K={K}: ours {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.0f} TF | hipBLASLt {dv*1e6:8.1f} us {2*M*N*K/dv/1e12:7.0f} TF")
