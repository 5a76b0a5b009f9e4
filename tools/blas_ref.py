"""What the vendor library (hipBLASLt via torch.matmul) reaches on the layer's GEMM shapes -- a yardstick for gemm.hip."""
import torch, sys
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
def t(f, n=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for N, K in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    dt = t(lambda: torch.matmul(A, B.t()))
    print(f"NT M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.0f} TF")
for N, K in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:      # wgrad: C[N,K] = A[M,N]^T B[M,K]
    A = torch.randn(M, N, device=dev).bfloat16(); B = torch.randn(M, K, device=dev).bfloat16()
    dt = t(lambda: torch.matmul(A.t(), B))
    print(f"TN M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.0f} TF")
