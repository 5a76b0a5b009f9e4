import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spokennlp_amd import lib as L, ops
dev = torch.device("cuda:0")
lib = L.load()
print("lib", os.environ.get("AMDSEG_LIB", "in-tree"), "DP_BN", os.environ.get("AMDSEG_DP_BN"))
st = torch.cuda.current_stream().cuda_stream
for M, N, K in [(256, 768, 768), (768, 3072, 768), (4096, 3072, 768)]:
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16(); B = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    H, U = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU, bias=bias)
    Hn = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU, bias=bias, out2=False) if False else None
    Hd, D = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU | L.EPI_KEEP_DERIV, bias=bias)
    Q = torch.zeros(M, N, dtype=torch.uint8, device=dev); H8 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    rc = lib.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, H8.data_ptr(), N, M, N, K, ops.EPI_BIAS_GELU | L.EPI_KEEP_DERIV | L.EPI_DERIV_U8, bias.data_ptr(), None, 0, Q.data_ptr(), N, 0, st)
    Hi = torch.empty(M, N, dtype=torch.bfloat16, device=dev)      # inference form: no second output
    rc2 = lib.amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, Hi.data_ptr(), N, M, N, K, ops.EPI_BIAS_GELU, bias.data_ptr(), None, 0, None, 0, 0, st)
    torch.save(dict(H=H.cpu(), Hd=Hd.cpu(), H8=H8.cpu(), Q=Q.cpu(), Hi=Hi.cpu()), f"gpurun_out/dbg_{os.environ.get('TAGX','x')}_{M}_{N}.pt")
    print(M, N, K, rc, rc2, "| H != Hd:", int((H != Hd).sum()), "| H != H8:", int((H != H8).sum()), "| H != Hi:", int((H != Hi).sum()))
