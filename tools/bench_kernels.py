#!/usr/bin/env python
"""Micro-benchmarks of the libamdseg kernels at the bert-base / M=16384 shapes (HIP-event timing on the launch stream).  BK_M overrides M for
the GEMM legs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spokennlp_amd import ops

dev = torch.device("cuda:0")
M, H, I, B, L, heads = int(os.environ.get("BK_M", 16384)), 768, 3072, 32, 512, 12      # BK_M=8192: the 4 x 2048 launch shape of run_finetune.sh


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    which = sys.argv[1:] or ["nt", "tn", "attn", "rows"]
    if "nt" in which:
        for N, K, epi in [(2304, 768, ops.EPI_BIAS), (768, 768, ops.EPI_BIAS), (3072, 768, ops.EPI_BIAS_GELU), (768, 3072, ops.EPI_BIAS),
                          (3072, 768, ops.EPI_GELU_BWD), (768, 3072, ops.EPI_ADD_RES), (768, 768, ops.EPI_NONE), (768, 2304, ops.EPI_ADD_RES)]:
            A = torch.randn(M, K, device=dev).bfloat16(); Bm = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            bias = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).bfloat16()
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); out2 = torch.empty_like(out)
            kw = dict(bias=bias if epi in (1, 2) else None, R=R if epi in (3, 4) else None, out=out, out2=out2 if epi == 2 else None)
            t = timeit(lambda: ops.gemm_nt(A, Bm, epi, **kw))
            print(f"gemm_nt N={N:5d} K={K:5d} epi={epi}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF")
    if "tn" in which:
        shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
        As = [torch.randn(M, n, device=dev).bfloat16() for n, _ in shapes]
        Bs = [torch.randn(M, k, device=dev).bfloat16() for _, k in shapes]
        Cs = [torch.zeros(n, k, device=dev) for n, k in shapes]
        t = timeit(lambda: ops.gemm_tn_grouped(As, Bs, Cs, accumulate=True), reps=10)
        fl = sum(2.0 * M * n * k for n, k in shapes)
        print(f"gemm_tn grouped layer wgrad: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF")
    if "attn" in which:
        qkv = torch.randn(M, 3 * H, device=dev).bfloat16(); mb = torch.zeros(B, L, device=dev)
        dctx = torch.randn(M, H, device=dev).bfloat16()
        for p in (0.0, 0.1):
            t = timeit(lambda: ops.attn_fwd(qkv, mb, B, L, heads, p=p, seed=1))
            fl = 4.0 * L * L * 64 * B * heads
            print(f"attn_fwd p={p}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF")
            ctx, lse = ops.attn_fwd(qkv, mb, B, L, heads, p=p, seed=1)
            t = timeit(lambda: ops.attn_bwd(qkv, mb, ctx, dctx, lse, B, L, heads, p=p, seed=1))
            print(f"attn_bwd p={p}: {t*1e6:8.1f} us  {2.5*fl/t/1e12:7.1f} TF (5 matmul-equivalents)")
        # dropout decided once per layer (amdseg_attn_keepmask): the generator and the three kernels reading its lane masks
        p = 0.1
        t = timeit(lambda: ops.attn_keepmask(B, L, heads, p, 1, dev))
        print(f"attn_keepmask (both layouts, {B * heads * L * L / 4 / 1e6:.1f} MB): {t*1e6:8.1f} us")
        keep = ops.attn_keepmask(B, L, heads, p, 1, dev)
        t = timeit(lambda: ops.attn_fwd_keep(qkv, mb, B, L, heads, p, keep))
        print(f"attn_fwd_keep p={p}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF")
        ctx, lse = ops.attn_fwd_keep(qkv, mb, B, L, heads, p, keep)
        t = timeit(lambda: ops.attn_bwd_keep(qkv, mb, ctx, dctx, lse, B, L, heads, p, keep))
        print(f"attn_bwd_keep p={p}: {t*1e6:8.1f} us  {2.5*fl/t/1e12:7.1f} TF (5 matmul-equivalents)")
    if "rows" in which:
        y = torch.randn(M, H, device=dev).bfloat16(); x = torch.randn(M, H, device=dev).bfloat16()
        g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
        for p in (0.0, 0.1):
            t = timeit(lambda: ops.add_ln_fwd(y, x, g, b, 1e-12, p=p, seed=3))
            print(f"add_ln_fwd p={p}: {t*1e6:8.1f} us  {4*M*H*2/t/1e9:7.0f} GB/s")
            out, mean, rstd = ops.add_ln_fwd(y, x, g, b, 1e-12, p=p, seed=3)
            dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dbias = torch.zeros(H, device=dev)
            part = torch.empty(ops.ln_partials_numel(M, H), device=dev)
            t = timeit(lambda: ops.ln_bwd(x, y, mean, rstd, g, p=p, seed=3, dgamma=dg, dbeta=db, dbias=dbias, partials=part))
            print(f"ln_bwd p={p}: {t*1e6:8.1f} us  {(4 if p else 3)*M*H*2/t/1e9:7.0f} GB/s")
        t = timeit(lambda: ops.colsum(torch.empty(0) if False else y))
        print(f"colsum [M,768]: {t*1e6:8.1f} us")


if __name__ == "__main__":
    main()
