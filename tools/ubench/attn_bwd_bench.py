#!/usr/bin/env python
"""attention backward: the one-kernel form (amdseg_attn_bwd_merged) against the two-kernel form (amdseg_attn_bwd_keep / amdseg_attn_bwd):
max difference of dQ / dK / dV (both are bf16 roundings of fp32 sums taken in different orders) and the time per launch, on the headline shape
(32 x 512, 12 heads) with the bench's kind of trailing padding.  usage: python tools/attn_bwd_bench.py [B L heads p]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spokennlp_amd import ops, lib as L  # noqa: E402


def main():
    B, Lq, heads, p = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (32, 512, 12, 0.1)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    H = heads * 64
    M = B * Lq
    qkv = (torch.randn(M, 3 * H, device=dev) * 0.8).bfloat16()
    lens = torch.full((B,), Lq, dtype=torch.int64)
    g = torch.Generator().manual_seed(1)
    short = torch.rand(B, generator=g) < 0.3                         # ~30 % of the windows end in padding
    lens[short] = torch.randint(40, Lq, (int(short.sum()),), generator=g)
    am = (torch.arange(Lq)[None, :] < lens[:, None]).to(dev)
    mask_bias = torch.where(am, 0.0, -10000.0).float().contiguous()
    kend = lens.to(torch.int32).to(dev)
    order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev)
    zero = torch.zeros(1, dtype=torch.int32, device=dev)              # pad_guard == 0: the dO rows of padding are exact zeros
    keep = ops.attn_keepmask(B, Lq, heads, p, 1234, dev, kend=kend) if p > 0 else None
    if p > 0:
        ctx, lse = ops.attn_fwd_keep(qkv, mask_bias, B, Lq, heads, p, keep)
    else:
        ctx, lse = ops.attn_fwd(qkv, mask_bias, B, Lq, heads)
    dctx = (torch.randn(M, H, device=dev) * 0.5).bfloat16()
    dctx = (dctx.view(B, Lq, H) * am[:, :, None]).reshape(M, H).contiguous()

    def two():
        return ops.attn_bwd_keep(qkv, mask_bias, ctx, dctx, lse, B, Lq, heads, p, keep) if p > 0 else ops.attn_bwd(qkv, mask_bias, ctx, dctx, lse, B, Lq, heads)

    part = torch.zeros(L.load().amdseg_attn_bwd_merged_scratch_bytes(B, Lq, heads) // 4, dtype=torch.float32, device=dev)

    def one(guard=True):
        return ops.attn_bwd_merged(qkv, mask_bias, ctx, dctx, lse, B, Lq, heads, p, keep, kend=kend, seq_order=order, pad_guard=zero if guard else None, dq_part=part)

    ref = two().float().view(B, Lq, 3, H)
    for guard in (True, False):
        got = one(guard).float().view(B, Lq, 3, H)
        torch.cuda.synchronize()
        for i, name in enumerate(("dQ", "dK", "dV")):
            d = (got[:, :, i] - ref[:, :, i]).abs()
            sc = ref[:, :, i].abs().max().item()
            print(f"guard={guard} {name}: max|diff| {d.max().item():.4g}  (scale {sc:.4g}, rel {d.max().item() / sc:.3g}), mean|diff| {d.mean().item():.3g}, nan {int(torch.isnan(got[:, :, i]).sum())}")
        bad = (~am)[:, :, None].expand(B, Lq, H)
        print("   padded rows all zero (dQ, dK, dV):", [bool((got[:, :, i][bad] == 0).all()) for i in range(3)] if guard else "n/a")

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t2, t1 = timeit(two), timeit(one)
    fl = 10.0 * B * heads * Lq * Lq * 64
    print(f"two kernels {t2:.1f} us   merged {t1:.1f} us (KT={os.environ.get('AMDSEG_ATTN_MERGED_KT', '2')})   algorithmic {fl / 1e9:.1f} GFLOP -> merged {fl / t1 / 1e6:.0f} TFLOP/s")
    # run-to-run reproducibility of the merged form
    a1, a2 = one(), one()
    print("merged bit-reproducible:", bool(torch.equal(a1, a2)))


if __name__ == "__main__":
    main()
