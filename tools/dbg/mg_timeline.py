"""timeline probe of attn_bwd_merged_kernel (AMDSEG_MG_DEBUG=256): per workgroup start / after the partner wait / end, and the XCC it ran on"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["AMDSEG_MG_DEBUG"] = str(256 | int(sys.argv[1]) if len(sys.argv) > 1 else 256)
from spokennlp_amd import ops, lib as L
B, Lq, heads, p = 32, 512, 12, 0.1
dev = torch.device("cuda"); torch.manual_seed(0); H = heads * 64; M = B * Lq
qkv = (torch.randn(M, 3 * H, device=dev) * 0.8).bfloat16()
lens = torch.full((B,), Lq, dtype=torch.int64)
g = torch.Generator().manual_seed(1)
short = torch.rand(B, generator=g) < 0.3
lens[short] = torch.randint(40, Lq, (int(short.sum()),), generator=g)
am = (torch.arange(Lq)[None, :] < lens[:, None]).to(dev)
mask_bias = torch.where(am, 0.0, -10000.0).float().contiguous()
kend = lens.to(torch.int32).to(dev)
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev)
zero = torch.zeros(1, dtype=torch.int32, device=dev)
keep = ops.attn_keepmask(B, Lq, heads, p, 1234, dev, kend=kend)
ctx, lse = ops.attn_fwd_keep(qkv, mask_bias, B, Lq, heads, p, keep)
dctx = ((torch.randn(M, H, device=dev) * 0.5).bfloat16().view(B, Lq, H) * am[:, :, None]).reshape(M, H).contiguous()
nb = L.load().amdseg_attn_bwd_merged_scratch_bytes(B, Lq, heads)
part = torch.zeros(nb // 4, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.attn_bwd_merged(qkv, mask_bias, ctx, dctx, lse, B, Lq, heads, p, keep, kend=kend, seq_order=order, pad_guard=zero, dq_part=part)
torch.cuda.synchronize()
nf = B * heads * Lq * 64
tail = part[nf:].view(torch.int32)[2 + 4096:].contiguous().view(torch.int64)[: 768 * 4].view(768, 4).cpu()
t0 = tail[:, 0].min()
print("wg  kb xcc start(us) spin_end(us) end(us)   [wall clock ticks of 10 ns]")
for i in list(range(0, 768, 24)) + [383, 384, 385, 767]:
    s_, w_, e_, x_ = [int(v) for v in tail[i]]
    print(f"{i:4d} {i // 384} {x_ & 15:3d} {(s_ - t0) / 100:9.1f} {(w_ - t0) / 100 if w_ else -1:9.1f} {(e_ - t0) / 100:9.1f}")
d = (tail[:, 2] - tail[:, 0]).float() / 100
w = (tail[:, 1] - tail[:, 0]).float() / 100
print("duration us: kb0 mean %.1f max %.1f | kb1 mean %.1f max %.1f ; wait kb1 mean %.1f max %.1f" % (d[:384].mean(), d[:384].max(), d[384:].mean(), d[384:].max(), w[384:][tail[384:, 1] > 0].mean(), w[384:][tail[384:, 1] > 0].max()))
print("total span us %.1f" % ((tail[:, 2].max() - t0) / 100))
for x in range(8):
    sel = (tail[:, 3] & 15) == x
    print("xcc", x, "wgs", int(sel.sum()), "kb0", int(sel[:384].sum()), "kb1", int(sel[384:].sum()))
