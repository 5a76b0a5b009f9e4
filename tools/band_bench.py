import torch, sys
sys.path.insert(0, "/root/repo")
from spokennlp_amd import ops
dev = torch.device("cuda")
B, L, heads, w = 8, 4096, 12, 256
H = heads * 64
qkv = torch.randn(B * L, 3 * H, device=dev).bfloat16()
mask = torch.zeros(B, L, device=dev)
dctx = torch.randn(B * L, H, device=dev).bfloat16()
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for G in (1, 0):
    for p in (0.0, 0.1):
        ctx, lse = ops.attn_band_fwd(qkv, mask, B, L, heads, w, G, p=p, seed=1)
        tf = t(lambda: ops.attn_band_fwd(qkv, mask, B, L, heads, w, G, p=p, seed=1))
        tb = t(lambda: ops.attn_band_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, w, G, p=p, seed=1))
        fl = 4 * B * heads * L * 513 * 64
        print(f"G={G} p={p}: fwd {tf:.0f} us ({fl/tf/1e6:.0f} TF)  bwd {tb:.0f} us ({2.5*fl/tb/1e6:.0f} TF)")
