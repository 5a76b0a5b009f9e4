"""does the first training step of an engine read memory it never wrote?  poison the caching allocator's free blocks with NaN (or a large
finite value) before building the model: anything that leaks shows up in the loss / gradients of iteration 0"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.test_oracle_golden import lf_case, load_case, flags_of
dev = torch.device("cuda:0")
family = sys.argv[1] if len(sys.argv) > 1 else "longformer"
def poison(val):
    xs = [torch.full((1 << 26,), val, dtype=torch.float32, device=dev) for _ in range(8)]       # 2 GiB of blocks of the pattern
    ys = [torch.full((1 << 20,), val, dtype=torch.float32, device=dev) for _ in range(64)]
    zs = [torch.full((1 << 12,), val, dtype=torch.float32, device=dev) for _ in range(256)]
    del xs, ys, zs
def build():
    if family == "longformer":
        from tests.test_gpu_longformer import build_lf
        z, sd, batch, arch = lf_case("lf_tiny_L128_w16")
        return build_lf(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train(), batch
    from tests.test_gpu_model import build_model
    z, sd, batch, arch = load_case("tiny_L64")
    return build_model(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train(), batch
res = []
for val in (0.0, float("nan"), 3.0e4, 0.0):
    poison(val)
    m, batch = build()
    random.seed(7)
    loss = m(**{k: v.to(dev) for k, v in batch.items()})[0]
    loss.backward()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    bad = [n for n, t in g.items() if not torch.isfinite(t).all()]
    res.append((loss.item(), g))
    print(f"poison {val}: loss {loss.item():.6f}, non-finite grads in {len(bad)} params {bad[:4]}")
    del m
base = res[0][1]
for i, (l, g) in enumerate(res[1:], 1):
    diffs = {n: float((g[n] - base[n]).abs().max()) for n in g if torch.isfinite(g[n]).all() and "embeddings" not in n and "loss_calculator" not in n}
    worst = sorted(diffs.items(), key=lambda kv: -kv[1])[:3]
    print(f"run {i} vs run 0: loss diff {abs(l - res[0][0]):.3e}, params differing {sum(1 for v in diffs.values() if v > 0)}, worst {worst}")
