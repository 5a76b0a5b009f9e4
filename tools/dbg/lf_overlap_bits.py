import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.test_oracle_golden import lf_case, flags_of
from tests.test_gpu_longformer import build_lf
dev = torch.device("cuda:0")
z, sd, batch, arch = lf_case("lf_tiny_L128_w16")
runs = []
for overlap in (False, 1, 2, 4, 7, 3):
    m = build_lf(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train()
    eng = m.engine(); eng.lf_overlap = overlap
    outs = []
    for it in range(3):
        m.zero_grad(set_to_none=False)
        random.seed(7 + it)
        loss = m(**{k: v.to(dev) for k, v in batch.items()})[0]
        loss.backward()
        outs.append((loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
    runs.append(outs)
def cmp(a, b, tag):
    bad = {}
    for it in range(3):
        for n in a[it][1]:
            d = float((a[it][1][n] - b[it][1][n]).abs().max())
            if d > 0: bad[n] = max(bad.get(n, 0), d)
    print(tag, "loss equal:", [a[i][0] == b[i][0] for i in range(3)], "params differing:", len(bad), sorted(bad.items(), key=lambda kv: -kv[1])[:6])
names = ["off", "fwd", "mid", "tail", "all", "fwd+mid"]
def nd(a, b):
    k = 0; mx = 0.0
    for it in range(3):
        for n in a[it][1]:
            if "embeddings" in n or "loss_calculator" in n: continue
            d = float((a[it][1][n] - b[it][1][n]).abs().max())
            if d > 0: k += 1; mx = max(mx, d)
    return k, mx
for i in range(len(runs)):
    print(names[i].ljust(8), [nd(runs[i], runs[j])[0] for j in range(len(runs))])

n = "longformer.encoder.layer.1.output.LayerNorm.weight"
for it in range(3):
    print("iter", it, "off vs mid:", float((runs[0][it][1][n] - runs[2][it][1][n]).abs().max()), "off vs tail:", float((runs[0][it][1][n] - runs[3][it][1][n]).abs().max()))
