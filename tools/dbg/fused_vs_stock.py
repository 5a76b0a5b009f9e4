import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.test_oracle_golden import load_case, flags_of
from tests.test_gpu_model import build_model, to_dev
from tests.test_gpu_trainer import _samples
from spokennlp_amd.trainer import AmdsegFusedAdamW
from transformers import Trainer

dev = torch.device("cuda:0")
z, sd, batch, arch = load_case("tiny_L64")
flags = flags_of(z, "train_full")
samples = _samples(arch, 16)
def mb(i):
    s = samples[4 * i:4 * i + 4]
    return {k: torch.stack([x[k] for x in s]).to(dev) for k in s[0]}
res = {}
for name in ("stock", "fused"):
    m = build_model(arch, flags, sd, dev).train()
    decay = Trainer.get_decay_parameter_names(None, m)
    if name == "stock":
        groups = [dict(params=[p for n, p in m.named_parameters() if n in decay], weight_decay=0.01),
                  dict(params=[p for n, p in m.named_parameters() if n not in decay], weight_decay=0.0)]
        opt = torch.optim.AdamW(groups, lr=1e-3)
    else:
        opt = AmdsegFusedAdamW(m, lr=1e-3, weight_decay=0.01, decay_names=decay, max_grad_norm=1.0)
    hist = []
    for step in range(3):
        for j in range(2):
            random.seed(step * 2 + j)
            loss = m(**mb((step * 2 + j) % 4))[0] / 2
            loss.backward()
        if name == "stock":
            gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        else:
            gn = opt.grad_norm(1.0)
        gn = float(gn)
        opt.step()
        m.zero_grad()
        hist.append((float(loss), gn, {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}))
    res[name] = hist
for step in range(3):
    a, b = res["stock"][step], res["fused"][step]
    print("step", step, "loss", a[0], b[0], "gn", a[1], b[1])
    worst = sorted(((float((a[2][k] - b[2][k]).abs().max()), k) for k in a[2]), reverse=True)[:5]
    print("  worst", worst)
