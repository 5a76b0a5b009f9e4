import sys, os, random, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_fullsize import _fullsize_case
from tests.test_gpu_model import build_model, to_dev
dev = torch.device("cuda:0")
z, sd, batch, arch, flags_of = _fullsize_case()
m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
random.seed(int(z["train_full.random_seed"]))
loss, logits, cos = m(**to_dev(batch, dev)); loss.backward()
params = dict(m.named_parameters())
rows = []
for n, v in zip(z["train_full.gradnorm_names"].tolist(), z["train_full.gradnorm_vals"].tolist()):
    if v <= 1e-6: continue
    gn = float(params[n].grad.float().norm()); rows.append((abs(gn - v) / v, n, gn, v))
rows.sort(reverse=True)
for r in rows[:6]: print(f"{r[0]:.4f} {r[1]} {r[2]:.4f} {r[3]:.4f}")
