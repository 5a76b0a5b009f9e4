import sys, random, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_bigbird import build_bb
from tests.test_oracle_golden import bb_case, flags_of
dev = torch.device("cuda:0")
def run(lazy, steps=3):
    z, sd, batch, arch = bb_case("bb_tiny_L1024")
    m = build_bb(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1)
    b = {k: v.to(dev) for k, v in batch.items()}
    m.config.amdseg_deterministic = True
    m.train(); m.amdseg_seed = 11; random.seed(3)
    losses = []
    for _ in range(steps):
        loss = m(**b)[0]
        m.engine().lazy_zero = lazy
        loss.backward(); losses.append(loss.item())
        m.engine().adamw_step(1e-3, max_grad_norm=1.0)
    torch.cuda.synchronize()
    eng = m.engine()
    return losses, eng.fp.flat_p.detach().clone(), eng
a = run(True); b = run(False); c = run(False)
print("losses lazy", a[0]); print("losses eager", b[0]); print("losses eager2", c[0])
print("lazy-eager max", float((a[1]-b[1]).abs().max()), "eager-eager max", float((b[1]-c[1]).abs().max()))
eng = a[2]; fp = eng.fp
d = (a[1]-b[1]).abs()
for n, o in fp.offsets.items():
    k = fp.params[n].numel()
    mx = float(d[o:o+k].max())
    if mx > 0: print(n, mx)
