"""the side-stream global-row chain must not change a bit: same model / batch with lf_overlap on and off, 3 train steps each"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import argparse, random, torch
import bench
dev = torch.device("cuda:0")
args = argparse.Namespace(model="longformer", workload="full_da", seq_len=2048, seqs_per_gpu=4, mode="train", precision="bf16")
res = {}
for ov in (False, False, True, True):
    model, cfg = bench.build(args, dev)
    eng = model.engine()
    eng.lf_overlap = ov
    if ov and eng._lf_side is None:
        eng._lf_side = torch.cuda.Stream(device=dev)
    batches, _ = bench.make_batches(args, 2, seed=0, device=dev)
    outs = []
    for i in range(3):
        random.seed(i)
        loss = model(**batches[i % 2])[0]
        loss.backward()
        outs.append((loss.detach().clone(), eng.fp.flat_g.clone()))
        eng.adamw_step(1e-4, max_grad_norm=1.0)
    torch.cuda.synchronize()
    res.setdefault(ov, []).append(outs)
a0, a1, b0, b1 = res[False][0], res[False][1], res[True][0], res[True][1]
names = eng.fp.offsets
def where(idx):
    best = None
    for n, o in names.items():
        if o <= idx and (best is None or o > names[best]):
            best = n
    return best
for i in range(3):
    for name, u, v in (("off vs off", a0, a1), ("off vs on ", a0, b0), ("on vs on  ", b0, b1)):
        dl = float((u[i][0] - v[i][0]).abs())
        d = (u[i][1] - v[i][1]).abs()
        k = int(d.argmax())
        print(f"step {i} {name}: |dloss| = {dl:.3e}, max |dgrad| = {float(d[k]):.3e} at {where(k)} (value {float(u[i][1][k]):.3e})")
