"""amdseg_bert_cfg.pad_guard: backward GEMMs drop the rows of trailing padding.  Same batch, skip on / off: parameter gradients compared
bit for bit, per-kernel launch times printed.   python tools/dbg/padrows_probe.py [bert|longformer]"""
import os, sys, argparse, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "bert"
args = argparse.Namespace(model=fam, workload="full_da", seq_len=512 if fam == "bert" else 4096, seqs_per_gpu=32 if fam == "bert" else (8 if fam == "ponet" else 4),
                          mode="train", precision="bf16")
model, cfg = bench.build(args, dev)
eng = model.engine()
batches, _ = bench.make_batches(args, 2, seed=0, device=dev)
b = batches[0]
res = {}
for skip in (True, False, True):
    eng.skip_padded_rows_bwd = skip
    for _ in range(3):
        model.zero_grad(set_to_none=False)
        random.seed(5); model._step_seed = 100
        loss = model(**b)[0]; loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    bench.prof_arm()
    for _ in range(5):
        loss = model(**b)[0]; loss.backward()
    k = bench.prof_collect(5)
    print(f"skip {skip}: loss {loss.item():.6f} guard {None if eng._pad_guard is None else int(eng._pad_guard.item())} " +
          " ".join(f"{n.replace('_kernel','')} {v['avg_launch_us']:.1f}" for n, v in k.items()))
    res.setdefault(skip, []).append(g)
def cmp(ga_, gb_, tag):
    bad, worst = 0, 0.0
    for n, ga in ga_.items():
        gb = gb_[n]
        if not torch.equal(ga, gb):
            bad += 1; worst = max(worst, float((ga - gb).abs().max()) / max(1e-12, float(gb.abs().max())))
    print(f"{tag}: {bad} of {len(ga_)} parameters differ, worst max-abs / max = {worst:.3e}")
    rows = sorted(((float((ga_[n] - gb_[n]).abs().max()) / max(1e-12, float(gb_[n].abs().max())), float(gb_[n].abs().max()), n) for n in ga_), reverse=True)
    for r, mx, n in rows[:12]:
        print(f"    {r:.3e} (max {mx:.3e}) {n}")
cmp(res[True][0], res[True][1], "skip vs skip (run-to-run noise of the atomic scatters)")
cmp(res[True][0], res[False][0], "skip vs dense")

# the zero-row claim, looked at directly: the workspaces of the last (dense) backward
eng.skip_padded_rows_bwd = False
model.zero_grad(set_to_none=False)
loss = model(**b)[0]; loss.backward(); torch.cuda.synchronize()
for key, A in eng._arenas.items():
    if not key[2] or "kend" not in A:
        continue
    B_, L_ = key[0], key[1]
    pos = torch.arange(L_, device=dev)[None, :] >= A["kend"][:, None].long()
    print("padded rows:", int(pos.sum()), "of", B_ * L_, "padded 64-tiles:", int((torch.arange(0, L_, 64, device=dev)[None, :] >= A["kend"][:, None].long()).sum()))
    ws = A["ws"]
    for name in ("dqkv", "du", "dctx", "dz1", "dz2", "dx1"):
        t = ws.get(name) if isinstance(ws, dict) else None
        if t is None or not torch.is_tensor(t):
            continue
        rows = t.reshape(B_ * L_, -1)[pos.reshape(-1)]
        print(f"  ws.{name}: padded rows max abs {float(rows.float().abs().max()):.3e}, nonzero elements {int((rows != 0).sum())}")
