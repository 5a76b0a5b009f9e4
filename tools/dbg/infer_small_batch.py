"""where a small-batch inference step's wall time goes: encoder only (engine.forward in a loop, one sync at the end) vs the whole model forward"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import types
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = types.SimpleNamespace(model="bert", workload="full_da", precision="bf16", mode="infer", seq_len=512, seqs_per_gpu=B)
dev = torch.device("cuda:0")
model, cfg = bench.build(args, dev)
model.eval()
batches, pairs = bench.make_batches(args, 8, seed=0, device=dev)
eng = model.engine()
b = batches[0]
ids = torch.cat((b["input_ids"][:, 0], b["input_ids"][:, 1])); am = torch.cat((b["attention_mask"][:, 0], b["attention_mask"][:, 1])); tt = torch.zeros_like(ids)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
with torch.no_grad():
    print("B", B, "encoder only: host ms/call %.3f, wall ms/call %.3f" % t(lambda: eng.forward(ids, am, tt, False)))
    print("model forward:      host ms/call %.3f, wall ms/call %.3f" % t(lambda: model(**b)))
    os.environ["X"] = "1"
    eng.eval_weight_check = False
    print("encoder, no weight check: host %.3f wall %.3f" % t(lambda: eng.forward(ids, am, tt, False)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eng.forward(ids, am, tt, False)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out, _ = eng.forward(ids, am, tt, False)
        ref, _ = eng.forward(ids, am, tt, False)
        g.replay(); torch.cuda.synchronize()
        print("graph replay equals eager:", bool(torch.equal(out, ref)))
        print("encoder as a hipGraph: host %.3f wall %.3f" % t(lambda: g.replay()))
    except Exception as e:
        print("graph capture failed:", type(e).__name__, str(e)[:300])
