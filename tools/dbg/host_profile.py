#!/usr/bin/env python
"""where the HOST time of a small-batch training step goes (cProfile over 30 steps of bert-base 8 x 512 after warm-up); python tools/dbg/host_profile.py"""
import argparse, cProfile, os, pstats, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
args = argparse.Namespace(model="bert", seq_len=512, seqs_per_gpu=int(os.environ.get("SEQS", 8)), workload="full_da", mode="train", precision="bf16")
dev = torch.device("cuda:0")
model, cfg = bench.build(args, dev)
eng = model.engine()
batches, _ = bench.make_batches(args, 4, seed=0, device=dev)
def step(i):
    random.seed(i)
    loss = model(**batches[i % 4])[0]
    loss.backward()
    eng.finish_grad_sync()
    eng.adamw_step(5e-5, max_grad_norm=1.0, grad_scale=1.0)
for i in range(10): step(i)
torch.cuda.synchronize()
# host-only issue time: the GPU is given a long head start of nothing to do -> measure how long the host needs to ENQUEUE steps (queue never full: few steps)
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) / 3 * 1e3, (t2 - t0) / 3 * 1e3))
print("host enqueue ms/step vs wall ms/step (3-step bursts):", [(round(a, 2), round(b, 2)) for a, b in ts])
pr = cProfile.Profile(); pr.enable()
for i in range(30): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
