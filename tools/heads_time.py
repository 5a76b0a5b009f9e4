#!/usr/bin/env python
"""GPU time between the end of the encoder forward and the start of the encoder backward (= heads forward + loss + heads backward,
including any idle time the GPU spends waiting for the host there), untraced, at the bench.py workload."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    args = argparse.Namespace(model="bert", seq_len=512, seqs_per_gpu=32, workload="full_da", mode="train", layers=None)
    dev = torch.device("cuda:0")
    ap_defaults = dict(steps=20, warmup=5)
    model, cfg = bench.build(args, dev)
    eng = model.engine()
    batches, _ = bench.make_batches(args, 8, seed=0, device=dev)
    ev = {}
    f0, b0 = eng.forward, eng.backward

    def fwd(*a, **k):
        out = f0(*a, **k)
        ev["f_end"] = torch.cuda.Event(enable_timing=True); ev["f_end"].record()
        return out

    def bwd(*a, **k):
        ev["b_start"] = torch.cuda.Event(enable_timing=True); ev["b_start"].record()
        return b0(*a, **k)
    eng.forward, eng.backward = fwd, bwd
    pairs = []
    for i in range(ap_defaults["steps"] + ap_defaults["warmup"]):
        random.seed(i)
        loss = model(**batches[i % len(batches)])[0]
        loss.backward()
        eng.adamw_step(5e-5, max_grad_norm=1.0)
        if i >= ap_defaults["warmup"]:
            pairs.append((ev["f_end"], ev["b_start"]))          # no synchronisation inside the loop: same pipelining as bench.py
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    print(f"heads section (GPU time, encoder forward end -> encoder backward start): median {ts[len(ts) // 2]:.3f} ms, min {ts[0]:.3f}, max {ts[-1]:.3f}")


if __name__ == "__main__":
    main()
