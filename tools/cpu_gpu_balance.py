#!/usr/bin/env python
"""Is the train step launch-bound?  CPU time to ENQUEUE one step (no synchronisation) vs GPU time of the step (events)."""
import random
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    model = os.environ.get("CB_MODEL", "bert"); seq_len = int(os.environ.get("CB_L", 512)); seqs_per_gpu = int(os.environ.get("CB_B", 32))
    workload = "full_da"; mode = "train"


def main():
    # split the forward's CPU time into "blocked on the label copy (= previous step still running)" and the rest
    waits = []
    _sync = torch.cuda.Event.synchronize

    def timed_sync(self):
        t = time.perf_counter(); _sync(self); waits.append(time.perf_counter() - t)
    torch.cuda.Event.synchronize = timed_sync
    dev = torch.device("cuda", 0)
    args = A()
    model, cfg = bench.build(args, dev)
    eng = model.engine()
    batches, _ = bench.make_batches(args, 8, 0, dev)

    def step(i):
        random.seed(i)
        loss = model(**batches[i % 8])[0]
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        eng.adamw_step(5e-5, max_grad_norm=1.0)
        return t1, t2

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    cpu = []
    ev[0].record()
    for i in range(n):
        t0 = time.perf_counter()
        t1, t2 = step(i + 5)
        t3 = time.perf_counter()
        ev[i + 1].record()
        cpu.append((t1 - t0, t2 - t1, t3 - t2))
    torch.cuda.synchronize()
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    f = sum(c[0] for c in cpu) / n * 1e3; b = sum(c[1] for c in cpu) / n * 1e3; o = sum(c[2] for c in cpu) / n * 1e3
    w = sum(waits[-n:]) / n * 1e3
    print(f"forward CPU time blocked in Event.synchronize: {w:.2f} ms per step -> host planning + head enqueue = {f - w:.2f} ms")
    print(f"CPU enqueue per step: forward {f:.2f} ms, backward {b:.2f} ms, optimiser {o:.2f} ms, total {f + b + o:.2f} ms;  GPU per step {sum(gpu) / n:.2f} ms")


if __name__ == "__main__":
    main()
