#!/usr/bin/env python
"""Where a training step's wall time goes without a profiler attached: HIP events on the compute stream around every layer's forward and
backward call (they include whatever the stream waits for), the embeddings / heads in between, and the optimiser.
usage: CB_MODEL=longformer CB_L=2048 CB_B=4 python tools/dbg/layer_times.py"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


class A:
    model = os.environ.get("CB_MODEL", "bert"); seq_len = int(os.environ.get("CB_L", 512)); seqs_per_gpu = int(os.environ.get("CB_B", 32))
    workload = "full_da"; mode = "train"


def main():
    dev = torch.device("cuda", 0)
    args = A()
    model, cfg = bench.build(args, dev)
    eng = model.engine()
    batches, _ = bench.make_batches(args, 8, 0, dev)
    marks = []

    def ev(tag):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
    of, ob = type(eng)._layer_forward, type(eng)._layer_backward

    def lf(self, *a, **k):
        ev("f0"); r = of(self, *a, **k); ev("f1"); return r

    def lb(self, *a, **k):
        ev("b0"); r = ob(self, *a, **k); ev("b1"); return r
    eng._layer_forward = lf.__get__(eng); eng._layer_backward = lb.__get__(eng)

    def step(i):
        random.seed(i)
        ev("s0")
        loss = model(**batches[i % 8])[0]
        loss.backward()
        ev("o0")
        eng.adamw_step(5e-5, max_grad_norm=1.0)
        ev("s1")
    for i in range(6):
        step(i)
    torch.cuda.synchronize(); marks.clear()
    n = 12
    for i in range(n):
        step(i + 6)
    torch.cuda.synchronize()
    acc = {}
    for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
        key = t0 + ">" + t1
        acc.setdefault(key, [0.0, 0]); acc[key][0] += e0.elapsed_time(e1); acc[key][1] += 1
    tot = sum(v[0] for k, v in acc.items() if k != "s1>s0") / n
    print(f"step (s0..s1) {tot:.3f} ms; between steps {acc.get('s1>s0', [0, 1])[0] / max(1, n - 1):.3f} ms")
    for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:8s} {t / n:8.3f} ms/step  ({c / n:.0f} x {t / c * 1e3:7.1f} us)")


if __name__ == "__main__":
    main()
