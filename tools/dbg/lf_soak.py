"""soak of the Longformer two-stream order: the same model and batches stepped N times with the global-row chain on the second stream (default) and on
the compute stream; a missing stream dependency shows up as a loss that differs between the two runs at some step (they agree to the fp32 atomics
noise of the loss heads otherwise).  usage: python tools/dbg/lf_soak.py [steps]"""
import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    model = "longformer"; seq_len = 1024; seqs_per_gpu = 4; workload = "full_da"; mode = "train"


def run(overlap, steps):
    dev = torch.device("cuda", 0)
    args = A()
    torch.manual_seed(0)
    model, cfg = bench.build(args, dev)
    eng = model.engine()
    eng.lf_overlap = overlap
    batches, _ = bench.make_batches(args, 8, 0, dev)
    losses = []
    for i in range(steps):
        random.seed(i)
        loss = model(**batches[i % 8])[0]
        loss.backward()
        eng.adamw_step(2e-5, max_grad_norm=1.0)
        losses.append(loss.item())
    return losses


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
a = run(True, steps)
b = run(False, steps)
worst = max(abs(x - y) / max(abs(y), 1e-6) for x, y in zip(a, b))
first = next((i for i, (x, y) in enumerate(zip(a, b)) if abs(x - y) > 5e-3 * max(abs(y), 1e-6)), None)
print(f"{steps} steps: loss {a[0]:.4f} -> {a[-1]:.4f} (two streams) / {b[0]:.4f} -> {b[-1]:.4f} (one stream); worst relative difference {worst:.2e}; "
      f"first step differing by > 0.5 %: {first}")
