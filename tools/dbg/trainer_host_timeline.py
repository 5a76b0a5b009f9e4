"""where does the host spend a Trainer step?  wall time blocked in Event.synchronize (the label fetch), in the dataloader, in
_prepare_inputs (H2D), and the rest, per step of the bench's via_trainer leg"""
import argparse, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import transformers
import bench
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[key] += time.perf_counter() - t; cnt[key] += 1
    setattr(obj, name, g)
wrap(torch.cuda.Event, "synchronize", "event.synchronize")
wrap(transformers.Trainer, "_prepare_inputs", "_prepare_inputs")
wrap(transformers.Trainer, "training_step", "training_step(total)")
wrap(transformers.Trainer, "get_batch_samples", "get_batch_samples")
from spokennlp_amd import trainer as T
wrap(T.AmdsegFusedAdamW, "step", "optimizer.step")
wrap(T.AmdsegFusedAdamW, "grad_norm", "grad_norm")
args = argparse.Namespace(model="bert", workload="full_da", seq_len=512, seqs_per_gpu=32, mode="train", precision="bf16")
out = bench.via_trainer(args, torch.device("cuda:0"), nsteps=60, nwarm=8, nan_filter=True)
print(out["ms_per_step"], "ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {v / 68 * 1e3:8.3f} ms/step over {cnt[k] / 68:.1f} calls/step")
