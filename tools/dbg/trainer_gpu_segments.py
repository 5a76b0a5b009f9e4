"""GPU-time segments of a Trainer step from CUDA events (no profiler): forward, backward, clip+optimiser, and the stretch from the end of
the optimiser to the start of the next forward (anything there is GPU idle or Trainer-side kernels)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import transformers
import bench
from spokennlp_amd import trainer as T
from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
f0 = M.forward
import functools
@functools.wraps(f0)
def fwd(self, *a, **k):
    ev("fwd0"); out = f0(self, *a, **k); ev("fwd1"); return out
M.forward = fwd
ts0 = transformers.Trainer.training_step
def ts(self, *a, **k):
    out = ts0(self, *a, **k); ev("bwd1"); return out
transformers.Trainer.training_step = ts
st0 = T.AmdsegFusedAdamW.step
def st(self, *a, **k):
    out = st0(self, *a, **k); ev("opt1"); return out
T.AmdsegFusedAdamW.step = st
args = argparse.Namespace(model="bert", workload="full_da", seq_len=512, seqs_per_gpu=32, mode="train", precision="bf16")
out = bench.via_trainer(args, torch.device("cuda:0"), nsteps=40, nwarm=8, nan_filter=True)
torch.cuda.synchronize()
print(out["ms_per_step"], "ms/step")
import collections
seg = collections.defaultdict(list)
for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
    seg[f"{t0}->{t1}"].append(e0.elapsed_time(e1))
for k, v in seg.items():
    v = v[len(v) // 3:]
    print(f"{k:12s} {sum(v) / len(v):8.3f} ms  (n={len(v)})")
