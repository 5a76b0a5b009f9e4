import os, sys, random, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.test_oracle_golden import load_case, flags_of
from tests.test_gpu_model import build_model
from tests.test_gpu_trainer import _samples, _DS, _args
from transformers import Trainer as HFTrainer, default_data_collator, TrainerCallback
from spokennlp_amd.trainer import Trainer

dev = torch.device("cuda:0")
z, sd, batch, arch = load_case("tiny_L64")
flags = flags_of(z, "train_full")
ds = _DS(_samples(arch))
res = {}
class CB(TrainerCallback):
    def __init__(self, m, tag): self.m, self.tag = m, tag
    def on_pre_optimizer_step(self, args, state, control, **kw):
        eng = self.m.engine()
        res[self.tag + "_g"] = eng.fp.flat_g.detach().cpu().clone()
        res[self.tag + "_p0"] = eng.fp.flat_p.detach().cpu().clone()
    def on_optimizer_step(self, args, state, control, **kw):
        eng = self.m.engine()
        res[self.tag + "_p1"] = eng.fp.flat_p.detach().cpu().clone()
for name, cls in (("stock", HFTrainer), ("fused", Trainer)):
    m = build_model(arch, flags, sd, dev)
    random.seed(3)
    tr = cls(model=m, args=_args("/tmp/o_" + name, max_steps=1), train_dataset=ds, data_collator=default_data_collator, callbacks=[CB(m, name)])
    tr.train()
    res[name + "_names"] = m.engine().fp.offsets
for k in ("_g", "_p0", "_p1"):
    a, b = res["stock" + k], res["fused" + k]
    print(k, "max diff", float((a - b).abs().max()), "norms", float(a.norm()), float(b.norm()))
d = (res["stock_p1"] - res["fused_p1"]).abs()
offs = res["stock_names"]
names = list(offs.keys())
for i, n in enumerate(names):
    o = offs[n]; e = offs[names[i + 1]] if i + 1 < len(names) else d.numel()
    mx = float(d[o:e].max())
    if mx > 1e-5:
        ds_ = (res["stock_p1"] - res["stock_p0"])[o:e].abs().max(); df_ = (res["fused_p1"] - res["fused_p0"])[o:e].abs().max()
        print(f"{n}: diff {mx:.2e} stock moved {float(ds_):.2e} fused moved {float(df_):.2e}")
