import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_mmvts_text import mmvts_case, build
dev = torch.device("cuda:0")
z, sd, ins = mmvts_case("mmvts_text_lf_L256")
m = build("lf", sd, dev, "bf16").eval()
with torch.no_grad():
    f = m(ins["input_ids"].to(dev), attention_mask=ins["attention_mask"].to(dev), token_type_ids=ins["token_type_ids"].to(dev))
eng = m.engine() if hasattr(m, "engine") else m._engine
print("windows", eng.windows, "L", ins["input_ids"].shape, "valid per seq", ins["attention_mask"].sum(1).tolist())
for key, A in eng._arenas.items():
    print(key)
    for i, la in enumerate(A["layers"]):
        for k in ("qkv", "ctx", "z1", "x1", "h", "z2"):
            t = la[k].float()
            bad = torch.isnan(t).any(1).nonzero().flatten()
            if bad.numel():
                print(" layer", i, k, "nan rows", bad[:10].tolist(), "count", bad.numel()); break
    print(" x0 nan", torch.isnan(A["x"][0].float()).any().item())
print("out nan rows", torch.isnan(f).any(-1).nonzero()[:10].tolist())
