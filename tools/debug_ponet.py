import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_ponet import build, make_inputs, ARCH
from oracle import ponet_oracle as PO, bert_ts_oracle as O
dev = torch.device("cuda")
m, cfg = build(dev)
sd = {k: v.detach().clone().float() for k, v in m.state_dict().items()}
ids, am, seg, lab = make_inputs(2, 64, 11)
ocfg = O.make_cfg(num_labels=2, **ARCH)
with torch.no_grad():
    x, hs = PO.ponet_encode(sd, ocfg, ids, am, torch.zeros_like(ids), seg, return_all=True)
m = m.to(dev).train()
out = m(input_ids=ids.to(dev), attention_mask=am.to(dev), segment_ids=seg.to(dev), labels=lab.to(dev), return_dict=False)
eng = m.engine()
A = eng._arena(2, 64, True)
valid = (am == 1)
for i in range(3):
    d = (A["x"][i].float().cpu().view(2, 64, -1) - hs[i]).abs()[valid].max().item()
    print("hidden", i, d, hs[i].abs().max().item())
# layer 0 internals
la = A["layers"][0]
p = "ponet.encoder.layer.0."
x0 = hs[0]
proj_ref = torch.cat([x0 @ sd[p + f"attention.self.{n}.weight"].t() + sd[p + f"attention.self.{n}.bias"] for n in PO.PROJ], -1)
print("proj", (la["qkv"].float().cpu().view(2, 64, -1) - proj_ref).abs()[valid].max().item())
hq, hk, ho, hl, hsg = proj_ref.split(128, -1)
ctx_ref = PO.pooling(hq, hk, ho, hl, hsg, valid, seg, 2)
print("ctx", (la["ctx"].float().cpu().view(2, 64, -1) - ctx_ref).abs().max().item(), ctx_ref.abs().max().item())
got = la["qkv"].float().cpu().view(2, 64, -1)
for k in range(5):
    print("block", k, (got[..., k*128:(k+1)*128] - proj_ref[..., k*128:(k+1)*128]).abs()[valid].max().item())
fp = eng.fp
for k, n in enumerate(PO.PROJ):
    w = fp.view(fp.flat_p, p + f"attention.self.{n}.weight")
    print(n, "flat==sd", torch.equal(w.cpu(), sd[p + f"attention.self.{n}.weight"]), fp.offsets[p + f"attention.self.{n}.weight"], fp.offsets[p + f"attention.self.{n}.bias"])
wq = fp.view(fp.flat_p, p + "attention.self.dense_q.weight", (640, 128))
sh = fp.view(eng.shadow, p + "attention.self.dense_q.weight", (640, 128))
print("shadow err", (sh.float() - wq).abs().max().item())
x_in = A["x"][0].float()
ref2 = x_in @ wq.t() + fp.view(fp.flat_p, p + "attention.self.dense_q.bias", (640,))
print("gemm vs flat", (la["qkv"].float() - ref2).abs().max().item())
