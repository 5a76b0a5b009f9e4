#!/usr/bin/env python
"""Which bf16 roundings of the fast path carry its logit error?  (verdict r02, weak point 1 / next-round item 6: fp32 residual stream.)

CPU study on the bert-base L = 512 golden case (tests/golden/bert_base_L512.npz, weights regenerated from their seed): the oracle's encoder
with a bf16 rounding inserted at each point where the HIP bf16 path stores or feeds bf16 (csrc/api.hip amdseg_bert_layer_fwd): GEMM operands
(activations and weights), qkv / ctx / dense outputs / u / h stores, attention probabilities, the pre-LayerNorm sums z1 / z2 and the LayerNorm
outputs x1 / x_out (the residual stream).  Reports max |dlogit| of the classifier over the valid tokens against the REFERENCE's golden logits
for: everything on (the fast path), the residual stream kept in fp32 (its bf16 image only feeds the GEMMs), and single sources alone.
Test infrastructure: imports oracle/, runs on CPU.   python tools/bf16_noise_study.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bert_ts_oracle as O          # noqa: E402
from tests.util import tiny_state_dict          # noqa: E402


def r(t, on):
    return t.bfloat16().float() if on else t


def encode(sd, cfg, ids, am, tt, k):
    """k: dict of switches -- w (weights), a (GEMM activation operands), o (GEMM outputs qkv / ctx / dense / u / h), p (probabilities),
    z (pre-LN sums), s (LayerNorm outputs = the residual stream as STORED; with s off and a on the stream is fp32 and only its image is rounded)"""
    pfx = "bert."
    mask_bias = (1.0 - am.float())[:, None, None, :] * -30000.0
    x = O.embeddings(sd, cfg, ids, tt, pfx)
    x = r(x, k["s"])
    nh = cfg.num_attention_heads
    for i in range(cfg.num_hidden_layers):
        p = f"{pfx}encoder.layer.{i}."
        B, L, H = x.shape
        d = H // nh

        def lin(t, name):
            return r(t, k["a"]) @ r(sd[p + name + ".weight"], k["w"]).t() + sd[p + name + ".bias"]

        q = r(lin(x, "attention.self.query"), k["o"]).view(B, L, nh, d).transpose(1, 2)
        kk = r(lin(x, "attention.self.key"), k["o"]).view(B, L, nh, d).transpose(1, 2)
        v = r(lin(x, "attention.self.value"), k["o"]).view(B, L, nh, d).transpose(1, 2)
        s = q @ kk.transpose(-1, -2) * (d ** -0.5) + mask_bias
        pr = r(torch.softmax(s, dim=-1), k["p"])
        ctx = r((pr @ v).transpose(1, 2).reshape(B, L, H), k["o"])
        z1 = r(lin(ctx, "attention.output.dense"), k["o"]) + x
        x1 = O.layer_norm(z1, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
        x1 = r(x1, k["s"])
        u = lin(x1, "intermediate.dense")
        h = r(O.gelu_erf(u), k["o"])
        z2 = r(lin(h, "output.dense"), k["o"]) + x1
        x = O.layer_norm(z2, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], cfg.layer_norm_eps)
        x = r(x, k["s"])
    return x


def main():
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    z = np.load(os.path.join(ROOT, "tests", "golden", "bert_base_L512.npz"), allow_pickle=False)
    arch = dict(zip(z["arch_keys"].tolist(), [int(v) for v in z["arch_vals"].tolist()]))
    sd = tiny_state_dict(arch, seed=int(z["seed"]), std=float(z["std"]))
    cfg = O.make_cfg(num_labels=2, **arch)
    ids = torch.from_numpy(z["in.input_ids"])[:, 0]
    am = torch.from_numpy(z["in.attention_mask"])[:, 0]
    tt = torch.from_numpy(z["in.token_type_ids"])[:, 0]
    ref = torch.from_numpy(z["full_eval.logits"])[:, 0]
    W, b = sd["loss_calculator.classifier.weight"] if "loss_calculator.classifier.weight" in sd else sd["classifier.weight"], None
    bname = "loss_calculator.classifier.bias" if "loss_calculator.classifier.bias" in sd else "classifier.bias"
    b = sd[bname]
    valid = am.bool()
    allon = dict(w=1, a=1, o=1, p=1, z=0, s=1)
    cases = [("fp32 (no rounding)", dict(w=0, a=0, o=0, p=0, z=0, s=0)),
             ("fast path: every bf16 store / operand", allon),
             ("... with the residual stream kept in fp32 (bf16 image feeds the GEMMs)", dict(allon, s=0)),
             ("... and the weights exact too (what a second product x . W_lo per GEMM would approach: 2x the GEMM flops)", dict(allon, s=0, w=0)),
             ("fast path with exact weights only (bf16 residual stream)", dict(allon, w=0)),
             ("only the residual stream stored in bf16", dict(w=0, a=0, o=0, p=0, z=0, s=1)),
             ("only GEMM activation operands bf16", dict(w=0, a=1, o=0, p=0, z=0, s=0)),
             ("only weights bf16", dict(w=1, a=0, o=0, p=0, z=0, s=0)),
             ("only GEMM / attention outputs stored bf16 (qkv, ctx, dense, h)", dict(w=0, a=0, o=1, p=0, z=0, s=0)),
             ("only attention probabilities bf16", dict(w=0, a=0, o=0, p=1, z=0, s=0))]
    print(f"bert-base L=512 golden case, {ids.shape[0]} sequences; max |logit| of the reference {ref[valid].abs().max():.2f}")
    with torch.no_grad():
        for name, k in cases:
            x = encode(sd, cfg, ids, am, tt, k)
            lg = x @ W.t() + b
            dl = (lg - ref)[valid].abs()
            print(f"{name:78s} max|dlogit| {dl.max().item():.4f}  mean {dl.mean().item():.5f}")


if __name__ == "__main__":
    main()
