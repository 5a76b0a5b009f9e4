"""PoNet path on the GPU against oracle/ponet_oracle.py -- the oracle restates the published algorithm;
the original encoder source is not in the reference tree, so this parity is against OUR statement only (unpinned)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

ARCH = dict(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
            max_position_embeddings=256, type_vocab_size=2)


def make_inputs(B, L, seed, long_run=False):
    """token ids, mask (ragged padding), monotone segment ids (CLS = 0, sentences 1.., pad = last + 1), labels at run ends"""
    r = random.Random(seed)
    ids = torch.zeros(B, L, dtype=torch.long); am = torch.zeros(B, L, dtype=torch.long)
    seg = torch.zeros(B, L, dtype=torch.long); lab = torch.full((B, L), -100, dtype=torch.long)
    for b in range(B):
        n = L if b == 0 else r.randrange(L // 2, L - 3)
        ids[b, :n] = torch.randint(5, 299, (n,)); am[b, :n] = 1
        pos, s = 1, 1
        while pos < n:
            ln = r.randrange(1, 12) if not (long_run and s == 2) else min(150, n - pos)
            e = min(pos + ln, n)
            seg[b, pos:e] = s
            lab[b, e - 1] = r.randrange(2)
            pos, s = e, s + 1
        seg[b, n:] = s
    return ids, am, seg, lab


@pytest.mark.parametrize("B,L,long_run", [(2, 64, False), (2, 256, True)])
def test_pooling_kernels_vs_oracle(dev, B, L, long_run):
    from oracle import ponet_oracle as PO
    from spokennlp_amd import ops
    torch.manual_seed(L)
    H, nh = 128, 2
    _, am, seg, _ = make_inputs(B, L, 3, long_run)
    proj = torch.randn(B * L, 5 * H).bfloat16()
    hq, hk, ho, hl, hs = [proj[:, k * H:(k + 1) * H].float().view(B, L, H).requires_grad_(True) for k in range(5)]
    valid = am == 1
    ctx_ref = PO.pooling(hq, hk, ho, hl, hs, valid, seg, nh)
    dctx = (torch.randn(B * L, H) * 0.5).bfloat16()
    ctx_ref.backward(dctx.float().view(B, L, H))
    # device side: only the segment / local / fusion part; g comes from the oracle's global aggregate
    with torch.no_grad():
        vf = valid.float()
        qbar = (hq * vf[..., None]).sum(1) / vf.sum(1, keepdim=True)
        a = torch.einsum("bhe,bjhe->bhj", qbar.view(B, nh, 64), hk.view(B, L, nh, 64)) / 8.0
        p = torch.softmax(a.masked_fill(~valid[:, None, :], float("-inf")), -1)
        g = torch.einsum("bhj,bjhe->bhe", p, hk.view(B, L, nh, 64)).reshape(B, H)
    pos = torch.arange(L).expand(B, L)
    diff = seg[:, 1:] != seg[:, :-1]
    one = torch.ones(B, 1, dtype=torch.bool)
    rs = torch.cummax(torch.where(torch.cat((one, diff), 1), pos, torch.zeros_like(pos)), 1).values.int().reshape(-1).to(dev)
    re = torch.flip(torch.cummin(torch.flip(torch.where(torch.cat((diff, one), 1), pos, torch.full_like(pos, L - 1)), (1,)), 1).values, (1,)).int().reshape(-1).to(dev)
    mb = ((1 - am.float()) * -1e30).to(dev)
    projd = proj.to(dev)
    part = torch.empty(3 * B * L, H, dtype=torch.bfloat16, device=dev); parg = torch.empty(3 * B * L, H, dtype=torch.int16, device=dev)
    ctx = torch.empty(B * L, H, dtype=torch.bfloat16, device=dev)
    ops.ponet_pool_fwd(projd, mb, rs, re, g.to(dev), part, parg, ctx, B, L, H)
    err = (ctx.float().cpu().view(B, L, H) - ctx_ref.detach()).abs()
    assert (err <= 0.01 * ctx_ref.detach().abs() + 0.02).all(), err.max().item()
    assert (ctx.float().cpu().view(B, L, H)[~valid] == 0).all()
    dproj = torch.full((B * L, 5 * H), 7.0, dtype=torch.bfloat16, device=dev)
    E = torch.empty(B * L, H, dtype=torch.bfloat16, device=dev); psum = torch.empty(3 * B * L, H, dtype=torch.float32, device=dev)
    ops.ponet_pool_bwd(projd, mb, rs, re, g.to(dev), part, parg, dctx.to(dev), dproj, E, psum, B, L, H)
    d = dproj.float().cpu()
    # bf16-quantised inputs tie now and then; torch.maximum / amax split the gradient among tied maxima while the kernels
    # route it to the first one -- compare only where the maximum is unique
    ninf = float("-inf")
    hlv = torch.where(valid[..., None], hl.detach(), torch.full((), ninf))
    padr = torch.full((B, 1, H), ninf)
    w3 = torch.stack((torch.cat((padr, hlv[:, :-1]), 1), hlv, torch.cat((hlv[:, 1:], padr), 1)))          # [3, B, L, H]
    tie_w = (w3 == w3.amax(0, keepdim=True)).sum(0) > 1                                                  # window of n has a tie
    tie_l = tie_w | torch.cat((tie_w[:, 1:], tie_w[:, -1:]), 1) | torch.cat((tie_w[:, :1], tie_w[:, :-1]), 1)
    same = (seg[:, :, None] == seg[:, None, :]) & valid[:, None, :] & valid[:, :, None]
    hsv = hs.detach()
    tie_s = torch.zeros(B, L, H, dtype=torch.bool)
    for b in range(B):
        vals = torch.where(same[b][:, :, None], hsv[b][None], torch.full((), ninf))                     # [n, j, H]
        tie_s[b] = (vals == vals.amax(1, keepdim=True)).sum(1) > 1
    ok = {2: torch.ones(B, L, H, dtype=torch.bool), 3: ~tie_l, 4: ~tie_s}
    for k, ref in ((2, ho.grad), (3, hl.grad), (4, hs.grad)):
        got = d[:, k * H:(k + 1) * H].view(B, L, H)
        e = (got - ref).abs()
        assert ok[k].float().mean() > 0.5
        assert (e <= 0.02 * ref.abs() + 0.05)[ok[k]].all(), (k, e[ok[k]].max().item())
    # E = dctx * Ho on valid rows
    eref = dctx.float().view(B, L, H) * ho.detach() * valid[..., None]
    assert ((E.float().cpu().view(B, L, H) - eref).abs() <= 0.01 * eref.abs() + 0.02).all()


def build(dev, sd=None, dropout=0.0):
    from spokennlp_amd.ponet import PoNetForTokenClassification, PoNetConfig
    cfg = PoNetConfig(num_labels=2, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, **ARCH)
    torch.manual_seed(0)
    m = PoNetForTokenClassification(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
            elif "LayerNorm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "classifier.weight" in n:
                p.normal_(0, 0.3)
            elif p.dim() == 2 and "embeddings" not in n:
                p.normal_(0, 0.08)
    return m, cfg


@pytest.mark.parametrize("L,long_run", [(64, False), (256, True)])
def test_model_vs_oracle(dev, L, long_run):
    from oracle import ponet_oracle as PO
    from oracle import bert_ts_oracle as O
    m, cfg = build(dev)
    sd = {k: v.detach().clone().float().requires_grad_(True) for k, v in m.state_dict().items()}
    ids, am, seg, lab = make_inputs(2, L, 11, long_run)
    ocfg = O.make_cfg(num_labels=2, **ARCH)
    loss_o, logits_o = PO.token_classification_forward(sd, ocfg, ids, am, torch.zeros_like(ids), seg, lab)
    loss_o.backward()
    m = m.to(dev).train()
    out = m(input_ids=ids.to(dev), attention_mask=am.to(dev), token_type_ids=torch.zeros_like(ids).to(dev), segment_ids=seg.to(dev),
            labels=lab.to(dev), return_dict=False)
    loss, logits = out[0], out[1]
    loss.backward()
    valid = am == 1
    d = (logits.detach().cpu() - logits_o.detach()).abs()[valid].max().item()
    print(f"ponet L={L}: max|dlogit| {d:.4f} (max|logit| {logits_o.abs().max().item():.2f}) loss {loss.item():.4f} vs {loss_o.item():.4f}")
    assert d < 0.02 * logits_o.abs().max().item() + 0.05
    assert abs(loss.item() - loss_o.item()) < 0.03
    assert (logits.detach().cpu()[valid].argmax(-1) == logits_o.detach()[valid].argmax(-1)).float().mean().item() > 0.98
    worst = 1.0
    for n, p in m.named_parameters():
        go = sd[n].grad
        if go is None or float(go.norm()) < 1e-6:
            continue
        c = torch.nn.functional.cosine_similarity(p.grad.float().cpu().flatten(), go.flatten(), dim=0).item()
        worst = min(worst, c)
        assert c > 0.98, (n, c)
    print("worst grad cosine", worst)
    # inference path (eval, no grad) agrees with the training forward at dropout 0
    m.eval()
    with torch.no_grad():
        lg = m(input_ids=ids.to(dev), attention_mask=am.to(dev), segment_ids=seg.to(dev), return_dict=True).logits
    assert (lg - logits.detach()).abs().max().item() < 1e-3


def test_dropout_step_deterministic(dev):
    vals = []
    ids, am, seg, lab = make_inputs(2, 64, 5)
    for _ in range(2):
        m, _ = build(dev, dropout=0.1)
        m = m.to(dev).train(); m.amdseg_seed = 3
        loss = m(input_ids=ids.to(dev), attention_mask=am.to(dev), segment_ids=seg.to(dev), labels=lab.to(dev), return_dict=False)[0]
        loss.backward()
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters())).item()
        assert np.isfinite(loss.item()) and np.isfinite(gn)
        vals.append((loss.item(), gn))
    assert vals[0][0] == vals[1][0]
    assert abs(vals[0][1] - vals[1][1]) <= 1e-6 * vals[0][1]          # embedding-table grads use fp32 atomics (order may vary)
