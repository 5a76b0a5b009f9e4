"""BigBird path on the GPU: the block-list attention kernels against a plain torch restatement, the gelu_new GEMM
epilogues, and the HF-surface wrapper end to end against golden vectors produced by the reference (tests/golden/bb_*.npz)."""
import math
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from tests.test_oracle_golden import bb_case, flags_of  # noqa: E402


# ---------------------------------------------------------------------------------------------------- kernel level
def list_reference(qkv, mask_bias, B, L, heads, klist, kcnt, dctx=None):
    """plain torch fp32: for every (head, query block) ONE softmax over the listed key blocks, duplicates included"""
    H = heads * 64
    qkv = qkv.float().clone().requires_grad_(dctx is not None)
    q, k, v = [t.view(B, L, heads, 64).transpose(1, 2) for t in qkv.view(B, L, 3 * H).split(H, dim=-1)]
    nb = L // 64
    rows = []
    for h in range(heads):
        blocks = []
        for i in range(nb):
            idx = torch.cat([torch.arange(int(kb) * 64, int(kb) * 64 + 64) for kb in klist[h, i, :int(kcnt[h, i])]]).to(qkv.device)
            s = q[:, h, i * 64:(i + 1) * 64] @ k[:, h, idx].transpose(-1, -2) * 0.125 + mask_bias.view(B, 1, L)[:, :, idx]
            blocks.append(torch.softmax(s, -1) @ v[:, h, idx])
        rows.append(torch.cat(blocks, dim=1))
    ctx = torch.stack(rows, dim=1).transpose(1, 2).reshape(B * L, H)
    ctx = ctx * (mask_bias.reshape(B * L, 1) >= 0)                   # rows of padded queries are zeroed (context_layer * from_mask)
    if dctx is None:
        return ctx
    ctx.backward(dctx.float())
    return ctx.detach(), qkv.grad


@pytest.mark.parametrize("B,L,heads,train", [(2, 1024, 2, False), (2, 1024, 2, True), (1, 768, 4, True), (1, 2048, 2, True),
                                             (1, 4096, 2, False)])   # eval lists at L = 4096: key block 0 is visited 250 times (list reload path)
def test_list_attention_fwd_bwd(dev, B, L, heads, train):
    from spokennlp_amd import ops, bigbird_plan
    torch.manual_seed(L + heads)
    H = heads * 64
    t = bigbird_plan.build(L, heads, 3, seed=1, training=train, max_seqlen=4096)
    klist, kcnt, qlist, qcnt, korder, qorder = [torch.from_numpy(t[k]).to(dev) for k in ("klist", "kcnt", "qlist", "qcnt", "korder", "qorder")]
    qkv = torch.randn(B * L, 3 * H, device=dev).bfloat16()
    mask = torch.zeros(B, L, device=dev)
    mask[0, L - 37:] = -10000.0                                   # padded tail in the first sequence, the reference's penalty
    dctx = (torch.randn(B * L, H, device=dev) * 0.5).bfloat16()
    assert L < 4096 or int(t["qcnt"].max()) > 64
    ctx, lse = ops.attn_list_fwd(qkv, mask, B, L, heads, klist, kcnt, t["stride"], korder=korder if train else None)
    dqkv = ops.attn_list_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, klist, kcnt, qlist, qcnt, t["stride"],
                             korder=korder if train else None, qorder=qorder if train else None)
    ref_ctx, ref_dqkv = list_reference(qkv, mask, B, L, heads, t["klist"], t["kcnt"], dctx)
    assert (ctx.float() - ref_ctx).abs().max().item() < 0.03
    assert torch.isfinite(dqkv.float()).all()
    err = (dqkv.float() - ref_dqkv).abs().max().item()
    assert err < 0.03 * max(1.0, ref_dqkv.abs().max().item()), err
    # per-section so that a wrong section cannot hide
    for i, name in enumerate(["dq", "dk", "dv"]):
        a, b = dqkv.float()[:, i * H:(i + 1) * H], ref_dqkv[:, i * H:(i + 1) * H]
        assert ((a - b).norm() / b.norm()).item() < 2e-2, name


@pytest.mark.parametrize("B,L,heads,train", [(2, 1024, 2, False), (1, 768, 4, True), (1, 4096, 2, False)])
def test_list_attention_fp32_parity_kernel(dev, B, L, heads, train):
    """amdseg_attn_list_f32 (inference parity mode) against the plain torch fp32 restatement: 1e-5"""
    from spokennlp_amd import ops, bigbird_plan
    torch.manual_seed(L)
    t = bigbird_plan.build(L, heads, 3, seed=2, training=train, max_seqlen=4096)
    klist, kcnt = torch.from_numpy(t["klist"]).to(dev), torch.from_numpy(t["kcnt"]).to(dev)
    qkv = torch.randn(B * L, 3 * heads * 64, device=dev)
    mask = torch.zeros(B, L, device=dev)
    mask[0, L - 37:] = -10000.0
    ctx = ops.attn_list_f32(qkv, mask, B, L, heads, klist, kcnt, t["stride"])
    ref = list_reference(qkv, mask, B, L, heads, t["klist"], t["kcnt"])
    assert (ctx - ref).abs().max().item() < 1e-5


def test_list_equals_full_attention_when_every_block_is_listed(dev):
    from spokennlp_amd import ops
    B, L, heads = 2, 512, 2
    nb = L // 64
    qkv = torch.randn(B * L, 3 * heads * 64, device=dev).bfloat16()
    mask = torch.zeros(B, L, device=dev)
    klist = torch.arange(nb, dtype=torch.int32, device=dev).repeat(heads, nb, 1).contiguous()
    kcnt = torch.full((heads, nb), nb, dtype=torch.int32, device=dev)
    full, lse_full = ops.attn_fwd(qkv, mask, B, L, heads)
    lst, lse = ops.attn_list_fwd(qkv, mask, B, L, heads, klist, kcnt, nb)
    assert (full.float() - lst.float()).abs().max().item() < 1e-2      # 128-row vs 64-row workgroups: same arithmetic per row
    assert (lse - lse_full).abs().max().item() < 1e-4


def test_gelu_new_epilogues(dev):
    """EPI_BIAS_GELU / EPI_GELU_BWD with AMDSEG_EPI_ACT_TANH against torch's tanh-GELU, on both tile paths"""
    from spokennlp_amd import ops
    from spokennlp_amd import lib as L
    torch.manual_seed(0)
    for M, N, K in [(256, 256, 128), (512, 768, 1536)]:
        A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev) * 0.1
        pre = A.float() @ W.float().t() + bias
        out, pre_out = ops.gemm_nt(A, W, epilogue=L.EPI_BIAS_GELU | L.EPI_ACT_TANH, bias=bias)
        ref = torch.nn.functional.gelu(pre, approximate="tanh")
        assert (out.float() - ref).abs().max().item() < 0.02
        assert (pre_out.float() - pre).abs().max().item() < 0.03
        dy = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        Wt = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        R = (torch.randn(M, N, device=dev)).bfloat16()
        g = ops.gemm_nt(dy, Wt, epilogue=L.EPI_GELU_BWD | L.EPI_ACT_TANH, R=R)
        r32 = R.float().requires_grad_(True)
        torch.nn.functional.gelu(r32, approximate="tanh").sum().backward()
        refg = (dy.float() @ Wt.float().t()) * r32.grad
        assert (g.float() - refg).abs().max().item() < 0.03 * max(1.0, refg.abs().max().item())


# ---------------------------------------------------------------------------------------------------- model level
def build_bb(arch, flags, sd, dev, dropout=0.0, precision=None):
    from transformers import BigBirdConfig
    from spokennlp_amd.bigbird_for_ts import BigBirdWithDAForSentenceLabelingTopicSegmentation as M
    cfg = BigBirdConfig(num_labels=2, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, **arch)
    for k, v in flags.items():
        setattr(cfg, k, v)
    if precision:
        cfg.amdseg_precision = precision
    m = M(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
    return m.to(dev)


@pytest.mark.parametrize("case", ["bb_tiny_L1024", "bb_tiny_L768", "bb_tiny_L128", "bb_tiny_L1000"])
@pytest.mark.parametrize("variant", ["plain_eval", "full_eval"])
def test_bigbird_eval_vs_reference_golden(dev, case, variant):
    from oracle import bert_ts_oracle as O
    z, sd, batch, arch = bb_case(case)
    m = build_bb(arch, flags_of(z, variant), sd, dev).eval()
    random.seed(int(z[f"{variant}.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**{k: v.to(dev) for k, v in batch.items()})
    ref = torch.from_numpy(z[f"{variant}.logits"])
    d = (logits.cpu() - ref).abs().max().item()
    print(f"{case}/{variant}: max|dlogit| {d:.2e}")
    assert d < 0.08 and abs(loss.item() - float(z[f"{variant}.loss"])) < 0.05
    assert O.decode_predictions(logits.cpu()[:, 0], batch["labels"][:, 0]) == O.decode_predictions(ref[:, 0], batch["labels"][:, 0])


@pytest.mark.parametrize("case", ["bb_tiny_L1024", "bb_tiny_L768", "bb_tiny_L1000"])
@pytest.mark.parametrize("variant", ["plain_eval", "full_eval"])
def test_bigbird_block_sparse_fp32_parity(dev, case, variant):
    """block-sparse attention in fp32 parity mode: the north-star tolerance (1e-3) against the reference's logits"""
    z, sd, batch, arch = bb_case(case)
    m = build_bb(arch, flags_of(z, variant), sd, dev, precision="fp32").eval()
    random.seed(int(z[f"{variant}.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**{k: v.to(dev) for k, v in batch.items()})
    d = (logits.cpu() - torch.from_numpy(z[f"{variant}.logits"])).abs().max().item()
    print(f"{case}/{variant} fp32: max|dlogit| {d:.2e}")
    assert d < 1e-3 and abs(loss.item() - float(z[f"{variant}.loss"])) < 1e-3
    assert m.engine().attention_type == "block_sparse"


def test_bigbird_full_attention_fallback_fp32_parity(dev):
    """L <= 704: the reference switches to full attention; the fp32 parity kernels then give the north-star tolerance"""
    z, sd, batch, arch = bb_case("bb_tiny_L128")
    m = build_bb(arch, flags_of(z, "plain_eval"), sd, dev, precision="fp32").eval()
    with torch.no_grad():
        loss, logits, cos = m(**{k: v.to(dev) for k, v in batch.items()})
    assert (logits.cpu() - torch.from_numpy(z["plain_eval.logits"])).abs().max().item() < 1e-3
    assert m.engine().attention_type == "original_full"


@pytest.mark.parametrize("case", ["bb_tiny_L1024", "bb_tiny_L768", "bb_tiny_L128", "bb_tiny_L1000"])
def test_bigbird_train_grads_vs_reference_golden(dev, case):
    """training mode: per-layer, per-head numpy-seeded random blocks as the reference draws them"""
    z, sd, batch, arch = bb_case(case)
    m = build_bb(arch, flags_of(z, "train_full"), sd, dev).train()
    random.seed(int(z["train_full.random_seed"]))
    loss, logits, cos = m(**{k: v.to(dev) for k, v in batch.items()})
    loss.backward()
    assert abs(loss.item() - float(z["train_full.loss"])) < 0.05
    params = dict(m.named_parameters())
    checked = 0
    for k in z.files:
        if not k.startswith("train_full.grad."):
            continue
        n = k[len("train_full.grad."):]
        ref = torch.from_numpy(z[k])
        g = params[n].grad.float().cpu()
        if float(ref.norm()) < 1e-5:
            assert float(g.norm()) < 1e-2, n
            continue
        c = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
        rel = abs(float(g.norm()) - float(ref.norm())) / float(ref.norm())
        assert c > 0.99 and rel < 0.06, (n, c, rel)
        checked += 1
    assert checked > 30
    names = z["train_full.gradnorm_names"].tolist()
    vals = z["train_full.gradnorm_vals"].tolist()
    for n, v in zip(names, vals):                              # embeddings: norms only (fixture size)
        if "embeddings" in n and v > 1e-5:
            gn = float(params[n].grad.float().norm())
            assert abs(gn - v) / v < 0.06, (n, gn, v)


def test_bigbird_dropout_step_deterministic(dev):
    z, sd, batch, arch = bb_case("bb_tiny_L768")
    vals = []
    for _ in range(2):
        m = build_bb(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train()
        m.amdseg_seed = 5
        random.seed(1)
        loss, _, _ = m(**{k: v.to(dev) for k, v in batch.items()})
        loss.backward()
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
        assert math.isfinite(loss.item()) and math.isfinite(gn)
        vals.append((loss.item(), gn))
    # same dropout masks, same loss bit for bit; the embedding-table gradients are accumulated with fp32 atomics (as torch's own
    # embedding backward), so with 12 blocks of rows per table the gradient norm may differ in the last bits between runs
    assert vals[0][0] == vals[1][0]
    assert abs(vals[0][1] - vals[1][1]) <= 1e-6 * vals[0][1]


def test_backward_drops_the_rows_of_trailing_padding_without_changing_a_bit(dev):
    """amdseg_bert_cfg.pad_guard with block-sparse (list) attention: masked keys get p = 0 in the list kernels as in the full ones, so the
    rows of trailing padding carry exact-zero gradients; engine level, fixed incoming gradient: layer gradients bit-identical on / off"""
    from transformers import BigBirdConfig
    from spokennlp_amd.bigbird_for_ts import BigBirdWithDAForSentenceLabelingTopicSegmentation as M
    cfg = BigBirdConfig(num_labels=2, vocab_size=300, hidden_size=768, num_attention_heads=12, num_hidden_layers=2, intermediate_size=3072,
                        max_position_embeddings=1024, block_size=64, num_random_blocks=3, attention_type="block_sparse",
                        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    m = M(cfg).to(dev)
    eng = m.engine()
    B, L = 4, 1024
    lens = [1024, 900, 500, 130]
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 300, (B, L), generator=g).to(dev)
    am = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, :n] = 1
    am = am.to(dev)
    tt = torch.zeros_like(ids)
    dseq = torch.randn(B, L, 768, generator=g).to(dev) * am[:, :, None].float()
    names = [n for n in eng.fp.offsets if ".encoder.layer." in n]

    def run(skip):
        eng.skip_padded_rows_bwd = skip
        _, ectx = eng.forward(ids, am, tt, True, seed=7, p_out=0.1)
        eng.backward(ectx, dseq, accumulate=False)
        torch.cuda.synchronize()
        return {n: eng.fp.view(eng.fp.flat_g, n).clone() for n in names}

    on = run(True)
    assert int(eng._pad_guard.item()) == 0
    off = run(False)
    for n in names:
        assert float(on[n].abs().max()) > 0
        assert torch.equal(on[n], off[n]), n
