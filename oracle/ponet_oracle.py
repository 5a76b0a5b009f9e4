"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU statement (plain torch) of the PoNet token-classification path of alimeeting4mug
(src/models/modeling_ponet.py:34-109: PoNetModel(..., segment_ids) -> dropout -> Linear(H, num_labels) -> CE with labels
forced to -100 where attention_mask != 1).

PARITY STATUS: UNPINNED.  The encoder arithmetic is NOT in the reference tree: modeling_ponet.py:24-30 imports
`modelscope.models.nlp.ponet.PoNetModel` (modelscope==1.1.0, alimeeting4mug/requirements.txt:56), which is absent from
/root/reference and not installed here, and the reference holds no test, golden vector or fixture for it.  What follows
restates the published algorithm (Tan et al., "PoNet: Pooling Network for Efficient Token Mixing in Long Sequences",
ICLR 2022, cited at alimeeting4mug/README.md:14) in the form of its public implementation as far as it can be
reconstructed without the source:

  per layer, five projections of the layer input x:  Hq, Hk, Ho, Hl, Hs = x W*^T + b*        (* in q, k, o, local, segment)
  global aggregation (multi-head, d = 64):  qbar_h = mean over valid tokens of Hq_h ;  a_j = qbar_h . Hk_{j,h} / sqrt(d)
      + key mask ;  p = softmax_j(a) (dropout) ;  g_h = sum_j p_j Hk_{j,h}                    (value = key)
  segment max-pooling:  S_n = elementwise max of Hs over the valid tokens with segment_ids == segment_ids[n]
  local max-pooling:    L_n = elementwise max of Hl over the valid tokens n-1, n, n+1
  fusion:               ctx_n = (g + S_n) * Ho_n + L_n ;  padded tokens give 0
  then the BERT block: dense + dropout + residual LayerNorm, GELU FFN + residual LayerNorm; BERT embeddings.
Details of the original that cannot be checked (treatment of [CLS]/[SEP] inside the pooling branches, dropout placement)
are fixed as written here; the HIP path (spokennlp_amd/ponet.py) is tested against THIS statement only.
segment_ids are expected non-decreasing along the sequence (what the reference's feature builder emits,
ponet_topic_segmentation.py:564-596): a segment is a contiguous run.
"""
import math

import torch
import torch.nn.functional as F

from .bert_ts_oracle import gelu_erf, layer_norm

PFX = "ponet."
PROJ = ("dense_q", "dense_k", "dense_o", "dense_local", "dense_segment")


def embeddings(sd, cfg, input_ids, token_type_ids, prefix=PFX):
    B, L = input_ids.shape
    e = F.embedding(input_ids, sd[prefix + "embeddings.word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", None))
    e = e + sd[prefix + "embeddings.token_type_embeddings.weight"][token_type_ids]
    e = e + sd[prefix + "embeddings.position_embeddings.weight"][torch.arange(L)].unsqueeze(0)
    return layer_norm(e, sd[prefix + "embeddings.LayerNorm.weight"], sd[prefix + "embeddings.LayerNorm.bias"], cfg.layer_norm_eps)


def segment_max_runs(hs, valid, segment_ids, ninf):
    """the segment maximum for NON-DECREASING segment ids (what the feature builder emits, ponet_topic_segmentation.py:564-596): a
    segment is a contiguous run, so S is one amax per run.  O(L H) instead of the O(L^2 H) mask of `pooling`'s general form; the two are
    compared on small inputs in tests/test_oracle_golden.py."""
    B, L, H = hs.shape
    out = []
    for b in range(B):
        seg = segment_ids[b]
        cuts = [0] + (torch.nonzero(seg[1:] != seg[:-1]).flatten() + 1).tolist() + [L]
        rows = []
        for a, e in zip(cuts[:-1], cuts[1:]):
            v = torch.where(valid[b, a:e, None], hs[b, a:e], torch.full((), ninf, dtype=hs.dtype))
            rows.append(v.amax(0, keepdim=True).expand(e - a, H))
        out.append(torch.cat(rows, 0))
    return torch.stack(out)


def special_positions(valid):
    """[B, L] bool: position 0 ([CLS]) and the last valid position ([SEP]) of every sequence"""
    B, L = valid.shape
    sp = torch.zeros_like(valid)
    sp[:, 0] = True
    last = (valid.long() * torch.arange(1, L + 1)[None, :]).amax(1) - 1
    for b in range(B):
        if last[b] >= 0:
            sp[b, last[b]] = True
    return sp & valid


def pooling(hq, hk, ho, hl, hs, valid, segment_ids, nh, runs=False, pool_valid=None):
    """the token-mixing block on the five projections ([B, L, H] each); valid [B, L] bool; returns ctx [B, L, H].
    pool_valid (config.ponet_special_tokens_mixing = False): the tokens that take part in the pooling branches -- [CLS] / [SEP] then enter
    no local or segment window and get no mixing output (ctx = 0, residual path only), while still being keys of the global aggregation.
    One of the two readings of the unavailable original; a user holding the checkpoint picks the one that reproduces it."""
    B, L, H = hq.shape
    d = H // nh
    vf = valid.to(hq.dtype)
    nvalid = vf.sum(1).clamp(min=1.0)
    qbar = (hq * vf[..., None]).sum(1) / nvalid[:, None]                                   # [B, H]
    a = torch.einsum("bhe,bjhe->bhj", qbar.view(B, nh, d), hk.view(B, L, nh, d)) / math.sqrt(d)
    a = a.masked_fill(~valid[:, None, :], float("-inf"))
    p = torch.softmax(a.float(), dim=-1).to(hq.dtype)
    g = torch.einsum("bhj,bjhe->bhe", p, hk.view(B, L, nh, d)).reshape(B, 1, H)
    if pool_valid is not None:
        valid = pool_valid
    ninf = torch.finfo(hq.dtype).min
    # segment max over valid tokens with the same id
    if runs:
        S = segment_max_runs(hs, valid, segment_ids, ninf)
    else:
        same = (segment_ids[:, :, None] == segment_ids[:, None, :]) & valid[:, None, :]       # [B, n, j]
        S = torch.stack([torch.where(same[b][:, :, None], hs[b][None, :, :], torch.full((), ninf, dtype=hq.dtype)).amax(1) for b in range(B)])
    # local max over valid n-1, n, n+1
    hlm = torch.where(valid[..., None], hl, torch.full((), ninf, dtype=hq.dtype))
    pad = torch.full((B, 1, H), ninf, dtype=hq.dtype)
    Lm = torch.maximum(torch.maximum(torch.cat((pad, hlm[:, :-1]), 1), hlm), torch.cat((hlm[:, 1:], pad), 1))
    ctx = (g + S) * ho + Lm
    return torch.where(valid[..., None], ctx, torch.zeros((), dtype=ctx.dtype))


def encoder_layer(sd, cfg, x, valid, segment_ids, i, prefix=PFX, runs=False):
    p = f"{prefix}encoder.layer.{i}."

    def lin(t, name):
        return t @ sd[p + name + ".weight"].t() + sd[p + name + ".bias"]

    hq, hk, ho, hl, hs = [lin(x, "attention.self." + n) for n in PROJ]
    pv = None if cfg.get("ponet_special_tokens_mixing", True) else (valid & ~special_positions(valid))
    ctx = pooling(hq, hk, ho, hl, hs, valid, segment_ids, cfg.num_attention_heads, runs=runs, pool_valid=pv)
    x1 = layer_norm(lin(ctx, "attention.output.dense") + x, sd[p + "attention.output.LayerNorm.weight"],
                    sd[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
    h = gelu_erf(lin(x1, "intermediate.dense"))
    return layer_norm(lin(h, "output.dense") + x1, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"],
                      cfg.layer_norm_eps)


def ponet_encode(sd, cfg, input_ids, attention_mask, token_type_ids, segment_ids, return_all=False, prefix=PFX, runs=False):
    valid = attention_mask == 1
    x = embeddings(sd, cfg, input_ids, token_type_ids, prefix)
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        x = encoder_layer(sd, cfg, x, valid, segment_ids, i, prefix, runs=runs)
        hs.append(x)
    return (x, hs) if return_all else x


def token_classification_forward(sd, cfg, input_ids, attention_mask, token_type_ids, segment_ids, labels=None, runs=False):
    """modeling_ponet.py:47-109 (eval / dropout 0): returns (loss or None, logits [B, L, num_labels])."""
    seq = ponet_encode(sd, cfg, input_ids, attention_mask, token_type_ids, segment_ids, runs=runs)
    logits = seq @ sd["classifier.weight"].t() + sd["classifier.bias"]
    loss = None
    if labels is not None:
        active = torch.where(attention_mask.view(-1) == 1, labels.view(-1), torch.full_like(labels.view(-1), -100))
        loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), active, ignore_index=-100)
    return loss, logits
