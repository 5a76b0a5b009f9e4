"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain torch tensor algebra) of the Longformer encoder as the reference's wrapper uses it
(emnlp2023-topic_segmentation/src/models/longformer_for_ts.py:34-129: CLS is the only global token, :55-58), to be
plugged into bert_ts_oracle.model_forward(encode=longformer_encode).  The arithmetic follows
[hf] transformers/models/longformer/modeling_longformer.py (third party, semantics per SURVEY.md 8(a) a4):
  embeddings       :384-443  RoBERTa-style position ids = cumsum(non-pad) * non-pad + pad_id (:368-381), one token type
  self-attention   :482-640  q / sqrt(d); band |i-j| <= w (w = attention_window // 2) minus padded and global keys,
                             global keys re-added as leading columns; fp32 softmax; rows of padded queries zeroed
  global rows      :964-1058 separate query_global / key_global / value_global projections over ALL tokens, padded
                             keys masked, result overwrites the global token's row
  output / FFN     :1061-1131 (the BERT block with config.layer_norm_eps)
HF evaluates the band with overlapping 2w x 2w chunks; here it is a dense L x L masked softmax (same values).

Parity status: PINNED -- tests/test_oracle_golden.py compares hidden states / logits / loss / gradients with golden
vectors produced by importing the reference wrapper over transformers' LongformerModel (tools/gen_golden.py).
"""
import torch
import torch.nn.functional as F

from .bert_ts_oracle import gelu_erf, layer_norm

PFX = "longformer."


def position_ids_from_input_ids(input_ids, pad_id):
    mask = (input_ids != pad_id).long()
    return torch.cumsum(mask, dim=1) * mask + pad_id


def embeddings(sd, cfg, input_ids, token_type_ids, prefix=PFX):
    pad = cfg.pad_token_id
    e = F.embedding(input_ids, sd[prefix + "embeddings.word_embeddings.weight"], padding_idx=pad)
    e = e + sd[prefix + "embeddings.token_type_embeddings.weight"][token_type_ids]
    e = e + F.embedding(position_ids_from_input_ids(input_ids, pad), sd[prefix + "embeddings.position_embeddings.weight"], padding_idx=pad)
    return layer_norm(e, sd[prefix + "embeddings.LayerNorm.weight"], sd[prefix + "embeddings.LayerNorm.bias"], cfg.layer_norm_eps)


def allowed_mask(L, w, is_global, is_pad):
    """(B, L, L) bool: key j visible from LOCAL query i  <=>  not pad(j) and (global(j) or |i-j| <= w)."""
    i = torch.arange(L)
    band = (i[:, None] - i[None, :]).abs() <= w
    return (band[None] | is_global[:, None, :]) & ~is_pad[:, None, :]


def encoder_layer(sd, cfg, x, is_global, is_pad, i, prefix=PFX):
    p = f"{prefix}encoder.layer.{i}."
    B, L, H = x.shape
    nh = cfg.num_attention_heads
    d = H // nh
    aw = cfg.attention_window[i] if isinstance(cfg.attention_window, (list, tuple)) else cfg.attention_window
    w = aw // 2

    def lin(t, name):
        return t @ sd[p + name + ".weight"].t() + sd[p + name + ".bias"]

    def heads(t):
        return t.view(B, -1, nh, d).transpose(1, 2)

    q = heads(lin(x, "attention.self.query") / (d ** 0.5))
    k = heads(lin(x, "attention.self.key"))
    v = heads(lin(x, "attention.self.value"))
    s = q @ k.transpose(-1, -2)
    ok = allowed_mask(L, w, is_global, is_pad)[:, None]
    pr = torch.softmax(s.masked_fill(~ok, float("-inf")).float(), dim=-1)
    pr = pr.masked_fill(is_pad[:, None, :, None], 0.0)               # rows of padded queries are zeroed (:579)
    ctx = (pr @ v).transpose(1, 2).reshape(B, L, H)
    # global rows: full attention with the *_global projections; every example has the same number of global tokens here
    # (none at all when LongformerModel is called with global_attention_mask=None: the mmvts text encoder)
    ng = int(is_global[0].sum())
    if ng > 0:
        gidx = is_global.nonzero(as_tuple=False)                          # (B*ng, 2)
        xg = x[gidx[:, 0], gidx[:, 1]].view(B, ng, H)
        qg = heads(lin(xg, "attention.self.query_global") / (d ** 0.5))
        kg = heads(lin(x, "attention.self.key_global"))
        vg = heads(lin(x, "attention.self.value_global"))
        sg = (qg @ kg.transpose(-1, -2)).masked_fill(is_pad[:, None, None, :], torch.finfo(x.dtype).min)
        pg = torch.softmax(sg.float(), dim=-1)
        og = (pg @ vg).transpose(1, 2).reshape(B * ng, H)
        ctx = ctx.index_put((gidx[:, 0], gidx[:, 1]), og)
    x1 = layer_norm(lin(ctx, "attention.output.dense") + x, sd[p + "attention.output.LayerNorm.weight"],
                    sd[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
    h = gelu_erf(lin(x1, "intermediate.dense"))
    return layer_norm(lin(h, "output.dense") + x1, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"],
                      cfg.layer_norm_eps)


def longformer_encode(sd, cfg, input_ids, attention_mask, token_type_ids, return_all=False, prefix=PFX, global_attention_mask=None):
    """LongformerModel.forward without pooler; global_attention_mask defaults to CLS only (longformer_for_ts.py:55-58)."""
    B, L = input_ids.shape
    if global_attention_mask is None:
        global_attention_mask = torch.zeros_like(input_ids)
        global_attention_mask[:, 0] = 1
    is_pad = attention_mask == 0
    is_global = (global_attention_mask == 1) & ~is_pad
    x = embeddings(sd, cfg, input_ids, token_type_ids, prefix)
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        x = encoder_layer(sd, cfg, x, is_global, is_pad, i, prefix)
        hs.append(x)
    return (x, hs) if return_all else x
