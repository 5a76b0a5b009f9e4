"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain torch tensor algebra) of the BigBird encoder as the reference's wrapper uses it
(emnlp2023-topic_segmentation/src/models/bigbird_for_ts.py:19-113: a BigBirdModel under `self.bert`, :27), to be plugged into
bert_ts_oracle.model_forward(encode=bigbird_encode).  The arithmetic follows [hf] transformers/models/big_bird/
modeling_big_bird.py (third party; the reference pins transformers in requirements.txt):
  BigBirdEmbeddings.forward           word + token_type + position, dropout, THEN LayerNorm
  BigBirdModel.forward                L <= (5 + 2r) * block  ->  attention_type "original_full" (BERT attention, finfo.min key mask)
  bigbird_block_sparse_attention      five query-block groups against concatenated key blocks: restated as one softmax per
                                      (batch, head, query block) over the explicitly gathered key blocks, duplicates included;
                                      additive mask -10000 * (1 - key mask); context rows * from_mask.  (For padded QUERY rows the
                                      reference's band / random masks also carry the query mask, which only changes rows that
                                      are multiplied by zero.)
  random blocks                       np.random.seed(layer_idx) + _bigbird_block_rand_mask / _bigbird_block_rand_mask_with_head
                                      in training; all zeros in eval.  `rand_fn` supplies them (spokennlp_amd.bigbird_plan.rand_blocks
                                      calls the installed transformers' own plan functions; this oracle does not re-derive them).
  BigBirdIntermediate                 hidden_act "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))

Parity status: PINNED -- tests/test_oracle_golden.py compares final hidden state / logits / loss / gradients with golden
vectors produced by importing the reference wrapper over transformers' BigBirdModel (tools/gen_golden.py --bigbird-only).
"""
import math

import torch
import torch.nn.functional as F

from .bert_ts_oracle import gelu_erf, layer_norm

PFX = "bert."
BLOCK = 64


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def embeddings(sd, cfg, input_ids, token_type_ids, prefix=PFX):
    L = input_ids.shape[1]
    e = sd[prefix + "embeddings.word_embeddings.weight"][input_ids] + sd[prefix + "embeddings.token_type_embeddings.weight"][token_type_ids]
    e = e + sd[prefix + "embeddings.position_embeddings.weight"][:L][None]
    return layer_norm(e, sd[prefix + "embeddings.LayerNorm.weight"], sd[prefix + "embeddings.LayerNorm.bias"], cfg.layer_norm_eps)


def key_block_rows(nb, rand_h):
    """rand_h: [nb-2, r] ints.  List of key-block lists, one per query block (see spokennlp_amd/bigbird_plan.py docstring)."""
    rows = []
    for i in range(nb):
        if i == 0 or i == nb - 1:
            rows.append(list(range(nb)))
        elif i == 1:
            rows.append([0, 1, 2, nb - 1] + [int(v) for v in rand_h[0]])
        elif i == nb - 2:
            rows.append([0, nb - 3, nb - 2, nb - 1] + [int(v) for v in rand_h[nb - 3]])
        else:
            rows.append([0, i - 1, i, i + 1] + [int(v) for v in rand_h[i - 1]] + [nb - 1])
    return rows


def block_sparse_attention(q, k, v, key_mask, rand):
    """q, k, v: [B, nh, L, d] (q unscaled); key_mask: [B, L] 0/1 float; rand: [nh, nb-2, r].  Returns [B, L, nh*d]."""
    B, nh, L, d = q.shape
    nb = L // BLOCK
    out = torch.zeros(B, nh, L, d, dtype=q.dtype)
    pen = (1.0 - key_mask) * -10000.0                                   # [B, L]
    for h in range(nh):
        rows = key_block_rows(nb, rand[h])
        for i, blocks in enumerate(rows):
            idx = torch.cat([torch.arange(kb * BLOCK, (kb + 1) * BLOCK) for kb in blocks])
            qi = q[:, h, i * BLOCK:(i + 1) * BLOCK]                      # [B, 64, d]
            s = qi @ k[:, h, idx].transpose(-1, -2) / math.sqrt(d) + pen[:, None, idx]
            p = torch.softmax(s, dim=-1)
            out[:, h, i * BLOCK:(i + 1) * BLOCK] = p @ v[:, h, idx]
    out = out * key_mask[:, None, :, None]                              # context_layer * from_mask
    return out.transpose(1, 2).reshape(B, L, nh * d)


def encoder_layer(sd, cfg, x, attention_mask, i, rand, prefix=PFX):
    p = f"{prefix}encoder.layer.{i}."
    B, L, H = x.shape
    nh = cfg.num_attention_heads
    d = H // nh

    def lin(t, name):
        return t @ sd[p + name + ".weight"].t() + sd[p + name + ".bias"]

    def heads(t):
        return t.view(B, L, nh, d).transpose(1, 2)

    q, k, v = heads(lin(x, "attention.self.query")), heads(lin(x, "attention.self.key")), heads(lin(x, "attention.self.value"))
    if rand is None:                                                    # original_full
        s = q @ k.transpose(-1, -2) / math.sqrt(d) + ((1.0 - attention_mask.to(x.dtype)) * torch.finfo(x.dtype).min)[:, None, None, :]
        ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, H)
    else:
        ctx = block_sparse_attention(q, k, v, attention_mask.to(x.dtype), rand)
    x1 = layer_norm(lin(ctx, "attention.output.dense") + x, sd[p + "attention.output.LayerNorm.weight"],
                    sd[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
    act = gelu_new if getattr(cfg, "hidden_act", "gelu_new") in ("gelu_new", "gelu_pytorch_tanh", "gelu_fast") else gelu_erf
    h = act(lin(x1, "intermediate.dense"))
    return layer_norm(lin(h, "output.dense") + x1, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def make_encode(rand_fn, training):
    """rand_fn(seq_len, num_heads, num_rand_blocks, seed, training, max_seqlen) -> [heads, nb-2, r] (bigbird_plan.rand_blocks)."""

    def bigbird_encode(sd, cfg, input_ids, attention_mask, token_type_ids, return_all=False, prefix=PFX):
        B, L0 = input_ids.shape
        sparse = getattr(cfg, "attention_type", "block_sparse") == "block_sparse" and L0 > (5 + 2 * cfg.num_random_blocks) * cfg.block_size
        if sparse and cfg.block_size != BLOCK:
            raise ValueError("oracle restates block_size 64 only")
        L = L0
        if sparse and L0 % BLOCK:
            # [hf] BigBirdModel._pad_to_block_size: right-pad ids with pad_token_id, mask / token types with 0; the output is cut back
            # to the original length ([hf] BigBirdModel.forward: sequence_output[:, :-padding_len])
            pad = BLOCK - L0 % BLOCK
            F = torch.nn.functional
            input_ids = F.pad(input_ids, (0, pad), value=int(getattr(cfg, "pad_token_id", 0) or 0))
            attention_mask = F.pad(attention_mask, (0, pad), value=0)
            token_type_ids = F.pad(token_type_ids, (0, pad), value=0)
            L = L0 + pad
        x = embeddings(sd, cfg, input_ids, token_type_ids, prefix)
        hs = [x]
        for i in range(cfg.num_hidden_layers):
            rand = rand_fn(L, cfg.num_attention_heads, cfg.num_random_blocks, i, training, cfg.max_position_embeddings) if sparse else None
            x = encoder_layer(sd, cfg, x, attention_mask, i, rand, prefix)
            hs.append(x)
        if L != L0:
            x, hs = x[:, :L0], [h[:, :L0] for h in hs]
        return (x, hs) if return_all else x

    return bigbird_encode
