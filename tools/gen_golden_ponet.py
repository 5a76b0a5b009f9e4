#!/usr/bin/env python
"""Pin the PoNet encoder on the REAL `modelscope.models.nlp.ponet` -- the one-command recipe.

alimeeting4mug/src/models/modeling_ponet.py:24-30 imports `PoNetModel` / `PoNetPreTrainedModel` from modelscope==1.1.0
(alimeeting4mug/requirements.txt:56).  That package is not in /root/reference and not installed in the build image, so
oracle/ponet_oracle.py restates the paper and says "parity unpinned".  On a machine that HAS the package:

    pip install modelscope==1.1.0          # (needs network; not possible in the build image)
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_ponet.py [--base]

imports it (never copies it), runs the reference's own `PoNetForTokenClassification` when /root/reference is present
(otherwise the same head built here on modelscope's `PoNetModel`: dropout -> Linear(H, num_labels) -> CE with labels forced
to -100 where attention_mask != 1, modeling_ponet.py:81-98), and writes DATA ONLY:

    tests/golden/ponet_tiny.npz            2 layers, H = 64, 4 heads, I = 128, vocab 100, L = 64, B = 2
                                           state dict, the four int inputs, labels, every hidden state, logits, eval loss,
                                           train-mode loss + every parameter gradient (dropout 0)
    tests/golden/ponet_base_L4096.npz      (--base) 12 x 768, L = 4096, B = 1: inputs, logits, the first and the last hidden state;
                                           weights are regenerated from the recorded seed, not stored (420 MB)

It then diffs BOTH readings of the detail the paper leaves open (`config.ponet_special_tokens_mixing`, oracle/ponet_oracle.py:76-79)
against what the package computed, prints max |difference| per hidden state for each, and records the matching reading in the
fixture (`reading`: 1 = mixing True, 0 = mixing False, -1 = neither within 1e-4 -- then the oracle is WRONG somewhere and the
per-layer table says where).  tests/test_oracle_golden.py::test_ponet_oracle_vs_modelscope_golden and
tests/test_gpu_ponet.py::test_model_vs_modelscope_golden activate as soon as the fixture exists; until then they report
"skipped: fixture absent" and rows a11 / f4 stay "parity unpinned".

Exit codes: 0 fixtures written; 3 modelscope's PoNet not importable (nothing written)."""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
OUT = os.path.join(ROOT, "tests", "golden")
REF_SRC = "/root/reference/alimeeting4mug/src"

TINY = dict(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
            max_position_embeddings=64, type_vocab_size=2)
BASE = dict(vocab_size=21129, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
            max_position_embeddings=4096, type_vocab_size=2)

# our parameter names (oracle/ponet_oracle.py, spokennlp_amd/ponet.py) <- candidates in the package, tried in order.  The HF-hub port of the
# paper's code and the modelscope backbone use the same module tree as BERT with the five projections inside `attention.self`; if a
# release names them differently, add the spelling here -- the generator fails with the unmatched names listed, it never guesses silently.
ALIASES = {
    "attention.self.dense_q": ("attention.self.dense_q", "attention.self.query", "attention.self.dense_query"),
    "attention.self.dense_k": ("attention.self.dense_k", "attention.self.key", "attention.self.dense_key"),
    "attention.self.dense_o": ("attention.self.dense_o", "attention.self.dense_out", "attention.self.value"),
    "attention.self.dense_local": ("attention.self.dense_local", "attention.self.local"),
    "attention.self.dense_segment": ("attention.self.dense_segment", "attention.self.segment"),
}


def import_ponet():
    try:
        from modelscope.models.nlp.ponet import PoNetModel, PoNetPreTrainedModel  # noqa: F401
        try:
            from modelscope.models.nlp.ponet import PoNetConfig
        except ImportError:
            from modelscope.models.nlp.ponet.configuration import PoNetConfig
        return PoNetModel, PoNetConfig
    except Exception as e:                                   # noqa: BLE001 -- any failure of the foreign import means "not pinnable here"
        print(f"gen_golden_ponet: modelscope's PoNet is not importable here ({type(e).__name__}: {e}).\n"
              "  install modelscope==1.1.0 (alimeeting4mug/requirements.txt:56) and re-run; nothing was written.", file=sys.stderr)
        return None, None


def build_model(PoNetModel, PoNetConfig, arch, seed):
    cfg = PoNetConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **arch)
    torch.manual_seed(seed)
    cls = None
    if os.path.isdir(REF_SRC):                               # the reference's own wrapper (imported where it lies, never copied)
        sys.path.insert(0, REF_SRC)
        try:
            from models.modeling_ponet import PoNetForTokenClassification as cls  # noqa: N813
        except Exception as e:                               # noqa: BLE001
            print(f"  (reference wrapper not importable: {type(e).__name__}: {e}; using the head restated here)", file=sys.stderr)
    if cls is not None:
        m = cls(cfg)
        kind = "reference wrapper alimeeting4mug/src/models/modeling_ponet.py"
    else:
        class Head(torch.nn.Module):                         # modeling_ponet.py:34-46,81-98 on the package's encoder
            def __init__(self):
                super().__init__()
                self.ponet = PoNetModel(cfg, add_pooling_layer=False)
                self.dropout = torch.nn.Dropout(cfg.hidden_dropout_prob)
                self.classifier = torch.nn.Linear(cfg.hidden_size, cfg.num_labels)

            def forward(self, input_ids, attention_mask, token_type_ids, segment_ids, labels=None, output_hidden_states=True, **kw):
                out = self.ponet(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, segment_ids=segment_ids,
                                 output_hidden_states=output_hidden_states, return_dict=True)
                logits = self.classifier(self.dropout(out[0]))
                loss = None
                if labels is not None:
                    active = torch.where(attention_mask.view(-1) == 1, labels.view(-1), torch.full_like(labels.view(-1), -100))
                    loss = torch.nn.functional.cross_entropy(logits.view(-1, cfg.num_labels), active, ignore_index=-100)
                return type("O", (), dict(loss=loss, logits=logits, hidden_states=out.hidden_states))()
        m = Head()
        kind = "head restated on modelscope PoNetModel"
    with torch.no_grad():
        m.classifier.weight.normal_(0, 0.3)                  # O(1) logits so that a tolerance means something
    return m, cfg, kind


def make_inputs(B, L, vocab, seed):
    """[CLS] + sentences ending in a labelled [EOS]-like token, ragged padding; segment ids as the reference's feature builder emits them
    (ponet_topic_segmentation.py:564-596,638,668): CLS 0, sentence s >= 1, padding = last + 1"""
    r = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(B, L, dtype=torch.long); am = torch.zeros(B, L, dtype=torch.long)
    seg = torch.zeros(B, L, dtype=torch.long); lab = torch.full((B, L), -100, dtype=torch.long)
    for b in range(B):
        n = L if b == 0 else r.randrange(L // 2, L - 3)
        ids[b, :n] = torch.randint(5, vocab - 1, (n,), generator=g); am[b, :n] = 1
        pos, s = 1, 1
        while pos < n:
            e = min(pos + r.randrange(1, 12), n)
            seg[b, pos:e] = s
            lab[b, e - 1] = r.randrange(2)
            pos, s = e, s + 1
        seg[b, n:] = s
    return ids, am, torch.zeros_like(ids), seg, lab


def to_ours(sd):
    """package state dict -> the names oracle/ponet_oracle.py reads (`ponet.` prefix, the five projections); fails on anything unmatched"""
    out, used = {}, set()
    for k, v in sd.items():
        nk = k
        for ours, cands in ALIASES.items():
            for c in cands:
                if ("." + c + ".") in k:
                    nk = k.replace(c, ours); break
        out[nk] = v.detach().clone().float()
        used.add(k)
    need = [f"ponet.encoder.layer.0.{a}.weight" for a in ALIASES]
    missing = [n for n in need if n not in out]
    if missing:
        raise SystemExit("gen_golden_ponet: cannot map the package's parameter names onto the oracle's; unmatched: " + ", ".join(missing)
                         + "\n  package names of layer 0: " + ", ".join(k for k in sd if ".layer.0." in k))
    return out


def diff_readings(sd, arch, inputs, hidden, eps):
    from oracle import bert_ts_oracle as O
    from oracle import ponet_oracle as PO
    ids, am, tt, seg, _ = inputs
    table, reading = {}, -1
    for flag in (True, False):
        cfg = O.make_cfg(num_labels=2, ponet_special_tokens_mixing=flag, layer_norm_eps=eps, **arch)
        with torch.no_grad():
            _, hs = PO.ponet_encode(sd, cfg, ids, am, tt, seg, return_all=True)
        valid = (am == 1)[..., None]
        errs = [float(((a - torch.as_tensor(b)) * valid).abs().max()) for a, b in zip(hs, hidden)]
        table[flag] = errs
        print(f"  oracle with ponet_special_tokens_mixing={flag}: max |hidden state - package| per layer (valid tokens): "
              + " ".join(f"{e:.2e}" for e in errs))
        if max(errs) < 1e-4 and reading < 0:
            reading = 1 if flag else 0
    return reading, table


def run(arch, B, L, seed, name, PoNetModel, PoNetConfig, store_weights, with_grads):
    m, cfg, kind = build_model(PoNetModel, PoNetConfig, arch, seed)
    inputs = make_inputs(B, L, arch["vocab_size"], seed + 1)
    ids, am, tt, seg, lab = inputs
    m.eval()
    with torch.no_grad():
        o = m(input_ids=ids, attention_mask=am, token_type_ids=tt, segment_ids=seg, labels=lab, output_hidden_states=True, return_dict=True)
    hidden = [h.detach().float().numpy() for h in o.hidden_states]
    sd = to_ours(m.state_dict())
    reading, table = diff_readings(sd, arch, inputs, hidden, getattr(cfg, "layer_norm_eps", 1e-12))
    z = dict(input_ids=ids.numpy(), attention_mask=am.numpy(), token_type_ids=tt.numpy(), segment_ids=seg.numpy(), labels=lab.numpy(),
             logits=o.logits.detach().float().numpy(), eval_loss=np.float32(o.loss.item()), reading=np.int64(reading), seed=np.int64(seed),
             layer_norm_eps=np.float64(getattr(cfg, "layer_norm_eps", 1e-12)), generated_with=np.array(kind),
             err_mixing_true=np.array(table[True]), err_mixing_false=np.array(table[False]))
    for k, v in arch.items():
        z["arch." + k] = np.int64(v)
    if store_weights:
        for k, v in sd.items():
            z["sd." + k] = v.numpy()
        for i, h in enumerate(hidden):
            z[f"hidden.{i}"] = h
    else:
        z["hidden.first"], z["hidden.last"] = hidden[0], hidden[-1]
    if with_grads:
        m.train()
        m.zero_grad()
        o = m(input_ids=ids, attention_mask=am, token_type_ids=tt, segment_ids=seg, labels=lab, return_dict=True)
        o.loss.backward()
        z["train_loss"] = np.float32(o.loss.item())
        grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        for k, v in to_ours(grads).items():
            z["grad." + k] = v.numpy()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **z)
    print(f"wrote {path} ({kind}); reading = {reading} (1: special tokens mix, 0: they do not, -1: the oracle disagrees with the package)")
    return reading


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", action="store_true", help="also write the 12 x 768, L = 4096 fixture (minutes of CPU time)")
    args = ap.parse_args()
    PoNetModel, PoNetConfig = import_ponet()
    if PoNetModel is None:
        return 3
    r = run(TINY, 2, 64, 0, "ponet_tiny.npz", PoNetModel, PoNetConfig, store_weights=True, with_grads=True)
    if args.base:
        run(BASE, 1, 4096, 0, "ponet_base_L4096.npz", PoNetModel, PoNetConfig, store_weights=False, with_grads=False)
    if r < 0:
        print("NEITHER reading reproduces the package: fix oracle/ponet_oracle.py (and spokennlp_amd/ponet.py with it) before trusting row a11.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
