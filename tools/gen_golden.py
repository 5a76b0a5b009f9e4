#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (never copying it).

Runs only in the build container, where /root/reference exists; the fixtures (inputs + expected outputs, data only)
travel with the repo, the reference does not.  Recipe = SURVEY.md Appendix A-1 / A-4:
  * sys.path -> /root/reference/emnlp2023-topic_segmentation/src, import models.bert_for_ts
  * custom config flags set by hand (the driver does it at ts_sentence_seq_labeling.py:196-197)
  * train mode needs the harness-side `torch` proxy (the reference's `loss += ...` on a CPU leaf raises otherwise)
Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/emnlp2023-topic_segmentation/src"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from transformers import BertConfig  # noqa: E402
import models.bert_for_ts as ref_bt  # noqa: E402
import models.longformer_for_ts as ref_lf  # noqa: E402
import models.electra_for_ts as ref_el  # noqa: E402
import models.bigbird_for_ts as ref_bb  # noqa: E402
import models.modules.loss_calculator as ref_lc  # noqa: E402
from transformers import LongformerConfig, ElectraConfig, BigBirdConfig  # noqa: E402
from spokennlp_amd import data  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class TorchProxy:
    """what `.to('cuda')` yields in the reference environment: a non-leaf copy of the zero loss."""

    def __getattr__(self, k):
        return getattr(torch, k)

    def tensor(self, *a, **kw):
        t = torch.tensor(*a, **kw)
        return t.clone() if t.requires_grad else t


ref_bt.torch = TorchProxy()
ref_lc.torch = TorchProxy()
ref_lf.torch = TorchProxy()
ref_el.torch = TorchProxy()
ref_bb.torch = TorchProxy()

FULL = dict(do_da_ts=True, do_cssl=True, do_tssp=True, ts_loss_weight=1.0, ts_score_predictor="lt", ts_score_predictor_cos_temp=1,
            focal_loss_gamma=0.0, weight_label_zero=0.5, cl_loss_weight=0.5, cl_temp=0.1, cl_anchor_level="eop_list",
            cl_positive_k=1, cl_negative_k=3, tssp_loss_weight=1.0, tssp_ablation="none", num_tssp_labels=3)
PLAIN = dict(do_da_ts=False, do_cssl=False, do_tssp=False, ts_loss_weight=1.0, ts_score_predictor="lt", ts_score_predictor_cos_temp=1,
             focal_loss_gamma=0.0, weight_label_zero=0.5, cl_loss_weight=0.0, cl_temp=1, cl_anchor_level="eop_matrix",
             cl_positive_k=1, cl_negative_k=1, tssp_loss_weight=0.0, tssp_ablation="none", num_tssp_labels=3)


def make_model(arch, flags, seed, kind="bert"):
    C = {"bert": BertConfig, "longformer": LongformerConfig, "electra": ElectraConfig, "bigbird": BigBirdConfig}[kind]
    cfg = C(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **arch)
    for k, v in flags.items():
        setattr(cfg, k, v)
    torch.manual_seed(seed)
    if kind == "bert":
        m = ref_bt.BertWithDAForSentenceLabelingTopicSegmentation(cfg)
    elif kind == "electra":
        m = ref_el.ElectraWithDAForSentenceLabelingTopicSegmentation(cfg)
        # reference quirk: electra_for_ts.py builds `self.electra` (:25) but its forward calls `self.bert(...)` (:52) and
        # raises AttributeError as shipped; the harness aliases the attribute (not registered as a sub-module, so the
        # state-dict names stay `electra.*`) to obtain the values the code evidently intends
        object.__setattr__(m, "bert", m.electra)
    elif kind == "bigbird":
        m = ref_bb.BigBirdWithDAForSentenceLabelingTopicSegmentation(cfg)
    else:
        m = ref_lf.LongformerWithDAForSentenceLabelingTopicSegmentation(cfg)
    with torch.no_grad():       # O(1) logits so that parity is meaningful
        m.loss_calculator.classifier.weight.normal_(0, 0.3)
        m.loss_calculator.tssp.classifier.weight.normal_(0, 0.3)
        for n, p in m.named_parameters():
            if "LayerNorm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
    return m, cfg


def run_case(name, arch, L, B, seed, variants, kind="bert"):
    if L >= 512:                    # long documents so that the windows are mostly full (one of them ends in padding)
        docs = data.synth_docs(8, seed=seed + 11, vocab=arch["vocab_size"], mean_sents=L // 9, sd_sents=L // 40, mean_boundaries=6,
                               mu_tok=1.9, sigma_tok=0.4)
    else:
        docs = data.synth_docs(8, seed=seed + 11, vocab=arch["vocab_size"], mean_sents=14, sd_sents=5, mean_boundaries=3,
                               mu_tok=1.4 + 0.2 * (L > 64), sigma_tok=0.4)
    batch = data.batches_from_docs(docs, L, B, seed=seed)[0]
    if kind == "longformer":        # RoBERTa convention: pad id 1 (position ids depend on it); no real token has id 1
        batch["input_ids"] = torch.where(batch["attention_mask"] == 0, torch.full_like(batch["input_ids"], arch["pad_token_id"]),
                                         batch["input_ids"])
    akeys = [k for k in arch if k != "attention_window"]
    out = {"arch_keys": np.array(akeys), "arch_vals": np.array([arch[k] for k in akeys]), "L": L, "B": B}
    if "attention_window" in arch:
        out["attention_window"] = np.array(arch["attention_window"])
    for k, v in batch.items():
        out["in." + k] = v.numpy()
    saved_sd = False
    for vname, flags, mode, rseed, ts_over in variants:
        fl = dict(flags); fl.update(ts_over)
        m, cfg = make_model(arch, fl, seed, kind)
        if not saved_sd:
            for k, v in m.state_dict().items():
                if "position_ids" in k or "token_type_ids" in k.split(".")[-1]:
                    continue
                out["sd." + k] = v.numpy().copy()
            saved_sd = True
        random.seed(rseed)
        if mode == "eval_raises":       # record that (and how) the reference itself fails on this configuration
            m.eval()
            try:
                with torch.no_grad():
                    m(**batch)
                out[f"{vname}.raised"] = np.array("")
            except Exception as e:      # noqa: BLE001
                out[f"{vname}.raised"] = np.array(type(e).__name__)
                out[f"{vname}.message"] = np.array(str(e)[:200])
            out[f"{vname}.flags_keys"] = np.array(list(fl.keys())); out[f"{vname}.flags_vals"] = np.array([str(v) for v in fl.values()])
            print(name, vname, "raised", out[f"{vname}.raised"])
            continue
        if mode == "eval":
            m.eval()
            with torch.no_grad():
                res = m(**batch, output_hidden_states=True)
            loss, logits, cos = res[0], res[1], res[2]
            out[f"{vname}.loss"] = loss.numpy(); out[f"{vname}.logits"] = logits.numpy(); out[f"{vname}.cos"] = cos.numpy()
            if len(res) > 3 and vname == "plain_eval":
                for i, h in enumerate(res[3]):
                    if L < 512 or i == len(res[3]) - 1:          # long cases: final hidden state only (fixture size)
                        out[f"{vname}.hidden{i}"] = h.numpy()
        else:
            m.train()
            loss, logits, cos = m(**batch)[:3]
            loss.backward()
            out[f"{vname}.loss"] = loss.detach().numpy(); out[f"{vname}.logits"] = logits.detach().numpy()
            gn = {}
            for n, p in m.named_parameters():
                if p.grad is None:
                    gn[n] = -1.0
                    continue
                gn[n] = float(p.grad.norm())
                if (vname.endswith("full") or "cos" in vname) and (L < 512 or "embeddings" not in n):
                    out[f"{vname}.grad.{n}"] = p.grad.numpy().copy()
            out[f"{vname}.gradnorm_names"] = np.array(list(gn.keys()))
            out[f"{vname}.gradnorm_vals"] = np.array(list(gn.values()), dtype=np.float64)
        out[f"{vname}.flags_keys"] = np.array(list(fl.keys())); out[f"{vname}.flags_vals"] = np.array([str(v) for v in fl.values()])
        out[f"{vname}.random_seed"] = rseed
        print(name, vname, "loss", float(loss))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    arch = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                max_position_embeddings=128, type_vocab_size=2)
    variants = [
        ("plain_eval", PLAIN, "eval", 0, {}),
        ("full_eval", FULL, "eval", 5, {}),
        ("train_full", FULL, "train", 7, {}),
        ("train_eop_matrix", FULL, "train", 7, dict(cl_anchor_level="eop_matrix", cl_temp=0.5)),
        ("train_eot_list", FULL, "train", 9, dict(cl_anchor_level="eot_list")),
        ("train_focal", PLAIN, "train", 3, dict(focal_loss_gamma=2.0, weight_label_zero=0.7)),
        ("train_wce", PLAIN, "train", 3, dict(weight_label_zero=0.3)),
    ]
    run_case("tiny_L64", arch, 64, 2, 0, variants)
    run_case("tiny_L128", arch, 128, 2, 1, variants[:3])
    run_case("electra_tiny_L64", dict(arch, embedding_size=128), 64, 2, 4, variants[:3], kind="electra")
    lf = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              max_position_embeddings=130, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5)
    run_case("lf_tiny_L64_w8", dict(lf, attention_window=[16, 16]), 64, 2, 2, variants[:3], kind="longformer")
    run_case("lf_tiny_L128_w16", dict(lf, attention_window=[32, 32]), 128, 2, 3, variants[:3], kind="longformer")
    if "--bigbird" in sys.argv or "--all" in sys.argv:
        main_bigbird(variants)


def main_cos():
    """ts_score_predictor="cos" (loss_calculator.py:45-48, utils.py:111-138): the token score is sigmoid(cos(eop_i, eop_{i+1}) / temp), the
    loss BCE-with-logits over the (B, k) matrix INCLUDING its -100 padding (targets of -100: the reference's quirk, kept).  Same model
    seed / documents as tiny_L64 / tiny_L128, separate files so the existing fixtures stay byte-identical.  With the DA pass on, the
    reference concatenates (B, k_anchor) with (B, k_da) scores (bert_for_ts.py:108): recorded as what it does on this batch."""
    arch = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                max_position_embeddings=128, type_vocab_size=2)
    cos1 = dict(ts_score_predictor="cos", ts_score_predictor_cos_temp=1)
    cos05 = dict(ts_score_predictor="cos", ts_score_predictor_cos_temp=0.5)
    variants = [
        ("eval_cos", PLAIN, "eval", 0, cos1),
        ("eval_cos_t05", PLAIN, "eval", 0, cos05),
        ("train_cos", PLAIN, "train", 3, cos1),
        ("train_cos_t05", PLAIN, "train", 3, cos05),
        ("full_eval_cos", FULL, "eval", 5, cos1),      # k_anchor == k_da on these batches (the DA half permutes the same sentences): the cat at bert_for_ts.py:108 works
    ]
    run_case("tiny_L64_cos", arch, 64, 2, 0, variants)
    run_case("tiny_L128_cos", arch, 128, 2, 1, variants)


def main_bigbird(variants):
    """BigBird ([hf] BigBirdModel under bigbird_for_ts.py:27): block_size 64, 3 random blocks.  L = 1024 takes the reference's
    "old plan" random-block path, L = 768 the `_get_rand_attn_plan` path, L = 128 falls back to full attention."""
    bb = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              max_position_embeddings=1024, type_vocab_size=2, block_size=64, num_random_blocks=3, attention_type="block_sparse",
              pad_token_id=0, bos_token_id=1, eos_token_id=2, sep_token_id=3)
    run_case("bb_tiny_L1024", bb, 1024, 1, 6, variants[:3], kind="bigbird")
    run_case("bb_tiny_L768", bb, 768, 1, 7, variants[:3], kind="bigbird")
    run_case("bb_tiny_L128", bb, 128, 2, 8, variants[:3], kind="bigbird")


def main_mmvts():
    """mmvts text branch: the reference's TextEncoder (mmvts/src/models/text_encoder/text_encoder.py) over BertModel and over
    LongformerModel with global_attention_mask=None; eval features + gradients of sum(features * fixed weights) in train mode."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_mmvts_text_encoder", "/root/reference/mmvts/src/models/text_encoder/text_encoder.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    base = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cases = [("mmvts_text_bert_L128", BertConfig(max_position_embeddings=128, type_vocab_size=2, **base), "tiny_bert", 128, 0),
             ("mmvts_text_lf_L256", LongformerConfig(max_position_embeddings=258, type_vocab_size=1, pad_token_id=1, bos_token_id=0,
                                                     eos_token_id=2, layer_norm_eps=1e-5, attention_window=[32, 64], **base),
              "tiny_longformer_zh", 256, 1)]
    for name, cfg, enc_name, L, pad in cases:
        cfg.text_encoder_name_or_path, cfg.init_model = enc_name, False
        torch.manual_seed(21)
        m = mod.TextEncoder(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "LayerNorm.weight" in n:
                    p.add_(0.1 * torch.randn_like(p))
                if n.endswith("bias"):
                    p.add_(0.05 * torch.randn_like(p))
        g = torch.Generator().manual_seed(5)
        B = 2
        ids = torch.randint(3, 200, (B, L), generator=g)
        am = torch.ones(B, L, dtype=torch.long)
        am[1, L - 41:] = 0
        ids = torch.where(am == 0, torch.full_like(ids, pad), ids)
        tt = torch.zeros(B, L, dtype=torch.long)
        wts = torch.randn(B, L, cfg.hidden_size, generator=g) * am.unsqueeze(-1)
        out = {"in.input_ids": ids.numpy(), "in.attention_mask": am.numpy(), "in.token_type_ids": tt.numpy(), "in.loss_weights": wts.numpy(),
               "kind": np.array(m.encoder_type)}
        for k, v in m.state_dict().items():
            if "position_ids" in k or k.endswith("token_type_ids"):
                continue
            out["sd." + k] = v.numpy().copy()
        m.eval()
        with torch.no_grad():
            out["eval.features"] = m(ids, attention_mask=am, token_type_ids=tt).numpy()
        m.train()
        f = m(ids, attention_mask=am, token_type_ids=tt)
        (f * wts).sum().backward()
        out["train.features"] = f.detach().numpy()
        for n, p in m.named_parameters():
            if p.grad is not None:
                out["train.grad." + n] = p.grad.numpy().copy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_fullsize():
    """bert-base shape (12 x 768, vocab 30523, L = 512, 2 samples = 4 sequences, the run_finetune.sh flags): the REFERENCE run on a
    state dict that tests/util.tiny_state_dict regenerates from its seed on any machine, so only inputs and outputs are stored."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import tiny_state_dict
    arch = dict(vocab_size=30523, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=512, type_vocab_size=2)
    sd = tiny_state_dict(arch, seed=20230924, std=0.03)
    docs = data.synth_docs(6, seed=99, vocab=30523, mean_sents=60, sd_sents=10, mean_boundaries=5)
    batch = data.batches_from_docs(docs, 512, 2, seed=3)[0]
    out = {"seed": 20230924, "std": 0.03, "arch_keys": np.array(list(arch)), "arch_vals": np.array([arch[k] for k in arch])}
    for k, v in batch.items():
        out["in." + k] = v.numpy()
    for vname, flags, mode, rseed in (("full_eval", FULL, "eval", 5), ("train_full", FULL, "train", 7)):
        m, cfg = make_model(arch, flags, 0, "bert")
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        random.seed(rseed)
        if mode == "eval_raises":       # record that (and how) the reference itself fails on this configuration
            m.eval()
            try:
                with torch.no_grad():
                    m(**batch)
                out[f"{vname}.raised"] = np.array("")
            except Exception as e:      # noqa: BLE001
                out[f"{vname}.raised"] = np.array(type(e).__name__)
                out[f"{vname}.message"] = np.array(str(e)[:200])
            out[f"{vname}.flags_keys"] = np.array(list(fl.keys())); out[f"{vname}.flags_vals"] = np.array([str(v) for v in fl.values()])
            print(name, vname, "raised", out[f"{vname}.raised"])
            continue
        if mode == "eval":
            m.eval()
            with torch.no_grad():
                loss, logits, cos = m(**batch)[:3]
            out[f"{vname}.logits"] = logits.numpy(); out[f"{vname}.cos"] = cos.numpy()
        else:
            m.train()
            loss, logits, cos = m(**batch)[:3]
            loss.backward()
            names, norms = [], []
            for n, p in m.named_parameters():
                names.append(n); norms.append(-1.0 if p.grad is None else float(p.grad.norm()))
                if p.grad is not None and (n.endswith("bias") or "LayerNorm" in n) and (".layer.0." in n or ".layer.11." in n or "loss_calculator" in n):
                    out[f"{vname}.grad.{n}"] = p.grad.numpy().copy()
            out[f"{vname}.gradnorm_names"] = np.array(names); out[f"{vname}.gradnorm_vals"] = np.array(norms)
        out[f"{vname}.loss"] = loss.detach().numpy()
        out[f"{vname}.flags_keys"] = np.array(list(flags.keys())); out[f"{vname}.flags_vals"] = np.array([str(v) for v in flags.values()])
        out[f"{vname}.random_seed"] = rseed
        print("bert_base_L512", vname, "loss", float(loss))
    path = os.path.join(OUT, "bert_base_L512.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_fullsize_lf():
    """longformer-base-4096 shape (12 x 768, window 512, vocab 50266, L = 4096, 1 sample = 2 sequences), eval only"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import longformer_state_dict
    arch = dict(vocab_size=50266, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=4098, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5)
    sd = longformer_state_dict(arch, seed=20230925, std=0.03)
    docs = data.synth_docs(6, seed=98, vocab=50266, mean_sents=200, sd_sents=30, mean_boundaries=8)
    batch = data.batches_from_docs(docs, 4096, 1, seed=4)[0]
    batch["input_ids"] = torch.where(batch["attention_mask"] == 0, torch.ones_like(batch["input_ids"]), batch["input_ids"])
    batch["input_ids"] = torch.where((batch["input_ids"] == 1) & (batch["attention_mask"] == 1), torch.full_like(batch["input_ids"], 5), batch["input_ids"])
    akeys = list(arch)
    out = {"seed": 20230925, "std": 0.03, "arch_keys": np.array(akeys), "arch_vals": np.array([arch[k] for k in akeys]),
           "attention_window": np.array([512] * 12)}
    for k, v in batch.items():
        out["in." + k] = v.numpy()
    m, cfg = make_model(dict(arch, attention_window=[512] * 12), FULL, 0, "longformer")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    random.seed(5)
    m.eval()
    import time
    t0 = time.time()
    with torch.no_grad():
        loss, logits, cos = m(**batch)[:3]
    print("longformer_base_L4096 full_eval loss", float(loss), "in", round(time.time() - t0, 1), "s; valid tokens", int(batch["attention_mask"].sum()))
    out["full_eval.logits"] = logits.numpy(); out["full_eval.cos"] = cos.numpy(); out["full_eval.loss"] = loss.numpy()
    out["full_eval.flags_keys"] = np.array(list(FULL.keys())); out["full_eval.flags_vals"] = np.array([str(v) for v in FULL.values()])
    out["full_eval.random_seed"] = 5
    path = os.path.join(OUT, "longformer_base_L4096.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_fullsize_lf_train():
    """ADDS one train step of the reference (loss, every parameter's gradient norm, the bias / LayerNorm / *_global gradients of the first
    and last layer) on the inputs of longformer_base_L4096.npz to that fixture: BASELINE config 5 is a TRAINING configuration"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import longformer_state_dict
    path = os.path.join(OUT, "longformer_base_L4096.npz")
    z = dict(np.load(path, allow_pickle=False))
    arch = {k: int(v) if float(v) == int(float(v)) else float(v) for k, v in zip(z["arch_keys"].tolist(), z["arch_vals"].tolist())}
    arch["layer_norm_eps"] = 1e-5
    sd = longformer_state_dict(arch, seed=int(z["seed"]), std=float(z["std"]))
    batch = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("in.")}
    m, cfg = make_model(dict(arch, attention_window=[512] * 12), FULL, 0, "longformer")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    random.seed(7)
    m.train()
    import time
    t0 = time.time()
    loss, logits, cos = m(**batch)[:3]
    loss.backward()
    print("longformer_base_L4096 train_full loss", float(loss), "in", round(time.time() - t0, 1), "s")
    names, norms = [], []
    for n, p in m.named_parameters():
        names.append(n); norms.append(-1.0 if p.grad is None else float(p.grad.norm()))
        small = n.endswith("bias") or "LayerNorm" in n
        if p.grad is not None and small and (".layer.0." in n or ".layer.11." in n or "loss_calculator" in n):
            z[f"train_full.grad.{n}"] = p.grad.numpy().copy()
    z["train_full.gradnorm_names"] = np.array(names); z["train_full.gradnorm_vals"] = np.array(norms)
    z["train_full.loss"] = loss.detach().numpy(); z["train_full.logits_anchor_labelled"] = logits.detach()[:, 0][batch["labels"][:, 0] != -100].numpy()
    z["train_full.flags_keys"] = np.array(list(FULL.keys())); z["train_full.flags_vals"] = np.array([str(v) for v in FULL.values()])
    z["train_full.random_seed"] = 7
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_fullsize_bb():
    """bigbird-roberta-base shape (12 x 768, block 64, 3 random blocks, gelu_new, vocab 50359), L = 4096, 2 sequences, eval only"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import tiny_state_dict
    arch = dict(vocab_size=50359, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=4096, type_vocab_size=2, block_size=64, num_random_blocks=3, attention_type="block_sparse",
                pad_token_id=0, bos_token_id=1, eos_token_id=2, sep_token_id=3)
    num_arch = {k: v for k, v in arch.items() if not isinstance(v, str)}
    sd = {k: v for k, v in tiny_state_dict(num_arch, seed=20230926, std=0.03).items() if "pooler" not in k}   # BigBirdModel.pooler is a bare Linear, unused here
    docs = data.synth_docs(6, seed=97, vocab=50359, mean_sents=200, sd_sents=30, mean_boundaries=8)
    batch = data.batches_from_docs(docs, 4096, 1, seed=6)[0]
    out = {"seed": 20230926, "std": 0.03, "arch_keys": np.array(list(arch)), "arch_vals": np.array([arch[k] for k in arch])}
    for k, v in batch.items():
        out["in." + k] = v.numpy()
    m, cfg = make_model(arch, FULL, 0, "bigbird")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    random.seed(5)
    m.eval()
    import time
    t0 = time.time()
    with torch.no_grad():
        loss, logits, cos = m(**batch)[:3]
    print("bigbird_base_L4096 full_eval loss", float(loss), "in", round(time.time() - t0, 1), "s; valid tokens", int(batch["attention_mask"].sum()))
    out["full_eval.logits"] = logits.numpy(); out["full_eval.cos"] = cos.numpy(); out["full_eval.loss"] = loss.numpy()
    out["full_eval.flags_keys"] = np.array(list(FULL.keys())); out["full_eval.flags_vals"] = np.array([str(v) for v in FULL.values()])
    out["full_eval.random_seed"] = 5
    path = os.path.join(OUT, "bigbird_base_L4096.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    if "--electra-small-only" in sys.argv:
        # electra-small's shape family: embedding_size != hidden_size -> ElectraModel.embeddings_project (electra_for_ts.py:25 builds the stock ElectraModel)
        arch_s = dict(vocab_size=200, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, max_position_embeddings=128,
                      type_vocab_size=2, embedding_size=128)
        variants = [("plain_eval", PLAIN, "eval", 0, {}), ("full_eval", FULL, "eval", 5, {}), ("train_full", FULL, "train", 7, {})]
        run_case("electra_small_tiny_L64", arch_s, 64, 2, 6, variants, kind="electra")
    elif "--cos-only" in sys.argv:
        main_cos()
    elif "--fullsize-bb-only" in sys.argv:
        main_fullsize_bb()
    elif "--fullsize-lf-train-only" in sys.argv:
        main_fullsize_lf_train()
    elif "--fullsize-lf-only" in sys.argv:
        main_fullsize_lf()
    elif "--fullsize-only" in sys.argv:
        main_fullsize()
    elif "--mmvts-only" in sys.argv:
        main_mmvts()
    elif "--bigbird-unaligned-only" in sys.argv:
        # L = 1000 is not a multiple of block_size: the reference pads to 1024 itself (`_pad_to_block_size`) and cuts the output back
        bb = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                  max_position_embeddings=1024, type_vocab_size=2, block_size=64, num_random_blocks=3, attention_type="block_sparse",
                  pad_token_id=0, bos_token_id=1, eos_token_id=2, sep_token_id=3)
        run_case("bb_tiny_L1000", bb, 1000, 1, 9, [("plain_eval", PLAIN, "eval", 0, {}), ("full_eval", FULL, "eval", 5, {}),
                                                    ("train_full", FULL, "train", 7, {})], kind="bigbird")
    elif "--longformer-unaligned-only" in sys.argv:
        # L = 100 is not a multiple of the attention window: the reference pads to 128 itself (`_pad_to_window_size`) and cuts back
        lf = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                  max_position_embeddings=130, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5)
        run_case("lf_tiny_L100_w16", dict(lf, attention_window=[32, 32]), 100, 3, 12,
                 [("plain_eval", PLAIN, "eval", 0, {}), ("full_eval", FULL, "eval", 5, {}), ("train_full", FULL, "train", 7, {})], kind="longformer")
    elif "--bert-unaligned-only" in sys.argv:
        arch = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    max_position_embeddings=128, type_vocab_size=2)
        run_case("tiny_L100_B3", arch, 100, 3, 13, [("plain_eval", PLAIN, "eval", 0, {}), ("full_eval", FULL, "eval", 5, {}),
                                                    ("train_full", FULL, "train", 7, {})])
    elif "--bigbird-only" in sys.argv:
        main_bigbird([("plain_eval", PLAIN, "eval", 0, {}), ("full_eval", FULL, "eval", 5, {}), ("train_full", FULL, "train", 7, {})])
    else:
        main()
