#!/usr/bin/env python
"""BASELINE config 1 golden (run_inference.sh plumbing): the REFERENCE model class, imported from /root/reference (never copied), run on
32 synthetic Wiki-727K-shaped documents windowed by the feature builder (which tests/golden/preprocess.npz pins bit-exactly to the
reference's own closures).  Stores inputs' recipe + the reference's outputs only: anchor logits at the labelled positions of every
window, the cos-sim side output, and the per-document predictions the decode (ts_sentence_seq_labeling.py:1138-1191) derives from them.
Two cases, both on all 32 documents: a tiny model (L = 128) and the bert-base shape (L = 512).
Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_config1.py
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/emnlp2023-topic_segmentation/src")

from transformers import BertConfig  # noqa: E402
import models.bert_for_ts as ref_bt  # noqa: E402
from spokennlp_amd import data, preprocess as P  # noqa: E402
from spokennlp_amd.inference import MODEL_COLUMNS  # noqa: E402
from util import tiny_state_dict  # noqa: E402

# at inference the driver overwrites the saved config with the argument defaults (ts_sentence_seq_labeling.py:196): every auxiliary
# flag is off, one encoder pass, logits[:, 1] duplicates logits[:, 0]
PLAIN = dict(do_da_ts=False, do_cssl=False, do_tssp=False, ts_loss_weight=1.0, ts_score_predictor="lt", ts_score_predictor_cos_temp=1,
             focal_loss_gamma=0.0, weight_label_zero=0.5, cl_loss_weight=0.0, cl_temp=1, cl_anchor_level="eop_matrix",
             cl_positive_k=1, cl_negative_k=1, tssp_loss_weight=0.0, tssp_ablation="none", num_tssp_labels=3)
CASES = {
    "config1_tiny": dict(arch=dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                   max_position_embeddings=128, type_vocab_size=2), sd_seed=31, std=0.05, ndocs=32, L=128, bs=4,
                         docs=dict(seed=2024, mean_sents=30, sd_sents=10, mean_boundaries=4, mu_tok=1.8, sigma_tok=0.5)),
    "config1_bert_base": dict(arch=dict(vocab_size=30523, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                        intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2), sd_seed=20230927,
                              std=0.03, ndocs=32, L=512, bs=2, docs=dict(seed=2025, mean_sents=52, sd_sents=25, mean_boundaries=5.23)),
}


def main():
    for name, c in CASES.items():
        arch = c["arch"]
        sd = tiny_state_dict(arch, seed=c["sd_seed"], std=c["std"])
        cfg = BertConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **arch)
        for k, v in PLAIN.items():
            setattr(cfg, k, v)
        m = ref_bt.BertWithDAForSentenceLabelingTopicSegmentation(cfg)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        m.eval()
        docs = data.synth_docs(c["ndocs"], vocab=arch["vocab_size"], **c["docs"])
        sent_ids = [[s.tolist() for s in d["sentences"]] for d in docs]
        labels = [[0 if v == 1 else 1 for v in d["labels"]] for d in docs]
        random.seed(42)
        cols = P.prepare_features(sent_ids, labels, list(range(len(docs))), c["L"], arch["vocab_size"] - 1, data.CLS_ID, data.PAD_ID)
        n = len(cols["input_ids"])
        lab_logits, cos_rows, counts = [], [], []
        with torch.no_grad():
            for i in range(0, n, c["bs"]):
                idx = list(range(i, min(i + c["bs"], n)))
                batch = {k: torch.tensor([cols[k][j] for j in idx], dtype=torch.long) for k in MODEL_COLUMNS}
                _, logits, cos = m(**batch)[:3]
                for r, j in enumerate(idx):
                    sel = batch["labels"][r, 0] != -100
                    lab_logits.append(logits[r, 0][sel].numpy())
                    counts.append(int(sel.sum()))
                    cos_rows.append(cos[r][:int(sel.sum())].numpy())
                print(name, "windows", i + len(idx), "/", n, flush=True)
        out = dict(sd_seed=c["sd_seed"], std=c["std"], ndocs=c["ndocs"], L=c["L"], arch_keys=np.array(list(arch)),
                   arch_vals=np.array([arch[k] for k in arch]), docs_keys=np.array(list(c["docs"])),
                   docs_vals=np.array([float(v) for v in c["docs"].values()]), n_windows=n, counts=np.array(counts),
                   labelled_logits=np.concatenate(lab_logits, 0).astype(np.float32), cos=np.concatenate(cos_rows, 0).astype(np.float32),
                   input_ids_checksum=np.array([int(np.sum(np.array(cols["input_ids"], dtype=np.int64) * (1 + np.arange(2 * c["L"]).reshape(1, 2, c["L"]) % 97)))]))
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
