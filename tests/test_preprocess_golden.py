"""a9: the restated feature builder (spokennlp_amd/preprocess.py) is BIT-EXACT with the reference's closures
(ts_sentence_seq_labeling.py:336-934) on the golden vectors of tools/gen_golden_preprocess.py; a10: decode + writer."""
import json
import os
import random

import numpy as np
import pytest

from spokennlp_amd import preprocess as P

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz"))
CASES = [str(c) for c in G["cases"]]
COLS = ["labels", "input_ids", "token_type_ids", "attention_mask", "sent_level_labels", "extract_eop_segment_ids",
        "eop_index_for_aggregate_batch_eop_features", "sent_pair_orders", "sent_token_mask", "example_id", "sentence_range"]


def load_case(name):
    nd, L, seed, bos, cls, pad = [int(v) for v in G[name + ".meta"]]
    tok, off = G[name + ".sent_tokens"], G[name + ".sent_off"]
    nsent = G[name + ".doc_nsent"]; labs = G[name + ".sent_labels"].tolist()
    docs, dl, s = [], [], 0
    for n in nsent:
        docs.append([[bos] + tok[off[i]:off[i + 1]].tolist() for i in range(s, s + n)])
        dl.append(labs[s:s + n]); s += n
    return docs, dl, L, seed, bos, cls, pad, str(G[name + ".ablation"])


@pytest.mark.parametrize("name", CASES)
def test_prepare_features_bit_exact(name):
    docs, dl, L, seed, bos, cls, pad, abl = load_case(name)
    assert str(G[name + ".error"]) == ""
    random.seed(seed)
    cols = P.prepare_features(docs, dl, list(range(len(docs))), L, bos, cls, pad, tssp_ablation=abl)
    for c in COLS:
        got = np.array(cols[c], dtype=np.int32)
        exp = G[name + "." + c]
        assert got.shape == exp.shape, (c, got.shape, exp.shape)
        assert np.array_equal(got, exp), c


def test_window_invariants():
    """size-independent properties on a larger random corpus: every sentence is labelled in exactly one anchor window,
    windows are exactly max_seq_length long, masks agree with padding, eop indices count the labelled sentences."""
    r = random.Random(5)
    bos, cls, pad, L = 5, 2, 0, 128
    docs, labels = [], []
    for _ in range(20):
        n = r.randrange(5, 80)
        docs.append([[bos] + [r.randrange(10, 999) for _ in range(r.randrange(1, 40))] for _ in range(n)])
        lab = [0 if r.random() < 0.2 else 1 for _ in range(n)]; lab[-1] = 0
        labels.append(lab)
    random.seed(1)
    cols = P.prepare_features(docs, labels, list(range(20)), L, bos, cls, pad)
    ids = np.array(cols["input_ids"]); lab = np.array(cols["labels"]); am = np.array(cols["attention_mask"])
    assert ids.shape[1:] == (2, L)
    assert ((ids != pad) == (am == 1)).all()
    assert (ids[:, :, 0] == cls).all()
    per_doc = np.zeros(20, dtype=int)
    for w in range(len(ids)):
        per_doc[cols["example_id"][w][0]] += int((lab[w, 0] != -100).sum())
        k = int((lab[w, 0] != -100).sum())
        assert cols["eop_index_for_aggregate_batch_eop_features"][w][0][:k + 1] == list(range(k + 1))
        assert max(cols["extract_eop_segment_ids"][w][0]) == k
        assert set(lab[w, 0][ids[w, 0] != bos].tolist()) <= {-100}
    # all but the window-final (shared) sentences are labelled once; the document's last sentence is never labelled
    for d in range(20):
        assert per_doc[d] <= len(docs[d]) - 1


def test_decode_and_writer(tmp_path):
    rng = np.random.default_rng(0)
    N, L = 5, 16
    logits = rng.standard_normal((N, 2, L, 2)).astype(np.float32)
    labels = np.full((N, 2, L), -100); labels[:, :, [1, 4, 9]] = rng.integers(0, 2, (N, 2, 3))
    dec = P.decode_anchor_predictions(logits, labels)
    for i, rec in enumerate(dec):
        assert rec["pred_ids"] == [int(np.argmax(logits[i, 0, p])) for p in (1, 4, 9)]
        assert rec["predictions"] == [P.LABEL_LIST[v] for v in rec["pred_ids"]]
        assert rec["int_labels"] == labels[i, 0, [1, 4, 9]].tolist()
        assert P.boundary_indices(rec["pred_ids"]) == [j for j, v in enumerate(rec["pred_ids"]) if v == 0]
    docs = P.merge_windows_to_documents(dec, [0, 0, 1, 1, 1], 2)
    assert len(docs[0]["predictions"]) == 6 and len(docs[1]["predictions"]) == 9
    path = tmp_path / "predict.txt"
    P.write_prediction_file(str(path), docs)
    lines = open(path).read().splitlines()
    assert len(lines) == 2 and json.loads(lines[1])["int_labels"] == docs[1]["int_labels"]


# ---------------------------------------------------------------------------------------------------- PoNet (a12)
PONET_CASES = [str(c) for c in G["ponet_cases"]]


@pytest.mark.parametrize("name", PONET_CASES)
def test_ponet_features_bit_exact(name):
    nd, L, seed, eos, cls, pad, para = [int(v) for v in G[name + ".meta"]]
    tok, off = G[name + ".sent_tokens"], G[name + ".sent_off"]
    nsent = G[name + ".doc_nsent"]; labs = G[name + ".sent_labels"].tolist()
    docs, dl, s = [], [], 0
    for n in nsent:
        docs.append([tok[off[i]:off[i + 1]].tolist() + [eos] for i in range(s, s + n)])
        dl.append(labs[s:s + n]); s += n
    cols = P.ponet_prepare_features(docs, dl, list(range(nd)), L, eos, cls, pad, use_paragraph_segment=bool(para))
    for c in ("input_ids", "token_type_ids", "attention_mask", "segment_ids", "example_id", "labels"):
        got, exp = np.array(cols[c], dtype=np.int32), G[name + "." + c]
        assert got.shape == exp.shape and np.array_equal(got, exp), c
    assert [b - a for a, b in cols["sentence_range"]] == G[name + ".num_sentences"].tolist()


def test_es_collect_predictions():
    """extractive summarisation decode (ponet_extractive_summarization.py:853-905): windows -> documents, the appended "O" when a
    window's last sentence end was cut by the window edge"""
    feats = P.ponet_prepare_features([[[10, 11, 7], [12, 7], [13, 14, 15, 7], [16, 7]]], [[0, 1, 0, 1]], [0], 8, 7, 2, 0)
    nwin = len(feats["input_ids"])
    assert nwin >= 2
    nsent = [r[1] - r[0] for r in feats["sentence_range"]]
    pred = [[0 if l != -100 else 5 for l in row] for row in feats["labels"]]          # predict "B-EOP" at every labelled [EOS]
    preds, golds = P.es_collect_predictions(pred, feats["labels"], nsent, feats["example_id"], 1)
    assert len(preds[0]) == len(golds[0]) == sum(nsent)
    lab_flat = [l for row in feats["labels"] for l in row if l != -100]
    assert sum(1 for v in golds[0] if v in (0, 1)) == len(golds[0]) and len(lab_flat) <= len(golds[0])
    assert P.es_selected_sentences([0, 1, 1, 0]) == [0, 3]
    with pytest.raises(ValueError):
        P.es_collect_predictions(pred, feats["labels"], [n + 2 for n in nsent], feats["example_id"], 1)
