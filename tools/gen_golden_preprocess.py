#!/usr/bin/env python
"""Generate tests/golden/preprocess_*.npz: golden vectors of the reference's feature builder.

The preprocessing functions of the reference are closures nested in `main()` of
emnlp2023-topic_segmentation/src/ts_sentence_seq_labeling.py (:336-934) and cannot be imported.  This script parses that
file with `ast` IN MEMORY (nothing is copied into the repo), compiles the nested FunctionDef nodes it needs into a
namespace that supplies their free variables (random, a stub tokenizer, label_to_id, target_specical_ids, config,
max_seq_length, the column names), runs them on small seeded toy documents and stores ONLY inputs + outputs (integers).

The stub tokenizer stands in for HuggingFace's: every "word" of a sentence is a decimal token id, "[BOS]" maps to bos id.
Runs only where /root/reference exists.   Usage:  python tools/gen_golden_preprocess.py
"""
import ast
import os
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/emnlp2023-topic_segmentation/src/ts_sentence_seq_labeling.py"
OUT = os.path.join(ROOT, "tests", "golden")
WANTED = ["get_extract_eop_segment_ids", "get_sample_sent_token_mask", "shuffle_and_replace_doc_topics", "shuffle_topic_sents",
          "get_example_sent_index_to_start_end_token_index", "prepare_augmented_data", "prepare_features_with_dynamic_num_sentence"]
BOS, CLS, PAD = 5, 2, 0
INT_COLS = ["labels", "input_ids", "token_type_ids", "attention_mask", "sent_level_labels", "extract_eop_segment_ids",
            "eop_index_for_aggregate_batch_eop_features", "sent_pair_orders", "sent_token_mask"]


class StubTokenizer:
    bos_token = "[BOS]"
    bos_token_id, cls_token_id, pad_token_id = BOS, CLS, PAD

    def __call__(self, sentences, is_split_into_words=True, add_special_tokens=False, return_token_type_ids=True,
                 return_attention_mask=True):
        assert is_split_into_words and not add_special_tokens
        ids = []
        for doc in sentences:
            row = []
            for s in doc:
                assert s.startswith(self.bos_token)
                row.append(BOS)
                row.extend(int(w) for w in s[len(self.bos_token):].split())
            ids.append(row)
        return {"input_ids": ids, "token_type_ids": [[0] * len(r) for r in ids], "attention_mask": [[1] * len(r) for r in ids]}


def load_closures(max_seq_length, tssp_ablation):
    tree = ast.parse(open(SRC).read())
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    fns = [n for n in ast.walk(main) if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(f.name for f in fns) == sorted(WANTED)
    mod = ast.Module(body=fns, type_ignores=[])
    ns = dict(random=random, tokenizer=StubTokenizer(), target_specical_ids={BOS}, label_to_id={"B-EOP": 0, "O": 1},
              config=types.SimpleNamespace(tssp_ablation=tssp_ablation), max_seq_length=max_seq_length,
              label_column_name="labels", context_column_name="sentences", example_id_column_name="example_id")
    exec(compile(mod, "<reference closures>", "exec"), ns)
    return ns


def toy_docs(n, seed, mean_sents, max_tok, unk_frac=0.0):
    """documents as (sentences: list of token-id lists WITHOUT bos, labels: list of 'B-EOP'/'O'/'X')."""
    r = random.Random(seed)
    docs = []
    for _ in range(n):
        ns = max(2, int(r.gauss(mean_sents, mean_sents / 3)))
        sents = [[r.randrange(10, 200) for _ in range(r.randrange(1, max_tok + 1))] for _ in range(ns)]
        labels = ["B-EOP" if r.random() < 0.25 else "O" for _ in range(ns)]
        labels[-1] = "B-EOP"
        if unk_frac:
            labels = [("X" if (l == "O" and r.random() < unk_frac) else l) for l in labels]
        docs.append((sents, labels))
    return docs


def pack(list_of_lists):
    """ragged list of int lists -> (flat int32, offsets int32)"""
    flat = np.array([v for row in list_of_lists for v in row], dtype=np.int32)
    off = np.cumsum([0] + [len(r) for r in list_of_lists]).astype(np.int32)
    return flat, off


CASES = [  # name, n_docs, mean_sents, max_tok, max_seq_length, seed, tssp_ablation, unk_frac
    ("L24_s0", 4, 9, 6, 24, 0, "none", 0.0),
    ("L24_s42", 5, 12, 9, 24, 42, "none", 0.0),
    ("L64_s0", 4, 20, 12, 64, 0, "none", 0.0),
    ("L64_s7_single", 1, 25, 10, 64, 7, "none", 0.0),       # one document: no topic replacement branch
    ("L16_long_sents", 3, 8, 30, 16, 3, "none", 0.0),        # sentences longer than the window: truncation branch
    ("L32_wo_intra", 4, 12, 8, 32, 1, "wo_intra_topic", 0.0),
    ("L32_wo_inter", 4, 12, 8, 32, 2, "wo_inter_topic", 0.0),
    ("L32_sso", 4, 12, 8, 32, 3, "sso", 0.0),
    ("L32_sso_intra", 4, 12, 8, 32, 4, "sso_and_intra_topic", 0.0),
    ("L48_unknown_labels", 4, 14, 8, 48, 5, "none", 0.3),
]


PONET_SRC = "/root/reference/alimeeting4mug/src/topic_segment/ponet_topic_segmentation.py"
EOS = 7
PONET_CASES = [  # name, n_docs, mean_sents, max_tok, max_seq_length, seed, use_paragraph_segment, unknown-label fraction
    ("ponet_L32_sent", 3, 12, 8, 32, 0, False, 0.0),
    ("ponet_L32_para", 3, 12, 8, 32, 1, True, 0.0),
    ("ponet_L64_para_unk", 4, 25, 10, 64, 2, True, 0.4),
    ("ponet_L16_long", 2, 8, 30, 16, 3, False, 0.0),
    # PoNet extractive summarisation: the same closure name in ponet_extractive_summarization.py:611-768; sentence-level
    # segment ids, a label on every sentence's [EOS] ("es_" prefix selects that source file)
    ("es_ponet_L32", 3, 12, 8, 32, 4, False, 0.0),
    ("es_ponet_L64_unk", 4, 25, 10, 64, 5, False, 0.3),
    ("es_ponet_L16_long", 2, 8, 30, 16, 6, False, 0.0),
]
ES_SRC = "/root/reference/alimeeting4mug/src/extractive_summarization/ponet_extractive_summarization.py"


class PonetStubTokenizer:
    eos_token_id, cls_token_id, pad_token_id = EOS, CLS, PAD

    def __call__(self, sentences, is_split_into_words=True, add_special_tokens=False, return_token_type_ids=True,
                 return_attention_mask=True):
        ids = []
        for doc in sentences:
            row = []
            for s in doc:
                assert s.endswith("[EOS]")
                row.extend(int(w) for w in s[:-5].split())
                row.append(EOS)
            ids.append(row)
        return {"input_ids": ids, "token_type_ids": [[0] * len(r) for r in ids], "attention_mask": [[1] * len(r) for r in ids]}


def _closure(src):
    tree = ast.parse(open(src).read())
    main_fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    fn = [n for n in ast.walk(main_fn) if isinstance(n, ast.FunctionDef) and n.name == "prepare_input_features"]
    assert len(fn) == 1
    return compile(ast.Module(body=fn, type_ignores=[]), "<reference closure>", "exec")


def ponet_cases(out, names):
    codes = {False: _closure(PONET_SRC), True: _closure(ES_SRC)}
    for name, nd, ms, mt, L, seed, para, unk in PONET_CASES:
        code = codes[name.startswith("es_")]
        ns = dict(tokenizer=PonetStubTokenizer(), target_specical_ids={EOS}, label_to_id={"B-EOP": 0, "O": 1},
                  use_paragraph_segment=para, max_seq_length=L, question_column_name="labels", context_column_name="sentences",
                  example_id_column_name="example_id")
        exec(code, ns)
        docs = toy_docs(nd, 2000 + seed, ms, mt, unk)
        examples = {"labels": [d[1] for d in docs], "sentences": [[" ".join(map(str, x)) for x in d[0]] for d in docs],
                    "example_id": list(range(nd))}
        res = ns["prepare_input_features"](examples)
        names.append(name)
        out[name + ".meta"] = np.array([nd, L, seed, EOS, CLS, PAD, int(para)], dtype=np.int32)
        sent_flat, sent_off = pack([s for d in docs for s in d[0]])
        out[name + ".sent_tokens"] = sent_flat; out[name + ".sent_off"] = sent_off
        out[name + ".doc_nsent"] = np.array([len(d[0]) for d in docs], dtype=np.int32)
        lab = {"B-EOP": 0, "O": 1}
        out[name + ".sent_labels"] = np.array([lab.get(l, -100) for d in docs for l in d[1]], dtype=np.int32)
        for c in ("input_ids", "token_type_ids", "attention_mask", "segment_ids", "example_id", "labels"):
            out[name + "." + c] = np.array(res[c], dtype=np.int32)
        out[name + ".num_sentences"] = np.array([len(w) for w in res["sentences"]], dtype=np.int32)
        print(name, "windows", len(res["input_ids"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    out = {}
    names = []
    for name, nd, ms, mt, L, seed, abl, unk in CASES:
        ns = load_closures(L, abl)
        docs = toy_docs(nd, 1000 + seed, ms, mt, unk)
        examples = {"labels": [d[1] for d in docs], "sentences": [[" ".join(map(str, x)) for x in d[0]] for d in docs],
                    "example_id": list(range(nd))}
        random.seed(seed)
        try:
            res = ns["prepare_features_with_dynamic_num_sentence"](examples)
            err = ""
        except (AssertionError, IndexError) as e:       # the reference itself fails on this input: record that it does
            res, err = None, type(e).__name__
        names.append(name)
        out[name + ".meta"] = np.array([nd, L, seed, BOS, CLS, PAD], dtype=np.int32)
        out[name + ".ablation"] = np.array(abl)
        out[name + ".error"] = np.array(err)
        sent_flat, sent_off = pack([s for d in docs for s in d[0]])
        out[name + ".sent_tokens"] = sent_flat; out[name + ".sent_off"] = sent_off
        out[name + ".doc_nsent"] = np.array([len(d[0]) for d in docs], dtype=np.int32)
        lab = {"B-EOP": 0, "O": 1}
        out[name + ".sent_labels"] = np.array([lab.get(l, -100) for d in docs for l in d[1]], dtype=np.int32)
        if res is not None:
            out[name + ".example_id"] = np.array(res["example_id"], dtype=np.int32)
            for c in INT_COLS:
                out[name + "." + c] = np.array(res[c], dtype=np.int32)
            # the "sentences" column holds strings "<sentence index>-<text>": keep only the anchor sentence index range
            rng = [[int(w[0][0].split("-")[0]), int(w[0][-1].split("-")[0]) + 1] for w in res["sentences"]]
            out[name + ".sentence_range"] = np.array(rng, dtype=np.int32)
            print(name, "windows", len(res["input_ids"]), err)
        else:
            print(name, "reference raised", err)
    out["cases"] = np.array(names)
    pn = []
    ponet_cases(out, pn)
    out["ponet_cases"] = np.array(pn)
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **out)
    print("wrote", os.path.join(OUT, "preprocess.npz"), os.path.getsize(os.path.join(OUT, "preprocess.npz")), "bytes")


if __name__ == "__main__":
    main()
