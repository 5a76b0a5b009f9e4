"""f-2: jsonl schema + feature pipeline + batching (CPU)."""
import json
import random

import torch

from spokennlp_amd import loader as LD


class WordTokenizer:
    """stand-in for the HF tokenizer (third party): one id per whitespace word, [BOS] -> 5"""
    bos_token = "[BOS]"
    bos_token_id, cls_token_id, pad_token_id = 5, 2, 0

    def __call__(self, sents, add_special_tokens=False):
        out = []
        for s in sents:
            assert s.startswith(self.bos_token)
            out.append([5] + [10 + (hash(w) % 150) for w in s[len(self.bos_token):].split()])
        return {"input_ids": out}


def test_jsonl_roundtrip_and_pipeline(tmp_path):
    r = random.Random(0)
    docs = []
    for d in range(6):
        n = r.randrange(5, 30)
        labels = [1 if r.random() < 0.2 else 0 for _ in range(n)]; labels[-1] = 1
        docs.append({"file": f"doc{d}", "sentences": [" ".join(f"w{r.randrange(99)}" for _ in range(r.randrange(2, 12))) for _ in range(n)],
                     "labels": labels if d % 2 else [str(v) for v in labels]})
    path = tmp_path / "test.jsonl"
    LD.write_jsonl(str(path), docs)
    exs = list(LD.read_jsonl(str(path)))
    assert [e["example_id"] for e in exs] == list(range(6))
    assert exs[0]["labels"][-1] == "B-EOP" and set(exs[1]["labels"]) <= {"B-EOP", "O"}
    tok = WordTokenizer()
    sent_ids, lab_ids, ids = LD.tokenize_examples(exs, tok)
    assert all(s[0] == 5 for d in sent_ids for s in d) and lab_ids[0][-1] == 0
    random.seed(0)
    feats = LD.build_features(sent_ids, lab_ids, ids, 64, 5, 2, 0)
    n = len(feats["input_ids"])
    b0, b1 = LD.batch_indices(n, 2, 0, 2), LD.batch_indices(n, 2, 1, 2)
    assert not set(sum(b0, [])) & set(sum(b1, [])) and all(len(b) == 2 for b in b0 + b1)
    got = list(LD.DevicePrefetcher(feats, b0, "cpu"))
    assert len(got) == len(b0)
    for idx, batch in zip(b0, got):
        assert batch["input_ids"].shape == (2, 2, 64) and batch["input_ids"].dtype == torch.long
        assert batch["labels"].tolist() == [feats["labels"][j] for j in idx]
    # a malformed line is reported with its line number
    with open(path, "a") as f:
        f.write(json.dumps({"sentences": ["a"], "labels": [0, 1]}) + "\n")
    try:
        list(LD.read_jsonl(str(path)))
        assert False
    except ValueError as e:
        assert ":7:" in str(e)
