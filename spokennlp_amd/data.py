"""Synthetic Wiki-727K-shaped inputs for the topic-segmentation path (no network: no corpus, no tokenizer vocab).

Documents follow the jsonl schema of the reference's converters (preprocess_data.py:160-165,
datasets/wiki727k/wiki727k.py:73-81): a list of sentences and a 0/1 label per sentence, 1 = last sentence of a
section, final sentence always 1.  Sentences are drawn directly as token ids (uniform in [1000, vocab-2]); every
sentence is prefixed by the `[BOS]` special token (ts_sentence_seq_labeling.py:282-286, id = vocab-1 for BERT).

`windows_from_doc` restates the reference's sliding window over sentences (ts_sentence_seq_labeling.py:811-917):
greedy fill up to max_seq_length-1 tokens after [CLS], neighbouring windows share one sentence, the last [BOS] of a
multi-sentence window is unlabelled, padding with id 0 / label -100; and builds the auxiliary index tensors of
:336-364,888-917 (extract_eop_segment_ids, eop_index_for_aggregate_batch_eop_features, sent_token_mask,
sent_level_labels).  The augmented ("DA") half is an in-topic sentence shuffle with the 3-way TSSP labels of
shuffle_topic_sents (:461-505, tssp_ablation == "none"); cross-document topic replacement (:366-459) is not
generated here.

Label ids: "B-EOP" (topic boundary) = 0, "O" = 1  (label list order of the reference driver).
"""
import numpy as np

B_EOP, O_LABEL, IGN = 0, 1, -100
CLS_ID, PAD_ID = 101, 0
COLUMNS = ("input_ids", "attention_mask", "token_type_ids", "labels", "sent_level_labels", "extract_eop_segment_ids",
           "eop_index_for_aggregate_batch_eop_features", "sent_pair_orders", "sent_token_mask")


def synth_docs(n_docs, seed=1234, vocab=30523, mean_sents=52, sd_sents=25, mean_boundaries=5.23, mu_tok=3.1, sigma_tok=0.5):
    """list of docs; doc = dict(sentences=[np.int64 arrays incl. leading BOS], labels=[0/1 per sentence, 1 = section end])"""
    rng = np.random.default_rng(seed)
    bos = vocab - 1
    lo = 1000 if vocab > 2000 else 110
    docs = []
    for _ in range(n_docs):
        ns = int(np.clip(round(rng.normal(mean_sents, sd_sents)), 4, 300))
        pb = min(0.9, mean_boundaries / max(ns - 1, 1))
        labels = (rng.random(ns) < pb).astype(np.int64)
        labels[-1] = 1
        sents = []
        for _s in range(ns):
            nt = int(np.clip(round(rng.lognormal(mu_tok, sigma_tok)), 3, 120))
            sents.append(np.concatenate(([bos], rng.integers(lo, vocab - 1, nt))).astype(np.int64))
        docs.append(dict(sentences=sents, labels=labels.tolist()))
    return docs


def _shuffle_in_topics(sents, sec_end, rng):
    """in-topic shuffle keeping each topic's last sentence last; returns (sentences, section_end flags, tssp labels)."""
    out_s, out_e, out_t = [], [], []
    start = 0
    n = len(sents)
    for i in range(n):
        if sec_end[i] == 1 or i == n - 1:
            idx = list(range(start, i))
            rng.shuffle(idx)
            idx.append(i)
            for j, si in enumerate(idx):
                out_s.append(sents[si]); out_e.append(1 if j == len(idx) - 1 and sec_end[i] == 1 else 0)
                out_t.append(2 if j == 0 else (0 if idx[j - 1] == si - 1 else 1))
            start = i + 1
    return out_s, out_e, out_t


def _features(sent_list, sec_end, tssp, L):
    """one padded sample from a list of sentences (already chosen to fit): returns dict of length-L int lists"""
    ids, lab, pair = [CLS_ID], [IGN], [IGN]
    for s, e, t in zip(sent_list, sec_end, tssp):
        ids += s.tolist()
        lab += [B_EOP if e == 1 else O_LABEL] + [IGN] * (len(s) - 1)
        pair += [t] + [IGN] * (len(s) - 1)
    ids, lab, pair = ids[:L], lab[:L], pair[:L]
    bos_pos = [i for i in range(1, len(ids)) if lab[i] != IGN]
    if len(bos_pos) >= 1:
        lab[bos_pos[-1]] = IGN                      # last sentence of a window is never predicted (:843-849)
    n = len(ids)
    am = [1] * n + [0] * (L - n)
    ids = ids + [PAD_ID] * (L - n); lab = lab + [IGN] * (L - n); pair = pair + [IGN] * (L - n)
    is_bos = [False] * L
    for p in bos_pos:
        is_bos[p] = True
    seg, k = [0] * L, 0
    for i in range(1, L):
        if is_bos[i] and lab[i] != IGN:
            k += 1; seg[i] = k
    eop_index = list(range(k + 1)) + [0] * (L - k - 1)
    stm = [IGN] * L
    for i in range(1, L):
        if is_bos[i]:
            stm[i] = 0 if lab[i] == 0 else 1
    sll = [IGN] + [lab[i] for i in range(1, L) if is_bos[i]]
    sll += [IGN] * (L - len(sll))
    return dict(input_ids=ids, attention_mask=am, token_type_ids=[0] * L, labels=lab, sent_level_labels=sll,
                extract_eop_segment_ids=seg, eop_index_for_aggregate_batch_eop_features=eop_index, sent_pair_orders=pair,
                sent_token_mask=stm)


def windows_from_doc(doc, L, rng):
    """sliding window over sentences; returns list of (anchor, da) feature-dict pairs."""
    sents, sec_end = doc["sentences"], doc["labels"]
    n = len(sents)
    out = []
    left = 0
    while left < n:
        tot, right = 0, left
        while right < n and (tot + len(sents[right]) < L - 1 or right == left):
            tot += len(sents[right]); right += 1
            if tot >= L - 1:
                break
        ws, we = sents[left:right], sec_end[left:right]
        da_s, da_e, da_t = _shuffle_in_topics(ws, we, rng)
        anchor = _features(ws, we, [IGN] * len(ws), L)
        da = _features(da_s, da_e, da_t, L)
        anchor["sent_pair_orders"] = da["sent_pair_orders"]      # both slots hold the DA labels (:882)
        out.append((anchor, da))
        if right >= n:
            break
        left = right - 1 if right - left > 1 else right        # neighbouring windows share one sentence
    return out


def batches_from_docs(docs, L, batch_size, seed=0, as_torch=True):
    """(B,2,L) int64 batches of the 9 model-input columns, in document order (no shuffling)."""
    rng = np.random.default_rng(seed)
    samples = []
    for d in docs:
        for pair in windows_from_doc(d, L, rng):
            if sum(1 for v in pair[0]["labels"] if v != IGN) >= 1:
                samples.append(pair)
    batches = []
    for i in range(0, len(samples) - batch_size + 1, batch_size):
        chunk = samples[i:i + batch_size]
        b = {c: np.array([[a[c], d[c]] for a, d in chunk], dtype=np.int64) for c in COLUMNS}
        if as_torch:
            import torch
            b = {k: torch.from_numpy(v) for k, v in b.items()}
        batches.append(b)
    return batches


def dense_batch(B, L=512, vocab=30523, sent_len=25, seed=0, as_torch=True):
    """pure-throughput variant: every sample exactly L real tokens, [BOS] at positions 1, 1+sent_len, ...;
    boundaries placed every ~5 sentences."""
    rng = np.random.default_rng(seed)
    bos = vocab - 1
    chunk = []
    for _ in range(B):
        sents, ends = [], []
        remaining = L - 1
        while remaining > 0:
            n = min(sent_len, remaining)
            sents.append(np.concatenate(([bos], rng.integers(1000, vocab - 1, n - 1))).astype(np.int64))
            ends.append(1 if rng.random() < 0.2 else 0)
            remaining -= n
        da_s, da_e, da_t = _shuffle_in_topics(sents, ends, rng)
        a = _features(sents, ends, [IGN] * len(sents), L)
        d = _features(da_s, da_e, da_t, L)
        a["sent_pair_orders"] = d["sent_pair_orders"]
        chunk.append((a, d))
    b = {c: np.array([[a[c], d[c]] for a, d in chunk], dtype=np.int64) for c in COLUMNS}
    if as_torch:
        import torch
        b = {k: torch.from_numpy(v) for k, v in b.items()}
    return b
