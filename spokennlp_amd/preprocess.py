"""Host-side feature builder of the topic-segmentation path: restatement of the preprocessing closures of the reference
driver (emnlp2023-topic_segmentation/src/ts_sentence_seq_labeling.py:336-934) on PRE-TOKENISED documents.

Tokenisation itself stays HuggingFace's (third party, unchanged): the caller passes, per document, the token ids of every
sentence INCLUDING the leading `[BOS]` special token (what `tokenizer(sentences, is_split_into_words=True,
add_special_tokens=False)` yields for `bos_token + sentence`, :723-739).  Everything after that -- token labels,
data augmentation (topic shuffle/replace, in-topic sentence shuffle with the TSSP labels), the sliding window over
sentences, padding and the auxiliary index tensors -- is integer work restated here and pinned BIT-EXACTLY against
golden vectors produced by the reference closures (tools/gen_golden_preprocess.py, tests/test_preprocess_golden.py).
The functions draw from Python's process-global `random` in exactly the reference's order, so the same seed gives the
same augmentation.

Label ids follow the driver's label list ["B-EOP", "O"] -> {"B-EOP": 0, "O": 1} (:137-187,312-330); unknown -> -100.
"""
import random

B_EOP, O_LABEL = 0, 1
TSSP_ABLATIONS = ("none", "wo_intra_topic", "wo_inter_topic", "sso", "sso_and_intra_topic")


def get_extract_eop_segment_ids(sample_input_ids, sample_token_seq_labels, special_ids):
    """:336-349 -- running index 1..k at labelled [BOS] positions, 0 elsewhere."""
    out, eop_id = [0], 1
    for i in range(1, len(sample_input_ids)):
        if sample_input_ids[i] in special_ids and sample_token_seq_labels[i] != -100:
            out.append(eop_id); eop_id += 1
        else:
            out.append(0)
    return out


def get_sample_sent_token_mask(sample_input_ids, sample_token_seq_labels, special_ids):
    """:351-364 -- at every [BOS]: 0 if its label is 0 (end of topic) else 1; -100 elsewhere."""
    out = [-100]
    for i in range(1, len(sample_input_ids)):
        if sample_input_ids[i] in special_ids:
            out.append(0 if sample_token_seq_labels[i] == 0 else 1)
        else:
            out.append(-100)
    return out


def sent_index_to_token_span(example_input_ids, n_sentences, special_ids):
    """:590-603 -- sentence index -> (first token, last token) inclusive."""
    starts = [i for i, t in enumerate(example_input_ids) if t in special_ids]
    ends = [starts[j] - 1 for j in range(1, len(starts))] + [len(example_input_ids) - 1]
    assert len(starts) == n_sentences
    return {i: (starts[i], ends[i]) for i in range(n_sentences)}


def _topic_bounds(sent_labels):
    ends = [i for i, v in enumerate(sent_labels) if v == B_EOP]
    starts = [0] + [ends[i] + 1 for i in range(len(ends) - 1)]
    return starts, ends


def shuffle_and_replace_doc_topics(all_input_ids, num_examples, example_index, sent_labels, spans, input_ids, topic_starts, topic_ends,
                                   all_topics, all_labels, all_spans):
    """:366-459 -- shuffle the topics of a document; with p=0.5 (and >1 document in the batch) replace each topic with p=0.5
    by a random topic of another document.  Returns (ids, sentence label ids, provenance, replaced_flag, topic_orders)."""
    da_ids, da_labels, prov = [], [], []
    replaced = False
    topic_indices = list(range(len(topic_starts)))
    random.shuffle(topic_indices)
    topic_orders = list(topic_indices)
    p1 = random.random()

    def emit(ex, s_id, tag):
        lo, hi = (all_spans[ex] if ex is not None else spans)[s_id]
        src = all_input_ids[ex] if ex is not None else input_ids
        da_ids.extend(src[lo:hi + 1])
        da_labels.append((all_labels[ex] if ex is not None else sent_labels)[s_id])
        prov.append(tag)

    if p1 > 0.5 and num_examples > 1:
        for i, topic_index in enumerate(topic_indices):
            p2 = random.random()
            if p2 > 0.5:
                replaced = True
                topic_orders[i] = -1
                choices = list(range(num_examples))
                choices.remove(example_index)
                other = random.choice(choices)
                n_topics_other = len(all_topics[other])
                t_other = random.choice(list(range(n_topics_other)))
                s0, s1 = all_topics[other][t_other]
                for s_id in range(s0, s1 + 1):
                    emit(other, s_id, (other, t_other, s_id))
            else:
                for s_id in range(topic_starts[topic_index], topic_ends[topic_index] + 1):
                    emit(None, s_id, (example_index, topic_index, s_id))
    else:
        for topic_index in topic_indices:
            for s_id in range(topic_starts[topic_index], topic_ends[topic_index] + 1):
                emit(None, s_id, (example_index, topic_index, s_id))
    return da_ids, da_labels, prov, replaced, topic_orders


def shuffle_topic_sents(input_ids, sent_labels, spans, topic_starts, topic_ends, tssp_ablation="none", topic_orders=None):
    """:461-588 -- shuffle the sentences inside each topic (its last sentence stays last) and emit, per token, the TSSP label
    at each [BOS] (-100 elsewhere).  Returns (ids, new sentence label ids, token-level pair-order labels, sentence order)."""
    da_ids, da_labels, pair, order = [], [], [], []
    for i in range(len(topic_starts)):
        start, end = topic_starts[i], topic_ends[i]
        idx = list(range(start, end))
        random.shuffle(idx)
        idx.append(end)
        for j, s in enumerate(idx):
            lo, hi = spans[s]
            da_ids.extend(input_ids[lo:hi + 1])
            order.append(s)
            if tssp_ablation == "none":
                lab = 2 if j == 0 else (0 if idx[j - 1] == s - 1 else 1)
            elif tssp_ablation == "wo_intra_topic":
                lab = 1 if j == 0 else 0
            elif tssp_ablation in ("wo_inter_topic", "sso"):
                other = 1 if tssp_ablation == "wo_inter_topic" else 2
                if j == 0:
                    if i == 0:
                        lab = other
                    elif topic_orders[i - 1] == -1 or topic_orders[i - 1] + 1 != topic_orders[i]:
                        lab = other
                    else:
                        lab = 0 if s == 0 else other
                else:
                    a_, b_ = idx[j - 1], s
                    if a_ == b_ - 1:
                        lab = 0
                    elif tssp_ablation == "sso" and a_ == b_ + 1:
                        lab = 1
                    else:
                        lab = other
            elif tssp_ablation == "sso_and_intra_topic":
                if j == 0:
                    lab = 2
                else:
                    a_, b_ = idx[j - 1], s
                    lab = 0 if a_ == b_ - 1 else (1 if a_ == b_ + 1 else 2)
            else:
                raise ValueError("not recognized tssp_ablation %s" % tssp_ablation)
            pair.append(lab)
            pair.extend([-100] * (hi - lo))
        da_labels += [O_LABEL] * (len(idx) - 1) + [B_EOP]
    return da_ids, da_labels, pair, order


def token_labels(input_ids, sent_labels, special_ids):
    out, s = [], -1
    for t in input_ids:
        if t in special_ids:
            s += 1
            out.append(sent_labels[s])
        else:
            out.append(-100)
    return out


def prepare_augmented_data(all_input_ids, all_sent_labels, special_ids, tssp_ablation="none"):
    """:605-716"""
    n = len(all_input_ids)
    all_spans = [sent_index_to_token_span(all_input_ids[e], len(all_sent_labels[e]), special_ids) for e in range(n)]
    all_topics = []
    for e in range(n):
        st, en = _topic_bounds(all_sent_labels[e])
        all_topics.append({i: (a, b) for i, (a, b) in enumerate(zip(st, en))})
    out_ids, out_labels, out_pair, out_replaced, out_tok_labels = [], [], [], [], []
    for e in range(n):
        st, en = _topic_bounds(all_sent_labels[e])
        ids1, lab1, _prov, replaced, topic_orders = shuffle_and_replace_doc_topics(
            all_input_ids, n, e, all_sent_labels[e], all_spans[e], all_input_ids[e], st, en, all_topics, all_sent_labels, all_spans)
        spans1 = sent_index_to_token_span(ids1, len(lab1), special_ids)
        st1, en1 = _topic_bounds(lab1)
        ids2, lab2, pair, _order = shuffle_topic_sents(ids1, lab1, spans1, st1, en1, tssp_ablation, topic_orders)
        out_ids.append(ids2); out_labels.append(lab2); out_pair.append(pair); out_replaced.append(replaced)
    for e in range(n):
        out_tok_labels.append(token_labels(out_ids[e], out_labels[e], special_ids))
    return out_ids, out_labels, out_tok_labels, out_pair, out_replaced


def prepare_features(docs_sentence_ids, docs_labels, example_ids, max_seq_length, bos_id, cls_id, pad_id, tssp_ablation="none",
                     special_ids=None):
    """:719-934 `prepare_features_with_dynamic_num_sentence` on token ids.
    docs_sentence_ids: per document, list of sentences, each a list of token ids starting with bos_id.
    docs_labels: per document, list of sentence label ids (0 = "B-EOP", 1 = "O", -100 unknown).
    Returns the reference's output columns (each row = [anchor, augmented])."""
    special_ids = set(special_ids) if special_ids is not None else {bos_id}
    n = len(docs_sentence_ids)
    all_ids = [[t for s in doc for t in s] for doc in docs_sentence_ids]
    all_tok_labels, all_bos_index = [], []
    for e in range(n):
        labs, idx, s = [], {}, -1
        for ti, t in enumerate(all_ids[e]):
            if t in special_ids:
                s += 1
                labs.append(docs_labels[e][s]); idx[s] = ti
            else:
                labs.append(-100)
        all_tok_labels.append(labs); all_bos_index.append(idx)
    da_ids, da_sent_labels, da_tok_labels, da_pair, _replaced = prepare_augmented_data(all_ids, docs_labels, special_ids, tssp_ablation)

    cols = {k: [] for k in ("example_id", "labels", "input_ids", "token_type_ids", "attention_mask", "sent_level_labels",
                            "extract_eop_segment_ids", "eop_index_for_aggregate_batch_eop_features", "sent_pair_orders",
                            "sent_token_mask", "sentence_range")}
    L = max_seq_length
    for e in range(n):
        ids, tl = all_ids[e], all_tok_labels[e]
        dids, dtl = da_ids[e], da_tok_labels[e]
        total = len(ids)
        bos_at = all_bos_index[e]
        acc = [i - 1 for i in range(1, len(ids)) if ids[i] == bos_id] + [len(ids) - 1]
        tok_left, sent_left, si = 0, 0, 0
        while si < len(acc):
            tok_right, sent_right = acc[si] + 1, si + 1
            if tok_right - tok_left >= L - 1 or tok_right == total:
                s_ids = ([cls_id] + ids[tok_left:tok_right])[:L]
                d_ids = ([cls_id] + dids[tok_left:tok_right])[:L]
                s_lab = ([-100] + tl[tok_left:tok_right])[:L]
                d_lab = ([-100] + dtl[tok_left:tok_right])[:L]
                s_tt = [0] * len(s_ids); s_am = [1] * len(s_ids)
                d_tt = [0] * len(d_ids); d_am = [1] * len(d_ids)
                pair = ([-100] + da_pair[e][tok_left:tok_right])[:L]
                first_sent = sent_left
                if sent_right - 1 == sent_left:
                    s_lab[bos_at[sent_left] - tok_left + 1] = -100
                    tok_left = tok_right
                else:
                    s_lab[bos_at[si] - tok_left + 1] = -100
                    tok_left = acc[si - 1] + 1
                if sent_right - 1 == sent_left or tok_right == total:
                    sent_left = sent_right
                    si += 1
                else:
                    sent_left = sent_right - 1
                while len(s_ids) < L:
                    s_ids.append(pad_id); s_lab.append(-100); s_tt.append(0); s_am.append(0)
                while len(d_ids) < L:
                    d_ids.append(pad_id); d_lab.append(-100); d_tt.append(0); d_am.append(0); pair.append(-100)
                a_stm = get_sample_sent_token_mask(s_ids, s_lab, special_ids)
                d_stm = get_sample_sent_token_mask(d_ids, d_lab, special_ids)
                assert sum(1 for v in pair if v != -100) == sum(1 for v in d_stm if v != -100)

                def sll(x_ids, x_lab):
                    out = [-100] + [x_lab[i] for i in range(1, len(x_ids)) if x_ids[i] in special_ids]
                    return out + [-100] * (L - len(out)) if len(out) != L else out

                def eidx(x_lab):
                    k = sum(1 for v in x_lab if v != -100)
                    return list(range(k + 1)) + [0] * (L - k - 1)

                cols["example_id"].append([example_ids[e], example_ids[e]])
                cols["sentence_range"].append([first_sent, sent_right])
                cols["input_ids"].append([s_ids, d_ids])
                cols["labels"].append([s_lab, d_lab])
                cols["token_type_ids"].append([s_tt, d_tt])
                cols["attention_mask"].append([s_am, d_am])
                cols["sent_token_mask"].append([a_stm, d_stm])
                cols["sent_pair_orders"].append([pair, pair])
                cols["sent_level_labels"].append([sll(s_ids, s_lab), sll(d_ids, d_lab)])
                cols["extract_eop_segment_ids"].append([get_extract_eop_segment_ids(s_ids, s_lab, special_ids),
                                                        get_extract_eop_segment_ids(d_ids, d_lab, special_ids)])
                cols["eop_index_for_aggregate_batch_eop_features"].append([eidx(s_lab), eidx(d_lab)])
            else:
                si += 1
    return cols


# ---------------------------------------------------------------------------------------------------- decoding / writer
LABEL_LIST = ("B-EOP", "O")


def decode_anchor_predictions(logits, labels, label_list=LABEL_LIST):
    """ts_sentence_seq_labeling.py:1138-1152 (ts_score_predictor == "lt"): argmax over the classes of the ANCHOR logits at
    positions whose label != -100.  logits: (N, 2, L, C) array-like, labels: (N, 2, L).  Returns per-sample
    (predicted label strings, predicted class ids, true label strings, true ids, kept logits)."""
    import numpy as np
    logits = np.asarray(logits); labels = np.asarray(labels)
    anchor_logits, anchor_labels = logits[:, 0], labels[:, 0]
    pred = np.argmax(anchor_logits, axis=2)
    out = []
    for p_row, l_row, lg_row in zip(pred, anchor_labels, anchor_logits):
        keep = l_row != -100
        ids = p_row[keep].tolist(); true = l_row[keep].tolist()
        out.append(dict(predictions=[label_list[i] for i in ids], pred_ids=ids, labels=[label_list[i] for i in true],
                        int_labels=[int(v) for v in true], predict_logits=[v.tolist() for v in lg_row[keep]]))
    return out


def boundary_indices(pred_ids):
    """positions (within the labelled sentences of a sample) predicted as topic boundary (class 0 = 'B-EOP')."""
    return [i for i, v in enumerate(pred_ids) if v == 0]


def merge_windows_to_documents(decoded, example_ids, num_examples, eop_pair_cos_sim=None, sentences=None):
    """:1173-1191 -- windows of one document are concatenated in order by example_id into the per-document record of the
    prediction file (`predict_<data>_max_seq<L>_ts_score_lt.txt`, one JSON object per line)."""
    out = [{"sentences": [], "labels": [], "int_labels": [], "predictions": [], "predict_logits": []} for _ in range(num_examples)]
    for i, (rec, ex) in enumerate(zip(decoded, example_ids)):
        if sentences is not None:
            out[ex]["sentences"].extend(sentences[i])
        out[ex]["labels"].extend(rec["labels"]); out[ex]["predictions"].extend(rec["predictions"])
        out[ex]["predict_logits"].extend(rec["predict_logits"]); out[ex]["int_labels"].extend(rec["int_labels"])
    if eop_pair_cos_sim is not None:
        for o in out:
            o["eop_pair_cos_sim"] = []
        for ex, cs in zip(example_ids, eop_pair_cos_sim):
            out[ex]["eop_pair_cos_sim"].extend([float(v) for v in cs if v != -100])
    return out


def write_prediction_file(path, documents):
    import json
    with open(path, "w") as f:
        f.writelines([json.dumps(d, ensure_ascii=False) + "\n" for d in documents])


# ---------------------------------------------------------------------------------------------------- PoNet (a12)
def ponet_prepare_features(docs_sentence_ids, docs_labels, example_ids, max_seq_length, eos_id, cls_id, pad_id,
                           use_paragraph_segment=False):
    """alimeeting4mug/src/topic_segment/ponet_topic_segmentation.py:527-691 `prepare_input_features` on token ids.
    docs_sentence_ids: per document, list of sentences, each a list of token ids ENDING with eos_id (the driver appends
    "[EOS]" to every sentence, :537-545).  docs_labels: per document the label id of every sentence (-100 = unlabelled).
    `segment_ids` (:564-596): 1-based sentence index per token, or -- with use_paragraph_segment -- a paragraph index that
    advances after every LABELLED [EOS]; [CLS] gets 0 (:638), padding gets num_sentences + 1 (:668)."""
    out = {k: [] for k in ("input_ids", "token_type_ids", "attention_mask", "segment_ids", "example_id", "labels", "sentence_range")}
    L = max_seq_length
    for e in range(len(docs_sentence_ids)):
        ids = [t for s in docs_sentence_ids[e] for t in s]
        labs = docs_labels[e]
        tok_lab, seg, para = [], [], []
        cur, cur_para = 1, 1
        for ti, t in enumerate(ids):
            if t == eos_id:
                tok_lab.append(labs[cur - 1]); seg.append(cur); cur += 1
            else:
                tok_lab.append(-100); seg.append(cur)
            para.append(cur_para)
            if tok_lab[ti] != -100:
                cur_para += 1
        seg_ids = para if use_paragraph_segment else seg
        total, nsent = len(ids), len(labs)
        acc = [i for i, x in enumerate(ids) if x == eos_id]
        left, sent_left, si = 0, 0, 0
        while si < len(acc):
            right, sent_right = acc[si] + 1, si + 1
            if right - left >= L - 1 or right == total:
                s_ids = ([cls_id] + ids[left:right])[:L]
                s_seg = ([0] + seg_ids[left:right])[:L]
                s_lab = ([-100] + tok_lab[left:right])[:L]
                s_tt = [0] * len(s_ids); s_am = [1] * len(s_ids)
                if sent_right - 1 == sent_left:
                    left = right
                    s_ids[-1] = eos_id
                    s_lab[-1] = -100
                else:
                    left = acc[si - 1] + 1
                    if s_lab[-1] != -100:
                        s_lab[-1] = -100
                if sent_right - 1 == sent_left or right == total:
                    rng = [sent_left, sent_right]
                    sent_left = sent_right
                    si += 1
                else:
                    rng = [sent_left, sent_right - 1]
                    sent_left = sent_right - 1
                while len(s_ids) < L:
                    s_ids.append(pad_id); s_tt.append(0); s_am.append(0); s_seg.append(nsent + 1); s_lab.append(-100)
                out["input_ids"].append(s_ids); out["token_type_ids"].append(s_tt); out["attention_mask"].append(s_am)
                out["segment_ids"].append(s_seg); out["labels"].append(s_lab); out["example_id"].append(example_ids[e])
                out["sentence_range"].append(rng)
            else:
                si += 1
    return out


# ---------------------------------------------------------------------------------------------------------------- PoNet extractive summarisation
def es_collect_predictions(pred_label_ids, labels, window_num_sentences, example_ids, num_examples, o_id=1):
    """alimeeting4mug/src/extractive_summarization/ponet_extractive_summarization.py:853-905 (`compute_metrics`, integer part):
    the argmax label id at every labelled [EOS] of every window, glued back per document in window order.  A window whose last
    sentence lost its label to the window edge (the feature builder sets the label of a cut-off final [EOS] to -100, :727-733)
    has one prediction fewer than sentences: the reference appends "O" to both prediction and label (:897-899).
    pred_label_ids / labels: per window, sequences of length L (labels -100 = not a sentence end);
    window_num_sentences: per window, the number of sentences it covers (`sentences` column of the feature builder).
    Returns (per-document predicted ids, per-document label ids)."""
    preds = [[] for _ in range(num_examples)]
    golds = [[] for _ in range(num_examples)]
    for p_row, l_row, nsent, ex in zip(pred_label_ids, labels, window_num_sentences, example_ids):
        p = [int(p) for p, l in zip(p_row, l_row) if l != -100]
        g = [int(l) for l in l_row if l != -100]
        if len(g) < nsent:
            p.append(o_id); g.append(o_id)
        if len(p) != nsent:
            raise ValueError(f"window of document {ex}: {len(p)} labelled sentence ends for {nsent} sentences")
        preds[ex].extend(p); golds[ex].extend(g)
    return preds, golds


def es_selected_sentences(doc_label_ids, key_id=0):
    """indices of the sentences labelled as summary sentences ("B-EOP" = id 0 in the reference's label list, :907-912)"""
    return [i for i, v in enumerate(doc_label_ids) if v == key_id]
