"""Example-level evaluator of the topic-segmentation path: restatement of
emnlp2023-topic_segmentation/src/metrics/seqeval.py:173-373 (`compute_window_metric`, `compute_metric_example_level`).
CPU only, pure Python/numpy -- it consumes the decoded predictions, not GPU tensors.

Third-party pieces the reference calls and that are NOT in its tree (parity UNPINNED for them -- no reference test or
golden vector exists, and the packages are absent here):
  * segeval==2.0.11 `pk` / `window_diff` (seqeval.py:24-25,196-197), restated from their published definitions
    (Beeferman et al. 1999; Pevzner & Hearst 2002) with segeval's defaults: masses in, window size k =
    max(2, round_half_even(mean reference segment mass / 2)), N - k probes, no Lamprier fix;
  * sklearn precision/recall/f1 on the flattened 0/1 lists (:231-234) -- binary, positive class 1, 0 on zero division;
  * seqeval==1.2.2 chunk P/R/F1 (:141-171): `seqeval_scores` below restates the package's default-mode `classification_report`
    (conlleval chunk extraction + per-type / micro precision, recall, F1 + tag accuracy) and is pinned on the worked example the
    reference file holds in its docstring (seqeval.py:96-105: overall_f1 0.5, PER f1 1.0) and on brute-force chunk counting; with the
    label set {"B-EOP", "O"} every "B-EOP" tag is a one-token chunk, so chunk-level scores equal the tag-level scores of "B-EOP".
Also here: the `compute_metrics` closure the Trainer calls during fine-tuning (ts_sentence_seq_labeling.py:1018-1074 ->
`make_compute_metrics`), the WikiSection sentence-level re-scoring of paragraph-level predictions (postprocess_predictions.py:7-75) and
the `*_str_metric.txt` writer (utils.py:8-48).
"""
import json
import os
from decimal import ROUND_HALF_EVEN, Decimal

import numpy as np


def mass_from_start_label_sequence(labels):
    """seqeval.py:178-191: [1,1,0,0,1,1] -> [1,1,3,1] (1 = last sentence of its segment)."""
    mass, cur = [], 0
    for v in labels:
        cur += 1
        if v == 1:
            mass.append(cur); cur = 0
    if cur > 0:
        mass.append(cur)
    return mass


def _positions(masses):
    return [i for i, m in enumerate(masses) for _ in range(m)]


def window_size(reference_masses):
    avg = Decimal(sum(reference_masses)) / Decimal(len(reference_masses))
    k = int((avg / 2).quantize(Decimal(1), rounding=ROUND_HALF_EVEN))
    return k if k > 1 else 2


def pk(hypothesis_masses, reference_masses, k=None):
    """P_k: fraction of probes (i, i+k) on which "same segment?" differs between hypothesis and reference."""
    ref, hyp = _positions(reference_masses), _positions(hypothesis_masses)
    if len(ref) != len(hyp):
        raise ValueError("segmentations cover a different number of units")
    k = k or window_size(reference_masses)
    n = len(ref) - k
    if n <= 0:
        return 0.0
    diff = sum(1 for i in range(n) if (ref[i] == ref[i + k]) != (hyp[i] == hyp[i + k]))
    return diff / n


def window_diff(hypothesis_masses, reference_masses, k=None):
    """WindowDiff: fraction of windows of k potential boundaries holding a different NUMBER of boundaries."""
    ref, hyp = _positions(reference_masses), _positions(hypothesis_masses)
    if len(ref) != len(hyp):
        raise ValueError("segmentations cover a different number of units")
    k = k or window_size(reference_masses)
    n = len(ref) - k
    if n <= 0:
        return 0.0
    diff = sum(1 for i in range(n) if (ref[i + k] - ref[i]) != (hyp[i + k] - hyp[i]))
    return diff / n


def binary_prf(references, predictions):
    tp = sum(1 for r, p in zip(references, predictions) if r == 1 and p == 1)
    npred, ntrue = sum(predictions), sum(references)
    p = tp / npred if npred else 0.0
    r = tp / ntrue if ntrue else 0.0
    f = 2 * p * r / (p + r) if (p + r) else 0.0
    return p, r, f


def compute_window_metric(predictions, references, prefix="", strict=False):
    """seqeval.py:173-237.  predictions / references: per example a 0/1 list, 1 = end sentence of a topic.
    strict: a malformed example (mass mismatch, empty list) raises instead of being dropped from the means -- the behaviour of the
    alimeeting4mug twin (challenge_evaluate.py:105-111: `print(i, e); raise RuntimeError`); the emnlp2023 function swallows it (:214-215)."""
    one_pk, one_wd = [], []
    for i, (y_pred, y_true) in enumerate(zip(predictions, references)):
        try:
            pm, tm = mass_from_start_label_sequence(y_pred), mass_from_start_label_sequence(y_true)
            assert sum(pm) == sum(tm)
            one_pk.append(1 - pk(pm, tm)); one_wd.append(1 - window_diff(pm, tm))
        except Exception as e:  # the emnlp2023 reference swallows per-example failures the same way (:214-215)
            if strict:
                raise RuntimeError(f"window metric: example {i} is malformed ({type(e).__name__}: {e})") from e
    t_pk = round(float(np.array(one_pk).mean()), 4)
    t_wd = round(float(np.array(one_wd).mean()), 4)
    flat_p, flat_r = sum(predictions, []), sum(references, [])
    p, r, f1 = binary_prf(flat_r, flat_p)
    return {prefix + "1-pk": t_pk, prefix + "1-wd": t_wd, prefix + "precision": round(p, 4), prefix + "recall": round(r, 4),
            prefix + "f1": round(f1, 4), prefix + "pk": 1 - t_pk, prefix + "wd": 1 - t_wd}


def _softmax0(logits):
    x = np.asarray(logits, dtype=np.float64)
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(axis=-1, keepdims=True))[:, 0]


def compute_metric_example_level(predictions_logits, labels, label_list=("B-EOP", "O"), threshold=None, topk=None,
                                 topk_with_threshold=False, f1_at_k=None, ts_score_predictor="lt", reverse_logits=False):
    """seqeval.py:248-373.  predictions_logits: per document a list of per-sentence logits (2 classes, "lt") or sigmoid
    cosine scores ("cos"); labels: per document the int labels (0 = "B-EOP" = topic boundary, 1 = "O")."""
    if ts_score_predictor == "lt":
        preds = [np.argmax(np.array(lg), axis=-1).tolist() for lg in predictions_logits]
        seg_scores = [_softmax0(lg).tolist() for lg in predictions_logits]
    else:
        preds = [(np.array(lg) > 0.5).astype(np.int32).tolist() for lg in predictions_logits]
        seg_scores = [[1 - v for v in lg] for lg in predictions_logits]
    if reverse_logits:
        preds = [[1 - v for v in p] for p in preds]
    keep = [[l != -100 for l in lab] for lab in labels]
    tp_ = [[p for p, k in zip(pr, kp) if k] for pr, kp in zip(preds, keep)]
    tl_ = [[l for l, k in zip(lab, kp) if k] for lab, kp in zip(labels, keep)]
    # chunk-level == tag-level for one-token "B-EOP" chunks (see module docstring)
    flat_p = [1 if p == 0 else 0 for row in tp_ for p in row]
    flat_l = [1 if l == 0 else 0 for row in tl_ for l in row]
    p, r, f1 = binary_prf(flat_l, flat_p)
    total = sum(len(row) for row in tl_)
    res = {"precision": p, "recall": r, "f1": f1,
           "accuracy": sum(1 for a, b in zip(sum(tp_, []), sum(tl_, [])) if a == b) / max(total, 1)}
    true_bin = [[int(not l) for l in row] for row in tl_]
    scores = [[s for s, k in zip(sc, kp) if k] for sc, kp in zip(seg_scores, keep)]
    if threshold is not None:
        cmp = (lambda v: v >= threshold) if ts_score_predictor == "lt" else (lambda v: v > threshold)
        pb = [[1 if cmp(v) else 0 for v in row] for row in scores]
        res.update(compute_window_metric(pb, true_bin, prefix="threshold_%s_example_level_" % str(threshold)))
    if topk is not None:
        prefix = "topk_%s_example_level_" % str(topk)
        ranked = [sorted([(v, i) for i, v in enumerate(row)], reverse=True) for row in scores]
        pb = []
        for row, rk in zip(scores, ranked):
            out = [0] * len(row)
            for i in range(min(len(rk), topk)):
                out[rk[i][1]] = 1
            pb.append(out)
        r_ = compute_window_metric(pb, true_bin, prefix=prefix)
        kth = [rk[min(len(rk), topk)][0] for rk in ranked]          # raises IndexError like the reference when len <= topk
        r_[prefix + "kth_scores_avg"] = round(sum(kth) / len(kth), 3)
        res.update(r_)
        if topk_with_threshold:
            assert threshold is not None
            pb = []
            for row, rk in zip(scores, ranked):
                out = [0] * len(row)
                for i in range(min(len(rk), topk)):
                    if rk[i][0] >= threshold:
                        out[rk[i][1]] = 1
                pb.append(out)
            res.update(compute_window_metric(pb, true_bin, prefix="topk_%s_with_threshold_%s_example_level_" % (str(topk), str(threshold))))
    if f1_at_k:
        pb = [[1 if v >= threshold else 0 for v in row] for row in scores]
        soft = []
        for pred, lab in zip(pb, true_bin):
            for i, pv in enumerate(pred):
                if pv == 0 or (pv == 1 and lab[i] == 1):
                    continue
                for j in range(max(0, i - f1_at_k), min(len(pred) - 1, i + f1_at_k) + 1):
                    if lab[j] == 1:
                        pred[i] = 0; pred[j] = 1
                        break
            soft.append(pred)
        res.update(compute_window_metric(soft, true_bin, prefix="f1@%d_example_level_" % f1_at_k))
    return res


# ---------------------------------------------------------------------------------------------------- seqeval chunk-level scores
def _tag_and_type(label, suffix=False):
    """seqeval 1.2.2 `get_entities`: "B-EOP" -> ("B", "EOP"); a bare "O" has the type "_" """
    if suffix:
        return label[-1], (label[:-1].rsplit("-", maxsplit=1)[0] or "_")
    return label[0], (label[1:].split("-", maxsplit=1)[-1] or "_")


def _end_of_chunk(prev_tag, tag, prev_type, type_):
    if prev_tag in ("E", "S"):
        return True
    if prev_tag in ("B", "I") and tag in ("B", "S", "O"):
        return True
    return prev_tag not in ("O", ".") and prev_type != type_


def _start_of_chunk(prev_tag, tag, prev_type, type_):
    if tag in ("B", "S"):
        return True
    if prev_tag in ("E", "S", "O") and tag in ("E", "I"):
        return True
    return tag not in ("O", ".") and prev_type != type_


def get_entities(seq, suffix=False):
    """the conlleval chunk extraction seqeval's default mode runs (what `classification_report(scheme=None)` of seqeval.py:141-150 calls):
    a list of tag lists is joined with an "O" between the sequences; returns [(type, first, last)] in positions of the joined list"""
    if any(isinstance(s, (list, tuple)) for s in seq):
        seq = [item for sub in seq for item in list(sub) + ["O"]]
    prev_tag, prev_type, begin, chunks = "O", "", 0, []
    for i, label in enumerate(list(seq) + ["O"]):
        tag, type_ = _tag_and_type(label, suffix)
        if _end_of_chunk(prev_tag, tag, prev_type, type_):
            chunks.append((prev_type, begin, i - 1))
        if _start_of_chunk(prev_tag, tag, prev_type, type_):
            begin = i
        prev_tag, prev_type = tag, type_
    return chunks


def _prf(tp, npred, ntrue):
    p = tp / npred if npred else 0.0                    # zero_division="warn" acts as 0 (seqeval.py:76-77)
    r = tp / ntrue if ntrue else 0.0
    return p, r, (2 * p * r / (p + r) if (p + r) else 0.0)


def seqeval_scores(predictions, references, suffix=False):
    """`Seqeval._compute` (seqeval.py:125-170) with its defaults (scheme=None, mode=None, no sample weights): per chunk type
    {"precision", "recall", "f1", "number"} and "overall_precision" / "overall_recall" / "overall_f1" (micro average over chunks) /
    "overall_accuracy" (tag accuracy).  A chunk counts when type, first and last position all agree."""
    if len(predictions) != len(references) or any(len(p) != len(r) for p, r in zip(predictions, references)):
        raise ValueError("Found input variables with inconsistent numbers of samples")          # seqeval's check_consistent_length
    true_by, pred_by = {}, {}
    for t, b, e in get_entities(references, suffix):
        true_by.setdefault(t, set()).add((b, e))
    for t, b, e in get_entities(predictions, suffix):
        pred_by.setdefault(t, set()).add((b, e))
    scores, tp_all, np_all, nt_all = {}, 0, 0, 0
    for t in sorted(set(true_by) | set(pred_by)):
        tr, pr = true_by.get(t, set()), pred_by.get(t, set())
        tp = len(tr & pr)
        p, r, f = _prf(tp, len(pr), len(tr))
        scores[t] = {"precision": p, "recall": r, "f1": f, "number": len(tr)}
        tp_all += tp; np_all += len(pr); nt_all += len(tr)
    p, r, f = _prf(tp_all, np_all, nt_all)
    scores["overall_precision"], scores["overall_recall"], scores["overall_f1"] = p, r, f
    flat_t = [x for row in references for x in row]
    flat_p = [x for row in predictions for x in row]
    scores["overall_accuracy"] = (sum(1 for a, b in zip(flat_t, flat_p) if a == b) / len(flat_t)) if flat_t else 0.0
    return scores


def make_compute_metrics(label_list=("B-EOP", "O"), ts_score_predictor="lt", return_entity_level_metrics=True):
    """the `compute_metrics` closure of ts_sentence_seq_labeling.py:1018-1074 for `transformers.Trainer(compute_metrics=...)`:
    p = ((logits (N,2,L,2), cos_sim (N,k)), (labels (N,2,L), sent_level_labels)); anchor half = index 0, DA half = index 1.  Returns the
    anchor's chunk scores and the DA half's under the `da_` prefix -- flattened (`EOP_f1`, `overall_f1`, `da_overall_f1`, ...: what
    `--metric_for_best_model overall_f1` of run_finetune.sh:80-82 selects on) when return_entity_level_metrics (arguments.py:219-222,
    default True), else the four overall numbers."""
    label_list = list(label_list)

    def compute_metrics(p):
        all_logits, all_labels = p
        logits, _cos = all_logits
        labels = all_labels[0] if isinstance(all_labels, (tuple, list)) else all_labels
        logits, labels = np.asarray(logits), np.asarray(labels)
        a_lab, d_lab = labels[:, 0], labels[:, 1]
        true_a = [[label_list[l] for l in row if l != -100] for row in a_lab]
        true_d = [[label_list[l] for l in row if l != -100] for row in d_lab]
        if ts_score_predictor == "lt":
            pa, pd = np.argmax(logits[:, 0], axis=2), np.argmax(logits[:, 1], axis=2)      # 0 -> "B-EOP" = boundary
            pred_a = [[label_list[q] for q, l in zip(pr, lr) if l != -100] for pr, lr in zip(pa, a_lab)]
            pred_d = [[label_list[q] for q, l in zip(pr, lr) if l != -100] for pr, lr in zip(pd, d_lab)]
        elif ts_score_predictor == "cos":
            # (:1043-1047) the logits ARE per-EOP scores; the reference builds no DA predictions on this branch and then reads
            # `da_true_predictions` (NameError at :1051) -- here the DA half is scored the same way as the anchor
            pa, pd = (logits[:, 0] > 0.5).astype(np.int32), (logits[:, 1] > 0.5).astype(np.int32)
            pred_a = [[label_list[q] for q in pr[:len(t)]] for pr, t in zip(pa, true_a)]
            pred_d = [[label_list[q] for q in pr[:len(t)]] for pr, t in zip(pd, true_d)]
        else:
            raise ValueError("not supported ts_score_predictor %s" % ts_score_predictor)
        results = seqeval_scores(pred_a, true_a)
        results.update({"da_" + k: v for k, v in seqeval_scores(pred_d, true_d).items()})
        if return_entity_level_metrics:
            final = {}
            for k, v in results.items():
                if isinstance(v, dict):
                    for n, x in v.items():
                        final[f"{k}_{n}"] = x
                else:
                    final[k] = v
            return final
        return {"precision": results["overall_precision"], "recall": results["overall_recall"], "f1": results["overall_f1"],
                "accuracy": results["overall_accuracy"]}

    return compute_metrics


# ---------------------------------------------------------------------------------------------------- WikiSection sentence-level re-scoring
def read_total_pred_and_labels(data_file, pred_file):
    """postprocess_predictions.py:7-26: the prediction file (one json line per document, tags "B-EOP" / "O") as 0 / 1 lists (1 = boundary)
    and the data file's sentence-level labels without each document's last one"""
    para_pred, para_lab, sent_lab = [], [], []
    with open(pred_file, "r") as f:
        for line in f:
            if not line.strip():
                continue
            t = json.loads(line)
            para_lab.append([0 if v == "O" else 1 for v in t["labels"]])
            para_pred.append([0 if v == "O" else 1 for v in t["predictions"]])
    with open(data_file, "r") as f:
        for line in f:
            if not line.strip():
                continue
            sent_lab.append(json.loads(line)["labels"][:-1])
    return para_pred, para_lab, sent_lab


def sent_level_metric_from_para_level_models(total_para_level_predictions, total_para_level_labels, total_sent_level_labels):
    """postprocess_predictions.py:50-75: a paragraph-level model predicts only at paragraph ends (the sentences whose label is not -100);
    its predictions are spread over the sentence positions (non-paragraph-end sentences: label 0, prediction 0) and both granularities are
    scored with the window metric.  Returns (sentence-level result, paragraph-level result) and leaves the inputs untouched (the
    reference rewrites its label lists in place)."""
    sent_pred, sent_lab = [], []
    for para_lab, slab, para_pred in zip(total_para_level_labels, total_sent_level_labels, total_para_level_predictions):
        assert len(para_lab) == len([v for v in slab if v != -100])
        sp, sl, pid = [0] * len(slab), list(slab), 0
        for i, v in enumerate(slab):
            if v != -100:
                assert v == para_lab[pid]
                sp[i] = para_pred[pid]; pid += 1
            else:
                sl[i] = 0
        sent_pred.append(sp); sent_lab.append(sl)
    return (compute_window_metric(sent_pred, sent_lab),
            compute_window_metric([list(x) for x in total_para_level_predictions], [list(x) for x in total_para_level_labels]))


def _prfkw_line(res):
    return " / ".join("%.2f" % (v * 100) for v in (res["precision"], res["recall"], res["f1"], res["pk"], res["wd"]))


def wiki_section_sent_level_metric(data_file, pred_file, disease_cnt=718, city_cnt=3893, out=print):
    """postprocess_predictions.py:29-47 (`get_wiki_section_sent_level_metric`, what run_inference.sh:51-57 runs after predicting on
    wiki_section): en_disease = the first 718 documents, en_city = the next 3893; prints the reference's lines and returns
    {name: {"sent_level": ..., "para_level": ...}}"""
    pp, pl, sl = read_total_pred_and_labels(data_file, pred_file)
    assert len(pp) == disease_cnt + city_cnt
    parts = {"wiki_section_disease": slice(0, disease_cnt), "wiki_section_city": slice(disease_cnt, None), "wiki_section": slice(None)}
    out(" / ".join(["p", "r", "f1", "pk", "wd"]))
    res = {}
    for name, sel in parts.items():
        s, p = sent_level_metric_from_para_level_models(pp[sel], pl[sel], sl[sel])
        out("data_name:  " + name)
        out("sent_level: " + _prfkw_line(s))
        out("para_level: " + _prfkw_line(p))
        out("\n")
        res[name] = {"sent_level": s, "para_level": p}
    return res


# ---------------------------------------------------------------------------------------------------- result-file helpers (utils.py)
def abridge_model_name(model_name_or_path):
    """utils.py:8-20: the short model tag of the cache / prediction file names"""
    for key, short in (("longformer", "lf"), ("bigbird", "bb"), ("bert", "bert"), ("electra", "ele")):
        if key in model_name_or_path:
            return short
    raise ValueError("not supported model_name")


def convert_res_format(file_path, threshold, out=print):
    """utils.py:23-48: `<name>_results.json` -> `<name>_results_str_metric.txt`, the last file run_inference.sh leaves
    (ts_sentence_seq_labeling.py:1222): the thresholded example-level P / R / F / Pk / WD as percentages with two decimals.
    `threshold` is custom_args.threshold (formatted with %s, as in the key names compute_metric_example_level writes)."""
    out_path = os.path.join(os.path.dirname(file_path), os.path.basename(file_path).split(".json")[0] + "_str_metric.txt")
    with open(file_path, "r") as f:
        res = json.load(f)
    vals = [res["threshold_%s_example_level_%s" % (threshold, k)] for k in ("precision", "recall", "f1", "pk", "wd")]
    line = "threshold_%s_example_level_metric\n" % threshold + " / ".join("%.2f" % (float(v) * 100) for v in vals)
    with open(out_path, "w") as f:
        f.write("p / r / f / pk / wd\n")
        f.write(line + "\n\n")
    out("p / r / f / pk / wd\n")
    out(line + "\n\n")
    return out_path


# ---------------------------------------------------------------------------------------------------- alimeeting4mug twins
def compute_window_metric_alimeeting(predictions, references, prefix=""):
    """alimeeting4mug/metrics/topic_seg_eval/topic_seg_eval.py:170-237 (== src/utils/challenge_evaluate.py:75-134): the emnlp2023 window
    metric plus the average number of predicted / true boundaries per example; no `pk` / `wd` complements."""
    n = len(predictions)
    if n == 0:
        raise ValueError("compute_window_metric_alimeeting: no examples")
    base = compute_window_metric(predictions, references, prefix=prefix, strict=True)      # challenge_evaluate.py:105-111 re-raises
    flat_p, flat_r = sum(predictions, []), sum(references, [])
    out = {k: base[k] for k in (prefix + "1-pk", prefix + "1-wd", prefix + "precision", prefix + "recall", prefix + "f1")}
    out[prefix + "avg_pred_cnt"] = round(sum(flat_p) * 1.0 / n, 2)
    out[prefix + "avg_true_cnt"] = round(sum(flat_r) * 1.0 / n, 2)
    return out


def topic_segment_score(pos_f1, one_minus_pk, one_minus_wd):
    """challenge_evaluate.py:137-139: the AliMeeting4MUG ranking score"""
    return 0.5 * pos_f1 + 0.25 * (one_minus_pk + one_minus_wd)


def topic_segment_evaluate_samples(label_samples, pred_samples):
    """challenge_evaluate.py:165-210 on in-memory samples (the reference reads them from the ModelScope dataset / a jsonl file).
    label sample: {"meeting_key", "sentences", "paragraph_segment_ids": [{"id"}], "topic_segment_ids": [{"id"}]} (1-based sentence ids);
    pred sample: {"meeting_key", "topic_segment_ids": [{"id"}]}.  Boundaries are scored at paragraph ends only, the last one dropped."""
    assert len(label_samples) == len(pred_samples), "NUMBER ERROR."
    total_preds, total_labels, preds_split, labels_split = [], [], [], []
    for ls, ps in zip(label_samples, pred_samples):
        assert ls["meeting_key"] == ps["meeting_key"], "meeting_key error."
        nsent = len(ls["sentences"])
        para = set(x["id"] for x in ls["paragraph_segment_ids"])
        preds, labels = [0] * nsent, [0] * nsent
        for x in ls["topic_segment_ids"]:
            labels[x["id"] - 1] = 1
        for x in ps["topic_segment_ids"]:
            preds[x["id"] - 1] = 1
        preds[-1] = 1; labels[-1] = 1
        labels = [v for i, v in enumerate(labels) if (i + 1) in para]
        preds = [v for i, v in enumerate(preds) if (i + 1) in para]
        total_labels.extend(labels[:-1]); total_preds.extend(preds[:-1])
        labels_split.append(labels[:-1]); preds_split.append(preds[:-1])
    _, _, pos_f1 = binary_prf(total_labels, total_preds)
    ws = compute_window_metric_alimeeting(preds_split, labels_split, prefix="test_")
    out = {"score": topic_segment_score(pos_f1, ws["test_1-pk"], ws["test_1-wd"])}
    ws.pop("test_avg_pred_cnt"); ws.pop("test_avg_true_cnt")
    out.update(ws)
    return out


# ---------------------------------------------------------------------------------------------------- ROUGE (extractive summarisation, 8(f)-4)
# alimeeting4mug/metrics/extractive_summarization_eval/extractive_summarization_eval.py:167-180 and metrics/rouge/rouge.py:121-135 call
# `rouge.Rouge().get_scores(hyps, refs, avg=True)` of the PyPI package rouge==1.0.1 (alimeeting4mug/requirements.txt:90; not in the
# reference tree, not installed here).  What follows restates that package's published algorithm for its three default metrics.  The
# only known answer the reference holds is the worked example in metrics/rouge/rouge.py:62-93 (the package's README example), and it is
# not self-consistent: its rouge-l numbers are the package's `exclusive=True` counting (SETS of words -- the 1.0.1 constructor default),
# its rouge-1 / rouge-2 numbers are `exclusive=False` counting (multisets, clipped overlap).  Both countings are restated, each pinned
# on the numbers it reproduces to the last digit (tests/test_evaluate.py); `exclusive=True` is the default here as in the package, so
# rouge-l is pinned and rouge-1 / rouge-2 are pinned only up to that choice.  Conventions of the package, kept as they are:
#   * a text is cut into sentences at every "." and whitespace-normalised; n-grams run over the words of ALL sentences joined;
#   * f = 2 p r / (p + r + 1e-8);
#   * rouge-l is the summary-level union LCS: for every reference sentence the words of its LCS with every hypothesis sentence are
#     collected in ONE running collection (a set / a list); r = its size / reference words, p = its size / hypothesis words
#     (distinct words / all words).
def _rouge_sentences(text):
    return [" ".join(s.split()) for s in text.split(".") if len(s) > 0]


def _rouge_words(sentences):
    return [w for s in sentences for w in s.split(" ")]


def _clipped_overlap(a, b):
    """the package's list intersection: every element of a that still finds an unused equal element in b"""
    left = {}
    for e in b:
        left[e] = left.get(e, 0) + 1
    n = 0
    for e in a:
        if left.get(e, 0) > 0:
            left[e] -= 1; n += 1
    return n


def _rouge_n(hyp, ref, n, exclusive=True):
    if not hyp:
        raise ValueError("Hypothesis is empty.")
    if not ref:
        raise ValueError("Reference is empty.")
    hw, rw = _rouge_words(hyp), _rouge_words(ref)
    hg = [tuple(hw[i:i + n]) for i in range(len(hw) - n + 1)]
    rg = [tuple(rw[i:i + n]) for i in range(len(rw) - n + 1)]
    if exclusive:
        hs, rs = set(hg), set(rg)
        ov, nh, nr = len(hs & rs), len(hs), len(rs)
    else:
        ov, nh, nr = _clipped_overlap(hg, rg), len(hg), len(rg)
    p = ov / nh if nh else 0.0
    r = ov / nr if nr else 0.0
    return {"f": 2.0 * ((p * r) / (p + r + 1e-8)), "p": p, "r": r}


def _lcs_words(x, y):
    """the words of one longest common subsequence of x and y (in order), traced back from the corner of the length table: on a match
    step diagonally; else up when table[i-1][j] > table[i][j-1], left otherwise (the package's tie rule -- it decides WHICH subsequence)"""
    n, m = len(x), len(y)
    t = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        xi, ti, tp = x[i - 1], t[i], t[i - 1]
        for j in range(1, m + 1):
            ti[j] = tp[j - 1] + 1 if xi == y[j - 1] else (tp[j] if tp[j] > ti[j - 1] else ti[j - 1])
    out, i, j = [], n, m
    while i > 0 and j > 0:
        if x[i - 1] == y[j - 1]:
            out.append(x[i - 1]); i -= 1; j -= 1
        elif t[i - 1][j] > t[i][j - 1]:
            i -= 1
        else:
            j -= 1
    return out[::-1]


def _rouge_l(hyp, ref, exclusive=True):
    if not hyp or not ref:
        raise ValueError("Collections must contain at least 1 sentence.")
    rw_all, hw_all = _rouge_words(ref), _rouge_words(hyp)
    m, n = (len(set(rw_all)), len(set(hw_all))) if exclusive else (len(rw_all), len(hw_all))
    union_set, union_len = set(), 0
    for rs in ref:
        rw = rs.split(" ")
        for hs in hyp:
            lcs = _lcs_words(rw, hs.split(" "))
            union_set.update(lcs); union_len += len(lcs)
    llcs = len(union_set) if exclusive else union_len
    r, p = llcs / m, llcs / n
    return {"f": 2.0 * ((p * r) / (p + r + 1e-8)), "p": p, "r": r}


def rouge_get_scores(hyps, refs, avg=False, exclusive=True):
    """rouge.Rouge(exclusive=...).get_scores: per pair {"rouge-1" | "rouge-2" | "rouge-l": {"f", "p", "r"}}; avg=True: the mean over pairs"""
    if isinstance(hyps, str):
        hyps, refs = [hyps], [refs]
    assert len(hyps) == len(refs)
    per = []
    for h, r in zip(hyps, refs):
        hs, rs = _rouge_sentences(h), _rouge_sentences(r)
        per.append({"rouge-1": _rouge_n(hs, rs, 1, exclusive), "rouge-2": _rouge_n(hs, rs, 2, exclusive), "rouge-l": _rouge_l(hs, rs, exclusive)})
    if not avg:
        return per
    return {m: {s: sum(x[m][s] for x in per) / len(per) for s in ("r", "p", "f")} for m in ("rouge-1", "rouge-2", "rouge-l")}


def rouge_compute(predictions, references, use_avg=True):
    """`Seqeval.rouge_compute` / `Rouge._compute` of the reference (extractive_summarization_eval.py:167-180): predictions / references
    are lists of sentence lists; each is joined with blanks, scored, and flattened to {"score": rouge-1 f, "rouge-1_r": ..}"""
    scores = rouge_get_scores([" ".join(p) for p in predictions], [" ".join(r) for r in references], avg=use_avg)
    result = {"score": scores["rouge-1"]["f"]}
    for k1 in scores:
        for k2 in scores[k1]:
            result["{}_{}".format(k1, k2)] = scores[k1][k2]
    return result


def es_rouge_metrics(documents, tokenize_func=lambda s: s.split(), key_label="B-EOP"):
    """the ROUGE part of the extractive-summarisation `compute_metrics`
    (alimeeting4mug/src/extractive_summarization/ponet_extractive_summarization.py:904-957).
    documents: [{"sentences": [str], "labels": [str], "predictions": [str], "multi_labels": [[str]] (optional)}] per document, as glued
    back from the windows (preprocess.es_collect_predictions).  Selected sentences are tokenised and joined; an empty selection is the
    one-blank summary [" "] as in the reference.  Returns the single-reference scores plus, when multi_labels are given, the
    `multi-ref-average_*` (mean over a document's references, then over documents) and `multi-ref-max_*` (the reference with the best
    rouge-l f per document) entries."""
    preds, refs = [], []
    for d in documents:
        ps = [" ".join(tokenize_func(s)) for s, p in zip(d["sentences"], d["predictions"]) if p == key_label] or [" "]
        ls = [" ".join(tokenize_func(s)) for s, l in zip(d["sentences"], d["labels"]) if l == key_label] or [" "]
        preds.append(ps); refs.append(ls)
    out = rouge_compute(preds, refs)
    if documents and all("multi_labels" in d for d in documents):
        ave_all, max_all = [], []
        for d, pred in zip(documents, preds):
            multi = []
            for ref in d["multi_labels"]:
                assert len(ref) == len(d["sentences"])
                rs = [" ".join(tokenize_func(s)) for s, l in zip(d["sentences"], ref) if l == key_label] or [" "]
                multi.append(rouge_compute([pred], [rs]))
            best = max(multi, key=lambda x: x["rouge-l_f"])
            max_all.append(best)
            ave_all.append({k: sum(m[k] for m in multi) / len(multi) for k in best})
        for k in max_all[0]:
            out["multi-ref-average_{}".format(k)] = sum(a[k] for a in ave_all) / len(ave_all)
            out["multi-ref-max_{}".format(k)] = sum(a[k] for a in max_all) / len(max_all)
    return out
