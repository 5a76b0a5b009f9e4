"""Example-level evaluator of the topic-segmentation path: restatement of
emnlp2023-topic_segmentation/src/metrics/seqeval.py:173-373 (`compute_window_metric`, `compute_metric_example_level`).
CPU only, pure Python/numpy -- it consumes the decoded predictions, not GPU tensors.

Third-party pieces the reference calls and that are NOT in its tree (parity UNPINNED for them -- no reference test or
golden vector exists, and the packages are absent here):
  * segeval==2.0.11 `pk` / `window_diff` (seqeval.py:24-25,196-197), restated from their published definitions
    (Beeferman et al. 1999; Pevzner & Hearst 2002) with segeval's defaults: masses in, window size k =
    max(2, round_half_even(mean reference segment mass / 2)), N - k probes, no Lamprier fix;
  * sklearn precision/recall/f1 on the flattened 0/1 lists (:231-234) -- binary, positive class 1, 0 on zero division;
  * seqeval==1.2.2 chunk P/R/F1 (:141-171) -- with the label set {"B-EOP", "O"} every "B-EOP" tag is a one-token chunk,
    so chunk-level scores equal the tag-level scores of "B-EOP".
"""
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
