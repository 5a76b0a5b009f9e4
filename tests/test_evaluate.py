"""f-1: example-level evaluator (spokennlp_amd/evaluate.py).  Pk / WindowDiff are third-party in the reference (segeval,
absent here: parity unpinned) -- checked against independent brute-force statements of the published definitions and
hand-worked cases; the surrounding reference logic (mass conversion, thresholds, P/R/F1) against hand-worked cases."""
import random

import numpy as np
import pytest

from spokennlp_amd import evaluate as E


def brute_pk_wd(hyp_b, ref_b, k):
    """hyp_b / ref_b: 0/1 per unit, 1 = a boundary FOLLOWS the unit."""
    n = len(ref_b)
    dp = dw = 0
    for i in range(n - k):
        rb, hb = sum(ref_b[i:i + k]), sum(hyp_b[i:i + k])
        dp += (rb == 0) != (hb == 0)
        dw += rb != hb
    return dp / (n - k), dw / (n - k)


def test_mass_conversion():
    assert E.mass_from_start_label_sequence([1, 1, 0, 0, 1, 1]) == [1, 1, 3, 1]
    assert E.mass_from_start_label_sequence([0, 0, 1, 0]) == [3, 1]
    assert E.mass_from_start_label_sequence([0, 0]) == [2]


def test_pk_wd_against_bruteforce():
    r = random.Random(0)
    for _ in range(200):
        n = r.randrange(6, 60)
        ref = [1 if r.random() < 0.25 else 0 for _ in range(n)]; ref[-1] = 1
        hyp = [1 if r.random() < 0.25 else 0 for _ in range(n)]; hyp[-1] = 1
        rm, hm = E.mass_from_start_label_sequence(ref), E.mass_from_start_label_sequence(hyp)
        k = E.window_size(rm)
        assert k == max(2, int(round(n / len(rm) / 2)))          # python round() is half-even as well
        bp, bw = brute_pk_wd(hyp, ref, k)
        assert abs(E.pk(hm, rm) - bp) < 1e-12 and abs(E.window_diff(hm, rm) - bw) < 1e-12
        assert E.pk(rm, rm) == 0 and E.window_diff(rm, rm) == 0
        assert E.window_diff(hm, rm) >= E.pk(hm, rm) - 1e-12        # WD counts every Pk miss plus near misses


def test_hand_worked_case():
    # 8 units, reference segments [4,4] -> k = 2; hypothesis [2,6]
    assert E.window_size([4, 4]) == 2
    # probes i=0..5: ref same-segment? T T F F T T ; hyp: F F T T T T -> 4 differ
    assert abs(E.pk([2, 6], [4, 4]) - 4 / 6) < 1e-12
    assert abs(E.window_diff([2, 6], [4, 4]) - 4 / 6) < 1e-12


def test_example_level_metrics():
    logits = [[[2.0, 0.0], [0.0, 1.0], [0.1, 0.0], [3.0, -1.0]], [[0.0, 2.0], [1.0, 0.0]]]
    labels = [[0, 1, 1, 0], [1, 0]]
    res = E.compute_metric_example_level(logits, labels, threshold=0.5)
    # argmax predictions: [0,1,0,0], [1,0] -> predicted boundaries 4, true 3, TP 3
    assert abs(res["precision"] - 3 / 4) < 1e-12 and res["recall"] == 1.0
    assert abs(res["accuracy"] - 5 / 6) < 1e-12
    key = "threshold_0.5_example_level_"
    assert res[key + "precision"] == 0.75 and res[key + "recall"] == 1.0 and res[key + "f1"] == round(2 * 0.75 / 1.75, 4)
    assert 0 <= res[key + "pk"] <= 1 and abs(res[key + "pk"] - (1 - res[key + "1-pk"])) < 1e-12
    # a higher threshold drops the marginal boundary (softmax([0.1, 0])[0] = 0.525)
    res2 = E.compute_metric_example_level(logits, labels, threshold=0.6)
    assert res2["threshold_0.6_example_level_precision"] == 1.0
    # ignored positions are removed before scoring
    res3 = E.compute_metric_example_level(logits, [[0, -100, 1, 0], [1, 0]], threshold=0.5)
    assert abs(res3["accuracy"] - 4 / 5) < 1e-12
    # f1@k moves a near-miss onto the true boundary
    lg = [[[0.0, 3.0], [3.0, 0.0], [0.0, 3.0], [3.0, 0.0]]]
    r4 = E.compute_metric_example_level(lg, [[0, 1, 1, 0]], threshold=0.5, f1_at_k=1)
    assert r4["f1@1_example_level_f1"] == 1.0 and r4["threshold_0.5_example_level_f1"] == 0.5


def test_pk_windowdiff_hand_worked_cases():
    """Pk (Beeferman et al. 1999) and WindowDiff (Pevzner & Hearst 2002) on cases small enough to enumerate by hand from the papers'
    definitions -- independent of the implementation, NOT outputs of segeval (absent here; parity with it stays unpinned):
      reference AABBB (masses 2,3), hypothesis AAABB (3,2): mean reference mass 2.5 -> k = max(2, round(1.25)) = 2, probes (0,2) (1,3) (2,4).
        same-segment?  ref: no, no, yes   hyp: yes, no, no   -> disagreements on probes 1 and 3 -> Pk = 2/3
        boundaries in the window: ref 1,1,0   hyp 0,1,1      -> differ on windows 1 and 3     -> WD = 2/3
      a near miss is penalised less than a miss by WindowDiff only through the window count: reference 4,4 vs hypothesis 8 (no boundary):
        k = 2, 6 probes; the reference boundary (between units 3|4) lies inside probes (2,4) and (3,5): Pk = WD = 2/6
      identical segmentations score 0; the all-boundaries hypothesis 1,1,1,1,1,1 against reference 3,3: k = 2 (round-half-even of 1.5),
        4 probes; every hypothesis probe spans boundaries (never "same"), the reference is "same" only on probe (0,2)... enumerated below."""
    from spokennlp_amd import evaluate as E
    assert E.window_size([2, 3]) == 2 and E.window_size([4, 4]) == 2 and E.window_size([3, 3]) == 2 and E.window_size([10, 10]) == 5
    assert E.window_size([5]) == 2            # round-half-even(2.5) = 2
    assert abs(E.pk([3, 2], [2, 3]) - 2 / 3) < 1e-12 and abs(E.window_diff([3, 2], [2, 3]) - 2 / 3) < 1e-12
    assert abs(E.pk([8], [4, 4]) - 2 / 6) < 1e-12 and abs(E.window_diff([8], [4, 4]) - 2 / 6) < 1e-12
    assert E.pk([4, 4], [4, 4]) == 0.0 and E.window_diff([4, 4], [4, 4]) == 0.0
    # reference 3,3 = AAABBB, hypothesis all boundaries; probes (0,2) (1,3) (2,4) (3,5):
    #   ref same? yes, no, no, yes ; hyp same? never -> Pk = 2/4.  boundaries in window: ref 0,1,1,0 ; hyp 2,2,2,2 -> WD = 4/4
    assert abs(E.pk([1] * 6, [3, 3]) - 0.5) < 1e-12 and abs(E.window_diff([1] * 6, [3, 3]) - 1.0) < 1e-12
    # WindowDiff >= Pk always (a window with a different boundary count is the only way "same segment?" can differ)
    import random
    rng = random.Random(0)
    for _ in range(200):
        n = rng.randrange(6, 40)
        def masses():
            cuts = sorted(rng.sample(range(1, n), rng.randrange(0, min(6, n - 1))))
            return [b - a for a, b in zip([0] + cuts, cuts + [n])]
        h, r = masses(), masses()
        assert E.window_diff(h, r) >= E.pk(h, r) - 1e-12


def test_alimeeting_challenge_scoring():
    """challenge_evaluate.py:137-210 on two in-memory meetings: boundaries are read at paragraph ends only, the final one is dropped"""
    from spokennlp_amd import evaluate as E
    lab = [dict(meeting_key="m1", sentences=list("abcdefgh"), paragraph_segment_ids=[dict(id=i) for i in (2, 4, 6, 8)],
                topic_segment_ids=[dict(id=4), dict(id=8)]),
           dict(meeting_key="m2", sentences=list("abcdef"), paragraph_segment_ids=[dict(id=i) for i in (1, 3, 5, 6)],
                topic_segment_ids=[dict(id=3), dict(id=6)])]
    pred = [dict(meeting_key="m1", topic_segment_ids=[dict(id=4)]), dict(meeting_key="m2", topic_segment_ids=[dict(id=5)])]
    out = E.topic_segment_evaluate_samples(lab, pred)
    # m1: paragraph-end labels [0,1,0|1] -> scored [0,1,0], preds [0,1,0]; m2: labels [0,1,0|1] -> [0,1,0], preds [0,0,1]
    assert out["test_precision"] == 0.5 and out["test_recall"] == 0.5 and out["test_f1"] == 0.5
    m1 = E.compute_window_metric_alimeeting([[0, 1, 0]], [[0, 1, 0]], "x_")
    assert m1["x_1-pk"] == 1.0 and m1["x_avg_pred_cnt"] == 1.0
    assert abs(out["score"] - E.topic_segment_score(0.5, out["test_1-pk"], out["test_1-wd"])) < 1e-12
    assert 0.0 <= out["test_1-pk"] <= 1.0


def test_trainer_device_side_nan_filter_matches_the_stock_formula():
    """spokennlp_amd.trainer.filter_nonfinite == the `logging_nan_inf_filter` branch of transformers.Trainer._inner_training_loop:
    a nan / inf step loss adds tr_loss / (1 + global_step - last_logged), a finite one adds itself"""
    import torch
    from spokennlp_amd.trainer import filter_nonfinite
    tr_loss = torch.tensor(6.0)
    for bad in (float("nan"), float("inf"), -float("inf")):
        assert float(filter_nonfinite(torch.tensor(bad), tr_loss, 3)) == 2.0
    assert float(filter_nonfinite(torch.tensor(1.25), tr_loss, 3)) == 1.25
    assert filter_nonfinite(torch.tensor(1.0, dtype=torch.bfloat16), tr_loss, 4).dtype == torch.bfloat16


def test_alimeeting_window_metric_is_strict_like_its_reference():
    """(round-2 advisor) challenge_evaluate.py:105-111 prints the failing example and raises RuntimeError; the emnlp2023 twin
    (seqeval.py:214-215) swallows it.  An empty input is an error, not a ZeroDivisionError / nan mean."""
    import pytest
    from spokennlp_amd import evaluate as E
    good_p, good_r = [[0, 1, 0, 1], [1, 0, 0, 1]], [[0, 1, 0, 1], [0, 0, 1, 1]]
    out = E.compute_window_metric_alimeeting(good_p, good_r, prefix="t_")
    assert out["t_avg_pred_cnt"] == 2.0 and out["t_avg_true_cnt"] == 2.0 and 0.0 <= out["t_1-pk"] <= 1.0
    bad_p, bad_r = good_p + [[0, 1]], good_r + [[0, 1, 1]]               # mass mismatch in the third example
    with pytest.raises(RuntimeError, match="example 2"):
        E.compute_window_metric_alimeeting(bad_p, bad_r)
    assert E.compute_window_metric(bad_p, bad_r)["1-pk"] == E.compute_window_metric(good_p, good_r)["1-pk"]   # emnlp2023: dropped silently
    with pytest.raises(ValueError):
        E.compute_window_metric_alimeeting([], [])


# ---------------------------------------------------------------------------------------------------- ROUGE (8(f)-4)
# the worked example the reference holds: alimeeting4mug/metrics/rouge/rouge.py:62-93 (inputs and the nine expected numbers; data)
_ROUGE_HYP = ("the #### transcript is a written version of each day 's cnn student news program use this transcript to he    lp students with "
              "reading comprehension and vocabulary use the weekly newsquiz to test your knowledge of storie s you     saw on cnn student news")
_ROUGE_REF = ("this page includes the show transcript use the transcript to help students with reading comprehension and     vocabulary at the "
              "bottom of the page , comment for a chance to be mentioned on cnn student news . you must be a teac    her or a student age # # or "
              "older to request a mention on the cnn student news roll call . the weekly newsquiz tests     students ' knowledge of even ts in the news")
_ROUGE_EXPECTED = {"rouge-1": {"f": 0.4786324739396596, "p": 0.6363636363636364, "r": 0.3835616438356164},
                   "rouge-2": {"f": 0.2608695605353498, "p": 0.3488372093023256, "r": 0.20833333333333334},
                   "rouge-l": {"f": 0.44705881864636676, "p": 0.5277777777777778, "r": 0.3877551020408163}}


def test_rouge_reproduces_the_references_worked_example():
    """rouge-l: the package-default set counting gives the example's numbers to the last digit; rouge-1 / rouge-2: the example's
    numbers are the multiset counting (the README example the reference copied mixes the two; evaluate.py says so)"""
    ex = E.rouge_get_scores(_ROUGE_HYP, _ROUGE_REF)[0]
    ne = E.rouge_get_scores(_ROUGE_HYP, _ROUGE_REF, exclusive=False)[0]
    for s in "fpr":
        assert ex["rouge-l"][s] == _ROUGE_EXPECTED["rouge-l"][s]
        assert ne["rouge-1"][s] == _ROUGE_EXPECTED["rouge-1"][s]
        assert ne["rouge-2"][s] == _ROUGE_EXPECTED["rouge-2"][s]
    # the other counting of each differs visibly, so the match above is not an accident of the example
    assert abs(ne["rouge-l"]["f"] - _ROUGE_EXPECTED["rouge-l"]["f"]) > 0.05
    assert abs(ex["rouge-1"]["f"] - _ROUGE_EXPECTED["rouge-1"]["f"]) > 0.01


def test_rouge_properties_and_the_es_metric_glue():
    rng = random.Random(5)
    vocab = ["w%d" % i for i in range(12)]
    for _ in range(50):
        h = " ".join(rng.choice(vocab + ["."]) for _ in range(rng.randint(3, 30))).strip(". ") or "w0"
        r = " ".join(rng.choice(vocab + ["."]) for _ in range(rng.randint(3, 30))).strip(". ") or "w1"
        for excl in (True, False):
            a, b = E.rouge_get_scores(h, r, exclusive=excl)[0], E.rouge_get_scores(r, h, exclusive=excl)[0]
            for m in ("rouge-1", "rouge-2"):                  # swapping the roles swaps precision and recall
                assert abs(a[m]["p"] - b[m]["r"]) < 1e-12 and abs(a[m]["f"] - b[m]["f"]) < 1e-12
            for m in a:                                       # (the list-counted rouge-l adds one LCS per sentence PAIR and may pass 1)
                if excl or m != "rouge-l":
                    assert 0.0 <= a[m]["p"] <= 1.0 + 1e-12 and 0.0 <= a[m]["r"] <= 1.0 + 1e-12
            same = E.rouge_get_scores(h, h, exclusive=excl)[0]
            assert abs(same["rouge-1"]["f"] - 1.0) < 1e-7 and same["rouge-1"]["p"] == 1.0
            if excl:
                assert same["rouge-l"]["p"] == 1.0
    # LCS words against a brute force over subsequences
    import itertools
    for _ in range(30):
        x = [rng.choice("abcd") for _ in range(rng.randint(1, 8))]
        y = [rng.choice("abcd") for _ in range(rng.randint(1, 8))]
        best = 0
        for k in range(len(x), 0, -1):
            if any(_is_subseq(c, y) for c in set(itertools.combinations(x, k))):
                best = k
                break
        got = E._lcs_words(x, y)
        assert len(got) == best and _is_subseq(got, x) and _is_subseq(got, y)
    # the extractive-summarisation glue: selected sentences joined, empty selection = the one-blank summary
    docs = [dict(sentences=["a b c", "d e", "f g h i"], labels=["B-EOP", "O", "B-EOP"], predictions=["B-EOP", "B-EOP", "O"],
                 multi_labels=[["B-EOP", "O", "B-EOP"], ["O", "B-EOP", "O"]]),
            dict(sentences=["x y", "z"], labels=["O", "O"], predictions=["O", "B-EOP"], multi_labels=[["O", "B-EOP"]])]
    res = E.es_rouge_metrics(docs)
    one = E.rouge_get_scores(["a b c d e", "z"], ["a b c f g h i", " "], avg=True)
    assert res["score"] == one["rouge-1"]["f"] and res["rouge-l_r"] == one["rouge-l"]["r"]
    assert res["multi-ref-max_rouge-1_f"] >= res["multi-ref-average_rouge-1_f"]
    d0 = [E.rouge_compute([["a b c", "d e"]], [r]) for r in (["a b c", "f g h i"], ["d e"])]
    d1 = E.rouge_compute([["z"]], [["z"]])
    assert abs(res["multi-ref-average_rouge-2_p"] - ((d0[0]["rouge-2_p"] + d0[1]["rouge-2_p"]) / 2 + d1["rouge-2_p"]) / 2) < 1e-12


def _is_subseq(c, y):
    it = iter(y)
    return all(any(a == b for b in it) for a in c)


# ---------------------------------------------------------------------------------------------------- seqeval chunk scores, compute_metrics, postprocess
def test_seqeval_scores_reproduce_the_worked_example_the_reference_file_holds():
    """emnlp2023-topic_segmentation/src/metrics/seqeval.py:96-105 (docstring of the metric): keys, overall_f1 0.5, PER f1 1.0"""
    predictions = [['O', 'O', 'B-MISC', 'I-MISC', 'I-MISC', 'I-MISC', 'O'], ['B-PER', 'I-PER', 'O']]
    references = [['O', 'O', 'O', 'B-MISC', 'I-MISC', 'I-MISC', 'O'], ['B-PER', 'I-PER', 'O']]
    res = E.seqeval_scores(predictions, references)
    assert list(res.keys()) == ['MISC', 'PER', 'overall_precision', 'overall_recall', 'overall_f1', 'overall_accuracy']
    assert res["overall_f1"] == 0.5 and res["PER"]["f1"] == 1.0
    assert res["MISC"] == {"precision": 0.0, "recall": 0.0, "f1": 0.0, "number": 1} and res["PER"]["number"] == 1
    assert res["overall_precision"] == 0.5 and res["overall_recall"] == 0.5 and res["overall_accuracy"] == 0.8


def _brute_chunks(rows):
    """independent chunk reader for B- / I- / O tags: a chunk starts at a B, or at an I that does not continue a chunk of its type"""
    out = set()
    for ri, row in enumerate(rows):
        i = 0
        while i < len(row):
            if row[i] == "O":
                i += 1
                continue
            typ = row[i][2:]
            j = i + 1
            while j < len(row) and row[j] == "I-" + typ:
                j += 1
            out.add((ri, typ, i, j - 1))
            i = j
    return out


def test_seqeval_scores_against_bruteforce_chunk_counting():
    rng = np.random.RandomState(5)
    tags = ["O", "B-A", "I-A", "B-B", "I-B"]
    for trial in range(30):
        refs = [[tags[i] for i in rng.randint(0, 5, size=rng.randint(1, 14))] for _ in range(rng.randint(1, 6))]
        preds = [[tags[i] for i in rng.randint(0, 5, size=len(r))] for r in refs]
        res = E.seqeval_scores(preds, refs)
        ct, cp = _brute_chunks(refs), _brute_chunks(preds)
        tp = len(ct & cp)
        p = tp / len(cp) if cp else 0.0
        r = tp / len(ct) if ct else 0.0
        f = 2 * p * r / (p + r) if p + r else 0.0
        assert abs(res["overall_precision"] - p) < 1e-12 and abs(res["overall_recall"] - r) < 1e-12 and abs(res["overall_f1"] - f) < 1e-12
        for typ in ("A", "B"):
            nt = sum(1 for c in ct if c[1] == typ)
            if typ in res:
                assert res[typ]["number"] == nt
            else:
                assert nt == 0 and not any(c[1] == typ for c in cp)
        flat = [(a, b) for pr, rr in zip(preds, refs) for a, b in zip(pr, rr)]
        assert abs(res["overall_accuracy"] - sum(a == b for a, b in flat) / len(flat)) < 1e-12
    with pytest.raises(ValueError):
        E.seqeval_scores([["O"]], [["O", "O"]])


def test_seqeval_scores_on_the_topic_segmentation_label_set_equal_tag_level_scores():
    """{"B-EOP", "O"}: every B-EOP is a one-token chunk (B after B closes the chunk), so chunk scores == binary scores of the tag"""
    rng = np.random.RandomState(1)
    refs = [["B-EOP" if v else "O" for v in rng.rand(n) < 0.2] for n in (7, 1, 30, 12)]
    preds = [["B-EOP" if v else "O" for v in rng.rand(len(r)) < 0.25] for r in refs]
    res = E.seqeval_scores(preds, refs)
    p, r, f = E.binary_prf([int(x == "B-EOP") for row in refs for x in row], [int(x == "B-EOP") for row in preds for x in row])
    assert abs(res["overall_precision"] - p) < 1e-12 and abs(res["overall_recall"] - r) < 1e-12 and abs(res["overall_f1"] - f) < 1e-12
    assert set(res) == {"EOP", "overall_precision", "overall_recall", "overall_f1", "overall_accuracy"}
    assert res["EOP"]["number"] == sum(x == "B-EOP" for row in refs for x in row)
    # nothing predicted, nothing true: zero division acts as 0 (seqeval's "warn")
    z = E.seqeval_scores([["O", "O"]], [["O", "O"]])
    assert z["overall_f1"] == 0.0 and z["overall_accuracy"] == 1.0


def test_compute_metrics_closure_of_the_finetune_script():
    """ts_sentence_seq_labeling.py:1018-1074: p = ((logits (N,2,L,2), cos_sim), (labels (N,2,L), sent_level_labels)) -> anchor scores + da_ scores"""
    rng = np.random.RandomState(3)
    N, L = 5, 16
    labels = np.full((N, 2, L), -100, dtype=np.int64)
    for n in range(N):
        for h in range(2):
            pos = np.sort(rng.choice(np.arange(1, L), size=rng.randint(2, 6), replace=False))
            labels[n, h, pos] = rng.randint(0, 2, size=len(pos))
    logits = rng.randn(N, 2, L, 2).astype(np.float32)
    cos = rng.rand(N, 4).astype(np.float32)
    cm = E.make_compute_metrics()
    res = cm(((logits, cos), (labels, np.zeros((N, 2, L)))))
    for half, prefix in ((0, ""), (1, "da_")):
        pred = logits[:, half].argmax(-1)
        keep = labels[:, half] != -100
        t = (labels[:, half][keep] == 0).astype(int).tolist()
        q = (pred[keep] == 0).astype(int).tolist()
        p, r, f = E.binary_prf(t, q)
        assert abs(res[prefix + "overall_precision"] - p) < 1e-12 and abs(res[prefix + "overall_recall"] - r) < 1e-12
        assert abs(res[prefix + "overall_f1"] - f) < 1e-12 and abs(res[prefix + "EOP_f1"] - f) < 1e-12
        assert res[prefix + "EOP_number"] == sum(t)
        assert abs(res[prefix + "overall_accuracy"] - float((pred[keep] == labels[:, half][keep]).mean())) < 1e-12
    # also the (predictions, label_ids) object transformers hands over, and the four-number form
    class P_:
        predictions, label_ids = (logits, cos), (labels, None)

        def __iter__(self):
            return iter((self.predictions, self.label_ids))
    short = E.make_compute_metrics(return_entity_level_metrics=False)(P_())
    assert set(short) == {"precision", "recall", "f1", "accuracy"} and short["f1"] == res["overall_f1"]
    # "cos" predictor: per-EOP sigmoid scores (N,2,k), > 0.5 = "O"
    sc = rng.rand(N, 2, L).astype(np.float32)
    rc = E.make_compute_metrics(ts_score_predictor="cos")(((sc, cos), (labels, None)))
    n0 = int((labels[0, 0] != -100).sum())
    t = [int(v == 0) for row in labels[:, 0] for v in row if v != -100]
    q = [int(not (s > 0.5)) for n in range(N) for s in sc[n, 0][:int((labels[n, 0] != -100).sum())]]
    assert n0 > 0 and abs(rc["overall_f1"] - E.binary_prf(t, q)[2]) < 1e-12
    with pytest.raises(ValueError):
        E.make_compute_metrics(ts_score_predictor="nope")(((logits, cos), (labels, None)))


def test_wiki_section_sentence_level_rescoring_and_the_str_metric_file(tmp_path):
    """postprocess_predictions.py:7-75 + utils.py:23-48 on a synthetic paragraph-level run"""
    import json
    rng = np.random.RandomState(9)
    docs, preds = [], []
    for d in range(7):
        ns = rng.randint(6, 25)
        sent = [-100] * ns
        para_ends = sorted(set(rng.choice(np.arange(ns - 1), size=rng.randint(2, min(6, ns - 1)), replace=False).tolist()))
        for i in para_ends:
            sent[i] = int(rng.rand() < 0.4)
        sent[-1] = 1                                                     # the document's last sentence (dropped by [:-1])
        plab = [sent[i] for i in para_ends]
        ppred = [int(rng.rand() < 0.4) for _ in para_ends]
        docs.append({"sentences": ["s"] * ns, "labels": sent})
        preds.append({"labels": ["B-EOP" if v else "O" for v in plab], "predictions": ["B-EOP" if v else "O" for v in ppred]})
    data_file, pred_file = tmp_path / "test.jsonl", tmp_path / "pred.txt"
    data_file.write_text("".join(json.dumps(d) + "\n" for d in docs))
    pred_file.write_text("".join(json.dumps(d) + "\n" for d in preds))
    pp, pl, sl = E.read_total_pred_and_labels(str(data_file), str(pred_file))
    assert [len(x) for x in sl] == [len(d["labels"]) - 1 for d in docs]
    sent_res, para_res = E.sent_level_metric_from_para_level_models(pp, pl, sl)
    assert sl[0].count(-100) > 0                                         # inputs untouched
    # expected: spread by hand
    sp, sll = [], []
    for d, p in zip(docs, preds):
        lab = d["labels"][:-1]
        it = iter([0 if v == "O" else 1 for v in p["predictions"]])
        sp.append([next(it) if v != -100 else 0 for v in lab]); sll.append([0 if v == -100 else v for v in lab])
    assert sent_res == E.compute_window_metric(sp, sll) and para_res == E.compute_window_metric(pp, pl)
    # same P / R / F at both granularities (only zeros are added); the window metrics differ
    assert sent_res["precision"] == para_res["precision"] and sent_res["recall"] == para_res["recall"]
    lines = []
    res = E.wiki_section_sent_level_metric(str(data_file), str(pred_file), disease_cnt=3, city_cnt=4, out=lines.append)
    assert set(res) == {"wiki_section_disease", "wiki_section_city", "wiki_section"} and res["wiki_section"]["sent_level"] == sent_res
    assert lines[0] == "p / r / f1 / pk / wd" and lines[1] == "data_name:  wiki_section_disease" and lines[2].startswith("sent_level: ")
    with pytest.raises(AssertionError):
        E.wiki_section_sent_level_metric(str(data_file), str(pred_file))   # the real split sizes (718 + 3893) do not match 7 documents
    # the metric file
    m = {"threshold_0.5_example_level_precision": 0.81337, "threshold_0.5_example_level_recall": 0.5, "threshold_0.5_example_level_f1": 0.61925,
         "threshold_0.5_example_level_pk": 0.1387, "threshold_0.5_example_level_wd": 0.14970000000000006}
    rp = tmp_path / "example_level_predict_x_results.json"
    rp.write_text(json.dumps(m))
    out = E.convert_res_format(str(rp), 0.5, out=lambda *_: None)
    assert out.endswith("example_level_predict_x_results_str_metric.txt")
    assert open(out).read() == "p / r / f / pk / wd\nthreshold_0.5_example_level_metric\n81.34 / 50.00 / 61.92 / 13.87 / 14.97\n\n"
    assert [E.abridge_model_name(x) for x in ("allenai/longformer-base-4096", "google/bigbird-roberta-base", "bert-base-uncased", "google/electra-base")] == ["lf", "bb", "bert", "ele"]
    with pytest.raises(ValueError):
        E.abridge_model_name("gpt2")
