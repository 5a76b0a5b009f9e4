"""f-1: example-level evaluator (spokennlp_amd/evaluate.py).  Pk / WindowDiff are third-party in the reference (segeval,
absent here: parity unpinned) -- checked against independent brute-force statements of the published definitions and
hand-worked cases; the surrounding reference logic (mass conversion, thresholds, P/R/F1) against hand-worked cases."""
import random

import numpy as np

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
