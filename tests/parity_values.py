"""The parity report of SURVEY 8(d): the HIP path against the REFERENCE's own golden outputs (tests/golden/*.npz, written by
tools/gen_golden*.py importing the reference classes), measured -- not only asserted -- so bench.py can put the numbers in the
driver-run JSON line and the GPU suite can write them per round (profiles/rNN_parity_values.json).

Quantities per precision ("bf16" = the fast path the headline is quoted on, "parity" = split-bf16 products / fp32 activations, the
path that meets the north-star 1e-3):
  * bert_base_L512 (4 x 512 tokens, run_finetune.sh flags): max / mean |dlogit| at all and at labelled positions, boundary decisions
    equal / total, minimum reference decision margin |logit0 - logit1|, eval-loss delta; ONE training step (dropout 0): loss delta,
    worst relative gradient-norm error over every parameter, worst relative error / cosine of the stored first / last layer gradients;
  * config1_bert_base (BASELINE config 1: 32 documents -> 105 windows of 512 tokens, run_inference.sh): the same logit / boundary
    quantities over every labelled position (this is where "predicted boundary indices bit-exact" is judged:
    emnlp2023-topic_segmentation/src/ts_sentence_seq_labeling.py:1138-1191, metrics/seqeval.py:248-296).
Nothing here imports oracle/: the comparison is against stored outputs of the reference."""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PLAIN = dict(do_da_ts=False, do_cssl=False, do_tssp=False, cl_loss_weight=0.0, tssp_loss_weight=0.0)


def _logit_stats(got, ref, lab=None):
    d = np.abs(got - ref)
    out = dict(max_dlogit=float(d.max()), mean_dlogit=float(d.mean()), max_abs_logit=float(np.abs(ref).max()))
    if lab is not None:
        out["max_dlogit_labelled"] = float(d[lab].max())
        out["mean_dlogit_labelled"] = float(d[lab].mean())
        got, ref = got[lab], ref[lab]
    same = got.argmax(-1) == ref.argmax(-1)
    margin = np.abs(ref[..., 0] - ref[..., 1])
    out.update(boundaries_equal=int(same.sum()), boundaries_total=int(same.size), min_ref_margin=float(margin.min()),
               min_ref_margin_among_flips=(float(margin[~same].min()) if (~same).any() else None))
    return out


def fullsize(dev, precision, train=True):
    """bert_base_L512.npz: eval forward (+ one training step) against the reference's stored outputs"""
    from tests.test_gpu_fullsize import _fullsize_case
    from tests.test_gpu_model import build_model, to_dev
    z, sd, batch, arch, flags_of = _fullsize_case()
    m = build_model(arch, flags_of(z, "full_eval"), sd, dev)
    m.config.amdseg_precision = precision
    m.eval()
    random.seed(int(z["full_eval.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**to_dev(batch, dev))
    lab = (batch["labels"] != -100).numpy()
    out = dict(eval=_logit_stats(logits.float().cpu().numpy(), z["full_eval.logits"], lab))
    out["eval"]["loss_delta"] = abs(float(loss) - float(z["full_eval.loss"]))
    out["eval"]["max_dcos"] = float(np.abs(cos.float().cpu().numpy() - z["full_eval.cos"]).max()) if "full_eval.cos" in z.files else None
    if train and precision in ("bf16", "parity"):
        del m
        m = build_model(arch, flags_of(z, "train_full"), sd, dev)
        m.config.amdseg_precision = precision
        m.train()
        random.seed(int(z["train_full.random_seed"]))
        loss, _, _ = m(**to_dev(batch, dev))
        loss.backward()
        ref_loss = float(z["train_full.loss"])
        params = dict(m.named_parameters())
        worst_gn, worst_name = 0.0, None
        for n, v in zip(z["train_full.gradnorm_names"].tolist(), z["train_full.gradnorm_vals"].tolist()):
            if v <= 1e-6:
                continue
            e = abs(float(params[n].grad.float().norm()) - v) / v
            if e > worst_gn:
                worst_gn, worst_name = e, n
        worst_rel, min_cos, checked = 0.0, 1.0, 0
        worst_rel_no_qk = 0.0
        for k in z.files:
            if not k.startswith("train_full.grad."):
                continue
            n = k[len("train_full.grad."):]
            ref = torch.from_numpy(z[k])
            if float(ref.norm()) < 1e-6:
                continue
            g = params[n].grad.float().cpu()
            rel = float((g - ref).norm() / ref.norm())
            worst_rel = max(worst_rel, rel)
            if not ("self.query.bias" in n or "self.key.bias" in n):       # near-cancelling sums, reported separately
                worst_rel_no_qk = max(worst_rel_no_qk, rel)
            min_cos = min(min_cos, float(torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0)))
            checked += 1
        out["train_step"] = dict(loss=float(loss), ref_loss=ref_loss, loss_rel_delta=abs(float(loss) - ref_loss) / abs(ref_loss),
                                 gradnorm_max_rel_err=worst_gn, gradnorm_worst_param=worst_name,
                                 stored_grads_checked=checked, stored_grad_max_rel_err=worst_rel,
                                 stored_grad_max_rel_err_excl_qk_bias=worst_rel_no_qk, stored_grad_min_cosine=min_cos)
    del m
    return out


def config1(dev, precision, name="config1_bert_base", bs=8):
    """BASELINE config 1: every window of the 32 documents through the HIP path, labelled logits against the reference's"""
    from tests.test_config1_golden import load, run_windows
    from tests.test_gpu_model import build_model
    z, arch, sd, cols, offs, columns, _ = load(name)
    m = build_model(arch, dict(PLAIN), sd, dev).eval()
    m.config.amdseg_precision = precision
    lg, cs = run_windows(lambda b: m(**b), cols, columns, bs, dev)
    got = np.concatenate(lg, 0)
    ref = z["labelled_logits"]
    out = _logit_stats(got, ref)
    out.update(windows=len(lg), documents=int(z["ndocs"]), max_dcos=float(np.abs(np.concatenate(cs, 0) - z["cos"]).max()))
    del m
    return out


def measure(dev, precisions=("bf16", "parity"), train=True, with_config1=True):
    out = dict(reference="outputs of the reference's own model class stored in tests/golden/bert_base_L512.npz and config1_bert_base.npz "
                         "(tools/gen_golden.py / gen_golden_config1.py); tolerance of the north star: 1e-3 on logits, boundaries bit-exact")
    t0 = time.time()
    for p in precisions:
        rec = dict(bert_base_L512=fullsize(dev, p, train))
        if with_config1:
            rec["config1_bert_base"] = config1(dev, p)
        e, c = rec["bert_base_L512"]["eval"], rec.get("config1_bert_base")
        rec["max_dlogit"] = max(e["max_dlogit"], c["max_dlogit"] if c else 0.0)
        rec["all_boundaries_equal"] = bool(e["boundaries_equal"] == e["boundaries_total"] and (c is None or c["boundaries_equal"] == c["boundaries_total"]))
        rec["meets_1e-3"] = bool(rec["max_dlogit"] < 1e-3 and rec["all_boundaries_equal"])
        out[p] = rec
    out["seconds"] = round(time.time() - t0, 1)
    return out


def _round(o, nd=3):
    if isinstance(o, float):
        return float(f"{o:.{nd}e}")
    if isinstance(o, dict):
        return {k: _round(v, nd) for k, v in o.items()}
    return o


def rounded(o):
    return _round(o)
