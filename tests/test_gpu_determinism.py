"""Bit-reproducibility of the training step (the reference trains under HF `set_seed` + torch's default non-deterministic scatter kernels;
here the order-dependent sums are removed instead):
  * the loss heads' scatter sums (CSSL feature rows, TSSP rows / weights) are 64-bit fixed-point integer sums (csrc/heads.hip) -- always;
  * the word / explicit-position embedding gradients are sums over a stable sort of the ids (amdseg_scatter_rows_sorted) under
    config.amdseg_deterministic / AMDSEG_DETERMINISTIC=1 (the default scatter uses fp32 atomics, like torch's embedding backward).
Everything else of a BERT step (column sums through partials, attention backward, AdamW, the gradient norm) has a fixed summation order."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from tests.test_oracle_golden import load_case, flags_of  # noqa: E402
from tests.test_gpu_model import build_model, to_dev  # noqa: E402


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_scatter_rows_sorted_equals_index_add_and_repeats_bitwise(dev, dtype):
    from spokennlp_amd import lib as L
    g = torch.Generator(device="cpu").manual_seed(5)
    M, H, V, pad = 1000, 200, 37, 3                         # H not a multiple of 256, every key ~27 times, one key skipped, some out of range
    dz = torch.randn(M, H, generator=g).to(dtype).to(dev)
    keys = torch.randint(0, V, (M,), generator=g)
    keys[::50] = -1
    keys[7::91] = V + 2
    keys = keys.to(dev)
    order = torch.sort(keys, stable=True)[1]
    adt = L.BF16 if dtype == torch.bfloat16 else L.F32
    outs = []
    for _ in range(2):
        table = torch.full((V, H), 0.5, dtype=torch.float32, device=dev)
        rc = L.load().amdseg_scatter_rows_sorted(dz.data_ptr(), keys.data_ptr(), order.data_ptr(), table.data_ptr(), M, H, V, pad, adt,
                                                 torch.cuda.current_stream().cuda_stream)
        L.check(rc, "amdseg_scatter_rows_sorted")
        outs.append(table)
    assert torch.equal(outs[0], outs[1])
    ok = (keys >= 0) & (keys < V) & (keys != pad)
    ref = torch.full((V, H), 0.5, dtype=torch.float64, device=dev).index_add_(0, keys[ok], dz[ok].double())
    assert float((outs[0].double() - ref).abs().max()) < 2e-5
    assert torch.equal(outs[0][pad], torch.full((H,), 0.5, device=dev))
    # the summation order is the batch order of each key's rows: equal to a serial fp32 sum, bit for bit
    k0 = int(keys[ok][0])
    serial = torch.full((H,), 0.0, dtype=torch.float32, device=dev)
    for r in torch.nonzero(keys == k0).flatten().tolist():
        serial = serial + dz[r].float()
    assert torch.equal(outs[0][k0], serial + 0.5)


def _three_steps(dev, precision, deterministic, case="tiny_L128", lazy_zero=None, skip_backward_at=None, bwd_cu_budget=0, keepmask_in_ln=None):
    if case == "bert_base_L512":                           # bert-base, 4 x 512 tokens (x 2 with the augmented half): the H = 768 kernels
        from tests.test_gpu_fullsize import _fullsize_case
        z, sd, batch, arch, fl = _fullsize_case()
        flags = fl(z, "train_full")
    else:
        z, sd, batch, arch = load_case(case)
        flags = flags_of(z, "train_full")
    m = build_model(arch, flags, sd, dev, dropout=0.1)
    m.config.amdseg_precision = precision
    m.config.amdseg_deterministic = deterministic
    m.train()
    m.amdseg_seed = 11
    random.seed(3)
    b = to_dev(batch, dev)
    if keepmask_in_ln is not None:                         # (before the first forward: the activation arenas carry the pairing)
        m.engine().keepmask_in_ln = keepmask_in_ln
    losses = []
    for it in range(3):
        loss, _, _ = m(**b)
        if lazy_zero is not None:
            m.engine().lazy_zero = lazy_zero
        if bwd_cu_budget:
            eng_ = m.engine()
            eng_._bwd_cu_budget, eng_._bwd_cu_budget_always = bwd_cu_budget, True
            if not hasattr(eng_, "_seen_budgets"):             # what the engine's context held WHILE backward ran (autograd's thread), per step
                eng_._seen_budgets, inner = [], eng_._backward

                def spy(*a, _inner=inner, _eng=eng_, **kw):
                    _eng._seen_budgets.append(_eng.ctx.cu_budget())
                    return _inner(*a, **kw)
                eng_._backward = spy
        if it != skip_backward_at:                         # (a step without a backward: every gradient counts as zero)
            loss.backward()
        losses.append(loss.item())
        m.engine().adamw_step(1e-3, max_grad_norm=1.0)
    torch.cuda.synchronize()
    eng = m.engine()
    assert eng.deterministic == deterministic
    if bwd_cu_budget:                                      # the budget was in force inside every backward and is gone behind it
        assert eng._seen_budgets == [bwd_cu_budget] * 3 and eng.ctx.cu_budget() == 0
    return losses, eng.fp.flat_p.detach().clone(), {n: p.detach().clone() for n, p in m.named_parameters()}


@pytest.mark.parametrize("precision,case", [("bf16", "tiny_L128"), ("parity", "tiny_L128"), ("bf16", "bert_base_L512")])
def test_deterministic_mode_three_training_steps_are_bit_identical(dev, precision, case):
    """two runs from the same weights, seeds and batch: dropout (stateless hash), CSSL sampling (`random`), three optimiser steps with
    clipping -- the same bits in every parameter.  (lr 1e-3 so that three steps move every weight well above its last bit)"""
    a = _three_steps(dev, precision, True, case)
    b = _three_steps(dev, precision, True, case)
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1])
    for n, p in a[2].items():
        assert torch.equal(p, b[2][n]), n
    assert a[0][0] != a[0][2]                              # ... and the steps did move the model


def test_deterministic_mode_equals_the_default_scatter_to_rounding(dev):
    """the sorted sums and the atomics add the same numbers in another order"""
    a = _three_steps(dev, "bf16", True)
    b = _three_steps(dev, "bf16", False)
    assert abs(a[0][0] - b[0][0]) == 0.0                   # the first forward does not depend on it
    d = (a[1] - b[1]).abs().max().item()
    assert d < 5e-3, d                                     # AdamW at lr 1e-3 amplifies last-bit gradient noise to at most ~lr per step


def test_heads_backward_is_bit_identical_run_to_run(dev):
    """default mode: everything upstream of the embedding tables is reproducible -- the encoder layers' and heads' gradients of two
    backward passes over the same batch are the same bits (they were not while CSSL / TSSP scattered with fp32 atomics: the sum over the
    anchors that list one feature row depended on the arrival order, and every layer below inherited it)."""
    z, sd, batch, arch = load_case("tiny_L128")
    grads = []
    for _ in range(2):
        m = build_model(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train()
        m.config.amdseg_precision = "parity"
        m.amdseg_seed = 5
        random.seed(9)
        loss, _, _ = m(**to_dev(batch, dev))
        loss.backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters()
                      if p.grad is not None and "word_embeddings" not in n})
    for n, g in grads[0].items():
        assert torch.equal(g, grads[1][n]), n


@pytest.mark.parametrize("precision", ["bf16", "parity"])
def test_lazy_gradient_zeroing_changes_no_bit(dev, precision):
    """engine.lazy_zero: the fused AdamW leaves the encoder layers' gradients in place and the next backward overwrites them
    (accumulate_grads = 0) instead of adding to zeros -- the same bits in every parameter after three steps, also when a step runs
    without a backward in between (the stale slice then counts as zero)."""
    for skip in (None, 1):
        a = _three_steps(dev, precision, True, lazy_zero=True, skip_backward_at=skip)
        b = _three_steps(dev, precision, True, lazy_zero=False, skip_backward_at=skip)
        assert a[0] == b[0]
        assert torch.equal(a[1], b[1])


@pytest.mark.parametrize("family", ["longformer", "bigbird", "ponet"])
def test_lazy_gradient_zeroing_changes_no_bit_in_the_other_engines(dev, family):
    """the same for the Longformer (global projections accumulate outside the layer call: they sit in the eagerly zeroed part of the flat buffer),
    BigBird and PoNet engines: two optimiser steps, lazy vs eager, deterministic embedding sums"""
    def run(lazy):
        if family == "longformer":
            from tests.test_gpu_longformer import build_lf, lf_case
            z, sd, batch, arch = lf_case("lf_tiny_L128_w16")
            m = build_lf(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1)
            b = {k: v.to(dev) for k, v in batch.items()}
        elif family == "bigbird":
            from tests.test_gpu_bigbird import build_bb
            from tests.test_oracle_golden import bb_case
            z, sd, batch, arch = bb_case("bb_tiny_L1024")
            m = build_bb(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1)
            b = {k: v.to(dev) for k, v in batch.items()}
        else:
            from tests.test_gpu_ponet import build, make_inputs
            m, _ = build(dev, dropout=0.1)
            m = m.to(dev)
            ids, am, seg, lab = (t.to(dev) for t in make_inputs(2, 256, 3))
            b = dict(input_ids=ids, attention_mask=am, segment_ids=seg, labels=lab)
        m.config.amdseg_deterministic = True
        m.train()
        m.amdseg_seed = 11
        random.seed(3)
        losses = []
        for _ in range(3):                                 # (lazy zeroing first matters in the SECOND backward; the third loss sees its update)
            out = m(**b)
            loss = out[0] if isinstance(out, (tuple, list)) else out.loss
            m.engine().lazy_zero = lazy
            loss.backward()
            losses.append(loss.item())
            m.engine().adamw_step(1e-3, max_grad_norm=1.0)
        torch.cuda.synchronize()
        assert m.engine().fp.grad_stale == lazy
        return losses, m.engine().fp.flat_p.detach().clone()
    a, b = run(True), run(False)
    if family in ("ponet", "bigbird"):
        # PoNet's pooling backward merges run pieces with fp32 atomics, BigBird's token-type rows / list attention likewise: two EAGER runs already
        # differ in the last bits (and Adam at lr 1e-3 turns a last-bit gradient difference into up to ~lr per step): equal to that noise only
        # (a weight whose gradient is noise around zero moves by +-lr per Adam step either way, so single weights may differ by several lr between
        # two runs; gradients added onto stale ones would move nearly EVERY weight by ~lr = 1e-3: the mean difference tells the two apart)
        assert abs(a[0][2] - b[0][2]) < 1e-2 * abs(b[0][2]), (a[0], b[0])
        assert float((a[1] - b[1]).abs().mean()) < 1e-4, float((a[1] - b[1]).abs().mean())
        return
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1])


def test_backward_under_a_cu_budget_changes_no_bit(dev):
    """data parallel: backward runs beside the RCCL channels of the bucket all-reduces and chooses its GEMM tile widths by rounds of workgroups over
    the CUs that are left (amdseg_ctx_set_cu_budget on the engine's context): another tile width is another launch geometry, not another summation order -- bert-base 4 x 512,
    three steps, the same bits in every parameter with 240 of 256 CUs budgeted"""
    a = _three_steps(dev, "bf16", True, "bert_base_L512", bwd_cu_budget=240)
    b = _three_steps(dev, "bf16", True, "bert_base_L512")
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1])


@pytest.mark.parametrize("case", ["tiny_L128", "bert_base_L512"])
def test_keepmasks_from_the_layer_norm_launch_change_no_bit(dev, case):
    """ABI 14: layer i + 1's attention-dropout keep masks are written by workgroups of layer i's second LayerNorm launch (acts.keep_next /
    keep_ready, amdseg_add_ln_fwd_keepmask) instead of a launch of their own: same generator, same seeds, same rows -- three training steps
    end in the same bits"""
    a = _three_steps(dev, "bf16", True, case, keepmask_in_ln=True)
    b = _three_steps(dev, "bf16", True, case, keepmask_in_ln=False)
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1])


def test_keepmask_pairing_launch_counts(dev):
    """what the pairing does to a bert-base step at 4 x 512 (x 2 with the augmented half): ONE keep-mask launch (layer 0's) instead of twelve, the same
    number of LayerNorm launches; unpaired, twelve.  Counted by the launch timer of the engine's context (AMDSEG_PROF_KEEPMASK = 8, _ADD_LN_FWD = 5)."""
    from tests.test_gpu_fullsize import _fullsize_case
    z, sd, batch, arch, fl = _fullsize_case()
    counts = {}
    for flag in (True, False):
        m = build_model(arch, fl(z, "train_full"), sd, dev, dropout=0.1)
        m.train()
        m.amdseg_seed = 3
        eng = m.engine()
        eng.keepmask_in_ln = flag
        b = to_dev(batch, dev)
        random.seed(1)
        m(**b)[0].backward()                               # (arenas built, graphs or not: the second step is the one counted)
        eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
        random.seed(1)
        m(**b)[0].backward()
        torch.cuda.synchronize()
        eng.ctx.prof_enable(False)
        counts[flag] = (eng.ctx.prof_read(8)[2], eng.ctx.prof_read(5)[2])
    nl = 12
    nfwd = counts[False][0] // nl                          # encoder passes per step
    assert nfwd >= 1 and counts[False][0] == nfwd * nl and counts[True][0] == nfwd, counts
    assert counts[True][1] == counts[False][1] == 2 * nl * nfwd, counts
