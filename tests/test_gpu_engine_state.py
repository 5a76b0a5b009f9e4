"""State-tracking guarantees of the host engine (round-1 advisor findings): the bf16 compute copies follow EVERY in-place write to the
fp32 masters (stock torch optimisers, load_state_dict on a live engine), outputs and saved activations of different calls never alias,
and the parameter / gradient re-homing is checked for every parameter, not a sentinel."""
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


def _oracle_logits(m, arch, flags, batch, seed):
    from oracle import bert_ts_oracle as O
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    cfg = O.make_cfg(num_labels=2, **arch, **flags)
    random.seed(seed)
    with torch.no_grad():
        _, lg, _ = O.model_forward(sd, cfg, batch)
    return lg


def test_torch_optimizer_step_is_seen_by_the_bf16_weights(dev):
    """one stock torch optimiser step with a large lr on ONE encoder matrix: the next forward must compute with the updated matrix
    (oracle on the updated state dict), not with the bf16 copy made before the step"""
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "full_eval")
    m = build_model(arch, flags, sd, dev).train()
    b = to_dev(batch, dev)
    w = m.bert.encoder.layer[0].intermediate.dense.weight
    opt = torch.optim.SGD([w], lr=30.0)
    random.seed(3)
    loss = m(**b)[0]
    loss.backward()
    before = w.detach().clone()
    opt.step()
    assert (w.detach() - before).abs().max().item() > 1e-2
    m.eval()
    random.seed(3)
    with torch.no_grad():
        _, lg, _ = m(**b)
    new = _oracle_logits(m, arch, flags, batch, 3)
    old = torch.from_numpy(z["full_eval.logits"])
    d_new = (lg.cpu() - new).abs().max().item()
    d_old = (lg.cpu() - old).abs().max().item()
    print(f"after the step: vs oracle(updated weights) {d_new:.4f}, vs logits of the initial weights {d_old:.4f}")
    assert d_new < 0.08 and d_old > 4 * d_new


def test_adamw_steps_then_oracle_on_state_dict(dev):
    """N steps of the HF-Trainer-like loop with torch.optim.AdamW (fused / foreach variants write through the Parameters), then the
    CPU oracle on model.state_dict(): logits must agree within the bf16 tolerance"""
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    m = build_model(arch, flags, sd, dev).train()
    b = to_dev(batch, dev)
    opt = torch.optim.AdamW(m.parameters(), lr=3e-3)
    for _ in range(5):
        random.seed(0)
        m(**b)[0].backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        m.zero_grad()
    m.eval()
    random.seed(1)
    with torch.no_grad():
        _, lg, _ = m(**b)
    ref = _oracle_logits(m, arch, flags, batch, 1)
    d = (lg.cpu() - ref).abs().max().item()
    moved = (ref - torch.from_numpy(z["full_eval.logits"])).abs().max().item()
    print(f"5 AdamW steps: max|dlogit| vs oracle on the trained state dict {d:.4f} (weights moved the logits by {moved:.3f})")
    assert d < 0.08 and moved > 0.3


def test_load_state_dict_on_a_live_engine(dev):
    """load_best_model_at_end / resume: load_state_dict AFTER the engine was built must change what the kernels compute"""
    from tests.util import tiny_state_dict
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "full_eval")
    m = build_model(arch, flags, sd, dev).eval()
    b = to_dev(batch, dev)
    with torch.no_grad():
        random.seed(2)
        m(**b)
    sd2 = tiny_state_dict(arch, seed=99)
    m.load_state_dict(sd2, strict=False)
    assert m.engine().fp.intact()
    with torch.no_grad():
        random.seed(2)
        _, lg, _ = m(**b)
    ref = _oracle_logits(m, arch, flags, batch, 2)
    assert (lg.cpu() - ref).abs().max().item() < 0.08
    assert (ref - torch.from_numpy(z["full_eval.logits"])).abs().max().item() > 0.3


def test_outputs_of_two_calls_do_not_alias(dev):
    from spokennlp_amd.mmvts_text_encoder import TextEncoder  # noqa: F401  (the class that hands the encoder output to the caller)
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "plain_eval"), sd, dev).eval()
    ids = batch["input_ids"][:, 0].to(dev)
    am = batch["attention_mask"][:, 0].to(dev)
    tt = batch["token_type_ids"][:, 0].to(dev)
    with torch.no_grad():
        a = m.encode(ids, am, tt)
        keep = a.clone()
        ids2 = torch.roll(ids, 1, 0)
        b = m.encode(ids2, torch.roll(am, 1, 0), tt)
    assert a.data_ptr() != b.data_ptr()
    assert torch.equal(a, keep) and not torch.equal(a, b)


def test_two_training_forwards_before_backward(dev):
    """siamese use: two encodes at the same shape, ONE backward through both -- gradients = sum of the separate runs"""
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "plain_eval")
    b = to_dev(batch, dev)
    b2 = {k: torch.flip(v, (0,)) for k, v in b.items()}

    def grads(model):
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    m = build_model(arch, flags, sd, dev).train()
    l1 = m(**b)[0]
    l2 = m(**b2)[0]
    (l1 + l2).backward()
    both = grads(m)
    m.zero_grad()
    m(**b)[0].backward()
    m(**b2)[0].backward()
    sep = grads(m)
    for n in sep:
        assert torch.allclose(both[n], sep[n], rtol=1e-3, atol=1e-5), n
    # with a single arena the first call's saved activations are gone: its backward must raise, not produce wrong gradients
    m.zero_grad()
    eng = m.engine()
    eng.max_live_arenas = 1
    for A in eng._arenas.values():
        A["busy"] = False
    l1 = m(**b)[0]
    l2 = m(**b2)[0]
    l2.backward()
    from spokennlp_amd.lib import AmdsegError
    with pytest.raises(AmdsegError, match="overwritten"):
        l1.backward()
    eng.max_live_arenas = 2


def test_dropped_graph_frees_its_arena(dev):
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "plain_eval"), sd, dev).train()
    b = to_dev(batch, dev)
    for _ in range(4):                       # train-mode forwards whose graphs are dropped without a backward
        loss = m(**b)[0]
        del loss
    eng = m.engine()
    assert sum(1 for k in eng._arenas if k[2]) == 1


def test_subset_optimizer_and_replaced_parameter(dev):
    """(a) an optimiser over a SUBSET of the parameters + zero_grad(set_to_none=True): the other parameters' gradients stay attached to
    the flat buffer and every gradient is re-attached where it was dropped; (b) replacing a parameter that is not the first one
    (re-initialised head) is detected and the engine rebuilt"""
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "plain_eval"), sd, dev).train()
    b = to_dev(batch, dev)
    sub = [p for n, p in m.named_parameters() if "embeddings" not in n]
    opt = torch.optim.SGD(sub, lr=1e-3)
    for _ in range(2):
        m(**b)[0].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    m(**b)[0].backward()
    eng = m.engine()
    for p, _, gptr, n in eng.fp._ptrs:
        if p.requires_grad and ".pooler." not in n:
            assert p.grad is not None and p.grad.data_ptr() == gptr, n
    w = m.bert.encoder.layer[1].output.dense.weight
    assert float(w.grad.abs().sum()) > 0
    m.zero_grad()
    old = m.engine()
    clf = m.loss_calculator.classifier
    clf.weight.data = torch.randn_like(clf.weight) * 0.3           # `p.data = new` on a parameter in the middle of the list
    assert not old.fp.intact()
    new = m.engine()
    assert new is not old and new.fp.intact()
    with torch.no_grad():
        m.eval()(**b)


@pytest.mark.parametrize("kw", [dict(fused=True), dict(foreach=True), dict(foreach=False)])
def test_every_torch_adamw_flavour_refreshes_the_bf16_weights(dev, kw):
    """torch's FUSED AdamW (HF Trainer's default, adamw_torch_fused) writes the parameters without bumping their version counters:
    the engine must still compute with the updated matrices (caught by the Trainer-vs-fused-Trainer comparison of round 2)"""
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    m = build_model(arch, flags, sd, dev).train()
    b = to_dev(batch, dev)
    opt = torch.optim.AdamW(m.parameters(), lr=5e-3, **kw)
    for _ in range(3):
        random.seed(0)
        m(**b)[0].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    m.eval()
    random.seed(1)
    with torch.no_grad():
        _, lg, _ = m(**b)
    ref = _oracle_logits(m, arch, flags, batch, 1)
    d = (lg.cpu() - ref).abs().max().item()
    assert d < 0.08, (kw, d)


@pytest.mark.parametrize("precision", ["bf16", "parity"])
def test_raw_master_write_between_two_eval_forwards_is_seen(dev, precision):
    """(round-2 verdict, robustness 12) `p.data.copy_(...)` bumps no version counter and runs no optimiser hook: the second inference forward
    finds the write through the device-side content checksum (amdseg_weights_changed) and re-derives its compute copies -- no
    mark_weights_dirty().  Unchanged weights: the conditional refresh is a no-op and the output is bit-identical."""
    from tests.test_oracle_golden import load_case, flags_of
    from tests.test_gpu_model import build_model, to_dev
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "full_eval"), sd, dev).eval()
    m.config.amdseg_precision = precision
    b = to_dev(batch, dev)
    with torch.no_grad():
        random.seed(0); a0 = m(**b)[1].clone()
        random.seed(0); a1 = m(**b)[1].clone()
        assert torch.equal(a0, a1)
        w = dict(m.named_parameters())["bert.encoder.layer.1.intermediate.dense.weight"]
        v0 = w._version
        w.data.copy_(w.data * 1.5 + 0.01)                   # behind every detector's back ...
        assert m.engine()._weights_version() is not None
        random.seed(0); a2 = m(**b)[1].clone()
        assert (a2 - a0).abs().max().item() > 1e-3          # ... and the forward runs on the new weights all the same
        # the reference answer for the new weights: a fresh model built from the modified state dict
        sd2 = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
        m2 = build_model(arch, flags_of(z, "full_eval"), sd2, dev).eval()
        m2.config.amdseg_precision = precision
        random.seed(0); a3 = m2(**b)[1]
        assert torch.equal(a2, a3)
        random.seed(0); a4 = m(**b)[1]
        assert torch.equal(a4, a2)


def test_checksum_state_follows_what_the_copies_were_derived_from(dev):
    """ the two ways the content checksum could serve stale weights:
    (a) eval (checksum S0 stored) -> load_state_dict (seen by the version counters; copies refreshed, checksum untouched) -> raw write BACK to
        the old weights -> eval: the content equals S0 again, yet the copies hold the load_state_dict weights;
    (b) eval bf16 -> raw write -> eval bf16 (bf16 copies refreshed, checksum updated) -> eval in "parity" precision: checksum unchanged, yet
        the split weight images are still the old ones.
    Both must run on the weights as they are."""
    from tests.test_oracle_golden import load_case, flags_of
    from tests.test_gpu_model import build_model, to_dev
    z, sd, batch, arch = load_case("tiny_L64")
    b = to_dev(batch, dev)
    name = "bert.encoder.layer.0.attention.output.dense.weight"

    def fresh(sd_, precision):
        m_ = build_model(arch, flags_of(z, "full_eval"), sd_, dev).eval()
        m_.config.amdseg_precision = precision
        return m_

    with torch.no_grad():
        # (a)
        m = fresh(sd, "bf16")
        random.seed(0); a0 = m(**b)[1].clone()
        sd_new = {k: (v * 1.25 + 0.02 if k == name else v.clone()) for k, v in sd.items()}
        m.load_state_dict(sd_new, strict=False)
        random.seed(0); a1 = m(**b)[1].clone()
        assert (a1 - a0).abs().max().item() > 1e-3
        w = dict(m.named_parameters())[name]
        w.data.copy_(sd[name].to(dev))                      # raw write back to the ORIGINAL content
        random.seed(0); a2 = m(**b)[1].clone()
        assert torch.equal(a2, a0)
        # (b)
        m = fresh(sd, "bf16")
        m.config.amdseg_precision = "parity"
        random.seed(0); p0 = m(**b)[1].clone()              # parity images built from the original weights
        m.config.amdseg_precision = "bf16"
        random.seed(0); m(**b)
        w = dict(m.named_parameters())[name]
        w.data.copy_(w.data * 1.25 + 0.02)
        random.seed(0); m(**b)                              # bf16 forward sees the write and refreshes ITS copies
        m.config.amdseg_precision = "parity"
        random.seed(0); p1 = m(**b)[1].clone()
        sd2 = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
        random.seed(0); p_ref = fresh(sd2, "parity")(**b)[1]
        assert (p1 - p0).abs().max().item() > 1e-3 and torch.equal(p1, p_ref)


def test_eval_forward_as_a_hipgraph_is_bit_identical_and_follows_the_weights(dev):
    """opt-in (AMDSEG_EVAL_GRAPHS=1): small-batch inference replays the encoder as ONE hipGraph from the third call at a shape on
    (engine._forward_graphed).  The replay must give the eager result bit for bit, see new inputs, see weights changed by an optimiser step or a raw
    write (the refresh of the bf16 copies stays outside the graph), and step aside for what it does not cover."""
    from spokennlp_amd import engine as E
    from tests.test_oracle_golden import load_case, flags_of
    from tests.test_gpu_model import build_model, to_dev
    z, sd, batch, arch = load_case("tiny_L128")
    fl = flags_of(z, "full_eval")
    m = build_model(arch, fl, sd, dev).eval()
    ref = build_model(arch, fl, sd, dev).eval()
    ref.engine().eval_graphs = False
    eng = m.engine()
    eng.eval_graphs = True                                   # (opt-in: AMDSEG_EVAL_GRAPHS=1)
    b1 = to_dev(batch, dev)
    b2 = {k: v.clone() for k, v in b1.items()}
    b2["input_ids"] = torch.where(b2["attention_mask"] > 0, (b2["input_ids"] * 7 + 3) % (arch["vocab_size"] - 5) + 4, b2["input_ids"])
    with torch.no_grad():
        for it in range(5):
            for b in (b1, b2):
                random.seed(5); got = m(**b)
                random.seed(5); want = ref(**b)
                assert torch.equal(got[1], want[1]) and torch.equal(got[0], want[0]), it
        assert any(e["graph"] is not None for e in eng._eval_graphs.values())
        # weights changed behind the graph's back: raw write to a master + an optimiser-style in-place update
        for mm in (m, ref):
            mm.bert.encoder.layer[0].attention.self.query.weight.data.mul_(1.5)
            mm.bert.embeddings.word_embeddings.weight.data.add_(0.01)
        random.seed(5); got = m(**b1)
        random.seed(5); want = ref(**b1)
        assert torch.equal(got[1], want[1])
        # pass-throughs and the armed launch timer take the eager path; results stay the same
        hs = m(**b1, output_hidden_states=True)
        assert torch.equal(hs[1], want[1]) and len(hs[3]) == arch["num_hidden_layers"] + 1
        E.GRAPHS_SUSPENDED = True
        try:
            assert torch.equal(m(**b1)[1], want[1])
        finally:
            E.GRAPHS_SUSPENDED = False
    # a training step after graphed inference, then inference again
    m.train()
    random.seed(1); loss = m(**b1)[0]; loss.backward(); m.engine().adamw_step(1e-3)
    ref.train()
    random.seed(1); loss_r = ref(**b1)[0]; loss_r.backward(); ref.engine().adamw_step(1e-3)
    m.eval(); ref.eval()
    with torch.no_grad():
        random.seed(5); got = m(**b1)
        random.seed(5); want = ref(**b1)
    assert torch.equal(got[1], want[1])
