"""The engine's fast path behind the reference's training surface (`spokennlp_amd.trainer.Trainer`, the one-line import swap of
ts_sentence_seq_labeling.py:43): fused AdamW + HIP gradient norm under the real `transformers.Trainer` loop, checkpoint / resume of
the flat optimiser state, and the engine's own data parallelism with gradient accumulation (run_finetune.sh:31,61,76)."""
import math
import os
import random
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from tests.test_oracle_golden import load_case, flags_of  # noqa: E402
from tests.test_gpu_model import build_model, to_dev  # noqa: E402


def _samples(arch, n=16, L=64):
    from spokennlp_amd import data
    docs = data.synth_docs(24, seed=11, vocab=arch["vocab_size"], mean_sents=10, sd_sents=3, mean_boundaries=2, mu_tok=1.4, sigma_tok=0.3)
    batches = data.batches_from_docs(docs, L, 1, seed=4)
    return [{k: v[0] for k, v in b.items()} for b in batches][:n]


class _DS(torch.utils.data.Dataset):
    def __init__(self, samples):
        self.s = samples

    def __len__(self):
        return len(self.s)

    def __getitem__(self, i):
        return self.s[i]


def _args(tmp, **kw):
    from transformers import TrainingArguments
    base = dict(output_dir=str(tmp), per_device_train_batch_size=4, per_device_eval_batch_size=4, max_steps=4, learning_rate=1e-3,
                lr_scheduler_type="linear", max_grad_norm=1.0, gradient_accumulation_steps=2, report_to=[], save_strategy="no",
                logging_steps=1, seed=7, dataloader_drop_last=True, remove_unused_columns=True, weight_decay=0.01)
    base.update(kw)
    return TrainingArguments(**base)


def test_fused_trainer_matches_stock_trainer(dev, tmp_path):
    """same data order, dropout 0: after 4 optimiser steps (8 micro-batches, clip 1.0, weight decay on the decay group only, linear lr)
    the fused path's weights equal the stock Trainer's (torch AdamW + clip_grad_norm_) to fp32 round-off"""
    from transformers import Trainer as HFTrainer, default_data_collator
    from spokennlp_amd.trainer import AmdsegFusedAdamW, Trainer
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    ds = _DS(_samples(arch))
    res = {}
    for name, cls in (("stock", HFTrainer), ("fused", Trainer)):
        m = build_model(arch, flags, sd, dev)
        random.seed(3)
        tr = cls(model=m, args=_args(tmp_path / name), train_dataset=ds, data_collator=default_data_collator)
        out = tr.train()
        assert out.global_step == 4 and math.isfinite(out.training_loss)
        if name == "fused":
            assert isinstance(tr._fused(), AmdsegFusedAdamW)
            norms = [h["grad_norm"] for h in tr.state.log_history if "grad_norm" in h]
            assert len(norms) == 4 and all(math.isfinite(g) and g > 0 for g in norms)
            res["fused_norms"] = norms
        else:
            res["stock_norms"] = [h["grad_norm"] for h in tr.state.log_history if "grad_norm" in h]
        res[name] = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
        res[name + "_loss"] = out.training_loss
    # the first optimiser step is identical to fp32 round-off (tools/dbg: 1e-7); later steps see Adam amplify bf16-level gradient noise
    # (m / sqrt(v) is +-1 for a parameter whose true gradient is ~0, e.g. key.bias), so the comparison is on the UPDATE as a whole
    assert abs(res["stock_loss"] - res["fused_loss"]) < 0.02 * abs(res["stock_loss"])
    for a, b in zip(res["stock_norms"], res["fused_norms"]):
        assert abs(a - b) <= 0.03 * max(1.0, abs(a)), (res["stock_norms"], res["fused_norms"])
    worst = 0.0
    for k, v in res["stock"].items():
        if "pooler" in k or "position_ids" in k or k not in sd or "key.bias" in k:
            continue
        ds, df = v - sd[k], res["fused"][k] - sd[k]
        if float(ds.norm()) < 1e-6:
            continue
        rel = float((ds - df).norm() / ds.norm())
        worst = max(worst, rel)
        assert rel < 0.15, (k, rel)
    print("fused vs stock Trainer: worst relative difference of the 4-step update =", worst)


def test_fused_optimizer_checkpoint_resume(dev, tmp_path):
    """Trainer checkpoints (optimizer.pt with the flat AdamW moments) -> resume_from_checkpoint continues exactly"""
    from transformers import default_data_collator
    from spokennlp_amd.trainer import Trainer
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    ds = _DS(_samples(arch))
    m = build_model(arch, flags, sd, dev)
    random.seed(3)
    tr = Trainer(model=m, args=_args(tmp_path / "a", max_steps=4, save_strategy="steps", save_steps=2), train_dataset=ds,
                 data_collator=default_data_collator)
    tr.train()
    full = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    ck = os.path.join(str(tmp_path / "a"), "checkpoint-2")
    assert os.path.exists(os.path.join(ck, "optimizer.pt"))
    m2 = build_model(arch, flags, sd, dev)
    random.seed(3)
    tr2 = Trainer(model=m2, args=_args(tmp_path / "a", max_steps=4, save_strategy="steps", save_steps=2), train_dataset=ds,
                  data_collator=default_data_collator)
    tr2.train(resume_from_checkpoint=ck)
    assert tr2.state.global_step == 4
    # the row gradients of the CSSL / TSSP heads are accumulated with fp32 atomics (order varies run to run at the 1e-7 level) and Adam turns
    # a sign flip of a ~zero gradient into a +-lr step, so the comparison is on the whole 4-step update, as in the test above
    for k, v in full.items():
        if "pooler" in k or "position_ids" in k or k not in sd or "key.bias" in k:
            continue
        du, dr = v - sd[k], m2.state_dict()[k].float().cpu() - sd[k]
        if float(du.norm()) < 1e-6:
            continue
        assert float((du - dr).norm() / du.norm()) < 0.1, k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        z, sd, batch, arch = load_case("tiny_L64")
        if rank == 1:                                # ranks start from DIFFERENT weights: enable_data_parallel must broadcast rank 0's
            sd = {k: v + 0.01 for k, v in sd.items()}
        m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
        eng = m.engine()
        assert eng.enable_data_parallel()
        samples = _samples(arch, 8)
        micro = [samples[rank * 4 + 0:rank * 4 + 2], samples[rank * 4 + 2:rank * 4 + 4]]
        for i, mb in enumerate(micro):
            b = {k: torch.stack([s[k] for s in mb]).to(dev) for k in mb[0]}
            random.seed(50 + 10 * rank + i)
            ctx = m.no_sync() if i == 0 else __import__("contextlib").nullcontext()
            with ctx:
                m(**b)[0].backward()
        eng.finish_grad_sync()
        assert eng.buckets.norm_is_complete()                     # sum of squares accumulated bucket by bucket behind the reductions
        norm, _ = eng.grad_norm_and_clip_coef(1.0, 0.5)
        torch.cuda.synchronize()
        want = 0.5 * float(eng.fp.flat_g.double().norm())
        assert abs(float(norm) - want) <= 1e-4 * want, (float(norm), want)
        torch.save(dict(g=eng.fp.flat_g.cpu(), p=eng.fp.flat_p.cpu()), os.path.join(out_dir, f"dp{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bf16_embed", ["0", "1"])
def test_native_dp_gradient_accumulation_two_ranks(dev, tmp_path, bf16_embed, monkeypatch):
    """2 ranks x 2 micro-steps under the engine's own bucketed exchange: every rank ends with sum over ranks and micro-steps of the
    single-process gradients -- each bucket reduced ONCE (the round-1 code reduced the accumulating buffer in every backward)"""
    import torch.multiprocessing as mp
    z, sd, batch, arch = load_case("tiny_L64")
    samples = _samples(arch, 8)
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    for rank in range(2):
        for i in range(2):
            mb = samples[rank * 4 + 2 * i:rank * 4 + 2 * i + 2]
            b = {k: torch.stack([s[k] for s in mb]).to(dev) for k in mb[0]}
            random.seed(50 + 10 * rank + i)
            m(**b)[0].backward()
    want = m.engine().fp.flat_g.cpu().clone()
    p0 = m.engine().fp.flat_p.cpu().clone()
    monkeypatch.setenv("AMDSEG_DP_BF16_EMBED", bf16_embed)        # "1": the word-embedding bucket travels in bf16 (opt-in)
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = torch.load(tmp_path / f"dp{rank}.pt")
        assert torch.equal(got["p"], p0), "parameters were not broadcast from rank 0"
        d = (got["g"] - want).abs().max().item()
        tol = (1e-4 if bf16_embed == "0" else 2e-2) * max(1.0, want.abs().max().item())
        assert d <= tol, (rank, d)


def _trainer_dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      ACCELERATE_TORCH_DEVICE="cuda:0")
    import torch.distributed as dist
    from transformers import default_data_collator
    from spokennlp_amd.trainer import NativeDataParallel, Trainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        z, sd, batch, arch = load_case("tiny_L64")
        m = build_model(arch, flags_of(z, "train_full"), sd, dev)
        ds = _DS(_samples(arch, 16))
        random.seed(3)
        tr = Trainer(model=m, args=_args(os.path.join(out_dir, f"r{rank}"), per_device_train_batch_size=2, max_steps=2, ddp_backend="gloo"),
                     train_dataset=ds, data_collator=default_data_collator)
        tr.train()
        assert isinstance(tr.model_wrapped, NativeDataParallel) and m.engine().buckets is not None
        torch.save({k: v.detach().float().cpu() for k, v in m.state_dict().items()}, os.path.join(out_dir, f"tr{rank}.pt"))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_fused_trainer_native_dp_two_ranks(dev, tmp_path):
    """the subclassed Trainer on 2 ranks (one GPU, gloo): no torch DDP wrapper, the engine's buckets + no_sync under gradient
    accumulation; both ranks end with identical weights, equal to the single-process run over the same global batches"""
    import torch.multiprocessing as mp
    from transformers import default_data_collator
    from spokennlp_amd.trainer import Trainer
    mp.spawn(_trainer_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "tr0.pt"), torch.load(tmp_path / "tr1.pt")
    z, sd, batch, arch = load_case("tiny_L64")
    moved = 0.0
    for k in a:
        assert torch.equal(a[k], b[k]), k
        if k in sd:
            moved = max(moved, (a[k] - sd[k]).abs().max().item())
    assert moved > 1e-4


def test_nan_inf_filter_runs_on_the_device_and_logs_like_the_stock_loop(dev, tmp_path):
    """`logging_nan_inf_filter=True` (the TrainingArguments default): the subclass substitutes non-finite step losses on the device
    (no host read per step) and hands the arguments back unchanged; with a NaN injected from the 3rd micro-batch on, every logged
    loss equals the stock loop's"""
    from transformers import Trainer as HFTrainer, default_data_collator
    from spokennlp_amd.trainer import Trainer
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    ds = _DS(_samples(arch))
    logs = {}
    for name, base in (("stock", HFTrainer), ("fused", Trainer)):
        class Inject(base):
            calls = 0
            seen_flag = []

            def compute_loss(self, model, inputs, *a, **kw):
                out = super().compute_loss(model, inputs, *a, **kw)
                type(self).calls += 1
                type(self).seen_flag.append(self.args.logging_nan_inf_filter)
                if type(self).calls >= 3:
                    out = (out[0] * float("nan"),) + tuple(out[1:]) if isinstance(out, tuple) else out * float("nan")
                return out

        m = build_model(arch, flags, sd, dev)
        random.seed(3)
        args = _args(tmp_path / name)
        assert args.logging_nan_inf_filter
        tr = Inject(model=m, args=args, train_dataset=ds, data_collator=default_data_collator)
        out = tr.train()
        assert args.logging_nan_inf_filter                       # handed back unchanged
        if name == "fused":
            assert Inject.seen_flag[0] and not any(Inject.seen_flag[1:])      # switched off inside the first training step
        logs[name] = [h["loss"] for h in tr.state.log_history if "loss" in h]
        logs[name + "_final"] = out.training_loss
    assert len(logs["stock"]) == len(logs["fused"]) == 4
    for a, b in zip(logs["stock"], logs["fused"]):
        assert math.isfinite(a) and math.isfinite(b) and abs(a - b) <= 2e-3 * max(1.0, abs(a)), logs
    assert abs(logs["stock_final"] - logs["fused_final"]) <= 2e-3 * max(1.0, abs(logs["stock_final"]))


def test_trainer_interleaved_evaluation_and_checkpoint(dev, tmp_path):
    """the subclass under the loop shapes a real run has: evaluation every 2 steps (the eval dataloader keeps accelerate's device placement,
    the train batches take the side-stream copy), a checkpoint in between, gradient accumulation 2, logging with the device-side NaN filter"""
    import json
    from transformers import default_data_collator
    from spokennlp_amd.trainer import Trainer
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    samples = _samples(arch, n=24)
    m = build_model(arch, flags, sd, dev)
    random.seed(3)
    args = _args(tmp_path, max_steps=4, eval_strategy="steps", eval_steps=2, save_strategy="steps", save_steps=2, save_total_limit=1,
                 per_device_eval_batch_size=4)
    seen = {}

    def metrics(p):
        seen["n"] = seen.get("n", 0) + 1
        return {"n_pred": float(len(p.predictions[0]))}

    tr = Trainer(model=m, args=args, train_dataset=_DS(samples[:16]), eval_dataset=_DS(samples[16:]), data_collator=default_data_collator,
                 compute_metrics=metrics)
    out = tr.train()
    assert out.global_step == 4 and math.isfinite(out.training_loss)
    evals = [h for h in tr.state.log_history if "eval_loss" in h]
    assert len(evals) == 2 and all(math.isfinite(h["eval_loss"]) for h in evals) and seen["n"] == 2
    assert getattr(tr, "_amdseg_side_copy", False)                     # the train batches went through the copy stream
    ck = [d for d in os.listdir(tmp_path) if d.startswith("checkpoint-")]
    assert ck == ["checkpoint-4"]
    st = json.load(open(tmp_path / ck[0] / "trainer_state.json"))
    assert st["global_step"] == 4
    # the saved weights are the live ones (the engine re-homed the parameters into its flat buffer long before the save)
    from safetensors.torch import load_file
    saved = load_file(str(tmp_path / ck[0] / "model.safetensors"))
    live = m.state_dict()
    k = "bert.encoder.layer.0.attention.self.query.weight"
    assert torch.equal(saved[k].cpu(), live[k].detach().cpu())


def test_trainer_hands_the_host_labels_to_the_model_no_copy_back(dev, tmp_path, monkeypatch):
    """`Trainer._prepare_inputs` uploads the pinned batch itself and leaves the host originals with the model (`amdseg_set_host_twins`): the
    heads build their index lists from those instead of copying the labels back and waiting for that copy (= for the previous step's GPU
    work).  Same seeds: the logged losses agree with and without the hand-over, and with it no forward waits on an event."""
    from transformers import default_data_collator
    from spokennlp_amd.trainer import Trainer
    from spokennlp_amd.bert_for_ts import TopicSegHeadsMixin
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    ds = _DS(_samples(arch))
    waits = {"n": 0}
    real_sync = torch.cuda.Event.synchronize

    def counting_sync(self):
        waits["n"] += 1
        return real_sync(self)
    monkeypatch.setattr(torch.cuda.Event, "synchronize", counting_sync)
    logs = {}
    for mode in ("twins", "copy_back"):
        if mode == "copy_back":
            monkeypatch.setattr(TopicSegHeadsMixin, "amdseg_set_host_twins", lambda self, pairs: None)
        m = build_model(arch, flags, sd, dev, dropout=0.1)
        m.amdseg_seed = 5
        random.seed(3)
        tr = Trainer(model=m, args=_args(tmp_path / mode, dataloader_pin_memory=True, gradient_accumulation_steps=1, max_steps=6),
                     train_dataset=ds, data_collator=default_data_collator)
        waits["n"] = 0
        tr.train()
        logs[mode] = ([h["loss"] for h in tr.state.log_history if "loss" in h], waits["n"])
        assert getattr(tr, "_amdseg_side_copy", False)
    assert len(logs["twins"][0]) == 6 and logs["twins"][0][0] == logs["copy_back"][0][0]        # first step: the same bits
    for a, b in zip(logs["twins"][0], logs["copy_back"][0]):                                      # later: the atomic scatters' run-to-run noise
        assert abs(a - b) <= 2e-3 * abs(b), logs                                                  # (Adam at lr 1e-3 amplifies it step by step)
    assert logs["copy_back"][1] >= 6                         # one wait per forward without the hand-over ...
    assert logs["twins"][1] <= logs["copy_back"][1] - 6      # ... none with it


def _train_step(m, batch, dev):
    random.seed(0)
    loss = m(**to_dev(batch, dev))[0]
    loss.backward()
    return loss


def test_fused_adamw_refuses_param_groups_it_cannot_honour(dev):
    """ONE lr / betas / eps for the flat pass: groups that differ in them raise instead of silently collapsing to the first group's values
    (layer-wise lr decay, a head with its own lr); decay / no-decay groups -- what HF's create_optimizer builds -- become the decay flag"""
    from spokennlp_amd import lib as L
    from spokennlp_amd.trainer import AmdsegFusedAdamW
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    named = list(m.named_parameters())
    head = [p for n, p in named if n.startswith("classifier") or "loss_calculator" in n]
    body = [p for n, p in named if not (n.startswith("classifier") or "loss_calculator" in n)]
    with pytest.raises(L.AmdsegError, match="ONE lr"):
        AmdsegFusedAdamW(m, param_groups=[dict(params=body, lr=1e-5), dict(params=head, lr=1e-3)])
    with pytest.raises(L.AmdsegError, match="ONE eps"):
        AmdsegFusedAdamW(m, param_groups=[dict(params=body, eps=1e-8), dict(params=head, eps=1e-6)])
    with pytest.raises(L.AmdsegError, match="one weight-decay value"):
        AmdsegFusedAdamW(m, param_groups=[dict(params=body, weight_decay=0.1), dict(params=head, weight_decay=0.01)])
    decay = [p for n, p in named if p.dim() >= 2]
    nodecay = [p for n, p in named if p.dim() < 2]
    opt = AmdsegFusedAdamW(m, lr=1e-3, param_groups=[dict(params=decay, weight_decay=0.01), dict(params=nodecay, weight_decay=0.0)])
    assert opt.param_groups[0]["weight_decay"] == 0.01 and opt.decay_names == {n for n, p in named if p.dim() >= 2}
    with pytest.raises(L.AmdsegError, match="one group"):
        opt.add_param_group(dict(params=[torch.nn.Parameter(torch.zeros(3, device=dev))]))
    # ... and the flags do what torch's two groups do
    ref = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    rn = dict(ref.named_parameters())
    topt = torch.optim.AdamW([dict(params=[rn[n] for n, p in named if p.dim() >= 2], weight_decay=0.01),
                              dict(params=[rn[n] for n, p in named if p.dim() < 2], weight_decay=0.0)], lr=1e-3)
    for _ in range(2):
        _train_step(m, batch, dev); opt.step(); opt.zero_grad()
        _train_step(ref, batch, dev); topt.step(); topt.zero_grad()
    for n, p in named:                                      # two steps of lr 1e-3: an element moves by up to 2e-3; the two bf16 runs differ in
        if "pooler" not in n and "key.bias" not in n:       # near-zero gradient elements (atomics order), where m / sqrt(v) amplifies noise
            d = (p.detach() - rn[n].detach()).abs()        # (the key bias has a zero gradient by construction: pure noise through Adam)
            assert float(d.max()) < 1e-3 and float(d.mean()) < 2e-5, n
    # without the decay flag the decayed matrices would differ by lr * wd * |w| * steps ~ 2e-5 * |w| systematically: check the flag bit itself
    eng = m.engine()
    flags = eng._chunk_flags.cpu()
    for n, p in named:
        o = eng.fp.offsets[n] // 64
        assert int(flags[o]) & 1 == (1 if p.dim() >= 2 else 0), n


def test_fused_adamw_follows_requires_grad_and_survives_an_engine_rebuild(dev):
    """(round-2 advisor) the frozen / decay flags follow the CURRENT requires_grad pattern, and the Adam moments + step count move to the new
    engine when the model re-homes its parameters (`p.data = new`, model.to) with an unchanged layout"""
    from spokennlp_amd.trainer import AmdsegFusedAdamW
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    w = dict(m.named_parameters())["bert.encoder.layer.0.intermediate.dense.weight"]
    w.requires_grad_(False)
    opt = AmdsegFusedAdamW(m, lr=1e-2, max_grad_norm=1.0)
    w0 = w.detach().clone()
    _train_step(m, batch, dev); opt.step(); opt.zero_grad()
    assert torch.equal(w.detach(), w0)                      # frozen: untouched
    w.requires_grad_(True)                                  # unfrozen mid-training: must start to move
    _train_step(m, batch, dev); opt.step(); opt.zero_grad()
    assert not torch.equal(w.detach(), w0)
    eng0 = m.engine()
    step0, m0 = eng0.opt_step, eng0.adam_m.clone()
    assert step0 == 2
    # the user replaces a parameter's storage: the model rebuilds its engine on the next forward
    b = dict(m.named_parameters())["bert.encoder.layer.1.output.dense.bias"]
    b.data = b.data.clone()
    _train_step(m, batch, dev)
    eng1 = m.engine()
    assert eng1 is not eng0
    opt.step(); opt.zero_grad()
    assert eng1.opt_step == step0 + 1                       # bias correction did not restart
    assert float((eng1.adam_m - m0).abs().max()) > 0 and float(eng1.adam_m.abs().max()) > 0


def test_grad_norm_twice_in_one_step_reduces_the_tail_bucket_once(dev):
    """(round-2 advisor) finish_grad_sync() is idempotent per optimiser step: a second grad_norm() -- a logging callback, grad_norm(inf)
    after grad_norm(max) -- must not all-reduce (i.e. multiply by the world size) the embeddings + heads slice again.  World 1 with the
    buckets forced on and a counting stand-in for the exchange."""
    from spokennlp_amd.dp import GradBuckets
    from spokennlp_amd.trainer import AmdsegFusedAdamW
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    opt = AmdsegFusedAdamW(m, lr=1e-3, max_grad_norm=1.0)
    eng = m.engine()
    calls = []

    class Counting(GradBuckets):
        def _reduce(self, a, b, group, bf16=False):
            calls.append((a, b))
            self._covered += b - a

        def norm_is_complete(self):
            return False

    eng.buckets = Counting(eng.fp)
    _train_step(m, batch, dev)
    n1 = opt.grad_norm().clone()
    # (round 4: the rest goes in two spans -- the embedding tables from inside backward, pooler + heads from finish_grad_sync)
    emb = eng.buckets.emb_slice
    tail = (emb[1], eng.buckets.rest_slice[1])
    count = lambda sl: len([c for c in calls if c == sl])      # noqa: E731
    assert count(emb) == 1 and count(tail) == 1 and len(calls) == eng.nlayers + 2
    n2 = opt.grad_norm(float("inf")).clone()
    assert count(emb) == 1 and count(tail) == 1 and torch.equal(n1, n2)
    opt.step(); opt.zero_grad()
    _train_step(m, batch, dev)
    opt.grad_norm()
    assert count(emb) == 2 and count(tail) == 2             # the next step reduces them again, once each


# ------------------------------------------------------------------------------------------------ multi-rank prediction (run_inference.sh:35)
N_PRED_WINDOWS = 7                                            # odd on purpose: the last batch of one rank is a wrapped-around duplicate


def _predict_and_write(tr, ds, path):
    """Trainer.predict -> the reference's decode + per-document merge + prediction file (ts_sentence_seq_labeling.py:1111-1191)"""
    import numpy as np
    from spokennlp_amd import preprocess as P
    out = tr.predict(ds)
    logits, cos = out.predictions[0], out.predictions[1]
    labels = out.label_ids[0] if isinstance(out.label_ids, (tuple, list)) else out.label_ids
    assert logits.shape[0] == len(ds) and cos.shape[0] == len(ds) and labels.shape[0] == len(ds)     # duplicates of the padded tail truncated
    decoded = P.decode_anchor_predictions(logits, labels)
    cos_rows = [[float(v) for v in row[:len(d["pred_ids"])]] for row, d in zip(cos, decoded)]
    ex = [i // 2 for i in range(len(ds))]
    docs = P.merge_windows_to_documents(decoded, ex, max(ex) + 1, eop_pair_cos_sim=cos_rows)
    P.write_prediction_file(path, docs)
    return np.asarray(logits), np.asarray(cos), decoded


def _predict_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      ACCELERATE_TORCH_DEVICE="cuda:0")
    import numpy as np
    import torch.distributed as dist
    from transformers import default_data_collator
    from spokennlp_amd.trainer import Trainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        z, sd, batch, arch = load_case("tiny_L64")
        m = build_model(arch, flags_of(z, "full_eval"), sd, dev)
        m.config.amdseg_precision = "parity"
        ds = _DS(_samples(arch, N_PRED_WINDOWS))
        tr = Trainer(model=m, args=_args(os.path.join(out_dir, f"p{rank}"), per_device_eval_batch_size=2, ddp_backend="gloo", dataloader_drop_last=False),
                     data_collator=default_data_collator)
        logits, cos, _ = _predict_and_write(tr, ds, os.path.join(out_dir, f"pred_rank{rank}.txt"))
        ev = tr.evaluate(ds)
        np.savez(os.path.join(out_dir, f"pred{rank}.npz"), logits=logits, cos=cos, eval_loss=ev["eval_loss"])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_predict_and_evaluate_two_ranks_equal_single_rank(dev, tmp_path):
    """run_inference.sh:35 launches prediction under torch.distributed: the Trainer's evaluation loop gathers (logits (N,2,L,2),
    cos_sim (N,k)) across ranks -- k (labelled [BOS] per window) differs per batch AND per rank and is padded with -100, and the 7 windows
    do not divide over 2 ranks x batches of 2, so the sharded loader wraps around and the gather truncates the duplicates.  Gathered
    predictions and the written prediction file must equal the single-process run bit for bit (each window's result is independent of its
    batch mates); the loss `evaluate` reports is the mean of per-batch losses, which the two runs batch differently -- finite on both."""
    import numpy as np
    import torch.multiprocessing as mp
    from transformers import default_data_collator
    from spokennlp_amd.trainer import Trainer
    z, sd, batch, arch = load_case("tiny_L64")
    samples = _samples(arch, N_PRED_WINDOWS)
    ks = [int((s["labels"][0] != -100).sum()) for s in samples]
    assert len(set(ks)) > 1, ks                                          # the cos_sim width really differs between windows
    m = build_model(arch, flags_of(z, "full_eval"), sd, dev)
    m.config.amdseg_precision = "parity"
    tr = Trainer(model=m, args=_args(tmp_path / "single", per_device_eval_batch_size=2, dataloader_drop_last=False), data_collator=default_data_collator)
    logits1, cos1, decoded1 = _predict_and_write(tr, _DS(samples), str(tmp_path / "pred_single.txt"))
    assert cos1.shape[1] == max(ks)
    for row, k in zip(cos1, ks):
        assert (row[k:] == -100).all() and (np.abs(row[:k - 1]) <= 1.0 + 1e-5).all()
    mp.spawn(_predict_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = open(tmp_path / "pred_single.txt", "rb").read()
    assert len(want.splitlines()) == (N_PRED_WINDOWS + 1) // 2
    for rank in range(2):
        got = np.load(tmp_path / f"pred{rank}.npz")
        assert got["logits"].shape == logits1.shape and np.array_equal(got["logits"], logits1), rank
        assert got["cos"].shape == cos1.shape and np.array_equal(got["cos"], cos1), rank
        assert open(tmp_path / f"pred_rank{rank}.txt", "rb").read() == want, rank
        assert math.isfinite(float(got["eval_loss"]))
