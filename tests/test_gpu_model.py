"""End-to-end parity of the HIP path (called through the C ABI by the HF-surface model) against
  (1) golden vectors produced by the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle on fresh seeded inputs.
bf16 fast path: MFMA operands and stored activations are bf16 (2^-9 relative rounding), accumulation / softmax /
LayerNorm statistics fp32.  Tolerances below are for that path and are asserted together with exact equality of
the decoded boundary predictions; the fp32 'parity mode' test asserts the north-star 1e-3 on logits."""
import math
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from tests.test_oracle_golden import load_case, flags_of  # noqa: E402


def build_model(arch, flags, sd, dev, dropout=0.0):
    from transformers import BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    cfg = BertConfig(num_labels=2, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, **arch)
    for k, v in flags.items():
        setattr(cfg, k, v)
    m = M(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    return m.to(dev)


def to_dev(batch, dev):
    return {k: v.to(dev) for k, v in batch.items()}


@pytest.mark.parametrize("case", ["tiny_L64", "tiny_L128", "tiny_L100_B3"])
@pytest.mark.parametrize("variant", ["plain_eval", "full_eval"])
def test_eval_vs_reference_golden(dev, case, variant):
    from oracle import bert_ts_oracle as O
    z, sd, batch, arch = load_case(case)
    m = build_model(arch, flags_of(z, variant), sd, dev).eval()
    random.seed(int(z[f"{variant}.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**to_dev(batch, dev))
    ref_logits = torch.from_numpy(z[f"{variant}.logits"])
    d = (logits.cpu() - ref_logits).abs()
    lab = batch["labels"] != -100
    print(f"{case}/{variant}: max|dlogit| all={d.max():.4f} labelled={d[lab].max():.4f} mean={d.mean():.5f} "
          f"logit std={ref_logits.std():.3f}")
    # bf16 fast path, logit std ~3: measured in round 6 0.041-0.046 (max) / 0.0095-0.011 (mean) over the six cases; bound = measured + 25 % (VERDICT r05 #13:
    # the 0.08 of rounds 1-5 could hide a doubling)
    assert d.max().item() < 0.058 and d.mean().item() < 0.014
    assert abs(loss.item() - float(z[f"{variant}.loss"])) < 0.05
    assert (cos.cpu() - torch.from_numpy(z[f"{variant}.cos"])).abs().max().item() < 0.02
    # predicted boundary indices bit-exact
    assert O.decode_predictions(logits.cpu()[:, 0], batch["labels"][:, 0]) == \
        O.decode_predictions(ref_logits[:, 0], batch["labels"][:, 0])
    assert logits.shape == ref_logits.shape and cos.shape == z[f"{variant}.cos"].shape


@pytest.mark.parametrize("variant", ["train_full", "train_eop_matrix", "train_eot_list", "train_focal", "train_wce"])
def test_train_grads_vs_reference_golden(dev, variant):
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, variant), sd, dev).train()      # dropout 0 in the golden run
    random.seed(int(z[f"{variant}.random_seed"]))
    loss, logits, cos = m(**to_dev(batch, dev))
    loss.backward()
    assert abs(loss.item() - float(z[f"{variant}.loss"])) < 0.05 * max(1.0, abs(float(z[f"{variant}.loss"])) / 5)
    names = z[f"{variant}.gradnorm_names"].tolist()
    vals = z[f"{variant}.gradnorm_vals"].tolist()
    params = dict(m.named_parameters())
    worst = 0.0
    for n, gv in zip(names, vals):
        g = params[n].grad
        assert g is not None, n
        mine = float(g.float().norm())
        if gv < 0:
            assert mine == 0.0, n
            continue
        rel = abs(mine - gv) / max(gv, 1e-3)
        worst = max(worst, rel)
        assert rel < 0.035, (n, mine, gv)               # measured worst over the five variants: 0.0177-0.0273 (+ 25 %; was 0.05)
    print(variant, "worst grad-norm rel err", worst)
    if variant == "train_full":
        for k in z.files:
            if k.startswith("train_full.grad."):
                n = k[len("train_full.grad."):]
                ref = torch.from_numpy(z[k])
                if float(ref.norm()) < 1e-5:          # e.g. key.bias: softmax is shift-invariant, its true gradient is 0
                    assert float(params[n].grad.float().norm()) < 1e-2, n
                    continue
                g = params[n].grad.float().cpu()
                cosine = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
                assert cosine > 0.995, (n, cosine)


def test_fresh_inputs_vs_oracle(dev):
    """new seeded weights + batch (not in any fixture): HIP path vs CPU oracle, eval + train, L=128 B=4."""
    from oracle import bert_ts_oracle as O
    from spokennlp_amd import data
    from tests.util import tiny_state_dict
    arch = dict(vocab_size=300, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512,
                max_position_embeddings=128, type_vocab_size=2)
    flags = dict(do_da_ts=True, do_cssl=True, do_tssp=True, cl_loss_weight=0.5, cl_temp=0.1, cl_anchor_level="eop_list",
                 cl_positive_k=1, cl_negative_k=3, tssp_loss_weight=1.0)
    sd = tiny_state_dict(arch, seed=123)
    docs = data.synth_docs(10, seed=77, vocab=300, mean_sents=16, sd_sents=5, mean_boundaries=3, mu_tok=1.8, sigma_tok=0.4)
    batch = data.batches_from_docs(docs, 128, 4, seed=5)[1]
    cfg = O.make_cfg(num_labels=2, **arch, **flags)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    random.seed(11)
    lo, logits_o, cos_o = O.model_forward(sdo, cfg, batch)
    lo.backward()
    m = build_model(arch, flags, sd, dev).train()
    random.seed(11)
    lm, logits_m, cos_m = m(**to_dev(batch, dev))
    lm.backward()
    assert abs(lm.item() - lo.item()) < 0.05 * max(1.0, abs(lo.item()) / 5)
    dl = (logits_m.detach().cpu() - logits_o.detach()).abs().max().item()
    print("fresh inputs: max|dlogit|", dl, "max|logit|", logits_o.abs().max().item())
    assert dl < 0.02 * logits_o.abs().max().item() + 0.05           # bf16 fast path: ~2^-7 of the logit scale after 3 layers
    for n, p in m.named_parameters():
        go = sdo[n].grad
        if go is None or float(go.norm()) < 1e-5:
            continue
        c = torch.nn.functional.cosine_similarity(p.grad.float().cpu().flatten(), go.flatten(), dim=0).item()
        assert c > 0.99, (n, c)


@pytest.mark.parametrize("Lq,B", [(100, 3), (50, 3), (72, 1)])
def test_unaligned_shapes_vs_oracle(dev, Lq, B):
    """windows that are not a multiple of 64 tokens / batches that do not fill a 128-row tile (the reference takes any shape):
    EncoderFn pads with masked tokens and cuts the output back -- eval in fp32 parity mode to 1e-3, one bf16 train step"""
    from oracle import bert_ts_oracle as O
    from spokennlp_amd import data
    from tests.util import tiny_state_dict
    arch = dict(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                max_position_embeddings=128, type_vocab_size=2)
    flags = dict(do_da_ts=True, do_cssl=True, do_tssp=True, cl_loss_weight=0.5, cl_temp=0.1, cl_anchor_level="eop_list",
                 cl_positive_k=1, cl_negative_k=3, tssp_loss_weight=1.0)
    sd = tiny_state_dict(arch, seed=7)
    docs = data.synth_docs(12, seed=5, vocab=300, mean_sents=14, sd_sents=4, mean_boundaries=3, mu_tok=1.6, sigma_tok=0.4)
    batch = data.batches_from_docs(docs, Lq, B, seed=3)[0]
    assert batch["input_ids"].shape == (B, 2, Lq)
    cfg = O.make_cfg(num_labels=2, **arch, **flags)
    # eval, fp32 parity mode
    random.seed(11)
    with torch.no_grad():
        lo, logits_o, cos_o = O.model_forward(sd, cfg, batch)
    m = build_model(arch, flags, sd, dev)
    m.config.amdseg_precision = "fp32"
    m.eval()
    random.seed(11)
    with torch.no_grad():
        lm, logits_m, cos_m = m(**to_dev(batch, dev))
    assert logits_m.shape == logits_o.shape
    assert (logits_m.cpu() - logits_o).abs().max().item() < 1e-3 and abs(lm.item() - lo.item()) < 1e-3
    assert (cos_m.cpu() - cos_o).abs().max().item() < 1e-3
    # one training step on the bf16 path
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    random.seed(12)
    lo, _, _ = O.model_forward(sdo, cfg, batch)
    lo.backward()
    m = build_model(arch, flags, sd, dev).train()
    random.seed(12)
    lm, _, _ = m(**to_dev(batch, dev))
    lm.backward()
    assert abs(lm.item() - lo.item()) < 0.05 * max(1.0, abs(lo.item()) / 5)
    for n, p in m.named_parameters():
        go = sdo[n].grad
        if go is None or float(go.norm()) < 1e-5:
            continue
        c = torch.nn.functional.cosine_similarity(p.grad.float().cpu().flatten(), go.flatten(), dim=0).item()
        assert c > 0.99, (n, c)


def test_dropout_training_step_is_finite_and_deterministic(dev):
    z, sd, batch, arch = load_case("tiny_L64")
    losses = []
    for rep in range(2):
        m = build_model(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train()
        m.amdseg_seed = 42
        random.seed(3)
        loss, _, _ = m(**to_dev(batch, dev))
        loss.backward()
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters())).item()
        assert np.isfinite(loss.item()) and np.isfinite(gn) and gn > 0
        losses.append((loss.item(), gn))
    # same seed -> bit-identical loss (stateless dropout RNG, deterministic reductions); the embedding-table gradients are
    # accumulated with fp32 atomics like torch's own embedding backward, so the norm is compared to 1e-6
    assert losses[0][0] == losses[1][0]
    assert abs(losses[0][1] - losses[1][1]) <= 1e-6 * losses[0][1]


@pytest.mark.parametrize("precision", ["bf16", "parity"])
def test_hidden_dropout_decisions_kept_by_forward_equal_the_rehashed_ones(dev, precision):
    """acts.drop1 / drop2 (ABI 8): the dropout + residual + LayerNorm forward stores its keep decisions (1 byte per 8 elements) and the
    LayerNorm backward reads them instead of re-hashing.  Same hash, same decisions: the encoder layers' gradients are BIT-IDENTICAL to
    the path that evaluates the hash in both directions (engine.hidden_keepbits = False), at a realised drop rate of ~p."""
    z, sd, batch, arch = load_case("tiny_L128")
    res = {}
    for keep in (True, False):
        m = build_model(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train()
        m.config.amdseg_precision = precision
        m.amdseg_seed = 7
        eng = m.engine()
        eng.hidden_keepbits = keep
        random.seed(3)
        loss, _, _ = m(**to_dev(batch, dev))
        loss.backward()
        torch.cuda.synchronize()
        res[keep] = (loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None and ".layer." in n})
        if keep:
            A = [a for a in eng._arenas.values() if a.get("layers") and "drop1" in a["layers"][0]]
            assert A, "no keep-bit buffers were allocated"
            bits = A[0]["layers"][0]["drop1"]
            rate = 1.0 - float(torch.tensor([bin(v).count("1") for v in bits.cpu().tolist()]).sum()) / (bits.numel() * 8)
            assert 0.08 < rate < 0.12, rate
    assert res[True][0] == res[False][0]
    for n, g in res[False][1].items():
        # (both precisions since the loss heads stopped scattering with fp32 atomics: their arrival-order noise used to reach every layer
        # below them, visible in fp32 activations)
        assert torch.equal(res[True][1][n], g), n


def test_h768_layernorm_backward_pair_kernel_equals_generic_kernel(dev):
    """csrc/elementwise.hip ln_bwd_pair768_kernel (H = 768, bf16: two rows per wave iteration, three chunks per lane, buffer addressing, the
    forward's keep bits) against the generic row kernel with the re-evaluated hash (engine.hidden_keepbits = False selects it): same
    decisions, same formula, another summation order -- one training step with dropout at H = 768 gives the same loss and gradients that
    agree to the bf16 rounding of the activation gradients"""
    from tests.util import tiny_state_dict
    from spokennlp_amd import data
    arch = dict(vocab_size=300, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1024,
                max_position_embeddings=128, type_vocab_size=2)
    z, _, _, _ = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    sd = tiny_state_dict(arch, seed=3)
    docs = data.synth_docs(8, seed=5, vocab=300, mean_sents=14, sd_sents=4, mean_boundaries=3, mu_tok=1.6, sigma_tok=0.4)
    batch = to_dev(data.batches_from_docs(docs, 128, 2, seed=2)[0], dev)
    res = {}
    for keep in (True, False):
        m = build_model(arch, flags, sd, dev, dropout=0.1).train()
        m.amdseg_seed = 5
        m.engine().hidden_keepbits = keep
        random.seed(3)
        loss = m(**batch)[0]
        loss.backward()
        res[keep] = (loss.item(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None and ".layer." in n})
    assert res[True][0] == res[False][0]                     # the forward is the same kernel either way
    worst = 0.0
    for n, g in res[False][1].items():
        if "key.bias" in n:                                  # softmax is invariant to a key bias: its gradient is rounding noise around 0
            continue
        d = float((res[True][1][n] - g).norm()) / max(float(g.norm()), 1e-6)
        worst = max(worst, d)
        assert d < 2e-2, (n, d)
    print("pair kernel vs generic ln_bwd: worst relative gradient difference", worst)


def test_fused_adamw_step_matches_torch(dev):
    """engine.adamw_step (clip 1.0 + AdamW over the flat buffers) vs clip_grad_norm_ + torch.optim.AdamW on a clone."""
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    random.seed(7)
    loss, _, _ = m(**to_dev(batch, dev))
    loss.backward()
    ref_params = {n: torch.nn.Parameter(p.detach().clone()) for n, p in m.named_parameters()}
    for n, p in m.named_parameters():
        ref_params[n].grad = p.grad.detach().clone()
    torch.nn.utils.clip_grad_norm_(list(ref_params.values()), 1.0)
    opt = torch.optim.AdamW(list(ref_params.values()), lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    opt.step()
    m.engine().adamw_step(5e-5, max_grad_norm=1.0)
    for n, p in m.named_parameters():
        assert (p.detach() - ref_params[n].detach()).abs().max().item() < 1e-6, n
    # the optimiser pass zeroed everything but the encoder layers' slice, which the next backward overwrites (engine.lazy_zero); any other
    # reader flushes it first
    fp = m.engine().fp
    assert fp.grad_stale == m.engine().lazy_zero and fp.flat_g[:fp.layers_begin].abs().max().item() == 0.0     # (AMDSEG_LAZY_ZERO=0: everything zeroed)
    fp.flush_stale()
    assert not fp.grad_stale and fp.flat_g.abs().max().item() == 0.0


@pytest.mark.parametrize("case", ["tiny_L64", "tiny_L128", "tiny_L100_B3"])
@pytest.mark.parametrize("variant", ["plain_eval", "full_eval"])
def test_fp32_parity_mode_logits_within_1e3(dev, case, variant):
    """north star: logits within 1e-3 of the fp32 CPU reference, predicted boundary indices bit-exact.
    config.amdseg_precision = 'fp32' runs inference on the exact-fp32 MFMA path (v_mfma_f32_32x32x2_f32)."""
    from oracle import bert_ts_oracle as O
    z, sd, batch, arch = load_case(case)
    fl = flags_of(z, variant)
    fl["amdseg_precision"] = "fp32"
    m = build_model(arch, fl, sd, dev).eval()
    random.seed(int(z[f"{variant}.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**to_dev(batch, dev))
    ref_logits = torch.from_numpy(z[f"{variant}.logits"])
    d = (logits.cpu() - ref_logits).abs().max().item()
    print(f"{case}/{variant} fp32 mode: max|dlogit| = {d:.2e}")
    assert d < 1e-3                                     # tolerance stated by BASELINE.json north_star
    assert d < 1e-4                                     # what exact-fp32 MFMA actually delivers
    assert abs(loss.item() - float(z[f"{variant}.loss"])) < 1e-4
    assert (cos.cpu() - torch.from_numpy(z[f"{variant}.cos"])).abs().max().item() < 1e-4
    assert O.decode_predictions(logits.cpu()[:, 0], batch["labels"][:, 0]) == \
        O.decode_predictions(ref_logits[:, 0], batch["labels"][:, 0])


# ------------------------------------------------------------------------------------------------ ELECTRA wrapper (f-3)
@pytest.mark.parametrize("case", ["electra_tiny_L64", "electra_small_tiny_L64"])
def test_electra_wrapper_vs_reference_golden(dev, case):
    """electra-base's shape family (embedding_size == hidden_size) and electra-small's (128 -> 256 through `embeddings_project`: embedding kernels at the
    embedding width + one NT GEMM, its weight / bias / input gradients in backward) against the reference's own outputs"""
    from transformers import ElectraConfig
    from spokennlp_amd.electra_for_ts import ElectraWithDAForSentenceLabelingTopicSegmentation as M
    from oracle import bert_ts_oracle as O
    z, sd, batch, arch = load_case(case)
    for variant, mode in (("full_eval", "eval"), ("train_full", "train")):
        cfg = ElectraConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **arch)
        for k, v in flags_of(z, variant).items():
            setattr(cfg, k, v)
        m = M(cfg)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing)
        m = m.to(dev)
        random.seed(int(z[f"{variant}.random_seed"]))
        if mode == "eval":
            m.eval()
            with torch.no_grad():
                loss, logits, cos = m(**to_dev(batch, dev))
            ref = torch.from_numpy(z[f"{variant}.logits"])
            assert (logits.cpu() - ref).abs().max().item() < 0.08
            assert O.decode_predictions(logits.cpu()[:, 0], batch["labels"][:, 0]) == O.decode_predictions(ref[:, 0], batch["labels"][:, 0])
        else:
            m.train()
            loss, _, _ = m(**to_dev(batch, dev))
            loss.backward()
            params = dict(m.named_parameters())
            for k in z.files:
                if k.startswith("train_full.grad."):
                    n = k[len("train_full.grad."):]
                    ref = torch.from_numpy(z[k])
                    if float(ref.norm()) < 1e-5:
                        continue
                    c = torch.nn.functional.cosine_similarity(params[n].grad.float().cpu().flatten(), ref.flatten(), dim=0).item()
                    assert c > 0.99, (n, c)
            if case == "electra_small_tiny_L64":
                assert params["electra.embeddings_project.weight"].grad is not None and float(params["electra.embeddings_project.bias"].grad.abs().sum()) > 0
        assert abs(loss.item() - float(z[f"{variant}.loss"])) < 0.05 * max(1.0, abs(float(z[f"{variant}.loss"])) / 5)


def test_stock_torch_optimizer_loop_like_hf_trainer(dev):
    """the single-GPU HF Trainer inner loop on the drop-in class: loss.backward(); clip_grad_norm_; optimizer.step();
    model.zero_grad() (set_to_none=True, Trainer's default) -- parameter gradients are views into the flat buffer, the
    weights the HIP kernels read follow the optimizer's in-place updates, and training makes progress."""
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3)
    b = to_dev(batch, dev)
    losses = []
    for step in range(6):
        random.seed(0)
        loss = m(**b)[0]
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        assert torch.isfinite(gn)
        opt.step()
        m.zero_grad()                                  # sets .grad = None: the next forward must re-attach the flat views
        losses.append(loss.item())
    assert all(p.grad is None for p in m.parameters())
    assert losses[-1] < losses[0] - 0.05, losses
    # the bf16 shadows were refreshed from the updated masters: eval logits differ from the initial weights' logits
    m.eval()
    with torch.no_grad():
        _, lg, _ = m(**b)
    assert (lg.cpu() - torch.from_numpy(z["full_eval.logits"])).abs().max().item() > 0.05


def test_hf_lifecycle_save_load_resize(dev, tmp_path):
    """the reference driver's model lifecycle (ts_sentence_seq_labeling.py:188-198,247-257,284; run_inference.sh:19,36):
    save_pretrained -> AutoConfig/from_pretrained(ignore_mismatched_sizes) -> resize_token_embeddings -> forward, also AFTER the engine
    has re-homed the parameters into its flat buffer (checkpoints written mid-training must hold every tensor)"""
    from transformers import AutoConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "full_eval")
    m = build_model(arch, flags, sd, dev).eval()
    random.seed(5)                                                        # the CSSL sampler draws from Python's global RNG
    with torch.no_grad():
        loss0, logits0, _ = m(**to_dev(batch, dev))                      # engine built: parameters are views of one flat buffer now
    d1 = tmp_path / "ckpt"
    m.save_pretrained(d1)
    cfg = AutoConfig.from_pretrained(d1)
    assert cfg.do_da_ts == flags["do_da_ts"] and cfg.cl_anchor_level == flags["cl_anchor_level"]
    m2 = M.from_pretrained(d1, config=cfg, ignore_mismatched_sizes=True)
    sd2 = m2.state_dict()
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), sd2[k]), k
    m2 = m2.to(dev).eval()
    random.seed(5)
    with torch.no_grad():
        loss1, logits1, _ = m2(**to_dev(batch, dev))
    assert torch.equal(logits0, logits1) and loss0.item() == loss1.item()
    # [BOS] is added to the tokenizer and the embedding table grows by one row (:284); old rows keep their values
    V = m2.config.vocab_size
    m2.resize_token_embeddings(V + 1)
    assert m2.bert.embeddings.word_embeddings.weight.shape[0] == V + 1
    with torch.no_grad():
        loss2, logits2, _ = m2(**to_dev(batch, dev))                     # the engine notices the replaced parameter and rebuilds
    assert torch.equal(logits1, logits2)
    b2 = {k: v.clone() for k, v in batch.items()}
    b2["input_ids"][:, :, 3] = V                                          # the new id is usable
    with torch.no_grad():
        _, logits3, _ = m2(**to_dev(b2, dev))
    assert torch.isfinite(logits3).all() and not torch.equal(logits3, logits2)
    # a training step, then a checkpoint: gradients do not leak into the saved tensors, optimiser-updated weights do
    m2.train()
    random.seed(0)
    loss, _, _ = m2(**to_dev(batch, dev))
    loss.backward()
    m2.engine().adamw_step(lr=1e-3)
    d2 = tmp_path / "ckpt2"
    m2.save_pretrained(d2)
    m3 = M.from_pretrained(d2, config=AutoConfig.from_pretrained(d2))
    w2, w3 = m2.state_dict(), m3.state_dict()
    for k in w2:
        assert torch.equal(w2[k].cpu(), w3[k]), k
    assert not torch.equal(w3["bert.encoder.layer.0.output.dense.weight"], sd2["bert.encoder.layer.0.output.dense.weight"])


def test_real_hf_trainer_train_and_predict(dev, tmp_path):
    """the actual `transformers.Trainer` (ts_sentence_seq_labeling.py:1077-1085 train, :1132 predict) drives the drop-in class:
    default collator -> (B,2,L) batches, label_names derived from the forward signature, torch AdamW on the parameter views,
    predict() returning (logits, cos_sim) as compute_metrics unpacks them (:1019-1021)"""
    from transformers import Trainer, TrainingArguments, default_data_collator
    from spokennlp_amd import data
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_full")
    docs = data.synth_docs(24, seed=11, vocab=arch["vocab_size"], mean_sents=10, sd_sents=3, mean_boundaries=2, mu_tok=1.4, sigma_tok=0.3)
    batches = data.batches_from_docs(docs, 64, 1, seed=4)
    samples = [{k: v[0] for k, v in b.items()} for b in batches][:16]
    assert len(samples) >= 8

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(samples)

        def __getitem__(self, i):
            return samples[i]

    m = build_model(arch, flags, sd, dev)
    w0 = m.state_dict()["bert.encoder.layer.0.output.dense.weight"].clone()
    # run_finetune.sh:74-82: evaluation by steps, --load_best_model_at_end, --metric_for_best_model overall_f1 -- the key the reference's
    # compute_metrics closure (ts_sentence_seq_labeling.py:1018-1074 -> evaluate.make_compute_metrics) returns
    from spokennlp_amd import evaluate as E
    args = TrainingArguments(output_dir=str(tmp_path / "out"), per_device_train_batch_size=4, per_device_eval_batch_size=4, max_steps=4,
                             learning_rate=1e-3, lr_scheduler_type="linear", max_grad_norm=1.0, gradient_accumulation_steps=2,
                             report_to=[], eval_strategy="steps", eval_steps=2, save_strategy="steps", save_steps=2, save_total_limit=2,
                             load_best_model_at_end=True, metric_for_best_model="overall_f1", eval_accumulation_steps=1000,
                             logging_steps=1, seed=7, dataloader_drop_last=True, remove_unused_columns=True)
    random.seed(3)
    tr = Trainer(model=m, args=args, train_dataset=DS(), eval_dataset=DS(), data_collator=default_data_collator,
                 compute_metrics=E.make_compute_metrics())
    assert set(tr.label_names) == {"labels", "sent_level_labels"}
    out = tr.train()
    assert out.global_step == 4 and math.isfinite(out.training_loss)
    assert not torch.equal(m.state_dict()["bert.encoder.layer.0.output.dense.weight"], w0)
    evals = [h for h in tr.state.log_history if "eval_overall_f1" in h]
    assert len(evals) == 2 and all(0.0 <= h["eval_overall_f1"] <= 1.0 and "eval_da_overall_f1" in h and "eval_EOP_number" in h for h in evals)
    best = max(evals, key=lambda h: h["eval_overall_f1"])
    assert tr.state.best_metric == best["eval_overall_f1"] and tr.state.best_model_checkpoint is not None
    # the best checkpoint was loaded back INTO the engine's flat buffer: evaluating now reproduces its score
    again = tr.evaluate()
    assert abs(again["eval_overall_f1"] - tr.state.best_metric) < 1e-12
    pred = tr.predict(DS())
    logits, cos = pred.predictions
    assert logits.shape == (len(samples), 2, 64, 2) and cos.shape[0] == len(samples)
    assert np.isfinite(logits).all()
    lab = pred.label_ids[0] if isinstance(pred.label_ids, (tuple, list)) else pred.label_ids
    assert lab.shape == (len(samples), 2, 64)


@pytest.mark.parametrize("variant", ["train_full", "train_eot_list", "train_wce", "train_focal"])
def test_fused_heads_equal_torch_heads(dev, variant):
    """csrc/heads.hip (token CE + CSSL lists + TSSP in a handful of launches) against the torch formulation of the same heads on the same
    encoder output: loss and every gradient (the reference goldens pin both paths separately; this pins them to each other tightly)"""
    z, sd, batch, arch = load_case("tiny_L64")
    res = {}
    for fused in (True, False):
        m = build_model(arch, flags_of(z, variant), sd, dev).train()
        m.config.amdseg_fused_heads = fused
        m.config.amdseg_precision = "parity"            # fp32-grade encoder: differences below come from the heads alone
        random.seed(int(z[f"{variant}.random_seed"]))
        loss, logits, cos = m(**to_dev(batch, dev))
        loss.backward()
        res[fused] = (loss.item(), logits.detach().float().cpu().clone(),
                      {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert abs(res[True][0] - res[False][0]) < 2e-5 * max(1.0, abs(res[False][0]))
    assert torch.equal(res[True][1], res[False][1])
    worst = 0.0
    for n, g in res[False][2].items():
        d = float((res[True][2][n] - g).norm()) / max(float(g.norm()), 1e-3)
        worst = max(worst, d)
        assert d < 1e-4, (n, d)
    print(variant, "fused vs torch heads: worst relative gradient difference", worst)


def test_focal_loss_of_a_half_without_labelled_rows_is_zero_not_nan(dev):
    """ FocalLoss on a segment whose labels are all -100: the reference's mean CE is NaN and it returns a constant 0
    (modules/utils.py:150-156) -- the DA half here; loss and every gradient stay finite, that half adds nothing, and the fused heads
    (csrc/heads.hip) and the torch formulation agree"""
    z, sd, batch, arch = load_case("tiny_L64")
    flags = flags_of(z, "train_focal")
    b = {k: v.clone() for k, v in batch.items()}
    b["labels"][:, 1, :] = -100                              # the augmented half carries no token label
    b = to_dev(b, dev)
    res = {}
    for fused in (True, False):
        m = build_model(arch, flags, sd, dev).train()
        m.config.amdseg_fused_heads = fused
        m.config.amdseg_precision = "parity"
        random.seed(int(z["train_focal.random_seed"]))
        loss, logits, _ = m(**b)
        loss.backward()
        assert torch.isfinite(loss)
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(g).all() for g in grads.values())
        res[fused] = (loss.item(), grads)
    assert abs(res[True][0] - res[False][0]) < 2e-5 * max(1.0, abs(res[False][0]))
    for n, g in res[False][1].items():
        assert float((res[True][1][n] - g).norm()) / max(float(g.norm()), 1e-3) < 1e-4, n


def test_trailing_padding_chunks_are_skipped_without_changing_a_bit(dev):
    """amdseg_bert_cfg.kend / seq_order: the full-attention kernels do not visit the 64-key chunks past a sequence's last unmasked key and
    walk the sequences longest first.  The skipped chunks contribute exact zeros, so the forward result is bit-identical; the backward is
    compared up to the atomics noise of the heads / embedding scatters (see test_global_row_side_stream_changes_nothing)"""
    from transformers import BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=300, hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=256,
                     max_position_embeddings=512, num_labels=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    m = M(cfg).to(dev)
    B, L = 8, 256                                             # 16 (batch, head) pairs: the XCD-aware placement is active too
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 300, (B, 1, L), generator=g)
    lens = [256, 200, 130, 64, 63, 256, 1, 129]
    am = torch.zeros(B, 1, L, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, 0, :n] = 1
    labels = torch.full((B, 1, L), -100)
    for b, n in enumerate(lens):
        labels[b, 0, 0:n:7] = torch.randint(0, 2, (len(range(0, n, 7)),), generator=g)
    batch = {k: v.to(dev) for k, v in dict(input_ids=ids, attention_mask=am, token_type_ids=torch.zeros_like(ids), labels=labels).items()}
    eng = m.engine()
    outs = {}
    for skip in (True, False):
        eng.skip_padded_chunks = skip
        m.eval()
        with torch.no_grad():
            _, logits, _ = m(**batch)
        m.train()
        m.zero_grad(set_to_none=False)
        random.seed(5)
        m._step_seed = 100                                    # the wrapper's dropout seed counts training forwards: same seed for both runs
        loss = m(**batch)[0]
        loss.backward()
        outs[skip] = (logits.clone(), loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(outs[True][0], outs[False][0])         # forward: bit-identical, padded positions included
    assert outs[True][1] == outs[False][1]                    # ... with dropout as well
    for n, ga in outs[True][2].items():
        gb = outs[False][2][n]
        tol = 1e-2 if "embeddings" in n else 2e-3
        assert float((ga - gb).abs().max()) <= tol * max(1e-3, float(ga.abs().max())), n


@pytest.mark.gpu
def test_backward_drops_the_rows_of_trailing_padding_without_changing_a_bit(dev):
    """amdseg_bert_cfg.pad_guard / pad_runs: when the gradient the backward starts from is an exact zero on the rows of trailing padding,
    those rows stay zero through every layer, the weight-gradient GEMM walks only the token tiles in front of kend and the input-gradient
    GEMMs skip their all-padding 256-row tiles.  Driven at the engine (a fixed incoming gradient: no atomic scatter in front of the encoder
    layers), the layer gradients must be bit-identical with the switch on and off.  With one non-zero element on a padded row the device-side
    guard must fall back to the dense walk."""
    from transformers import BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=300, hidden_size=768, num_attention_heads=12, num_hidden_layers=2, intermediate_size=3072,
                     max_position_embeddings=512, num_labels=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    m = M(cfg).to(dev)
    eng = m.engine()
    B, L = 4, 512
    lens = [512, 300, 200, 1]                                 # 256-row tiles made of padding: the second halves of sequences 2 and 3
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 300, (B, L), generator=g).to(dev)
    am = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, :n] = 1
    am = am.to(dev)
    tt = torch.zeros_like(ids)
    dseq0 = torch.randn(B, L, 768, generator=g).to(dev) * am[:, :, None].float()
    layer_names = [n for n in eng.fp.offsets if ".encoder.layer." in n]
    assert layer_names

    def run(skip, dseq):
        eng.skip_padded_rows_bwd = skip
        _, ectx = eng.forward(ids, am, tt, True, seed=7, p_out=0.1)
        eng.backward(ectx, dseq, accumulate=False)
        torch.cuda.synchronize()
        return {n: eng.fp.view(eng.fp.flat_g, n).clone() for n in layer_names}, int(eng._pad_guard.item()) if skip else None

    on, guard = run(True, dseq0)
    off, _ = run(False, dseq0)
    assert guard == 0
    for n in layer_names:
        assert torch.equal(on[n], off[n]), n
        assert float(on[n].abs().max()) > 0
    dseq1 = dseq0.clone()
    dseq1[2, 400, 5] = 1e-3                                   # a caller whose loss does look at a padded row
    on1, guard1 = run(True, dseq1)
    off1, _ = run(False, dseq1)
    assert guard1 == 1
    differs = 0
    for n in layer_names:
        assert torch.equal(on1[n], off1[n]), n
        differs += int(not torch.equal(on1[n], on[n]))
    assert differs > 0                                        # ... and that row's gradient did arrive in the weights
    # gradient accumulation: a second micro-batch with other lengths ADDS into the same buffers (accumulate = 1 in every kernel)
    def run_accum(skip):
        eng.skip_padded_rows_bwd = skip
        am2 = torch.zeros_like(am)
        for b, n in enumerate([64, 512, 129, 300]):
            am2[b, :n] = 1
        for i, (mask, seed) in enumerate(((am, 7), (am2, 8))):
            _, ectx = eng.forward(ids, mask, tt, True, seed=seed, p_out=0.1)
            eng.backward(ectx, dseq0 * mask[:, :, None].float(), accumulate=i > 0)
        torch.cuda.synchronize()
        return {n: eng.fp.view(eng.fp.flat_g, n).clone() for n in layer_names}
    acc_on, acc_off = run_accum(True), run_accum(False)
    for n in layer_names:
        assert torch.equal(acc_on[n], acc_off[n]), n
        assert not torch.equal(acc_on[n], on[n])
    # nothing but padding (no visible key anywhere, an all-zero incoming gradient): zero runs, the weight-gradient GEMM's K loop is empty
    am_keep = am.clone()
    am.zero_()
    on0, guard0 = run(True, torch.zeros_like(dseq0))
    off0, _ = run(False, torch.zeros_like(dseq0))
    am.copy_(am_keep)
    assert guard0 == 0
    for n in layer_names:
        assert torch.equal(on0[n], off0[n]) and float(on0[n].abs().max()) == 0.0, n


# ------------------------------------------------------------------------------------------------ ts_score_predictor = "cos"
# loss_calculator.py:45-48 / utils.py:111-138: the score is sigmoid(cos(eop_i, eop_next) / temp) and the "logits" output is the (B, 2, k) score
# matrix; goldens from the reference itself (tools/gen_golden.py --cos-only).  VERDICT r04: this branch had never been compared with anything.
@pytest.mark.parametrize("precision,tol_score,tol_cos", [("bf16", 5e-3, 2e-2), ("parity", 1e-4, 2e-4)])
@pytest.mark.parametrize("case", ["tiny_L64_cos", "tiny_L128_cos"])
@pytest.mark.parametrize("variant", ["eval_cos", "eval_cos_t05", "full_eval_cos"])
def test_cos_score_predictor_eval_vs_reference_golden(dev, case, variant, precision, tol_score, tol_cos):
    z, sd, batch, arch = load_case(case)
    fl = flags_of(z, variant)
    assert fl["ts_score_predictor"] == "cos"
    fl["amdseg_precision"] = precision
    m = build_model(arch, fl, sd, dev).eval()
    random.seed(int(z[f"{variant}.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**to_dev(batch, dev))
    ref_logits, ref_cos = torch.from_numpy(z[f"{variant}.logits"]), torch.from_numpy(z[f"{variant}.cos"])
    assert logits.shape == ref_logits.shape and cos.shape == ref_cos.shape
    pad = ref_cos == -100
    assert torch.equal(cos.cpu() == -100, pad)
    dc = (cos.cpu() - ref_cos).abs().max().item()
    ds = (logits.cpu() - ref_logits).abs().max().item()
    rl = abs(loss.item() - float(z[f"{variant}.loss"])) / abs(float(z[f"{variant}.loss"]))
    print(f"{case}/{variant}/{precision}: max|dcos| {dc:.2e} max|dscore| {ds:.2e} loss rel {rl:.2e}")
    assert dc < tol_cos * (2.0 if "t05" in variant else 1.0) and ds < tol_score and rl < (2e-4 if precision == "bf16" else 2e-6)
    # the decision of this predictor: score > 0.5 <=> cos > 0 at every real pair
    margin = ref_cos[~pad].abs()
    sure = margin > 2 * tol_cos
    assert torch.equal((cos.cpu()[~pad] > 0)[sure], (ref_cos[~pad] > 0)[sure])


@pytest.mark.parametrize("precision,tol", [("bf16", 0.06), ("parity", 2e-3)])
@pytest.mark.parametrize("case", ["tiny_L64_cos", "tiny_L128_cos"])
@pytest.mark.parametrize("variant", ["train_cos", "train_cos_t05"])
def test_cos_score_predictor_train_grads_vs_reference_golden(dev, case, variant, precision, tol):
    z, sd, batch, arch = load_case(case)
    fl = flags_of(z, variant)
    fl["amdseg_precision"] = precision
    m = build_model(arch, fl, sd, dev).train()
    random.seed(int(z[f"{variant}.random_seed"]))
    loss, logits, cos = m(**to_dev(batch, dev))
    loss.backward()
    ref_loss = float(z[f"{variant}.loss"])
    assert abs(loss.item() - ref_loss) < (2e-4 if precision == "bf16" else 2e-6) * abs(ref_loss)
    assert logits.shape == z[f"{variant}.logits"].shape
    params = dict(m.named_parameters())
    worst = 0.0
    for n, gv in zip(z[f"{variant}.gradnorm_names"].tolist(), z[f"{variant}.gradnorm_vals"].tolist()):
        g = params[n].grad
        mine = 0.0 if g is None else float(g.float().norm())
        if gv < 0:                                      # pooler and BOTH linear heads: no gradient in the reference either
            assert mine == 0.0, n
            continue
        rel = abs(mine - gv) / max(gv, 1e-3)
        worst = max(worst, rel)
        assert rel < tol, (n, mine, gv)
    wc = 1.0
    for k in z.files:
        if k.startswith(f"{variant}.grad."):
            n = k[len(f"{variant}.grad."):]
            ref = torch.from_numpy(z[k])
            if float(ref.norm()) < 1e-5:
                continue
            c = torch.nn.functional.cosine_similarity(params[n].grad.float().cpu().flatten(), ref.flatten(), dim=0).item()
            wc = min(wc, c)
            assert c > (0.99 if precision == "bf16" else 0.99999), (n, c)
    print(f"{case}/{variant}/{precision}: worst grad-norm rel err {worst:.2e}, worst cosine {wc:.6f}")


# ------------------------------------------------------------------------------------------------ pass-through: output_hidden_states
@pytest.mark.parametrize("precision,tol", [("bf16", 0.06), ("parity", 1e-3), ("fp32", 1e-4)])
@pytest.mark.parametrize("case", ["tiny_L64", "tiny_L128", "tiny_L100_B3"])
def test_output_hidden_states_per_layer_vs_reference_golden(dev, case, precision, tol):
    """bert_for_ts.py:57-63,111-112: `output_hidden_states=True` appends the anchor pass's hidden-state tuple (embedding output + one per
    layer) to the return.  The goldens hold the reference's own `plain_eval.hidden{i}`: a PER-LAYER parity check of the encoder."""
    z, sd, batch, arch = load_case(case)
    fl = flags_of(z, "plain_eval")
    fl["amdseg_precision"] = precision
    m = build_model(arch, fl, sd, dev).eval()
    random.seed(int(z["plain_eval.random_seed"]))
    with torch.no_grad():
        out = m(**to_dev(batch, dev), output_hidden_states=True)
        plain = m(**to_dev(batch, dev))
    assert len(out) == 4 and len(plain) == 3
    assert torch.equal(out[1], plain[1]) and torch.equal(out[0], plain[0])       # asking for the states changes nothing else
    hs = out[3]
    n = arch["num_hidden_layers"]
    assert isinstance(hs, tuple) and len(hs) == n + 1
    worst = []
    for i, h in enumerate(hs):
        ref = torch.from_numpy(z[f"plain_eval.hidden{i}"])
        assert h.shape == ref.shape and h.dtype == torch.float32
        keep = batch["attention_mask"][:, 0].bool()                # rows of padding are not defined by either side's contract
        d = (h.cpu() - ref)[keep].abs().max().item()
        worst.append(d)
        assert d < tol * max(1.0, float(ref[keep].abs().max()) / 4), (i, d)
    print(f"{case}/{precision}: max|dhidden| per layer {['%.1e' % w for w in worst]}")
    # training mode too (the training arena keeps every layer input): same values at dropout 0
    m.train()
    random.seed(0)
    out_t = m(**to_dev(batch, dev), output_hidden_states=True)
    # ("fp32" trains in "parity" precision -- split-bf16 products, 2^-16 relative -- so the training-mode values differ from the exact-fp32 eval ones at that level)
    assert len(out_t) == 4 and all((a - b).abs().max().item() < (1e-3 if precision != "bf16" else 2e-2) for a, b in zip(out_t[3], hs))


# ------------------------------------------------------------------------------------------------ the Auto* surface
def test_auto_model_for_token_classification_round_trip(dev, tmp_path):
    """north_star: the AutoModelForTokenClassification plug-in surface.  A config twin (model_type "amdseg-bert") resolves through
    AutoConfig / AutoModelForTokenClassification to the drop-in class; save_pretrained -> from_pretrained gives identical logits; the same
    checkpoint loads through the reference driver's direct-class path (ts_sentence_seq_labeling.py:250-257) with identical logits."""
    import spokennlp_amd
    from transformers import AutoConfig, AutoModelForTokenClassification, BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    z, sd, batch, arch = load_case("tiny_L64")
    fl = flags_of(z, "full_eval")
    cfg = spokennlp_amd.amdseg_config(BertConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **arch), **fl)
    assert cfg.model_type == "amdseg-bert"
    m = AutoModelForTokenClassification.from_config(cfg)
    assert isinstance(m, M)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    random.seed(5)
    with torch.no_grad():
        loss, logits, cos = m(**to_dev(batch, dev))
    ref = build_model(arch, fl, sd, dev).eval()                     # the direct class on a stock BertConfig
    random.seed(5)
    with torch.no_grad():
        loss_r, logits_r, cos_r = ref(**to_dev(batch, dev))
    assert torch.equal(logits, logits_r) and torch.equal(loss, loss_r) and torch.equal(cos, cos_r)
    m.save_pretrained(tmp_path)
    c2 = AutoConfig.from_pretrained(tmp_path)
    assert type(c2).__name__ == "AmdsegBertConfig" and c2.cl_loss_weight == fl["cl_loss_weight"] and c2.do_da_ts is True
    m2 = AutoModelForTokenClassification.from_pretrained(tmp_path).to(dev).eval()
    assert isinstance(m2, M) and type(m2).__name__ == "AmdsegBertForTokenClassification"
    random.seed(5)
    with torch.no_grad():
        _, logits2, _ = m2(**to_dev(batch, dev))
    assert torch.equal(logits2, logits)
    # ... and through the direct-class path of the reference driver, config passed explicitly
    dcfg = BertConfig.from_pretrained(tmp_path)
    m3 = M.from_pretrained(tmp_path, config=dcfg, ignore_mismatched_sizes=True).to(dev).eval()
    random.seed(5)
    with torch.no_grad():
        _, logits3, _ = m3(**to_dev(batch, dev))
    assert torch.equal(logits3, logits)


def test_fused_heads_propagate_a_diverged_step_as_nan(dev):
    """the heads' scatter sums are 64-bit fixed-point integers (bit-reproducible) -- a NaN / Inf contribution converts to 0 / saturates there.
    A diverged step must still reach the gradient norm as NaN (as the float atomics did): the contributing wave poisons the gradient row."""
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    random.seed(7)
    loss, _, _ = m(**to_dev(batch, dev))
    loss.backward()
    gn_ok = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    assert math.isfinite(gn_ok) and gn_ok > 0
    m.zero_grad(set_to_none=False)
    random.seed(7)
    loss, _, _ = m(**to_dev(batch, dev))
    (loss * float("inf")).backward()                       # an overflowed loss scale: every head gradient contribution is Inf / NaN
    gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    assert not math.isfinite(gn)
    emb = m.bert.embeddings.word_embeddings.weight.grad
    assert not bool(torch.isfinite(emb).all())             # ... all the way down the encoder


def test_direct_head_gradients_equal_autograd_ones_over_accumulation(dev):
    """in native mode the fused heads write the classifier / TSSP-classifier gradients straight into their .grad views (accumulate) and hand
    autograd None.  Over two accumulation micro-steps, and after zero_grad(set_to_none=True), they must equal what autograd computes for the same heads
    (amdseg_fused_heads=False: torch formulation)."""
    z, sd, batch, arch = load_case("tiny_L64")
    names = ["loss_calculator.classifier.weight", "loss_calculator.classifier.bias", "loss_calculator.tssp.classifier.weight", "loss_calculator.tssp.classifier.bias"]
    got = {}
    for fused in (True, False):
        m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
        m.config.amdseg_fused_heads = fused
        m.config.amdseg_precision = "parity"
        for rnd in range(2):                               # second round: after zero_grad(set_to_none=True) the views must be re-attached and zeroed
            for micro in range(2):
                random.seed(11 + micro)
                loss, _, _ = m(**to_dev(batch, dev))
                (loss * (0.5 + micro)).backward()
            params = dict(m.named_parameters())
            got[(fused, rnd)] = {n: params[n].grad.detach().float().cpu().clone() for n in names}
            m.zero_grad(set_to_none=True)
    for rnd in range(2):
        for n in names:
            a, b = got[(True, rnd)][n], got[(False, rnd)][n]
            assert float((a - b).norm()) <= 1e-4 * max(float(b.norm()), 1e-3), (rnd, n)
        for n in names:                                    # and the second round equals the first: nothing was carried over a zero_grad
            assert torch.allclose(got[(True, 0)][n], got[(True, 1)][n], rtol=1e-5, atol=1e-7), n


# ------------------------------------------------------------------------------------------------ pass-through: position_ids
@pytest.mark.parametrize("case", ["tiny_L64", "tiny_L100_B3"])
def test_position_ids_pass_through(dev, case):
    """bert_for_ts.py:60,74 forwards `position_ids[:, 0]` / `[:, 1]` to the encoder passes.  Served by the explicit-position path of the embedding kernels:
    arange ids give exactly the default, and permuted ids give exactly what a model whose position table is permuted the same way computes at the default
    positions -- forward and the position-table gradient (scattered through the permutation)."""
    z, sd, batch, arch = load_case(case)
    fl = flags_of(z, "train_full")
    B, _, Lq = batch["input_ids"].shape
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(Lq, generator=g)
    pos_default = torch.arange(Lq)[None, None, :].expand(B, 2, Lq).contiguous()
    pos_perm = perm[None, None, :].expand(B, 2, Lq).contiguous()

    def run(state, position_ids):
        m = build_model(arch, fl, state, dev).train()
        random.seed(5)
        kw = {} if position_ids is None else {"position_ids": position_ids.to(dev)}
        loss, logits, _ = m(**to_dev(batch, dev), **kw)
        loss.backward()
        return loss.item(), logits.detach().float().cpu(), m.bert.embeddings.position_embeddings.weight.grad.detach().float().cpu().clone()

    l0, lg0, gp0 = run(sd, None)
    l1, lg1, gp1 = run(sd, pos_default)
    assert l0 == l1 and torch.equal(lg0, lg1) and torch.allclose(gp0, gp1, rtol=1e-6, atol=1e-7)
    sd_b = dict(sd)
    tab = sd["bert.embeddings.position_embeddings.weight"].clone()
    tab[:Lq] = sd["bert.embeddings.position_embeddings.weight"][perm]
    sd_b["bert.embeddings.position_embeddings.weight"] = tab
    l2, lg2, gp2 = run(sd, pos_perm)                          # original table, permuted ids
    l3, lg3, gp3 = run(sd_b, None)                            # permuted table, default ids
    assert l2 == l3 and torch.equal(lg2, lg3)
    assert l2 != l0                                           # (the permutation does change the model)
    assert torch.allclose(gp2[perm], gp3[:Lq], rtol=1e-5, atol=1e-6)
    with pytest.raises(Exception):
        m = build_model(arch, fl, sd, dev).eval()
        m(**to_dev(batch, dev), position_ids=pos_perm[:, :, :-1].to(dev))
