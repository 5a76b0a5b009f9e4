"""mmvts text branch on the GPU: spokennlp_amd.mmvts_text_encoder.TextEncoder against golden vectors produced by the
reference's own TextEncoder (mmvts/src/models/text_encoder/text_encoder.py) over BertModel / LongformerModel."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from tests.test_oracle_golden import mmvts_case  # noqa: E402


def build(kind, sd, dev, precision=None):
    from transformers import BertConfig, LongformerConfig
    from spokennlp_amd.mmvts_text_encoder import TextEncoder
    base = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    if kind == "bert":
        cfg = BertConfig(max_position_embeddings=128, type_vocab_size=2, **base)
        cfg.text_encoder_name_or_path = "tiny_bert"
    else:
        cfg = LongformerConfig(max_position_embeddings=258, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                               layer_norm_eps=1e-5, attention_window=[32, 64], **base)
        cfg.text_encoder_name_or_path = "tiny_longformer_zh"
    cfg.init_model = False
    if precision:
        cfg.amdseg_precision = precision
    m = TextEncoder(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
    return m.to(dev)


@pytest.mark.parametrize("case,kind", [("mmvts_text_bert_L128", "bert"), ("mmvts_text_lf_L256", "lf")])
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_text_encoder_eval_vs_reference_golden(dev, case, kind, precision):
    z, sd, ins = mmvts_case(case)
    m = build(kind, sd, dev, precision).eval()
    assert m.encoder_type == kind
    with torch.no_grad():
        f = m(ins["input_ids"].to(dev), attention_mask=ins["attention_mask"].to(dev), token_type_ids=ins["token_type_ids"].to(dev))
    valid = ins["attention_mask"].bool()
    d = (f.cpu() - torch.from_numpy(z["eval.features"]))[valid].abs().max().item()
    print(f"{case}/{precision}: max|dfeature| {d:.2e}")
    assert d < (1e-3 if precision == "fp32" else 0.06)


@pytest.mark.parametrize("case,kind", [("mmvts_text_bert_L128", "bert"), ("mmvts_text_lf_L256", "lf")])
def test_text_encoder_train_grads_vs_reference_golden(dev, case, kind):
    """differentiable through the reference's torch layers on top: d sum(features * weights) / d parameters"""
    z, sd, ins = mmvts_case(case)
    m = build(kind, sd, dev).train()
    f = m(ins["input_ids"].to(dev), attention_mask=ins["attention_mask"].to(dev), token_type_ids=ins["token_type_ids"].to(dev))
    (f * ins["loss_weights"].to(dev)).sum().backward()
    params = dict(m.named_parameters())
    checked = 0
    for k in z.files:
        if not k.startswith("train.grad."):
            continue
        n = k[len("train.grad."):]
        ref = torch.from_numpy(z[k])
        if float(ref.norm()) < 1e-4:
            continue
        g = params[n].grad.float().cpu()
        c = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
        rel = abs(float(g.norm()) - float(ref.norm())) / float(ref.norm())
        assert c > 0.99 and rel < 0.06, (n, c, rel)
        checked += 1
    assert checked > 30
    if kind == "lf":          # no global token: the *_global projections are unused, as in the reference run
        assert all(float(params[n].grad.abs().max()) == 0 for n in params if "_global" in n and params[n].grad is not None)


def test_text_encoder_rejects_global_mask(dev):
    from spokennlp_amd import lib as L
    z, sd, ins = mmvts_case("mmvts_text_lf_L256")
    m = build("lf", sd, dev).eval()
    g = torch.zeros_like(ins["input_ids"]); g[:, 0] = 1
    with pytest.raises(L.AmdsegError):
        m(ins["input_ids"].to(dev), attention_mask=ins["attention_mask"].to(dev), global_attention_mask=g.to(dev))
