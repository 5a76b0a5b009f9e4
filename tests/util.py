"""shared test helpers (no product code, no oracle code)."""
import torch


def tiny_state_dict(arch, seed=0, std=0.05, clf_std=0.3):
    """random fp32 state dict with the HF parameter names of BertWithDAForSentenceLabelingTopicSegmentation."""
    g = torch.Generator().manual_seed(seed)
    H, I, V = arch["hidden_size"], arch["intermediate_size"], arch["vocab_size"]
    P, T, NL = arch["max_position_embeddings"], arch["type_vocab_size"], arch["num_hidden_layers"]

    def w(*s, sd=std):
        return torch.randn(*s, generator=g) * sd

    def ln():
        return 1.0 + 0.1 * torch.randn(H, generator=g), 0.05 * torch.randn(H, generator=g)

    sd = {"bert.embeddings.word_embeddings.weight": w(V, H), "bert.embeddings.position_embeddings.weight": w(P, H),
          "bert.embeddings.token_type_embeddings.weight": w(T, H)}
    sd["bert.embeddings.word_embeddings.weight"][0] = 0
    sd["bert.embeddings.LayerNorm.weight"], sd["bert.embeddings.LayerNorm.bias"] = ln()
    for i in range(NL):
        p = f"bert.encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            sd[p + n + ".weight"] = w(H, H); sd[p + n + ".bias"] = w(H)
        sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"] = ln()
        sd[p + "intermediate.dense.weight"] = w(I, H); sd[p + "intermediate.dense.bias"] = w(I)
        sd[p + "output.dense.weight"] = w(H, I); sd[p + "output.dense.bias"] = w(H)
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = ln()
    sd["bert.pooler.dense.weight"] = w(H, H); sd["bert.pooler.dense.bias"] = w(H)
    sd["loss_calculator.classifier.weight"] = w(2, H, sd=clf_std); sd["loss_calculator.classifier.bias"] = w(2)
    sd["loss_calculator.tssp.classifier.weight"] = w(3, H, sd=clf_std); sd["loss_calculator.tssp.classifier.bias"] = w(3)
    return sd


def longformer_state_dict(arch, seed=0, std=0.03, clf_std=0.3):
    """the same for LongformerWithDAForSentenceLabelingTopicSegmentation (`longformer.*` names, *_global projections)"""
    sd = {}
    for k, v in tiny_state_dict(arch, seed, std, clf_std).items():
        if "pooler" in k:                       # the reference builds LongformerModel(add_pooling_layer=False)
            continue
        sd[k.replace("bert.", "longformer.", 1) if k.startswith("bert.") else k] = v
    g = torch.Generator().manual_seed(seed + 1)
    H = arch["hidden_size"]
    for i in range(arch["num_hidden_layers"]):
        p = f"longformer.encoder.layer.{i}.attention.self."
        for n in ("query_global", "key_global", "value_global"):
            sd[p + n + ".weight"] = torch.randn(H, H, generator=g) * std
            sd[p + n + ".bias"] = torch.randn(H, generator=g) * std
    pad = arch.get("pad_token_id", 1)
    sd["longformer.embeddings.word_embeddings.weight"][pad] = 0
    sd["longformer.embeddings.position_embeddings.weight"][pad] = 0
    return sd
