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
