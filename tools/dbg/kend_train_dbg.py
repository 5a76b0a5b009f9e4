import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transformers import BertConfig
from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
for pa, ph in ((0.1, 0.0), (0.0, 0.1), (0.1, 0.1)):
    cfg = BertConfig(vocab_size=300, hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=256,
                     max_position_embeddings=512, num_labels=2, hidden_dropout_prob=ph, attention_probs_dropout_prob=pa)
    m = M(cfg).to(dev).train()
    B, L = 8, 256
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 300, (B, L), generator=g).to(dev)
    lens = [256, 200, 130, 64, 63, 256, 1, 129]
    am = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens): am[b, :n] = 1
    am = am.to(dev); tt = torch.zeros_like(ids)
    eng = m.engine()
    outs = []
    for skip in (True, False, True):
        eng.skip_padded_chunks = skip
        out, ctx = eng.forward(ids, am, tt, True, seed=1234, p_out=0.0)
        outs.append(out.clone())
    print(f"p_attn {pa} p_hidden {ph}: skip vs noskip max diff {float((outs[0]-outs[1]).abs().max()):.3e}; skip vs skip {float((outs[0]-outs[2]).abs().max()):.3e}")
    d = (outs[0]-outs[1]).abs().view(B, L, -1).amax(-1)
    print("   rows differing per sequence:", [(int((d[b] > 0).sum())) for b in range(B)])
