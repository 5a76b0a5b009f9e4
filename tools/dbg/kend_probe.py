import os, sys, argparse, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from spokennlp_amd import lib as L
dev = torch.device("cuda:0")
args = argparse.Namespace(model="bert", workload="full_da", seq_len=512, seqs_per_gpu=32, mode="train", precision="bf16")
model, cfg = bench.build(args, dev)
eng = model.engine()
batches, _ = bench.make_batches(args, 2, seed=0, device=dev)
b = {k: v.clone() for k, v in batches[0].items()}
for valid in ("23 full + 9 x 128", "interleaved"):
    am = b["attention_mask"]; am[:] = 1
    flat = am.view(-1, am.shape[-1]) if am.dim() == 3 else am
    if valid == "interleaved":
        for i in range(0, flat.shape[0], 4): flat[i, 128:] = 0
    else:
        flat[23:, 128:] = 0
    for skip in (True, False):
        eng.skip_padded_chunks = skip
        for _ in range(3):
            loss = model(**b)[0]; loss.backward()
        bench.prof_arm()
        for _ in range(5):
            loss = model(**b)[0]; loss.backward()
        k = bench.prof_collect(5)
        print(f"valid {valid} skip {skip}: fwd {k['attn_fwd_kernel']['avg_launch_us']} dq {k['attn_bwd_dq_kernel']['avg_launch_us']} dkv {k['attn_bwd_dkv_kernel']['avg_launch_us']}")
