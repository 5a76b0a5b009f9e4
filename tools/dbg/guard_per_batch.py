"""does amdseg_pad_rows_guard trip on the bench's batches?  guard value, padded-row fraction and the TN kernel's in-step time per batch"""
import os, sys, argparse, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "bert"
args = argparse.Namespace(model=fam, workload="full_da", seq_len=512 if fam == "bert" else 4096, seqs_per_gpu=32 if fam == "bert" else (8 if fam == "ponet" else 4),
                          mode="train", precision="bf16")
model, cfg = bench.build(args, dev)
eng = model.engine()
batches, _ = bench.make_batches(args, 8, seed=0, device=dev)
for i, b in enumerate(batches):
    for _ in range(2):
        model.zero_grad(set_to_none=False); loss = model(**b)[0]; loss.backward()
    bench.prof_arm()
    for _ in range(3):
        loss = model(**b)[0]; loss.backward()
    k = bench.prof_collect(3)
    am = b["attention_mask"].reshape(-1, args.seq_len)
    kend = ((am != 0).long() * torch.arange(1, args.seq_len + 1, device=dev)[None, :]).amax(dim=1)
    print(f"batch {i}: guard {int(eng._pad_guard.item())} valid rows {float((am != 0).float().mean()):.3f} tiles walked {float(((kend + 63) // 64).sum()) / (am.shape[0] * args.seq_len // 64):.3f} "
          f"TN {k['gemm_tn_dp_kernel']['avg_launch_us']:.1f} us dkv {k['attn_bwd_dkv_kernel']['avg_launch_us']:.1f}")
