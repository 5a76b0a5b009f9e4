import os, sys, random, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import argparse, bench
from spokennlp_amd.dp import GradBuckets
args = argparse.Namespace(model="bert", seq_len=512, seqs_per_gpu=32, workload="full_da", mode="train")
dev = torch.device("cuda:0")
model, cfg = bench.build(args, dev)
eng = model.engine()
eng.buckets = GradBuckets(eng.fp)          # what enable_data_parallel() does for world > 1
batches, _ = bench.make_batches(args, 4, seed=0, device=dev)
def step(i):
    random.seed(i)
    loss = model(**batches[i % 4])[0]
    loss.backward()
    eng.finish_grad_sync()
    eng.adamw_step(5e-5, max_grad_norm=1.0, grad_scale=1.0)
    return loss
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3, 13): l = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"nccl world=1 with per-layer buckets: {dt*1e3:.2f} ms/step, loss {l.item():.4f}")
dist.destroy_process_group()
