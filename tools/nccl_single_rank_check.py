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
# the same schedule through the C ABI's exchange (amdseg_allreduce_*, AMDSEG_DP_NATIVE_COMM=1) and with every bucket in bf16 on the wire
# (AMDSEG_DP_WIRE=bf16): one step each from the same state; with one rank the sum is the identity, so fp32 gradients must come out bit-identical
import copy
def one_step_grads(buckets):
    eng.buckets = buckets
    random.seed(99)
    loss = model(**batches[0])[0]
    loss.backward()
    eng.finish_grad_sync()
    torch.cuda.synchronize()
    g = eng.fp.flat_g.detach().clone()
    sched, onwire = list(buckets.log), buckets.bytes_on_wire()
    eng.fp.flat_g.zero_(); eng.fp.grad_stale = False; eng.fp.grad_is_zero = True
    buckets.reset_norm()
    return g, sched, onwire
model.amdseg_seed = 5; model._step_seed = 100
g_t, s_t, w_t = one_step_grads(GradBuckets(eng.fp))
model._step_seed = 100
nat = GradBuckets(eng.fp, native=True)
g_n, s_n, w_n = one_step_grads(nat)
model._step_seed = 100
g_b, s_b, w_b = one_step_grads(GradBuckets(eng.fp, wire="bf16"))
same = torch.equal(g_t, g_n)
rel_b = float((g_b - g_t).norm() / g_t.norm())
print(f"native comm: rccl ranks {nat.native.world}, schedule equal {s_t == s_n} ({len(s_n)} buckets, {w_n} B on the wire), gradients bit-identical {same}")
print(f"bf16 wire: schedule slices equal {[(a, b) for a, b, _ in s_b] == [(a, b) for a, b, _ in s_t]}, {w_b} B on the wire, relative gradient difference {rel_b:.2e}")
nat.native.close()
dist.destroy_process_group()
