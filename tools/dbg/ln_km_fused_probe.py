"""A/B of amdseg_add_ln_fwd + amdseg_attn_keepmask (two launches) against amdseg_add_ln_fwd_keepmask (one grid of interleaved workgroups) at the
bert-base 32 x 512 shape, then of the training step with engine.keepmask_in_ln on / off in one process (medians).  Run on the GPU box."""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
from spokennlp_amd import ops  # noqa: E402


def t_us(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def op_level():
    dev = torch.device("cuda:0")
    for B, L, heads, H in [(32, 512, 12, 768), (8, 512, 12, 768), (16, 1024, 16, 1024), (8, 1024, 12, 768), (64, 256, 12, 768), (16, 512, 16, 1024),
                           (4, 2048, 12, 768), (64, 128, 12, 768), (32, 512, 8, 512), (32, 512, 12, 1536)]:
        M = B * L
        y = torch.randn(M, H, device=dev).bfloat16()
        res = torch.randn(M, H, device=dev).bfloat16()
        g = torch.ones(H, device=dev)
        b = torch.zeros(H, device=dev)
        bits = torch.zeros(M * H // 8, dtype=torch.uint8, device=dev)
        ln = t_us(lambda: ops.add_ln_fwd(y, res, g, b, 1e-12, 0.1, 7))
        km = t_us(lambda: ops.attn_keepmask(B, L, heads, 0.1, 9, dev))
        both = t_us(lambda: (ops.add_ln_fwd(y, res, g, b, 1e-12, 0.1, 7), ops.attn_keepmask(B, L, heads, 0.1, 9, dev)))
        fused = t_us(lambda: ops.add_ln_fwd_keepmask(y, res, g, b, 1e-12, 0.1, 7, B, L, heads, 0.1, 9, drop_bits=bits))
        print(f"B={B} L={L} heads={heads} H={H}: add_ln {ln:.1f} us, keepmask {km:.1f} us, back to back {both:.1f} us, one launch {fused:.1f} us", flush=True)


def step_level(seqs=32, seq_len=512):
    import argparse
    import random
    import bench
    dev = torch.device("cuda:0")
    a = argparse.Namespace(model="bert", workload="full_da", precision="bf16", seqs_per_gpu=seqs, seq_len=seq_len, mode="train")
    runs = {}
    for flag in (True, False):
        model, cfg = bench.build(a, dev)
        eng = model.engine()
        eng.keepmask_in_ln = flag
        batches, _ = bench.make_batches(a, 8, seed=7, device=dev)

        def step(i, model=model, eng=eng, batches=batches):
            random.seed(i)
            loss = model(**batches[i % len(batches)])[0]
            loss.backward()
            eng.finish_grad_sync()
            eng.adamw_step(5e-5, max_grad_norm=1.0, grad_scale=1.0)
            return loss
        runs[flag] = step
    for f in runs:
        for i in range(10):
            runs[f](i)
    torch.cuda.synchronize()
    res = {True: [], False: []}
    for rep in range(4):                                   # alternate the two in one process: box speed and clocks are shared
        for f in (True, False):
            ts = []
            for i in range(25):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                runs[f](100 + i)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res[f].append(statistics.median(ts))
            print(f"{seqs} x {seq_len} keepmask_in_ln={f}: median step {statistics.median(ts):.3f} ms (min {min(ts):.3f})", flush=True)
    print({k: [round(x, 3) for x in v] for k, v in res.items()}, flush=True)


if __name__ == "__main__":
    op_level()
    if "--step" in sys.argv:
        step_level()
        step_level(8, 512)
