#!/usr/bin/env python
"""Round 6 probe: do the dropout keep masks of the NEXT step (attn_keepmask_kernel: VALU-bound, no LDS, small register footprint) hide under the
HBM-bound kernels that end a step (AdamW over 109.5 M parameters) when issued on a second stream?   python tools/dbg/km_overlap_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spokennlp_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
N = 109_486_856
p = torch.randn(N, device=dev) * 0.02
g = torch.randn(N, device=dev) * 1e-3
m = torch.zeros(N, device=dev); v = torch.zeros(N, device=dev)
sh = torch.empty(N, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, device=dev)
B, Lq, heads = 32, 512, 12
nb = lib.amdseg_attn_keepmask_bytes(B, Lq, heads)
keeps = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(12)]
side = torch.cuda.Stream()


def adamw(step):
    ops.adamw(p, g, m, v, sh, 5e-5, 0.9, 0.999, 1e-8, 0.0, step, gscale=coef, zero_grad=False)


def masks(stream):
    for i, k in enumerate(keeps):
        L.check(lib.amdseg_attn_keepmask(k.data_ptr(), B, Lq, heads, 0.1, 1234 + i, None, stream), "km")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


step = [0]


def only_adamw():
    step[0] += 1; adamw(step[0])


def only_masks():
    masks(torch.cuda.current_stream().cuda_stream)


def serial():
    only_adamw(); only_masks()


def overlapped():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    masks(side.cuda_stream)
    done = torch.cuda.Event(); done.record(side)
    only_adamw()
    main.wait_event(done)


for rep in range(3):
    print(f"rep {rep}: adamw {timed(only_adamw):.1f} us | 12 keep-mask launches {timed(only_masks):.1f} us | serial {timed(serial):.1f} us | "
          f"two streams {timed(overlapped):.1f} us", flush=True)
