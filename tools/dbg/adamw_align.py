#!/usr/bin/env python
"""AdamW pass time vs the RELATIVE placement of its five streams (p, g, m, v fp32 + bf16 shadow) in HBM: the kernel reads / writes element i
of every buffer at the same moment, so buffers whose base addresses differ by a multiple of the channel-interleave period hit the same
channels together.  Carves the buffers out of one allocation with a per-buffer skew and times amdseg_adamw (bert-base: 109.5 M parameters)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spokennlp_amd import ops

dev = torch.device("cuda:0")
N = 109_486_848            # bert-base(+[BOS]) parameters rounded to 64
GRAN = 256


def run(skews, reps=8):
    nb = N * 4
    span = ((nb + (1 << 21) - 1) >> 21) << 21          # 2 MiB-aligned slots
    big = torch.empty(5 * span + (64 << 20), dtype=torch.uint8, device=dev)
    base = big.data_ptr()
    off0 = (-base) % (1 << 21)
    views = []
    for k in range(5):
        o = off0 + k * span + skews[k]
        n = N * (2 if k == 4 else 4)
        t = big[o:o + n].view(torch.bfloat16 if k == 4 else torch.float32)
        views.append(t)
    p, g, m, v, sh = views
    p.normal_(); g.normal_(); m.zero_(); v.zero_()
    coef = torch.ones(1, device=dev)
    for _ in range(2):
        ops.adamw(p, g, m, v, sh, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, gscale=coef, zero_grad=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.adamw(p, g, m, v, sh, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, gscale=coef, zero_grad=True)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    print("bytes per pass:", N * 34 / 1e9, "GB")
    cases = {"aligned (2 MiB slots, skew 0)": [0, 0, 0, 0, 0]}
    for s in (256, 1024, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20):
        cases[f"skew k*{s}"] = [k * s for k in range(5)]
    cases["skew k*(4096+256)"] = [k * 4352 for k in range(5)]
    cases["skew k*(65536+4096+256)"] = [k * (65536 + 4352) for k in range(5)]
    cases["skew primes*256"] = [0, 3 * 256, 7 * 256, 13 * 256, 29 * 256]
    cases["skew primes*4096"] = [0, 3 * 4096, 7 * 4096, 13 * 4096, 29 * 4096]
    for name, sk in cases.items():
        t = run(sk)
        print(f"{name:32s} {t:8.1f} us  {N * 34 / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
