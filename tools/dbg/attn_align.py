#!/usr/bin/env python
"""attention backward (dQ / dK+dV) time vs the placement of its buffers: the in-step launch timer shows both kernels in two modes from one
process to the next (dQ 53-55 / 60-61 us, dK+dV 60-61 / 67-68 us) -- buffer addresses are the only thing that differs.  Carves
qkv / ctx / dctx / dqkv / lse / delta / keep out of one allocation with per-buffer skews and reads the launch timer per kernel class."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spokennlp_amd import lib as L

dev = torch.device("cuda:0")
B, Lq, heads, H = 32, 512, 12, 768
M = B * Lq
lib = L.load()
keep_bytes = lib.amdseg_attn_keepmask_bytes(B, Lq, heads)
SIZES = dict(qkv=M * 3 * H * 2, ctx=M * H * 2, dctx=M * H * 2, dqkv=M * 3 * H * 2, lse=B * heads * Lq * 4, delta=B * heads * Lq * 4, keep=keep_bytes)


def run(skews, reps=12):
    slot = 1 << 27                                           # 128 MiB slots
    big = torch.empty(len(SIZES) * slot + (1 << 26), dtype=torch.uint8, device=dev)
    off0 = (-big.data_ptr()) % (1 << 21)
    ptr = {}
    for i, (k, n) in enumerate(SIZES.items()):
        ptr[k] = big.data_ptr() + off0 + i * slot + skews.get(k, 0)
    qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
    C.memmove  # noqa
    torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(C.c_void_p(ptr["qkv"]), C.c_void_p(qkv.data_ptr()), C.c_size_t(SIZES["qkv"]), 3)
    dctx = torch.randn(M, H, device=dev).bfloat16()
    hip.hipMemcpy(C.c_void_p(ptr["dctx"]), C.c_void_p(dctx.data_ptr()), C.c_size_t(SIZES["dctx"]), 3)
    mb = torch.zeros(B, Lq, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.amdseg_attn_keepmask(ptr["keep"], B, Lq, heads, 0.1, 7, None, s), "keepmask")
    L.check(lib.amdseg_attn_fwd_keep(ptr["qkv"], mb.data_ptr(), ptr["ctx"], ptr["lse"], B, Lq, heads, 0.125, 0.1, ptr["keep"], s), "fwd")

    def bwd():
        L.check(lib.amdseg_attn_bwd_keep(ptr["qkv"], mb.data_ptr(), ptr["ctx"], ptr["dctx"], ptr["lse"], ptr["delta"], ptr["dqkv"], B, Lq, heads,
                                         0.125, 0.1, ptr["keep"], s), "bwd")
    for _ in range(3):
        bwd()
    lib.amdseg_prof_enable(1); lib.amdseg_prof_reset()
    for _ in range(reps):
        L.check(lib.amdseg_attn_fwd_keep(ptr["qkv"], mb.data_ptr(), ptr["ctx"], ptr["lse"], B, Lq, heads, 0.125, 0.1, ptr["keep"], s), "fwd")
        bwd()
    torch.cuda.synchronize()
    out = []
    for cls in (2, 3, 4):
        us, work, n = C.c_double(), C.c_double(), C.c_longlong()
        lib.amdseg_prof_read(cls, C.byref(us), C.byref(work), C.byref(n))
        out.append(us.value / max(n.value, 1))
    lib.amdseg_prof_enable(0)
    return out


def main():
    cases = {"all aligned to 2 MiB": {}}
    for s_ in (4096, 65536, 1 << 20):
        cases[f"every buffer skewed by k*{s_}"] = {k: i * s_ for i, k in enumerate(SIZES)}
    for k in ("qkv", "dqkv", "dctx", "ctx", "keep", "lse"):
        cases[f"{k} + 1 MiB + 4 KiB"] = {k: (1 << 20) + 4096}
    cases["dqkv + 192 KiB"] = {"dqkv": 196608}
    cases["keep + 64 KiB"] = {"keep": 65536}
    for name, sk in cases.items():
        f, dq, dkv = run(sk)
        print(f"{name:40s} fwd {f:6.1f}  dq {dq:6.1f}  dkv {dkv:6.1f} us")


if __name__ == "__main__":
    main()
