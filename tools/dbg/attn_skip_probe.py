import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spokennlp_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
B, L, heads = 32, 512, 12
qkv = (torch.randn(B * L, 3 * heads * 64, device=dev) * 0.5).bfloat16()
for valid in (512, 448, 256, 128):
    mb = torch.zeros(B, L, device=dev); mb[:, valid:] = -30000.0
    mb = mb.reshape(-1).contiguous()
    t = timeit(lambda: ops.attn_fwd(qkv, mb, B, L, heads, p=0.1, seed=3))
    print(f"valid {valid}: fwd {t*1e6:.1f} us")
