"""ln_bwd standalone: back-to-back launches over rotating buffers (no MALL hits), bf16 [16384, 768], dropout 0.1, all column partials.
AMDSEG_LIB=<other build> python tools/dbg/ln_bwd_probe.py  compares builds."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spokennlp_amd import ops
dev = torch.device("cuda:0")
M, H, NB = 16384, 768, 12
g = torch.Generator(device=dev).manual_seed(0)
dy = [torch.randn(M, H, device=dev, generator=g).bfloat16() for _ in range(NB)]
z = [torch.randn(M, H, device=dev, generator=g).bfloat16() for _ in range(NB)]
mean = torch.randn(M, device=dev, generator=g) * 0.1
rstd = torch.rand(M, device=dev, generator=g) + 0.5
gamma = torch.randn(H, device=dev, generator=g)
dg, db, dbi = (torch.zeros(H, device=dev) for _ in range(3))
part = torch.empty(ops.ln_partials_numel(M, H), device=dev)
def run(i):
    return ops.ln_bwd(dy[i % NB], z[i % NB], mean, rstd, gamma, p=0.1, seed=3, dgamma=dg, dbeta=db, dbias=dbi, accumulate=True, partials=part)
for i in range(10): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 200
e0.record()
for i in range(N): run(i)
e1.record(); torch.cuda.synchronize()
dz, dbr = run(0); torch.cuda.synchronize()
print(f"{os.environ.get('AMDSEG_LIB', 'in-tree')}: ln_bwd {e0.elapsed_time(e1) / N * 1e3:.1f} us per call (incl. its reduce + torch.empty)  checksum {float(dz.float().sum()):.4f} {float(dbr.float().sum()):.4f} {float(dg.sum()):.3f}")
