"""ln_bwd stand-alone, parts switched off one at a time (what does each cost?): full (dropout 0.1 + column partials + second stage), no dropout (no
dbranch store, no hash), no column sums (no partials, no second stage), neither; raw C-ABI calls over rotating buffers, preallocated outputs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spokennlp_amd import lib as L, ops
dev = torch.device("cuda:0")
M, H, NB = int(os.environ.get("LN_M", 16384)), 768, 12
g = torch.Generator(device=dev).manual_seed(0)
dy = [torch.randn(M, H, device=dev, generator=g).bfloat16() for _ in range(NB)]
z = [torch.randn(M, H, device=dev, generator=g).bfloat16() for _ in range(NB)]
dz = [torch.empty(M, H, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
dbr = [torch.empty(M, H, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
mean = torch.randn(M, device=dev, generator=g) * 0.1
rstd = torch.rand(M, device=dev, generator=g) + 0.5
gamma = torch.randn(H, device=dev, generator=g)
dg, db, dbi = (torch.zeros(H, device=dev) for _ in range(3))
part = torch.empty(ops.ln_partials_numel(M, H), device=dev)
lib = L.load()
s = torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()      # noqa: E731


def run(i, p, sums):
    k = i % NB
    rc = lib.amdseg_ln_bwd(P(dy[k]), P(z[k]), P(mean), P(rstd), P(gamma), P(dz[k]), P(dbr[k]) if p > 0 else None, P(part) if sums else None,
                           P(dg) if sums else None, P(db) if sums else None, P(dbi) if sums else None, M, H, p, 3, 1, L.BF16, s)
    assert rc == 0, rc


for name, p, sums in (("full", 0.1, True), ("no dropout", 0.0, True), ("no column sums", 0.1, False), ("neither", 0.0, False)):
    for i in range(10):
        run(i, p, sums)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 200
    e0.record()
    for i in range(N):
        run(i, p, sums)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / N * 1e3
    mb = M * H * 2 * (3 + (1 if p > 0 else 0)) / 1e6
    print(f"{name:16s}: {t:6.1f} us per call  ({mb:.0f} MB of rows -> {mb / t / 1e3 * 1e3 / 1e3:.2f} TB/s)")
