"""fused AdamW stand-alone: does the per-byte rate depend on how the five streams (p, g, m, v, bf16 copies) sit relative to each other in HBM?
The buffers are carved out of one arena with a skew of k * skew bytes in front of the k-th buffer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spokennlp_amd import ops
dev = torch.device("cuda:0")
for n in (109_486_848, 148_700_160):
    n = n // 64 * 64
    for skew in (0, 4096, 65536, 1 << 18, (1 << 18) + 4096, 1 << 20, (1 << 20) + 65536 + 4096, 3 << 19):
        span = ((n * 4 + (2 << 20) - 1) >> 21) << 21                       # every buffer starts 2 MiB aligned + its skew
        arena = torch.empty(5 * (span + (4 << 20)), dtype=torch.uint8, device=dev)
        def carve(k, nbytes, dtype):
            off = k * (span + (2 << 20)) + k * skew
            return arena[off:off + nbytes].view(dtype)
        p, g, m, v = (carve(k, n * 4, torch.float32) for k in range(4))
        sh = carve(4, n * 2, torch.bfloat16)
        for t in (p, g, m):
            t.normal_(0, 0.01)
        v.uniform_(0, 1e-4)
        flags = torch.ones(n // 64, dtype=torch.uint8, device=dev)
        gs = torch.ones(1, device=dev)
        for _ in range(3):
            ops.adamw(p, g, m, v, sh, 1e-5, 0.9, 0.999, 1e-8, 0.0, 3, gscale=gs, zero_grad=True, chunk_flags=flags)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.adamw(p, g, m, v, sh, 1e-5, 0.9, 0.999, 1e-8, 0.0, 3, gscale=gs, zero_grad=True, chunk_flags=flags)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"n = {n / 1e6:7.1f} M  skew {skew:8d}: {t * 1e6:7.1f} us  {n * 34 / t / 1e12:5.2f} TB/s")
        del arena, p, g, m, v, sh
        torch.cuda.empty_cache()
