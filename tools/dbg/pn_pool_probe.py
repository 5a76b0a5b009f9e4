"""time the PoNet pooling kernels on synthetic runs of a given length (rocprofv3 --kernel-trace --stats around this script)"""
import sys, torch
sys.path.insert(0, ".")
from spokennlp_amd import ops
B, L, H = 8, 4096, 768
runlen = int(sys.argv[1]) if len(sys.argv) > 1 else 70
dev = torch.device("cuda")
torch.manual_seed(0)
proj = torch.randn(B * L, 5 * H, device=dev).to(torch.bfloat16)
pos = torch.arange(L, device=dev)
rs = ((pos // runlen) * runlen).to(torch.int32).repeat(B).contiguous()
re = torch.clamp(rs + runlen - 1, max=L - 1).to(torch.int32)
mb = torch.zeros(B * L, device=dev)
g = torch.randn(B, H, device=dev)
part = torch.empty(3 * B * L, H, dtype=torch.bfloat16, device=dev); parg = torch.empty(3 * B * L, H, dtype=torch.int16, device=dev)
ctx = torch.empty(B * L, H, dtype=torch.bfloat16, device=dev)
dctx = torch.randn(B * L, H, device=dev).to(torch.bfloat16)
dproj = torch.empty_like(proj)
psum = torch.empty(3 * B * L, H, dtype=torch.float32, device=dev)
work = ops.ponet_plan(mb, rs, B, L)
for _ in range(20):
    ops.ponet_pool_fwd(proj, mb, rs, re, work, g, part, parg, ctx, B, L, H)
    ops.ponet_pool_bwd(proj, mb, rs, re, work, g, part, parg, dctx, dproj, psum, B, L, H)
torch.cuda.synchronize()
