"""inter-kernel dispatch cost on this box: N back-to-back tiny launches, eager (ctypes -> hipLaunchKernelGGL) vs a captured graph"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from spokennlp_amd import ops
dev = torch.device("cuda")
x = torch.ones(64 * 1024, device=dev); coef = torch.ones(1, device=dev)
big = torch.randn(16384, 768, device=dev).bfloat16(); W = (torch.randn(768, 768, device=dev) * .05).bfloat16(); out = torch.empty(16384, 768, device=dev, dtype=torch.bfloat16)
def run_small(n):
    for _ in range(n): ops.scale_(x, coef)
def run_mixed(n):
    for _ in range(n):
        ops.gemm_nt(big, W, ops.EPI_NONE, out=out); ops.scale_(x, coef)
def timeit(f, n):
    f(10); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); f(n); e1.record(); t1 = time.perf_counter(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) * 1e6 / n
print("eager tiny: gpu %.2f us/launch, cpu %.2f us/launch" % timeit(run_small, 2000))
print("eager gemm+tiny pair: gpu %.2f us/pair, cpu %.2f us/pair" % timeit(run_mixed, 500))
# single gemm alone
def run_gemm(n):
    for _ in range(n): ops.gemm_nt(big, W, ops.EPI_NONE, out=out)
print("eager gemm: gpu %.2f us, cpu %.2f us" % timeit(run_gemm, 500))
# graph
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run_mixed(3)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run_mixed(200)
torch.cuda.synchronize()
def replay(n):
    for _ in range(n): g.replay()
a, b = timeit(replay, 10)
print("graph gemm+tiny pair: gpu %.2f us/pair, cpu %.2f us/pair" % (a / 200, b / 200))
with torch.cuda.stream(s):
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        run_small(1000)
torch.cuda.synchronize()
def replay2(n):
    for _ in range(n): g2.replay()
a, b = timeit(replay2, 10)
print("graph tiny: gpu %.2f us/launch, cpu %.2f us/launch" % (a / 1000, b / 1000))
