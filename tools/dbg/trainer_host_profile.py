import os, sys, cProfile, pstats, io, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(model="bert", workload="full_da", seq_len=512, seqs_per_gpu=32, mode="train")
dev = torch.device("cuda:0")
pr = cProfile.Profile()
import bench as B
orig = B.via_trainer
pr.enable()
out = B.via_trainer(args, dev, nsteps=40, nwarm=8)
pr.disable()
print(out)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue()[:12000])
