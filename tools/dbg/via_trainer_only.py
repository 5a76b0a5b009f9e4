"""run only the bench's via_trainer leg (for rocprofv3 --kernel-trace + tools/trace_gaps.py)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(model="bert", workload="full_da", seq_len=512, seqs_per_gpu=32, mode="train", precision="bf16")
print(bench.via_trainer(args, torch.device("cuda:0"), nsteps=30, nwarm=8, nan_filter=True))
