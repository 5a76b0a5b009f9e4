"""Pure data parallelism over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl")
all-reduce of the flat fp32 gradient buffer over xGMI, launched per encoder layer as soon as that layer's weight
gradients exist so the exchange hides under the rest of backward.

The reference gets this from torch DDP via `python -m torch.distributed.launch` (run_finetune.sh:61) + HF Trainer; here
parameter gradients are written by HIP kernels straight into one flat buffer (engine.FlatParams), whose per-layer
slices are contiguous, so each bucket is a plain slice -- no gradient copies, no autograd hooks.
Buckets: [layer 11] ... [layer 0] [embeddings + heads], summed; the mean (1/world) is folded into the clip/AdamW
gradient scale (engine.adamw_step(grad_scale=1/world)).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:                      # AMDSEG_DIST_BACKEND=gloo: exercise the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("AMDSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items, rank, world):
    """DistributedSampler partition without shuffling: rank r takes items r, r+W, ... (tail padded by wrap-around)."""
    per = (n_items + world - 1) // world
    idx = [(rank + i * world) % n_items for i in range(per)]
    return idx


class GradBuckets:
    """contiguous slices of the flat gradient buffer in the order backward produces them."""

    def __init__(self, fp):
        names = list(fp.offsets.keys())
        offs = [fp.offsets[n] for n in names] + [fp.numel]
        first_layer = next(i for i, n in enumerate(names) if n.startswith(fp.encoder_prefix))
        per_layer = (len(names) - first_layer) // max(fp.nlayers, 1)
        self.layer_slices = []
        for li in range(fp.nlayers):
            a = offs[first_layer + li * per_layer]
            b = offs[first_layer + (li + 1) * per_layer]
            self.layer_slices.append((a, b))
        self.rest_slice = (0, offs[first_layer])
        self.flat_g = fp.flat_g
        self.handles = []

    def reduce_layer(self, li, group=None):
        a, b = self.layer_slices[li]
        self.handles.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True))

    def reduce_rest(self, group=None):
        a, b = self.rest_slice
        self.handles.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True))

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []


def allreduce_grads(engine, group=None):
    """non-overlapped fallback: one all-reduce over the whole flat gradient buffer."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(engine.fp.flat_g, op=dist.ReduceOp.SUM, group=group)
