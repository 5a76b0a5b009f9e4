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
        # RCCL keeps one CU per channel busy for the whole of an overlapped all-reduce, and the GEMM grids of this path are sized by rounds of
        # workgroups over the CUs (csrc/gemm_dp.hip: amdseg_cu_budget): 436 MB per 13-ms step need tens of GB/s, not every link saturated, so
        # the ring gets at most 32 channels unless the launcher says otherwise (the weight-gradient GEMM's 216 tiles need 216 free CUs; 224 are left)
        if os.environ.get("AMDSEG_NCCL_CHANNEL_CAP", "32") not in ("0", ""):          # (AMDSEG_NCCL_CHANNEL_CAP=0: leave RCCL's channel count alone)
            os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("AMDSEG_NCCL_CHANNEL_CAP", "32"))
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


def gather_sharded(local, n_items, group=None):
    """inverse of `shard_indices` over the world: every rank passes the rows it computed for ITS indices (in that order, `per` rows -- the
    wrapped tail included) and gets back the rows of items 0 .. n_items-1 in item order, the wrapped duplicates dropped (the reference's
    multi-GPU predict: DistributedSampler pads the tail, accelerate gathers, the Trainer truncates to len(dataset) -- run_inference.sh:35).
    `local`: tensor [per, ...]; works on any backend (gloo on CPU, RCCL on device tensors).  world 1 / no process group: `local[:n_items]`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local[:n_items]
    world = dist.get_world_size(group)
    per = (n_items + world - 1) // world
    if local.shape[0] != per:
        raise ValueError(f"gather_sharded: this rank holds {local.shape[0]} rows, shard_indices({n_items}, rank, {world}) has {per}")
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=group)
    # item i*world + r sits in parts[r][i]: interleave, then cut the wrap-around padding
    out = torch.stack(parts, dim=1).reshape(per * world, *local.shape[1:])
    return out[:n_items]


class NativeComm:
    """the gradient exchange of the C ABI (include/amdseg.h amdseg_allreduce_*: RCCL bound by libamdseg itself, csrc/comm.hip) behind the SAME
    bucket schedule as the torch.distributed one (GradBuckets._reduce): AMDSEG_DP_NATIVE_COMM=1.  The unique id travels over the existing
    process group (any backend); every rank then joins the RCCL communicator with its torch rank."""

    def __init__(self, group=None):
        import ctypes as C
        from . import lib as L
        self.L, self.lib = L, L.load()
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        uid = C.create_string_buffer(128)
        if rank == 0:
            L.check(self.lib.amdseg_allreduce_unique_id(uid), "amdseg_allreduce_unique_id")
        box = [bytes(uid.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        uid = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        L.check(self.lib.amdseg_allreduce_init(C.byref(h), uid, rank, world), "amdseg_allreduce_init")
        self.h = h
        r, w, pend = C.c_int(), C.c_int(), C.c_size_t()
        L.check(self.lib.amdseg_allreduce_info(h, C.byref(r), C.byref(w), C.byref(pend)), "amdseg_allreduce_info")
        self.rank, self.world = r.value, w.value          # the rank count as RCCL's communicator reports it (ncclCommCount)

    def allreduce(self, t, stream):
        """in place, ordered behind everything queued on `stream` so far; `stream` sees the result when this returns (stream-ordered)"""
        dt = self.L.F32 if t.dtype == torch.float32 else self.L.BF16
        self.L.check(self.lib.amdseg_allreduce_bucket(self.h, t.data_ptr(), t.numel(), dt, stream.cuda_stream), "amdseg_allreduce_bucket")
        self.L.check(self.lib.amdseg_allreduce_wait(self.h, stream.cuda_stream), "amdseg_allreduce_wait")

    def close(self):
        if getattr(self, "h", None):
            self.lib.amdseg_allreduce_destroy(self.h)
            self.h = None


class GradBuckets:
    """contiguous slices of the flat gradient buffer in the order backward produces them: [layer N-1] ... [layer 0] [embeddings + heads].

    On a GPU every exchange is issued from a SIDE stream: side waits for the main stream (the slice is final), RCCL reduces it, and the
    slice's sum of squares is accumulated right behind the reduction -- so after the last bucket the global gradient norm needs no
    extra pass over the 436 MB buffer, and the main stream (the rest of backward) never waits before `wait()`.
    Wire format (`wire`, AMDSEG_DP_WIRE): "fp32" (default: what torch DDP exchanges, 435.6 MB per bert-base step), "bf16_embed" (the
    word-embedding gradient -- 94 MB of the fully exposed tail bucket, produced last -- in bf16; AMDSEG_DP_BF16_EMBED=1 is the old spelling) or
    "bf16" (EVERY bucket cast to bf16, all-reduced, cast back: 217.8 MB on the xGMI links; the sum of W bf16 values carries 8 mantissa bits, so
    both bf16 forms deviate from the reference's fp32 exchange and are opt-in).
    Transport (`native`, AMDSEG_DP_NATIVE_COMM=1): the same schedule through the C ABI's amdseg_allreduce_* (NativeComm) instead of
    torch.distributed.all_reduce -- GPU + RCCL only.  `log` records every exchange of a step as (first, end, wire dtype) in issue order: the schedule
    is the same list whatever the transport (tests/test_dp_gloo.py, tests/test_gpu_ddp.py)."""

    def __init__(self, fp, bf16_embeddings=None, wire=None, native=None):
        names = list(fp.offsets.keys())
        offs = [fp.offsets[n] for n in names] + [fp.numel]
        # FlatParams lays the buffer out as [rest | layer 0 | ... | layer N-1]: `rest` = every parameter that is not one of the per-layer
        # names of `layer_order` (for Longformer that includes the query/key/value_global matrices, whose names ALSO start with the encoder
        # prefix -- so the layer slices are derived from the layout's own counts, never from a name prefix)
        first_layer = fp.n_rest
        per_layer = fp.per_layer
        assert len(names) == first_layer + fp.nlayers * per_layer, "FlatParams layout: rest + nlayers x per_layer names"
        self.layer_slices = []
        for li in range(fp.nlayers):
            a = offs[first_layer + li * per_layer]
            b = offs[first_layer + (li + 1) * per_layer]
            assert names[first_layer + li * per_layer].startswith(f"{fp.encoder_prefix}{li}."), names[first_layer + li * per_layer]
            self.layer_slices.append((a, b))
        self.rest_slice = (0, offs[first_layer])
        assert self.rest_slice[1] == fp.layers_begin or not fp.nlayers
        cov = self.rest_slice[1] - self.rest_slice[0] + sum(b - a for a, b in self.layer_slices)
        assert cov == fp.numel and all(self.layer_slices[i][1] == self.layer_slices[i + 1][0] for i in range(fp.nlayers - 1)), \
            "gradient buckets must tile the flat buffer exactly"
        # the leading part of the rest that ONLY the encoder's backward writes (embedding tables + their LayerNorm: 23.8 M of bert-base's 24.4 M
        # non-layer parameters, the fully exposed tail of the exchange): reduced as soon as the embedding backward is queued (engine.backward);
        # what follows (pooler, loss heads -- written by autograd nodes that may run after the encoder's) waits for finish_grad_sync()
        n_emb = 0
        while n_emb < first_layer and ".embeddings." in ("." + names[n_emb]):
            n_emb += 1
        self.emb_slice = (0, offs[n_emb])
        self._emb_reduced = False
        self.timing = False                    # bench: events behind the last layer bucket / the embeddings bucket / the rest (side stream)
        self.marks = {}
        self.flat_g = fp.flat_g
        self.handles = []
        self.cuda = fp.flat_g.is_cuda
        self.side = torch.cuda.Stream(device=fp.flat_g.device) if self.cuda else None
        self.sumsq = torch.zeros(1, device=fp.flat_g.device)
        self._partials = torch.empty(2048, device=fp.flat_g.device) if self.cuda else None
        self._covered = 0                      # elements whose squares are in `sumsq` since the last reset
        if wire is None:
            wire = os.environ.get("AMDSEG_DP_WIRE") or ("bf16_embed" if os.environ.get("AMDSEG_DP_BF16_EMBED", "0") == "1" else "fp32")
        if bf16_embeddings:
            wire = "bf16_embed"
        if wire not in ("fp32", "bf16_embed", "bf16"):
            raise ValueError(f"AMDSEG_DP_WIRE={wire!r}: expected fp32, bf16_embed or bf16")
        self.wire = wire
        self.log = []                          # (first, end, "fp32" | "bf16") per exchange since the last reset_norm(), in issue order
        self.last_log = []
        if native is None:
            native = os.environ.get("AMDSEG_DP_NATIVE_COMM", "0") == "1"
        self.native = None
        if native:
            if not self.cuda:
                raise RuntimeError("AMDSEG_DP_NATIVE_COMM=1: the C-ABI exchange (amdseg_allreduce_*) is RCCL on device buffers; this buffer is on the CPU")
            self.native = NativeComm()
        self._wire_buf = None
        if wire == "bf16" and self.cuda:       # one staging buffer: every exchange runs on the one side stream, in order
            biggest = max([b - a for a, b in self.layer_slices] + [self.rest_slice[1] - self.rest_slice[0]])
            self._wire_buf = torch.empty(biggest, dtype=torch.bfloat16, device=fp.flat_g.device)
        self.word_slice = None
        if wire == "bf16_embed":
            wn = next((n for n in names[:first_layer] if n.endswith("word_embeddings.weight")), None)
            if wn is not None:
                o = fp.offsets[wn]
                self.word_slice = (o, o + fp.params[wn].numel())
                self._word_bf16 = torch.empty(self.word_slice[1] - o, dtype=torch.bfloat16, device=fp.flat_g.device)

    def reset_norm(self):
        self.sumsq.zero_()
        self._covered = 0
        self._emb_reduced = False
        if self.log:
            self.last_log = self.log           # the schedule of the step that just ended (bench.py's dp record reads it)
        self.log = []

    def bytes_on_wire(self):
        """of the exchanges logged since the last reset"""
        return sum((b - a) * (2 if w == "bf16" else 4) for a, b, w in self.log)

    def norm_is_complete(self):
        return self.cuda and self._covered == self.flat_g.numel()

    def _exchange(self, t, group):
        """one all-reduce(sum) of `t` in place on the side stream (current), by the chosen transport"""
        if self.native is not None:
            self.native.allreduce(t, self.side)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)              # stream-ordered on `side`

    def _reduce(self, a, b, group, bf16=False):
        g = self.flat_g[a:b]
        bf16 = bf16 or self.wire == "bf16"
        self.log.append((a, b, "bf16" if bf16 else "fp32"))
        if not self.cuda:
            if bf16:                                                           # (CPU / gloo: the wire format is honoured, synchronously)
                w = g.to(torch.bfloat16)
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
                g.copy_(w)
            else:
                self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
            return
        from . import ops
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            if bf16:
                w = self._word_bf16 if (self.wire == "bf16_embed") else self._wire_buf[:b - a]
                w.copy_(g)
                self._exchange(w, group)
                g.copy_(w)
            else:
                self._exchange(g, group)
            ops.sumsq(g, self.sumsq, self._partials, accumulate=True)
        self._covered += b - a

    def _mark(self, key):
        if self.timing and self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.side)
            self.marks[key] = ev

    def reduce_layer(self, li, group=None):
        a, b = self.layer_slices[li]
        self._reduce(a, b, group)
        if li == 0:
            self._mark("layers_done")

    def _reduce_span(self, a, b, group):
        """[a, b) of the rest; the word-embedding table inside it in bf16 when that option is on"""
        if b <= a:
            return
        if self.word_slice is None or self.word_slice[1] <= a or self.word_slice[0] >= b:
            self._reduce(a, b, group)
            return
        wa, wb = self.word_slice
        self._reduce(wa, wb, group, bf16=True)
        if wa > a:
            self._reduce(a, wa, group)
        if b > wb:
            self._reduce(wb, b, group)

    def reduce_embeddings(self, group=None):
        """called by engine.backward right behind the embedding backward: the tail bucket starts without waiting for the host to come back
        from loss.backward() and call finish_grad_sync()"""
        if self._emb_reduced:
            return
        self._reduce_span(self.emb_slice[0], self.emb_slice[1], group)
        self._emb_reduced = True
        self._mark("embeddings_done")

    def reduce_rest(self, group=None):
        a, b = self.rest_slice
        if self._emb_reduced:
            a = self.emb_slice[1]
        self._reduce_span(a, b, group)
        self._emb_reduced = False
        self._mark("rest_done")

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)


def allreduce_grads(engine, group=None):
    """non-overlapped fallback: one all-reduce over the whole flat gradient buffer."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        engine.fp.flush_stale()                  # the slice the fused AdamW left un-zeroed counts as zero: make it so before anyone sums it
        dist.all_reduce(engine.fp.flat_g, op=dist.ReduceOp.SUM, group=group)
