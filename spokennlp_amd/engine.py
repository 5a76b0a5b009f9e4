"""Host-side engine of the MI355X BERT encoder: flat parameter storage, bf16 compute shadows, activation arenas, and
the forward / backward drivers that call libamdseg through the C ABI.  PyTorch tensors are storage + autograd glue.

Memory layout in HBM (one process per GPU):
  flat_p / flat_g / adam m, v : fp32 [n_params]      every nn.Parameter of the HF module tree is a VIEW into flat_p (and
                                                     its .grad a view into flat_g); q,k,v weights (and biases) of a layer
                                                     are adjacent, so the fused QKV projection reads one [3H,H] block
  shadow                      : bf16 [n_params]      compute copy of the encoder matrices (same layout), written by AdamW
  shadow_t (per layer)        : bf16 W^T blocks      for the dgrad GEMMs (refreshed after each optimiser step)
  activations                 : bf16 [layers][...]   saved-for-backward tensors (~25 KB/token/layer), one arena per M
"""
import ctypes as C
import contextlib
import weakref
from collections import OrderedDict

import torch

from . import lib as L
from . import ops

LAYER_ORDER = ["attention.self.query.weight", "attention.self.key.weight", "attention.self.value.weight",
               "attention.self.query.bias", "attention.self.key.bias", "attention.self.value.bias",
               "attention.output.dense.weight", "attention.output.dense.bias",
               "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
               "intermediate.dense.weight", "intermediate.dense.bias",
               "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]


MASK_BIAS = -30000.0


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatParams:
    """Re-homes the parameters of a module tree into one flat fp32 buffer (+ a flat gradient buffer)."""

    def __init__(self, module, device, encoder_prefix="bert.encoder.layer.", layer_order=None):
        layer_order = layer_order or LAYER_ORDER
        named = OrderedDict(module.named_parameters())
        order = []
        nlayers = 0
        while f"{encoder_prefix}{nlayers}.{layer_order[0]}" in named:
            nlayers += 1
        enc_names = set()
        for i in range(nlayers):
            for suffix in layer_order:
                n = f"{encoder_prefix}{i}.{suffix}"
                order.append(n); enc_names.add(n)
        rest = [n for n in named if n not in enc_names]
        order = rest + order
        self.offsets, off = OrderedDict(), 0
        for n in order:
            self.offsets[n] = off
            off = _align(off + named[n].numel())
        self.numel = off
        self.nlayers = nlayers
        self.n_rest = len(rest)                 # names [0, n_rest) of `offsets` are the non-layer front part; then nlayers x per_layer names
        self.per_layer = len(layer_order)
        # the encoder layers are one suffix of the flat buffers: [layers_begin, numel).  grad_stale: that slice of flat_g still holds the
        # PREVIOUS step's gradients (the fused AdamW did not zero it, engine.adamw_step) and counts as zero: the next backward overwrites it
        self.layers_begin = self.offsets[order[len(rest)]] if nlayers else off
        self.layers_dense = all(named[n].numel() % 64 == 0 for n in order[len(rest):])     # no alignment gaps inside the slice
        self.grad_stale = False
        self.encoder_prefix = encoder_prefix
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.params = named
        self.grad_is_zero = True
        with torch.no_grad():
            for n, p in named.items():
                v = self.view(self.flat_p, n, p.shape)
                v.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = v
        self._ptrs = [(p, self.flat_p.data_ptr() + 4 * self.offsets[n], self.flat_g.data_ptr() + 4 * self.offsets[n], n)
                      for n, p in named.items()]
        self.attach_grads()

    def view(self, flat, name, shape=None):
        shape = self.params[name].shape if shape is None else shape
        n = 1
        for s in shape:
            n *= s
        o = self.offsets[name]
        return flat[o:o + n].view(shape)

    def attach_grads(self):
        for n, p in self.params.items():
            p.grad = self.view(self.flat_g, n) if p.requires_grad else None

    def reattach_missing_grads(self):
        """every trainable parameter's .grad must be its view of the flat gradient buffer (the kernels write there).  A gradient that
        was dropped (`zero_grad(set_to_none=True)`, possibly by an optimiser that covers only a SUBSET of the parameters) or replaced
        is re-attached and its slice zeroed -- checked for EVERY parameter, not a sentinel."""
        missing = [(p, n) for p, _, gptr, n in self._ptrs
                   if p.requires_grad and (p.grad is None or p.grad.data_ptr() != gptr)]
        if not missing:
            return 0
        if len(missing) >= sum(1 for p, *_ in self._ptrs if p.requires_grad):
            if not self.grad_is_zero:                # the fused AdamW zeroes the buffer in its own pass
                self.flat_g.zero_()
                self.grad_is_zero = True
                self.grad_stale = False
            for p, n in missing:
                p.grad = self.view(self.flat_g, n)
        else:
            for p, n in missing:
                v = self.view(self.flat_g, n)
                if not (self.grad_stale and self.offsets[n] >= self.layers_begin):     # (a stale slot is overwritten by the next backward)
                    v.zero_()
                p.grad = v
        return len(missing)

    def flush_stale(self):
        """make the flat gradient buffer literally what it stands for: the encoder-layer slice that the fused AdamW left un-zeroed is
        zeroed now (a reader other than the next backward is about to look at it)"""
        if self.grad_stale:
            self.flat_g[self.layers_begin:].zero_()
            self.grad_stale = False

    def intact(self):
        """EVERY parameter still aliases its slot of the flat buffer (False after model.to(...), `p.data = new`, a re-initialised head,
        resize_token_embeddings ...: the owner then rebuilds the engine)."""
        dev = self.flat_p.device
        for p, pptr, _, _ in self._ptrs:
            if p.data_ptr() != pptr or p.device != dev:
                return False
        return True

    def lp(self, i, suffix):
        return f"{self.encoder_prefix}{i}.{suffix}"


GRAPHS_SUSPENDED = False            # bench.py sets this while the launch timer is armed (launches inside a replayed graph carry no events)
_LIVE_ENGINES = weakref.WeakSet()
_HOOKED = []


def _optimizer_stepped(opt, args, kwargs):
    """global torch.optim post-step hook: SOME optimiser wrote SOME parameters.  torch's fused AdamW (`_fused_adamw_`, the HF Trainer
    default `adamw_torch_fused`) does not bump the Parameters' version counters, so versions alone miss it."""
    if getattr(opt, "amdseg_fused", False):          # the engine's own optimiser refreshes the copies itself
        return
    for e in list(_LIVE_ENGINES):
        e._dirty = True


class BertEncoderEngine:
    """Runs embeddings + N BertLayers (+ final dropout) forward and backward on libamdseg kernels."""

    def __init__(self, module, config, device, bert_attr="bert", layer_order=None, nproj=3):
        L.load()      # fail loudly right away if the HIP library is absent
        self.cfg = config
        self.device = device
        self.H, self.I, self.heads = config.hidden_size, config.intermediate_size, config.num_attention_heads
        if self.H != self.heads * 64:
            raise L.AmdsegError("libamdseg attention kernels need head_dim == 64")
        act = getattr(config, "hidden_act", "gelu")
        if act == "gelu":
            self.act = 0                                    # exact erf GELU ([hf] activations.py GELUActivation)
        elif act in ("gelu_new", "gelu_pytorch_tanh", "gelu_fast"):
            self.act = 1                                    # the tanh form (three spellings of one formula)
        else:
            raise L.AmdsegError(f"hidden_act={act!r} is not implemented in the HIP epilogues (gelu / gelu_new only)")
        self.emb_dropout_pre_ln = False                     # BigBird: LayerNorm(dropout(sum)) instead of dropout(LayerNorm(sum))
        self.prefix = bert_attr + "."
        self.layer_order = layer_order or LAYER_ORDER
        self.nproj = nproj                                  # matrices fused into the input projection (q|k|v; PoNet: 5)
        self.proj_w0, self.proj_b0 = self.layer_order[0], self.layer_order[nproj]
        self.fp = FlatParams(module, device, encoder_prefix=self.prefix + "encoder.layer.", layer_order=self.layer_order)
        self.nlayers = self.fp.nlayers
        self.shadow = torch.zeros(self.fp.numel, dtype=torch.bfloat16, device=device)
        H, I = self.H, self.I
        self.shadow_t = [dict(wqkv_t=torch.empty(H, nproj * H, dtype=torch.bfloat16, device=device),
                              wo_t=torch.empty(H, H, dtype=torch.bfloat16, device=device),
                              w1_t=torch.empty(H, I, dtype=torch.bfloat16, device=device),
                              w2_t=torch.empty(I, H, dtype=torch.bfloat16, device=device)) for _ in range(self.nlayers)]
        self._shadow_version = None
        self._dirty = True                                  # the bf16 copies may be behind the fp32 masters (see refresh_shadows)
        self._fused_owner = False                           # the engine's fused AdamW is the one writing the parameters
        _LIVE_ENGINES.add(self)
        if not _HOOKED:
            from torch.optim.optimizer import register_optimizer_step_post_hook
            _HOOKED.append(register_optimizer_step_post_hook(_optimizer_stepped))
        self._ct_table = None
        import os as _os
        # the explicit library context of this engine (include/amdseg.h amdseg_ctx): the CU budget of backward's tile rules, the launch timer
        self.ctx = L.Ctx()
        # Engine options that are ATTRIBUTES, not environment switches (a test or a tool sets them on the engine before the first forward of a shape):
        # the trailing-padding chunks of full attention are not visited (False: every chunk, bit-identical results) ...
        self.skip_padded_chunks = True
        # ... and in backward the GEMM tiles made of rows of trailing padding (exact-zero gradients) are dropped after a run-time check of the
        # incoming gradient (amdseg_bert_cfg.pad_guard).  Every encoder family (the argument per mixer: DESIGN.md section 8)
        self.skip_padded_rows_bwd = True
        self._pad_guard = None
        self.eval_weight_check = _os.environ.get("AMDSEG_EVAL_WEIGHT_CHECK", "1") != "0"
        self._ck_state = None
        # attention-probability dropout decided once per layer (amdseg_attn_keepmask, acts.keep) instead of hashed per element in three kernels;
        # full softmax attention only (the band / list / pooling engines switch it off); AMDSEG_ATTN_HASH=1 keeps the hash path
        self.attn_keepmask = _os.environ.get("AMDSEG_ATTN_HASH", "0") != "1"
        self.keepmask_in_ln = True          # layer i + 1's masks from layer i's LayerNorm launch (set before the first forward; arenas are built once)
        # hidden-state dropout: the forward's add + LayerNorm kernels keep their decisions (1 byte per 8 elements, acts.drop1 / drop2) and the
        # LayerNorm backward reads them instead of re-hashing (every encoder family: the row kernels are shared); False = hash twice
        self.hidden_keepbits = True
        # bit-reproducible training steps: the word (and explicit position) embedding gradients are summed over a stable sort of the ids
        # (amdseg_scatter_rows_sorted) instead of scattered with fp32 atomics -- the only order-dependent sums of a BERT / Longformer step (the loss
        # heads accumulate in fixed point, every column sum goes through partials added in a fixed order).  config.amdseg_deterministic or
        # AMDSEG_DETERMINISTIC=1; off by default (the sort costs a few launches per step).  Not covered: token-type ids other than 0 mixed in
        # one batch (their table rows are still atomics) and the PoNet pooling backward (csrc/ponet.hip merges run pieces with atomics).
        self.deterministic = bool(getattr(self.cfg, "amdseg_deterministic", False)) or _os.environ.get("AMDSEG_DETERMINISTIC", "0") == "1"
        # the fused AdamW does not zero the encoder layers' gradients (78 % of bert-base's parameters): the next backward WRITES them
        # (accumulate_grads = 0 for its first call) instead of adding to zeros -- 4 B per parameter less written by the optimiser pass and 4 B
        # less read by the weight-gradient epilogues.  Every other reader of flat_g first calls fp.flush_stale().  Every engine family: the slice
        # holds exactly the parameters amdseg_bert_layer_bwd writes under cfg.accumulate_grads (PoNet's 5H projection and BigBird's layers are those
        # too; Longformer's global projections, accumulated by csrc/lf_global.hip, are not in layer_order and so sit in the eagerly zeroed front
        # part).  AMDSEG_LAZY_ZERO=0 switches it off.
        self.lazy_zero = _os.environ.get("AMDSEG_LAZY_ZERO", "1") != "0"
        self._arena_slot = 0
        self.max_live_arenas = 2                            # training arenas per shape that may be alive between forward and backward
        self.grad_sync = True                               # False inside no_sync(): accumulate locally, no bucket all-reduce
        self._rest_reduced = False                          # finish_grad_sync() ran for the gradients now in the flat buffer
        self._arenas = {}
        self._trigger = torch.zeros(1, device=device, requires_grad=True)
        self.adam_m = None
        self.adam_v = None
        self.opt_step = 0
        self._scratch = dict(sumsq=torch.zeros(1, device=device), coef=torch.ones(1, device=device),
                             norm=torch.zeros(1, device=device), partials=torch.empty(2048, device=device))
        self._build_param_structs()
        mats = set()
        for i in range(self.nlayers):
            for sfx in self.layer_order:
                if sfx.endswith(".weight") and self.fp.params[self.fp.lp(i, sfx)].dim() == 2:
                    mats.add(self.fp.lp(i, sfx))
        self._matrix_params = [self.fp.params[n] for n in self.fp.params if n in mats]
        self.buckets = None
        import os
        # opt-in (AMDSEG_OVERLAP_WGRAD=1): the grouped weight-gradient GEMM of layer i on a second stream under layer i - 1's backward -- its 216 tiles
        # leave 40 CUs idle and the next layer's input-gradient GEMMs fill them.  Round 1 measured it slower (19.7 vs 18.3 ms); round 5, same box,
        # interleaved: 12.54-12.61 -> 12.44-12.49 ms per step (+1 %, profiles/r05_default_switches.md).  Off by default all the same: the co-running
        # kernels' spans stretch (NT 58 -> 74 us, TN 200 -> 230 us) and every per-kernel roofline figure -- bench.py's in-step timer as much as a
        # rocprofv3 table of the same command -- would then describe two kernels sharing the chip instead of the kernel
        self.overlap_wgrad = os.environ.get("AMDSEG_OVERLAP_WGRAD", "0") == "1" and device.type == "cuda"
        self._wgrad_stream = torch.cuda.Stream(device=device, priority=0) if self.overlap_wgrad else None
        self._wgrad_done = [None, None]
        self._wgrad_last = None
        # per-call pass-through state set by the wrappers around ONE encode() (bert_for_ts.TopicSegHeadsMixin.forward) and cleared right after it
        # ELECTRA with embedding_size != hidden_size (electra-small: 128 -> 256): the embedding tables and their LayerNorm are E wide and a Linear
        # `embeddings_project` ([hf] models/electra/modeling_electra.py, ElectraModel.forward) maps them to H in front of the first layer
        self.E = int(getattr(config, "embedding_size", self.H) or self.H) if (self.prefix + "embeddings_project.weight") in self.fp.params else self.H
        self.emb_project = self.E != self.H
        self.eval_graphs = os.environ.get("AMDSEG_EVAL_GRAPHS", "0") == "1" and device.type == "cuda"
        self._eval_graphs = {}              # (B, L) -> captured inference forward
        self._hidden_sink = None            # list: output_hidden_states -> fp32 copies of the embedding output and every layer output
        self._explicit_pos = None           # int64 [B * L]: caller-given position ids

    def enable_data_parallel(self):
        """overlap per-layer RCCL all-reduce of the flat gradient slices with backward (dp.GradBuckets)."""
        import torch.distributed as dist
        from .dp import GradBuckets
        if dist.is_initialized() and dist.get_world_size() > 1:
            if self.buckets is None:
                dist.broadcast(self.fp.flat_p, src=0)         # what torch DDP does at construction: every rank starts from rank 0's weights
                self.refresh_shadows(force=True)
            self.buckets = GradBuckets(self.fp)
            if self.device.type == "cuda" and dist.get_backend() == "nccl":
                # the CUs the RCCL channels hold during the overlapped exchange are not available to the GEMM grids (include/amdseg.h)
                import os
                nch = int(os.environ.get("NCCL_MAX_NCHANNELS", "0") or 0)
                if nch > 0:
                    cus = torch.cuda.get_device_properties(self.device).multi_processor_count
                    self._bwd_cu_budget = max(cus - nch, cus // 2)       # applied around backward only: forward has every CU
                    if dist.get_rank() == 0 and not getattr(BertEncoderEngine, "_budget_logged", False):
                        BertEncoderEngine._budget_logged = True
                        import sys as _sys
                        print(f"spokennlp_amd: NCCL_MAX_NCHANNELS={nch}: backward tile rule counts on {self._bwd_cu_budget} of {cus} CUs "
                              f"(AMDSEG_NCCL_CHANNEL_CAP=0 leaves the channel count alone)", file=_sys.stderr)
        return self.buckets is not None

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation under the engine's own data parallelism (run_finetune.sh:31,76: gradient_accumulation_steps=2): inside
        this context backward only accumulates into the flat gradient buffer; the micro-step OUTSIDE it (the last one before the
        optimiser step) reduces every bucket once, so each bucket carries sum_ranks(sum_microsteps g) exactly once (torch DDP's no_sync)."""
        old, self.grad_sync = self.grad_sync, False
        try:
            yield
        finally:
            self.grad_sync = old

    def ddp_compat(self):
        """torch.distributed world > 1 WITHOUT the engine's own gradient exchange (`enable_data_parallel`): the caller is assumed to have
        wrapped the model in torch DistributedDataParallel, as `transformers.Trainer` under `torch.distributed.launch` does
        (run_finetune.sh:61).  DDP reduces a gradient when autograd accumulates it, so in that case the encoder's parameters are passed
        through autograd (EncoderFn returns their gradients) instead of being written behind its back into `param.grad` views."""
        import torch.distributed as dist
        return self.buckets is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def autograd_param_names(self):
        """the parameters whose gradients `backward` writes (everything under the encoder attribute except the unused pooler)"""
        if getattr(self, "_ag_names", None) is None:
            self._ag_names = [n for n in self.fp.params if n.startswith(self.prefix) and ".pooler." not in n]
        return self._ag_names

    def encode(self, input_ids, attention_mask, token_type_ids, train, seed, p_out):
        """differentiable encoder call for the wrappers: [N, L] int64 -> fp32 [N, L, H]"""
        if train and self.ddp_compat():
            return EncoderFn.apply(self._trigger, self, input_ids, attention_mask, token_type_ids, train, seed, p_out, *self.compat_params())
        if train:
            self.attach_grads_if_needed()
        return EncoderFn.apply(self._trigger, self, input_ids, attention_mask, token_type_ids, train, seed, p_out)

    def attach_grads_if_needed(self):
        """native mode: every param.grad is a view of the flat gradient buffer (re-attached after zero_grad(set_to_none=True))"""
        self.fp.reattach_missing_grads()

    def compat_params(self):
        """DDP-compatible mode: the parameters handed to autograd; no gradient may alias the flat buffer in this mode"""
        fp = self.fp
        lo = fp.flat_g.data_ptr()
        hi = lo + 4 * fp.numel
        for p in fp.params.values():
            if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                p.grad = None
        return [fp.params[n] for n in self.autograd_param_names()]

    def compat_backward(self, run_backward):
        """the kernels write this call's gradients into the (zeroed) flat buffer; autograd gets a snapshot, so accumulation over
        micro-steps, DDP's reduction hooks and any torch optimiser work on ordinary gradient tensors"""
        self.fp.flat_g.zero_()
        self.fp.grad_stale = False
        run_backward()
        snap = self.fp.flat_g.clone()
        self.fp.grad_is_zero = False
        return tuple(self.fp.view(snap, n) for n in self.autograd_param_names())

    def finish_grad_sync(self):
        """reduce the non-encoder slice (embeddings, heads) and wait for every outstanding bucket."""
        if self.buckets is not None and self.grad_sync and not self._rest_reduced:
            # once per optimiser step: a second call (a logging callback asking for the norm again, grad_norm(inf) and grad_norm(max)) would
            # all-reduce the embeddings + heads slice again, i.e. multiply it by the world size
            self.buckets.reduce_rest()
            self.buckets.wait()
            self._rest_reduced = True

    # ------------------------------------------------------------------------------------------------ parameters
    def _p(self, flat, i, suffix):
        return self.fp.view(flat, self.fp.lp(i, suffix))

    def _build_param_structs(self):
        self.lparams, self.lgrads, self.lparams32 = [], [], []
        fp = self.fp
        for i in range(self.nlayers):
            sh = lambda s: self._p(self.shadow, i, s).data_ptr()          # noqa: E731
            ps = lambda s: self._p(fp.flat_p, i, s).data_ptr()            # noqa: E731
            gs = lambda s: self._p(fp.flat_g, i, s).data_ptr()            # noqa: E731
            t = self.shadow_t[i]
            P = L.LayerParams(wqkv=sh(self.proj_w0), wo=sh("attention.output.dense.weight"),
                              w1=sh("intermediate.dense.weight"), w2=sh("output.dense.weight"),
                              wqkv_t=t["wqkv_t"].data_ptr(), wo_t=t["wo_t"].data_ptr(), w1_t=t["w1_t"].data_ptr(),
                              w2_t=t["w2_t"].data_ptr(),
                              bqkv=ps(self.proj_b0), bo=ps("attention.output.dense.bias"),
                              b1=ps("intermediate.dense.bias"), b2=ps("output.dense.bias"),
                              ln1_g=ps("attention.output.LayerNorm.weight"), ln1_b=ps("attention.output.LayerNorm.bias"),
                              ln2_g=ps("output.LayerNorm.weight"), ln2_b=ps("output.LayerNorm.bias"))
            G = L.LayerGrads(wqkv=gs(self.proj_w0), wo=gs("attention.output.dense.weight"),
                             w1=gs("intermediate.dense.weight"), w2=gs("output.dense.weight"),
                             bqkv=gs(self.proj_b0), bo=gs("attention.output.dense.bias"),
                             b1=gs("intermediate.dense.bias"), b2=gs("output.dense.bias"),
                             ln1_g=gs("attention.output.LayerNorm.weight"), ln1_b=gs("attention.output.LayerNorm.bias"),
                             ln2_g=gs("output.LayerNorm.weight"), ln2_b=gs("output.LayerNorm.bias"))
            self.lparams.append(P); self.lgrads.append(G)
            # fp32 parity mode reads the fp32 masters directly (no shadows, no transposes)
            self.lparams32.append(L.LayerParams(
                wqkv=ps(self.proj_w0), wo=ps("attention.output.dense.weight"),
                w1=ps("intermediate.dense.weight"), w2=ps("output.dense.weight"), wqkv_t=None, wo_t=None, w1_t=None, w2_t=None,
                bqkv=ps(self.proj_b0), bo=ps("attention.output.dense.bias"),
                b1=ps("intermediate.dense.bias"), b2=ps("output.dense.bias"),
                ln1_g=ps("attention.output.LayerNorm.weight"), ln1_b=ps("attention.output.LayerNorm.bias"),
                ln2_g=ps("output.LayerNorm.weight"), ln2_b=ps("output.LayerNorm.bias")))

    def _weights_version(self):
        """changes whenever any encoder matrix was written in place through its Parameter (torch optimisers, load_state_dict,
        load_best_model_at_end, checkpoint resume).  Each Parameter keeps its OWN version counter after `p.data = view`, so the flat
        buffer's counter says nothing; the engine's own fused AdamW writes through raw pointers and refreshes explicitly."""
        v = self.fp.flat_p._version
        for p in self._matrix_params:
            v += p._version
        return v

    # ---- "parity" precision: split-bf16 weight images (csrc/parity.hip), built on first use and refreshed with the bf16 copies
    supports_parity = True
    parity_needs_split_attn = False                         # band attention exists in split-bf16 form only (no fp32-MFMA fallback)

    def _parity_weights(self):
        if getattr(self, "_parity", None) is None:
            if not self.supports_parity or self.nproj != 3:
                raise L.AmdsegError('amdseg_precision="parity" is implemented for the BERT / ELECTRA encoder (full softmax attention)')
            H, I, dev = self.H, self.I, self.device
            bf = dict(dtype=torch.bfloat16, device=dev)
            self._parity = [dict(wqkv=torch.empty(3 * H, 3 * H, **bf), wo=torch.empty(H, 3 * H, **bf), w1=torch.empty(I, 3 * H, **bf),
                                 w2=torch.empty(H, 3 * I, **bf), wqkv_t=torch.empty(H, 9 * H, **bf), wo_t=torch.empty(H, 3 * H, **bf),
                                 w1_t=torch.empty(H, 3 * I, **bf), w2_t=torch.empty(I, 3 * H, **bf)) for _ in range(self.nlayers)]
            fp = self.fp
            self.lparams_parity = []
            for i in range(self.nlayers):
                ps = lambda s_: self._p(fp.flat_p, i, s_).data_ptr()            # noqa: E731
                t = self._parity[i]
                self.lparams_parity.append(L.LayerParams(
                    wqkv=t["wqkv"].data_ptr(), wo=t["wo"].data_ptr(), w1=t["w1"].data_ptr(), w2=t["w2"].data_ptr(),
                    wqkv_t=t["wqkv_t"].data_ptr(), wo_t=t["wo_t"].data_ptr(), w1_t=t["w1_t"].data_ptr(), w2_t=t["w2_t"].data_ptr(),
                    bqkv=ps(self.proj_b0), bo=ps("attention.output.dense.bias"), b1=ps("intermediate.dense.bias"), b2=ps("output.dense.bias"),
                    ln1_g=ps("attention.output.LayerNorm.weight"), ln1_b=ps("attention.output.LayerNorm.bias"),
                    ln2_g=ps("output.LayerNorm.weight"), ln2_b=ps("output.LayerNorm.bias")))
            self._split_parity_weights()
        return self.lparams_parity

    def _split_parity_weights(self):
        H = self.H
        if (H % 64) == 0 and (self.I % 64) == 0:
            # one launch for every matrix of the model (amdseg_split3_weights_batched)
            if getattr(self, "_sw_table", None) is None:
                Ws, outs, outts, Ns, Ks = [], [], [], [], []
                for i in range(self.nlayers):
                    t = self._parity[i]
                    for W, k in ((self.fp.view(self.fp.flat_p, self.fp.lp(i, self.proj_w0), (3 * H, H)), "wqkv"),
                                 (self._p(self.fp.flat_p, i, "attention.output.dense.weight"), "wo"),
                                 (self._p(self.fp.flat_p, i, "intermediate.dense.weight"), "w1"),
                                 (self._p(self.fp.flat_p, i, "output.dense.weight"), "w2")):
                        Ws.append(W.data_ptr()); outs.append(t[k].data_ptr()); outts.append(t[k + "_t"].data_ptr())
                        Ns.append(W.shape[0]); Ks.append(W.shape[1])
                n = len(Ws)
                self._sw_table = (n, (C.c_void_p * n)(*Ws), (C.c_void_p * n)(*outs), (C.c_void_p * n)(*outts), (C.c_int * n)(*Ns), (C.c_int * n)(*Ks))
            n, pw, po, pt, pn, pk = self._sw_table
            L.check(L.load().amdseg_split3_weights_batched(n, pw, po, pt, pn, pk, torch.cuda.current_stream().cuda_stream),
                    "amdseg_split3_weights_batched")
            return
        for i in range(self.nlayers):
            t = self._parity[i]
            mats = ((self.fp.view(self.fp.flat_p, self.fp.lp(i, self.proj_w0), (3 * H, H)), "wqkv"),
                    (self._p(self.fp.flat_p, i, "attention.output.dense.weight"), "wo"),
                    (self._p(self.fp.flat_p, i, "intermediate.dense.weight"), "w1"), (self._p(self.fp.flat_p, i, "output.dense.weight"), "w2"))
            for W, k in mats:
                ops.split3(W, t[k], order=1)
                ops.split3_transpose(W, t[k + "_t"])

    def mark_weights_dirty(self):
        """call after writing encoder weights by a route none of the detectors below can see (`p.data.copy_(...)`, raw pointers)"""
        self._dirty = True

    def refresh_shadows(self, force=False, copies_done=False):
        """bf16 compute copies of the encoder matrices (+ transposes).  Re-done when the masters MAY have changed:
          * a matrix Parameter's version counter moved (in-place torch ops, load_state_dict, foreach optimisers),
          * any torch optimiser stepped (global post-step hook: the fused torch AdamW bumps no version counter),
          * a training forward ran since the last refresh and the engine's own fused AdamW is not the writer -- whatever updates the
            weights between two training forwards (third-party optimisers writing through `.data`) is then covered too.
        The engine's fused AdamW refreshes explicitly (force) right after its pass."""
        ver = self._weights_version()
        if not force and not self._dirty and ver == self._shadow_version:
            return False
        self._dirty = False
        if self._ct_table is None:
            Ws, Wbs, Wts, Ns, Ks = [], [], [], [], []
            H = self.H
            for i in range(self.nlayers):
                t = self.shadow_t[i]
                qn = self.fp.lp(i, self.proj_w0)
                NP = self.nproj * H
                Ws.append(self.fp.view(self.fp.flat_p, qn, (NP, H))); Wbs.append(self.fp.view(self.shadow, qn, (NP, H))); Wts.append(t["wqkv_t"])
                for s_, key in (("attention.output.dense.weight", "wo_t"), ("intermediate.dense.weight", "w1_t"),
                                ("output.dense.weight", "w2_t")):
                    Ws.append(self._p(self.fp.flat_p, i, s_)); Wbs.append(self._p(self.shadow, i, s_)); Wts.append(t[key])
            n = len(Ws)
            self._ct_table = (n, (C.c_void_p * n)(*[w.data_ptr() for w in Ws]), (C.c_void_p * n)(*[w.data_ptr() for w in Wbs]),
                              (C.c_void_p * n)(*[w.data_ptr() for w in Wts]), (C.c_int * n)(*[w.shape[0] for w in Ws]),
                              (C.c_int * n)(*[w.shape[1] for w in Ws]))
        n, pw, pb, pt, pn, pk = self._ct_table
        # copies_done: the fused AdamW has just written the bf16 copies -> transposes only, read from them (half the bytes of the masters)
        rc = L.load().amdseg_cast_transpose_batched(n, None if copies_done else pw, pb, pt, pn, pk, torch.cuda.current_stream().cuda_stream)
        L.check(rc, "amdseg_cast_transpose_batched")
        if getattr(self, "_parity", None) is not None:
            self._split_parity_weights()
        self._shadow_version = self._weights_version()
        # the content checksums of _check_unseen_writes describe what the copies were derived from LAST TIME IT LOOKED; the copies now
        # come from the masters as they are at this moment, which that check has not seen: both of its states are stale
        self._ck_stale = {"bf16": True, "parity": True}
        return True

    def _check_unseen_writes(self, parity):
        """inference forwards: writes to the fp32 masters that none of refresh_shadows' detectors can see (`p.data.copy_(...)` between two
        eval forwards, raw pointers) are found by a 64-bit content checksum of the encoder's parameter region taken ON THE DEVICE
        (amdseg_weights_changed, ~80 us for bert-base) and the bf16 copies / transposes are re-derived by a refresh that is a no-op when
        nothing changed (amdseg_cast_transpose_batched_if) -- no host read, no `mark_weights_dirty()` needed.  "parity" precision re-splits its
        weight images on the host's say-so, so there the flag is read back (one sync per inference forward of that slower mode).
        The stored checksum must describe the masters THE COPIES WERE DERIVED FROM, so: there is one state per set of copies
        (the bf16 copies / the split images of "parity" precision -- a bf16 forward that refreshes only its copies leaves the images' state
        behind, and the next parity forward sees the difference), and a refresh by any other route (version counters, mark_weights_dirty,
        optimiser hooks: refresh_shadows) marks both stale -- a stale state is zeroed, which the kernel reads as "first call: changed", i.e.
        one redundant refresh instead of a missed one.  AMDSEG_EVAL_WEIGHT_CHECK=0 switches the check off."""
        if not self.eval_weight_check or self._ct_table is None:
            return
        fp = self.fp
        if self._ck_state is None:
            self._ck_state = {k: torch.zeros(2, dtype=torch.int64, device=self.device) for k in ("bf16", "parity")}
            self._ck_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
            first = min(fp.offsets[n] for n in fp.offsets if n.startswith(fp.encoder_prefix))
            self._ck_region = (fp.flat_p.data_ptr() + 4 * first, 4 * (fp.numel - first))
        which = "parity" if parity else "bf16"
        state = self._ck_state[which]
        stale = getattr(self, "_ck_stale", None)
        if stale is None:
            stale = self._ck_stale = {"bf16": True, "parity": True}
        if stale[which]:
            state.zero_()
        s = torch.cuda.current_stream().cuda_stream
        lib = L.load()
        L.check(lib.amdseg_weights_changed(self._ck_region[0], self._ck_region[1], state.data_ptr(), self._ck_flag.data_ptr(), s),
                "amdseg_weights_changed")
        if parity:
            if int(self._ck_flag.item()):
                self.refresh_shadows(force=True)            # every copy re-derived from the content `state` now describes (marks both stale)
            self._ck_stale["parity"] = False
            return
        n, pw, pb, pt, pn, pk = self._ct_table
        L.check(lib.amdseg_cast_transpose_batched_if(n, pw, pb, pt, pn, pk, self._ck_flag.data_ptr(), s), "amdseg_cast_transpose_batched_if")
        self._ck_stale["bf16"] = False

    # ------------------------------------------------------------------------------------------------ arenas
    def _acquire_arena(self, B, Lseq, train, fp32=False):
        """the arena a forward may write.  Inference arenas hold nothing past the call.  A TRAINING arena holds the activations saved for
        backward, so it stays `busy` from its forward until its backward ran (or its autograd node died); a second training forward at the
        same shape in between (siamese / contrastive calls, checkpoint recomputation) gets another arena, up to `max_live_arenas`;
        beyond that the oldest is recycled and ITS backward raises instead of silently reading another call's activations."""
        if not train:
            self._arena_slot = 0
            return self._arena(B, Lseq, train, fp32)
        pick, oldest = None, None
        for slot in range(self.max_live_arenas):
            A = self._arenas.get((B, Lseq, True, fp32, slot))
            if A is None or not A["busy"]:
                pick = slot
                break
            if oldest is None or A["stamp"] < oldest[1]:
                oldest = (slot, A["stamp"])
        self._arena_slot = oldest[0] if pick is None else pick
        A = self._arena(B, Lseq, train, fp32)
        self._arena_slot = 0
        self._arena_clock = getattr(self, "_arena_clock", 0) + 1
        A["gen"] += 1
        A["busy"], A["stamp"] = True, self._arena_clock
        return A

    @staticmethod
    def _release_arena(A, gen):
        if A["gen"] == gen:
            A["busy"] = False

    def _arena(self, B, Lseq, train, fp32=False):
        key = (B, Lseq, train, fp32, self._arena_slot if train else 0)
        if key in self._arenas:
            return self._arenas[key]
        dev, H, I, M = self.device, self.H, self.I, B * Lseq
        parity = fp32 == "parity"
        bf = torch.float32 if fp32 else torch.bfloat16
        nsave = self.nlayers if train else 1

        def e(*s, dt=bf):
            return torch.empty(*s, dtype=dt, device=dev)

        A = dict(x=[e(M, H) for _ in range(nsave + 1)] if train else [e(M, H), e(M, H)],
                 layers=[dict(qkv=e(M, self.nproj * H), ctx=e(M, H), z1=e(M, H), x1=e(M, H), u=e(M, I), h=e(M, I), z2=e(M, H),
                              lse=e(B * self.heads * Lseq, dt=torch.float32), mean1=e(M, dt=torch.float32),
                              rstd1=e(M, dt=torch.float32), mean2=e(M, dt=torch.float32), rstd2=e(M, dt=torch.float32))
                         for _ in range(nsave)],
                 emb_z=e(M, self.E), emb_mean=e(M, dt=torch.float32), emb_rstd=e(M, dt=torch.float32),
                 mask_bias=e(B, Lseq, dt=torch.float32), gen=0, busy=False, stamp=0)
        # "parity" precision: attention as split-bf16 products (csrc/attention_split.hip) needs the split image of q|k|v per layer
        # (engine.parity_split_attn = False keeps the fp32-MFMA attention of csrc/parity.hip where the family allows it)
        split_attn = parity and (self.parity_needs_split_attn or getattr(self, "parity_split_attn", True))
        if split_attn:
            for la in A["layers"]:
                la["qkv_s"] = e(M, 9 * H, dt=torch.bfloat16)
        if train and float(self.cfg.attention_probs_dropout_prob) > 0 and ((not fp32 and self.attn_keepmask) or split_attn):
            nbytes = L.load().amdseg_attn_keepmask_bytes(B, Lseq, self.heads)
            for la in A["layers"]:
                la["keep"] = e(nbytes, dt=torch.uint8)
        if train and float(self.cfg.hidden_dropout_prob) > 0 and self.hidden_keepbits and H % 8 == 0:
            for la in A["layers"]:
                la["drop1"] = e(M * H // 8, dt=torch.uint8)
                la["drop2"] = e(M * H // 8, dt=torch.uint8)
        if parity:               # split-bf16 images of the GEMM operands (see csrc/parity.hip); `h` itself is never materialised
            for la in A["layers"]:
                la.update(xs=e(M, 3 * H, dt=torch.bfloat16), ctx_s=e(M, 3 * H, dt=torch.bfloat16), x1_s=e(M, 3 * H, dt=torch.bfloat16),
                          h_s=e(M, 3 * I, dt=torch.bfloat16))
        if train:
            npart = 2 * ops.ln_partials_numel(M, H) + max((M + 127) // 128, (H + 127) // 128) * (I + self.nproj * H)   # amdseg.h: amdseg_bert_layer_ws
            def ws_set():
                d = dict(dz2=e(M, H), dbr2=e(M, H), du=e(M, I), dx1=e(M, H), dz1=e(M, H), dbr1=e(M, H), dctx=e(M, H),
                         dqkv=e(M, self.nproj * H), delta=e(B * self.heads * Lseq, dt=torch.float32),
                         partials=e(npart, dt=torch.float32))
                if parity:
                    d.update(d_out_s=e(M, 3 * H, dt=torch.bfloat16), du_s=e(M, 3 * I, dt=torch.bfloat16),
                             d_ao_s=e(M, 3 * H, dt=torch.bfloat16), dqkv_s=e(M, 9 * H, dt=torch.bfloat16))
                    if split_attn:
                        d["dctx_s"] = e(M, 3 * H, dt=torch.bfloat16)
                return d

            # two scratch sets: layer i's weight-gradient GEMM (second stream) still reads set i % 2 while layer i-1's backward
            # writes the other one
            A["ws_sets"] = [ws_set(), ws_set()]
            wkeys = ("dz2", "dbr2", "du", "dx1", "dz1", "dbr1", "dctx", "dqkv", "delta", "partials") + \
                (("d_out_s", "du_s", "d_ao_s", "dqkv_s") if parity else ()) + (("dctx_s",) if split_attn else ())
            A["ws_structs"] = [L.LayerWs(**{k: w[k].data_ptr() for k in wkeys}) for w in A["ws_sets"]]
            A["ws"] = dict(A["ws_sets"][0], dy=[e(M, H), e(M, H)])
            A["ws_struct"] = A["ws_structs"][0]
        if self.emb_project:                # the E-wide embedding output (input of the projection; kept for its weight gradient) + backward scratch
            A["emb_x"] = e(M, self.E)
            if train:
                A["demb"] = [e(M, self.E), e(M, self.E)]
        A["acts_struct"] = []
        for i in range(self.nlayers):
            la = A["layers"][i if train else 0]
            xin = A["x"][i] if train else A["x"][i % 2]
            xout = A["x"][i + 1] if train else A["x"][(i + 1) % 2]
            ptrs = {k: la[k].data_ptr() for k in ("qkv", "ctx", "z1", "x1", "u", "h", "z2", "lse", "mean1", "rstd1", "mean2", "rstd2")}
            if parity:
                ptrs.update({k: la[k].data_ptr() for k in ("xs", "ctx_s", "x1_s", "h_s")})
            if "keep" in la:
                ptrs["keep"] = la["keep"].data_ptr()
            if "qkv_s" in la:
                ptrs["qkv_s"] = la["qkv_s"].data_ptr()
            if "drop1" in la:
                ptrs["drop1"] = la["drop1"].data_ptr(); ptrs["drop2"] = la["drop2"].data_ptr()
            if not train and not fp32:
                ptrs["u"] = None                  # inference: the FFN GEMM skips the pre-activation output
            if not train and parity and M % 256 == 0 and self.I % 256 == 0 and getattr(self.cfg, "hidden_act", "gelu") == "gelu":
                ptrs["u"] = None                  # "parity" inference: the fused up-projection epilogue writes only the image of gelu(u)
            A["acts_struct"].append(L.LayerActs(x_in=xin.data_ptr(), x_out=xout.data_ptr(), **ptrs))
        # the keep masks of layer i + 1 written by the launch of layer i's second LayerNorm (amdseg.h: acts.keep_next / keep_ready): the VALU-bound
        # generator under an HBM-bound row kernel; layer 0 generates its own.  Same bits either way (self.keepmask_in_ln = False: every layer its own)
        if train and not fp32 and self.keepmask_in_ln and "keep" in A["layers"][0]:
            for i in range(self.nlayers - 1):
                A["acts_struct"][i].keep_next = A["layers"][i + 1]["keep"].data_ptr()
                A["acts_struct"][i + 1].keep_ready = 1
        A["x_final"] = A["x"][self.nlayers] if train else A["x"][self.nlayers % 2]
        self._arenas[key] = A
        return A

    def _cfg_struct(self, B, Lseq, p_hidden, p_attn, seed, accumulate):
        return L.BertCfg(B=B, L=Lseq, H=self.H, heads=self.heads, I=self.I, ln_eps=float(self.cfg.layer_norm_eps),
                         p_hidden=p_hidden, p_attn=p_attn, seed=seed, accumulate_grads=1 if accumulate else 0, dtype=L.BF16,
                         window=0, nglobal=0, nproj=0, mixer=0, phase=0, act=self.act, ctx=self.ctx.ptr)

    # ------------------------------------------------------------------------------------------------ forward / backward
    def _emb(self, name):
        return self.fp.params[self.prefix + "embeddings." + name]

    def forward(self, input_ids, attention_mask, token_type_ids, train, seed=0, p_out=0.0):
        """input_ids/attention_mask/token_type_ids: int64 [B, L] on device.  Returns fp32 [B, L, H] (after the wrapper's
        classifier dropout p_out when training) and a context for backward."""
        B, Lseq = input_ids.shape
        M = B * Lseq
        if (M % 128) or (Lseq % 64):
            raise L.AmdsegError(f"batch*seq must be a multiple of 128 and seq a multiple of 64 (got B={B}, L={Lseq})")
        fp32 = self._precision(train)
        if fp32 == "parity":
            self._parity_weights()
        if fp32 is not True:
            refreshed = self.refresh_shadows()
            if not train and not refreshed:
                self._check_unseen_writes(fp32 == "parity")
            if train and not self._fused_owner:
                self._dirty = True                          # an unknown optimiser is expected to write the weights after this step
        if not train and self._eval_graph_ok(fp32, attention_mask):
            return self._forward_graphed(input_ids, attention_mask, token_type_ids, seed, p_out, fp32)
        return self._forward_body(input_ids, attention_mask, token_type_ids, train, seed, p_out, fp32)

    # ---- small-batch inference: the encoder's ~100 launches as ONE hipGraph replay (opt-in: AMDSEG_EVAL_GRAPHS=1 / engine.eval_graphs = True)
    # At 4 x 512 tokens (run_inference.sh ships per_device_eval_batch_size 4) queueing the encoder's launches costs the host 1.21 ms per call
    # (tools/dbg/infer_small_batch.py).  After two eager calls at a shape the third is captured (torch.cuda.graph over the same `_forward_body`:
    # static input buffers, the arena's own activations, a static output) and later calls copy their inputs in and replay: 0.1 ms of host time,
    # bit-identical output (test_eval_forward_as_a_hipgraph_*).  The WALL time does not move -- 1.70 ms per model forward either way: the ~100
    # kernels of a 2048-token forward are latency-bound on the GPU (1.16 ms + 0.33 ms of weight check + heads) -- so this only frees the host
    # (tokenisation / gathers of a prediction loop) and stays off by default.  Only the plain engine (BERT / ELECTRA), bf16, hard masks, no
    # hidden-state / position pass-through; the weight checks stay outside the graph; while the launch timer is armed (bench.py's roofline) the
    # eager path runs, because launches inside a graph carry no events.
    def _eval_graph_ok(self, fp32, attention_mask):
        return (self.eval_graphs and not GRAPHS_SUSPENDED and type(self) is BertEncoderEngine and fp32 is False and not self.emb_project
                and self._hidden_sink is None and self._explicit_pos is None and not attention_mask.dtype.is_floating_point
                and not torch.cuda.is_current_stream_capturing())

    def _forward_graphed(self, input_ids, attention_mask, token_type_ids, seed, p_out, fp32):
        B, Lseq = input_ids.shape
        key = (B, Lseq)
        ent = self._eval_graphs.get(key)
        if ent is None:
            ent = self._eval_graphs[key] = dict(calls=0, graph=None)
            if len(self._eval_graphs) > 8:                  # (a prediction loop has one or two shapes; do not hoard graphs for a shape sweep)
                self._eval_graphs.pop(next(iter(self._eval_graphs)))
        ent["calls"] += 1
        if ent["graph"] is None:
            if ent["calls"] < 3:
                return self._forward_body(input_ids, attention_mask, token_type_ids, False, seed, p_out, fp32)
            ent["ids"] = input_ids.clone(); ent["am"] = attention_mask.clone(); ent["tt"] = token_type_ids.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out, _ = self._forward_body(ent["ids"], ent["am"], ent["tt"], False, seed, p_out, fp32)
            ent["graph"], ent["out"] = g, out
        ent["ids"].copy_(input_ids); ent["am"].copy_(attention_mask); ent["tt"].copy_(token_type_ids)
        ent["graph"].replay()
        return ent["out"].clone(), None                     # (a fresh tensor per call, as the eager path returns)

    def _forward_body(self, input_ids, attention_mask, token_type_ids, train, seed, p_out, fp32):
        B, Lseq = input_ids.shape
        M = B * Lseq
        A = self._acquire_arena(B, Lseq, train, fp32)
        dt = L.F32 if fp32 else L.BF16
        p_h = float(self.cfg.hidden_dropout_prob) if train else 0.0
        p_a = float(self.cfg.attention_probs_dropout_prob) if train else 0.0
        cfg = self._cfg_struct(B, Lseq, p_h, p_a, seed, True)
        cfg.dtype = L.F32S if fp32 == "parity" else dt
        lparams = self.lparams_parity if fp32 == "parity" else (self.lparams32 if fp32 else self.lparams)
        ids = input_ids.reshape(-1).contiguous()
        tts = token_type_ids.reshape(-1).contiguous()
        # additive key mask: 0 / MASK_BIAS.  A moderate magnitude on purpose: exp() of a masked score underflows to an exact 0 next to any
        # real score (as the reference's finfo.min does), while (mask - lse) and (score - max) keep their fp32 digits in rows whose
        # visible keys are ALL masked (padded queries of a band, fully padded sequences) -- with -1e30 those differences carry an
        # absolute error of ~1e23 and exp2 of them is inf (attention.hip folds mask and lse into the MFMA accumulator start)
        # ... and the padding plan (include/amdseg.h amdseg_pad_plan): kend = (last unmasked key) + 1 per sequence -- the attention kernels do not
        # visit the chunks past it (they add exact zeros) --, the sequences by decreasing kend (dispatch order), and for the backward the
        # runs of 64-token tiles in front of kend (amdseg_bert_cfg.pad_runs).  One launch pair per forward, kept with the arena for backward.
        if "kend" not in A or A["kend"].numel() != B:
            A["kend"] = torch.empty(B, dtype=torch.int32, device=self.device)
            A["seq_order"] = torch.empty(B, dtype=torch.int32, device=self.device)
            A["pad_runs"] = torch.zeros(B, 2, dtype=torch.int32, device=self.device)
            A["pad_counts"] = torch.zeros(2, dtype=torch.int32, device=self.device)
        am64 = attention_mask
        if am64.dtype.is_floating_point:                    # soft masks: the host formula (kend from "non-zero")
            torch.mul(1.0 - attention_mask.to(torch.float32), MASK_BIAS, out=A["mask_bias"])
            am64 = attention_mask != 0
        am64 = am64.to(torch.int64).contiguous()
        if B > 8192:
            raise L.AmdsegError(f"more than 8192 sequences in one forward (got {B}): split the batch (amdseg_pad_plan)")
        L.check(L.load().amdseg_pad_plan(am64.data_ptr(), B, Lseq, A["kend"].data_ptr(), A["seq_order"].data_ptr(), A["pad_runs"].data_ptr(),
                                         A["pad_counts"].data_ptr(), None if attention_mask.dtype.is_floating_point else A["mask_bias"].data_ptr(),
                                         MASK_BIAS, torch.cuda.current_stream().cuda_stream), "amdseg_pad_plan")
        cfg.kend = A["kend"].data_ptr() if self.skip_padded_chunks else None
        cfg.seq_order = A["seq_order"].data_ptr() if self.skip_padded_chunks else None
        lib = L.load()
        s = torch.cuda.current_stream().cuda_stream
        eps = float(self.cfg.layer_norm_eps)
        we, pe, te = self._emb("word_embeddings.weight"), self._emb("position_embeddings.weight"), self._emb("token_type_embeddings.weight")
        # caller-given position ids (the wrapper's `position_ids` argument, bert_for_ts.py:60) take the explicit-position path of the embedding
        # kernels (the one Longformer's pad-aware positions use); otherwise the family's own rule (BERT: 0 .. L-1 inside the kernel)
        pos = getattr(self, "_explicit_pos", None)
        if pos is not None:
            if pos.numel() != M:
                raise L.AmdsegError(f"position_ids: expected {M} ids for {B} x {Lseq} tokens, got {pos.numel()}")
            pos = pos.reshape(-1).to(torch.int64).contiguous()
        else:
            pos = self._position_ids(input_ids)
        if self.emb_project and fp32:
            raise L.AmdsegError("embeddings_project (ELECTRA with embedding_size != hidden_size) runs in bf16 precision only")
        emb_out = A["emb_x"] if self.emb_project else A["x"][0]
        rc = lib.amdseg_embed_ln_fwd(ids.data_ptr(), tts.data_ptr(), None if pos is None else pos.data_ptr(), we.data_ptr(), pe.data_ptr(), te.data_ptr(),
                                     self._emb("LayerNorm.weight").data_ptr(), self._emb("LayerNorm.bias").data_ptr(),
                                     A["emb_z"].data_ptr(), emb_out.data_ptr(), A["emb_mean"].data_ptr(), A["emb_rstd"].data_ptr(),
                                     M, Lseq, self.E, we.shape[0], te.shape[0], pe.shape[0], eps,
                                     -p_h if self.emb_dropout_pre_ln else p_h, seed * 1000003 + 17, dt, s)
        L.check(rc, "amdseg_embed_ln_fwd")
        if self.emb_project:                # x0 = emb_x Wp^T + bp (MFMA operand copy of the 2 x E x H matrix made per call: 64 KB)
            wp = self.fp.params[self.prefix + "embeddings_project.weight"]
            ops.gemm_nt(emb_out, wp.detach().to(torch.bfloat16), ops.EPI_BIAS, bias=self.fp.params[self.prefix + "embeddings_project.bias"].detach(),
                        out=A["x"][0])
        mb = A["mask_bias"].data_ptr()
        saved = []
        sink = getattr(self, "_hidden_sink", None)          # output_hidden_states: fp32 copies of the embedding output and of every layer output

        def keep_hidden(buf):
            h = torch.empty(M, self.H, dtype=torch.float32, device=self.device)
            L.check(lib.amdseg_dropout(buf.data_ptr(), h.data_ptr(), M * self.H, 0.0, 0, dt, L.F32, s), "amdseg_dropout(copy)")
            sink.append(h.view(B, Lseq, self.H))
        if sink is not None:
            keep_hidden(A["x"][0])
        for i in range(self.nlayers):
            saved.append(self._layer_forward(lib, cfg, lparams[i], A, i, mb, s, train))
            if sink is not None:
                keep_hidden(A["x"][i + 1] if train else A["x"][(i + 1) % 2])
        # the result is a FRESH tensor per call (the caller and autograd keep it; arena buffers are overwritten by the next forward)
        out = torch.empty(M, self.H, dtype=torch.float32, device=self.device)
        rc = lib.amdseg_dropout(A["x_final"].data_ptr(), out.data_ptr(), M * self.H, p_out if train else 0.0,
                                seed * 1000003 + 29, dt, L.F32, s)
        L.check(rc, "amdseg_dropout")
        ctx = dict(B=B, L=Lseq, ids=ids, tts=tts, pos=pos, seed=seed, p_h=p_h, p_a=p_a, p_out=p_out if train else 0.0,
                   layer_saved=saved, arena=A, gen=A["gen"], parity=fp32 == "parity")
        return out.view(B, Lseq, self.H), ctx

    def _precision(self, train):
        """False = bf16 fast path; True = exact-fp32 MFMA (inference only); "parity" = fp32 activations + split-bf16 contractions
        (forward and backward).  `config.amdseg_precision`: "bf16" (default) | "fp32" | "parity"; TRAINING with "fp32" runs "parity"
        (the exact-fp32 kernels have no backward; the split products carry 2^-16 relative error instead of 2^-24)."""
        want = getattr(self.cfg, "amdseg_precision", "bf16")
        if want == "bf16":
            return False
        if want not in ("fp32", "parity"):
            raise L.AmdsegError(f"amdseg_precision={want!r}: expected 'bf16', 'fp32' or 'parity'")
        return "parity" if (train or want == "parity") else True

    # hooks overridden by the Longformer engine
    def _position_ids(self, input_ids):
        return None

    def _layer_forward(self, lib, cfg, lp, A, i, mb, s, train):
        rc = lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(A["acts_struct"][i]), mb, i, s)
        L.check(rc, f"amdseg_bert_layer_fwd[{i}]")
        return None

    def _layer_backward(self, lib, cfg, A, i, mb, dy, other, s, saved):
        """critical path (everything but the weight gradients) on the current stream; the grouped weight-gradient GEMM of this
        layer on a second, lower-priority stream so that it fills the CUs its single wave of tiles leaves idle and runs under
        the next layer's backward"""
        if cfg.dtype == L.F32S:
            rc = lib.amdseg_bert_layer_bwd(C.byref(cfg), C.byref(self.lparams_parity[i]), C.byref(self.lgrads[i]),
                                           C.byref(A["acts_struct"][i]), C.byref(A["ws_struct"]), mb, dy.data_ptr(), other.data_ptr(), i, s)
            L.check(rc, f"amdseg_bert_layer_bwd[{i}] (parity)")
            return
        if not self.overlap_wgrad:
            rc = lib.amdseg_bert_layer_bwd(C.byref(cfg), C.byref(self.lparams[i]), C.byref(self.lgrads[i]),
                                           C.byref(A["acts_struct"][i]), C.byref(A["ws_struct"]), mb, dy.data_ptr(), other.data_ptr(), i, s)
            L.check(rc, f"amdseg_bert_layer_bwd[{i}]")
            return
        k = i & 1
        main = torch.cuda.current_stream()
        if self._wgrad_done[k] is not None:
            main.wait_event(self._wgrad_done[k])          # scratch set k is free again (its last reader was layer i+2's wgrad)
        args = (C.byref(cfg), C.byref(self.lparams[i]), C.byref(self.lgrads[i]), C.byref(A["acts_struct"][i]), C.byref(A["ws_structs"][k]),
                mb, dy.data_ptr(), other.data_ptr(), i)
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_bwd(*args, s), f"amdseg_bert_layer_bwd[{i}].1")
        cfg.phase = 6
        L.check(lib.amdseg_bert_layer_bwd(*args, s), f"amdseg_bert_layer_bwd[{i}].6")
        ev = torch.cuda.Event()
        ev.record(main)
        ws = self._wgrad_stream
        ws.wait_event(ev)
        cfg.phase = 4
        L.check(lib.amdseg_bert_layer_bwd(*args, ws.cuda_stream), f"amdseg_bert_layer_bwd[{i}].4")
        cfg.phase = 0
        done = torch.cuda.Event()
        done.record(ws)
        self._wgrad_done[k] = done
        self._wgrad_last = done

    def backward(self, ctx, dseq, accumulate=True):
        """dseq: fp32 [B, L, H] gradient of the encoder output.  Writes every parameter gradient into flat_g."""
        budget = getattr(self, "_bwd_cu_budget", 0) if ((self.buckets is not None and self.grad_sync) or getattr(self, "_bwd_cu_budget_always", False)) else 0
        if not budget:
            return self._backward(ctx, dseq, accumulate)
        # data parallel: the bucket all-reduces run beside this backward and their RCCL channels hold CUs (tile widths are chosen at launch time).
        # The budget lives in this engine's context (amdseg_ctx), which every composite call of this backward names in its cfg; the context is
        # also bound for the cfg-less GEMM calls of the Longformer / PoNet / embedding-projection backward.
        prev = self.ctx.set_cu_budget(budget)
        try:
            with self.ctx.bound():
                return self._backward(ctx, dseq, accumulate)
        finally:
            self.ctx.set_cu_budget(prev)

    def _backward(self, ctx, dseq, accumulate=True):
        B, Lseq = ctx["B"], ctx["L"]
        M = B * Lseq
        A = ctx["arena"]
        self.fp.grad_is_zero = False
        self._rest_reduced = False
        if A["gen"] != ctx["gen"]:
            raise L.AmdsegError(f"the activations saved by this forward (B={B}, L={Lseq}) were overwritten: more than "
                                f"{self.max_live_arenas} training forwards at this shape were alive before their backward ran "
                                f"(raise engine.max_live_arenas)")
        ws = A["ws"]
        lib = L.load()
        s = torch.cuda.current_stream().cuda_stream
        # the encoder layers' slice of flat_g may hold the previous step's gradients (lazy_zero): this backward overwrites them
        layer_acc = accumulate and not self.fp.grad_stale
        if not accumulate:
            self.fp.grad_stale = False
        cfg = self._cfg_struct(B, Lseq, ctx["p_h"], ctx["p_a"], ctx["seed"], layer_acc)
        cfg.kend = A["kend"].data_ptr() if (self.skip_padded_chunks and "kend" in A) else None
        cfg.seq_order = A["seq_order"].data_ptr() if (self.skip_padded_chunks and "seq_order" in A) else None
        adt = L.F32 if ctx.get("parity") else L.BF16            # dtype of the activation gradients
        if ctx.get("parity"):
            cfg.dtype = L.F32S
        dseq = dseq.contiguous()
        if self.skip_padded_rows_bwd and "pad_runs" in A and cfg.kend and (self.H % 4) == 0:
            # rows of trailing padding: is their incoming gradient an exact zero (it is whenever the loss ignores them)?  Then it stays
            # zero through every layer and the GEMMs of the backward drop those rows (include/amdseg.h, amdseg_bert_cfg.pad_guard)
            if self._pad_guard is None:
                self._pad_guard = torch.zeros(1, dtype=torch.int32, device=self.device)
            L.check(lib.amdseg_pad_rows_guard(dseq.data_ptr(), A["kend"].data_ptr(), B, Lseq, self.H, self._pad_guard.data_ptr(), s),
                    "amdseg_pad_rows_guard")
            cfg.pad_guard = self._pad_guard.data_ptr()
            cfg.pad_runs = A["pad_runs"].data_ptr()
            cfg.pad_counts = A["pad_counts"].data_ptr()
        dy, other = ws["dy"]
        rc = lib.amdseg_dropout(dseq.data_ptr(), dy.data_ptr(), M * self.H, ctx["p_out"], ctx["seed"] * 1000003 + 29, L.F32, adt, s)
        L.check(rc, "amdseg_dropout(bwd)")
        mb = A["mask_bias"].data_ptr()
        for i in reversed(range(self.nlayers)):
            self._layer_backward(lib, cfg, A, i, mb, dy, other, s, ctx["layer_saved"][i])
            dy, other = other, dy
            if self.buckets is not None and self.grad_sync:   # data parallel: this layer's gradient slice is final -> start its all-reduce
                if self.overlap_wgrad and self._wgrad_last is not None:
                    with torch.cuda.stream(self._wgrad_stream):      # ... ordered after the weight-gradient GEMM's stream
                        self.buckets.reduce_layer(i)
                else:
                    self.buckets.reduce_layer(i)
        self.fp.grad_stale = False                  # every layer's gradients have been written
        # embeddings: out = dropout(LN(z)) (BigBird: LN(dropout(z))); grads of LN affine + the three tables
        Ew = self.E
        if self.emb_project:
            # d(Wp) (+)= dx0^T emb_x, d(bp) (+)= colsum(dx0) in one grouped launch; d(emb_x) = dx0 Wp (the NT kernel on the transposed operand copy)
            wp = self.fp.params[self.prefix + "embeddings_project.weight"]
            gw = self.fp.view(self.fp.flat_g, self.prefix + "embeddings_project.weight")
            gb = self.fp.view(self.fp.flat_g, self.prefix + "embeddings_project.bias")
            dx0 = dy.view(M, self.H) if dy.dim() == 2 else dy.reshape(M, self.H)
            ops.gemm_tn_grouped([dx0], [A["emb_x"]], [gw], accumulate=bool(accumulate), colsums=[gb])
            dy, other = A["demb"]
            ops.gemm_nt(dx0, wp.detach().t().contiguous().to(torch.bfloat16), ops.EPI_NONE, out=dy)
        if ctx["p_h"] > 0 and not self.emb_dropout_pre_ln:
            rc = lib.amdseg_dropout(dy.data_ptr(), other.data_ptr(), M * Ew, ctx["p_h"], ctx["seed"] * 1000003 + 17, adt, adt, s)
            L.check(rc, "amdseg_dropout(emb bwd)")
            dy, other = other, dy
        g = lambda n: self.fp.view(self.fp.flat_g, self.prefix + "embeddings." + n)        # noqa: E731
        we, pe, te = g("word_embeddings.weight"), g("position_embeddings.weight"), g("token_type_embeddings.weight")
        if not accumulate:
            we.zero_(); pe.zero_(); te.zero_()
        # the LayerNorm backward's third column sum (its "dbias" = colsum of dz) IS the gradient of token-type row 0 when every row has type 0
        # (one-segment inputs); amdseg_embed_bwd then only moves the rows of other types out of it (type_vocab < 0) -- no hot-row atomics
        type0_sum = not (ctx["p_h"] > 0 and self.emb_dropout_pre_ln)      # (pre-LN dropout: dz is masked AFTER this LayerNorm backward)
        rc = lib.amdseg_ln_bwd(dy.data_ptr(), A["emb_z"].data_ptr(), A["emb_mean"].data_ptr(), A["emb_rstd"].data_ptr(),
                               self._emb("LayerNorm.weight").data_ptr(), other.data_ptr(), None, ws["partials"].data_ptr(),
                               g("LayerNorm.weight").data_ptr(), g("LayerNorm.bias").data_ptr(), te.data_ptr() if type0_sum else None,
                               M, Ew, 0.0, 0, 1 if accumulate else 0, adt, s)
        L.check(rc, "amdseg_ln_bwd(emb)")
        if ctx["p_h"] > 0 and self.emb_dropout_pre_ln:
            rc = lib.amdseg_dropout(other.data_ptr(), dy.data_ptr(), M * Ew, ctx["p_h"], ctx["seed"] * 1000003 + 17, adt, adt, s)
            L.check(rc, "amdseg_dropout(emb bwd, pre-LN)")
            dy, other = other, dy
        pad = self.cfg.pad_token_id if getattr(self.cfg, "pad_token_id", None) is not None else -1
        pos = ctx.get("pos")
        det = self.deterministic or bool(getattr(self.cfg, "amdseg_deterministic", False))
        rc = lib.amdseg_embed_bwd(other.data_ptr(), ctx["ids"].data_ptr(), ctx["tts"].data_ptr(), None if pos is None else pos.data_ptr(),
                                  we.data_ptr(), pe.data_ptr(), te.data_ptr(), M, Lseq, Ew, 0 if det else we.shape[0],
                                  -te.shape[0] if type0_sum else te.shape[0], 0 if (det and pos is not None) else pe.shape[0], pad, adt, s)
        L.check(rc, "amdseg_embed_bwd")
        if det:                                     # the word (and explicit position) tables without atomics: sums over a stable sort
            for keys, table, skip in ((ctx["ids"], we, pad), (pos, pe, -1)):
                if keys is None:
                    continue
                order = torch.sort(keys.reshape(-1), stable=True)[1]
                rc = lib.amdseg_scatter_rows_sorted(other.data_ptr(), keys.data_ptr(), order.data_ptr(), table.data_ptr(), M, Ew,
                                                    table.shape[0], skip, adt, s)
                L.check(rc, "amdseg_scatter_rows_sorted")
        self._embed_backward_fixup(pe, pad)
        if self.buckets is not None and self.grad_sync and not self._rest_reduced:
            self.buckets.reduce_embeddings()        # the tail bucket (the embedding tables: 94 MB of bert-base's exposed exchange) starts here
        if self.overlap_wgrad:                      # every consumer of flat_g (clip, AdamW, torch optimizers) is on the current stream
            main = torch.cuda.current_stream()
            for ev in self._wgrad_done:
                if ev is not None:
                    main.wait_event(ev)
        self._release_arena(A, ctx["gen"])

    def _embed_backward_fixup(self, dpos, pad):
        pass

    # ------------------------------------------------------------------------------------------------ optimiser
    def zero_grad(self):
        self.fp.flat_g.zero_()
        self.fp.grad_is_zero = True
        self.fp.grad_stale = False
        self._rest_reduced = False
        if self.buckets is not None:
            self.buckets.reset_norm()

    def grad_norm_and_clip_coef(self, max_norm, extra_scale=1.0):
        sc = self._scratch
        self.fp.flush_stale()                       # (no backward since the last lazy optimiser step: those gradients are zeros)
        if self.buckets is not None and self.buckets.norm_is_complete():
            ssq = self.buckets.sumsq                   # accumulated bucket by bucket right behind each all-reduce (dp.GradBuckets)
        else:
            ssq = sc["sumsq"]
            ops.sumsq(self.fp.flat_g, ssq, sc["partials"])
        ops.clip_coef(ssq, max_norm, extra_scale, sc["coef"], sc["norm"])
        return sc["norm"], sc["coef"]

    def set_param_flags(self, decay_names=None):
        """per-parameter optimiser flags for the fused AdamW: `decay_names` = the parameters weight decay applies to (None: all of
        them; HF Trainer excludes biases and LayerNorm weights, [hf] trainer.py:1168-1215); frozen parameters (requires_grad False,
        e.g. mmvts `freeze_text_encoder`) are never updated.  Stored as one byte per 64-element chunk of the flat buffer."""
        fp = self.fp
        flags = torch.zeros(fp.numel // 64, dtype=torch.uint8)
        for n, p in fp.params.items():
            o, c = fp.offsets[n] // 64, (p.numel() + 63) // 64
            flags[o:o + c] = (1 if (decay_names is None or n in decay_names) else 0) | (0 if p.requires_grad else 2)
        self._chunk_flags = flags.to(self.device)

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, grad_scale=1.0,
                   zero_grad=True, coef=None):
        """clip_grad_norm_(max_grad_norm) + torch.optim.AdamW step over the flat buffers, fused; refreshes the bf16 shadows.
        `coef`: a device scalar already produced by `grad_norm_and_clip_coef` for these gradients (the Trainer asks for the norm first)."""
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.fp.flat_p)
            self.adam_v = torch.zeros_like(self.fp.flat_p)
        self.opt_step += 1
        if coef is None:
            _, coef = self.grad_norm_and_clip_coef(max_grad_norm, grad_scale)
        # the bf16 compute copies ride the AdamW pass (its `shadow` output); what is left for the refresh are the transposes, made from them
        ride = getattr(self, "_parity", None) is None
        self.fp.flush_stale()
        fp, flags = self.fp, getattr(self, "_chunk_flags", None)
        lb = fp.layers_begin
        lazy = bool(zero_grad) and self.lazy_zero and fp.layers_dense and 0 < lb < fp.numel and not self.ddp_compat()
        with self.ctx.bound():              # (the launch timer of this engine's context also sees the cfg-less amdseg_adamw)
            for a, b, zg in (((0, lb, True), (lb, fp.numel, False)) if lazy else ((0, fp.numel, zero_grad),)):
                # (bf16 compute copies exist for the encoder layers' matrices only: the front part -- embeddings, heads -- is read in fp32)
                ops.adamw(fp.flat_p[a:b], fp.flat_g[a:b], self.adam_m[a:b], self.adam_v[a:b], self.shadow[a:b] if (ride and (b > lb or not lazy)) else None, lr, betas[0],
                          betas[1], eps, weight_decay, self.opt_step, gscale=coef, zero_grad=zg,
                          chunk_flags=None if flags is None else flags[a // 64:(b + 63) // 64])
        fp.grad_stale = lazy
        self.fp.grad_is_zero = bool(zero_grad)
        self._rest_reduced = False
        if self.buckets is not None:
            self.buckets.reset_norm()
        self._fused_owner = True
        self.refresh_shadows(force=True, copies_done=ride)


class EncoderFn(torch.autograd.Function):
    """autograd glue: forward/backward of the whole encoder run in HIP; parameter grads are written straight into
    the flat gradient buffer (param.grad views), so autograd only carries the activation gradient.

    The kernels work on 64-token blocks and 128-row GEMM tiles; any other [B, L] the caller hands over (the reference accepts every
    shape) is padded here with masked pad tokens / fully masked sequences and the output is cut back, which leaves the valid tokens'
    results unchanged (padded keys are masked exactly as the reference masks its own padding)."""

    @staticmethod
    def aligned_shape(B, Lq):
        Lp = -(-Lq // 64) * 64
        Bp = B
        while (Bp * Lp) % 128:
            Bp += 1
        return Bp, Lp

    @staticmethod
    def forward(ctx, trigger, engine, input_ids, attention_mask, token_type_ids, train, seed, p_out, *params):
        ctx.nparams = len(params)                           # > 0: DDP-compatible mode (engine.ddp_compat)
        B, Lq = input_ids.shape
        Bp, Lp = EncoderFn.aligned_shape(B, Lq)
        ctx.shapes = (B, Lq, Bp, Lp)
        if (Bp, Lp) != (B, Lq):
            pad_id = getattr(engine.cfg, "pad_token_id", None) or 0
            grow = (0, Lp - Lq, 0, Bp - B)
            input_ids = torch.nn.functional.pad(input_ids, grow, value=pad_id)
            attention_mask = torch.nn.functional.pad(attention_mask, grow, value=0)
            token_type_ids = torch.nn.functional.pad(token_type_ids, grow, value=0)
        xp = getattr(engine, "_explicit_pos", None)
        if xp is not None and (Bp, Lp) != (B, Lq):
            engine._explicit_pos = torch.nn.functional.pad(xp.reshape(B, Lq), (0, Lp - Lq, 0, Bp - B), value=0)
        out, ectx = engine.forward(input_ids, attention_mask, token_type_ids, train, seed, p_out)
        sink = getattr(engine, "_hidden_sink", None)
        if sink is not None and (Bp, Lp) != (B, Lq):
            sink[:] = [h[:B, :Lq].contiguous() for h in sink]
        ctx.engine, ctx.ectx = engine, ectx
        if train:                                           # graph dropped without a backward: the arena is free again
            weakref.finalize(ctx, BertEncoderEngine._release_arena, ectx["arena"], ectx["gen"])
        if (Bp, Lp) != (B, Lq):
            out = out[:B, :Lq].contiguous()
        return out

    @staticmethod
    def backward(ctx, dseq):
        B, Lq, Bp, Lp = ctx.shapes
        if (Bp, Lp) != (B, Lq):
            full = dseq.new_zeros((Bp, Lp, dseq.shape[-1]))
            full[:B, :Lq] = dseq
            dseq = full
        eng = ctx.engine
        if ctx.nparams:
            grads = eng.compat_backward(lambda: eng.backward(ctx.ectx, dseq, accumulate=True))
            return (torch.zeros(1, device=dseq.device),) + (None,) * 7 + grads
        eng.backward(ctx.ectx, dseq, accumulate=True)
        return (None,) * 8                              # (the trigger needs no gradient: None = nothing to accumulate, no fill / add kernel per step)


class RowDotFn(torch.autograd.Function):
    """small-C linear head (classifier H->2 over every token) on the HIP rowdot kernels."""

    @staticmethod
    def forward(ctx, x, W, b):
        M = x.numel() // x.shape[-1]
        x2 = x.reshape(M, x.shape[-1])
        ctx.save_for_backward(x2, W)
        ctx.shape = x.shape
        return ops.rowdot_fwd(x2, W, b).view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dl):
        x2, W = ctx.saved_tensors
        dl2 = dl.reshape(-1, W.shape[0]).contiguous().float()
        dW = torch.empty_like(W); db = torch.empty(W.shape[0], dtype=torch.float32, device=W.device)
        dx = ops.rowdot_bwd(x2, W, dl2, dW=dW, db=db, need_dx=True)
        return dx.view(ctx.shape), dW, db


class HeadInputsFn(torch.autograd.Function):
    """everything the loss heads read from the encoder output x [M, H] (fp32) in one step: the token classifier
    logits = x W^T + b over every row (HIP rowdot kernels) and the rows requested by the CSSL / TSSP / cos-sim heads
    (one gather).  Backward writes ONE dense gradient: the classifier's dx with the head-row gradients scatter-added."""

    @staticmethod
    def forward(ctx, x, W, b, rows):
        ctx.save_for_backward(x, W, rows)
        logits = ops.rowdot_fwd(x, W, b)
        feats = x.index_select(0, rows) if rows.numel() else x.new_zeros((0, x.shape[1]))
        return logits, feats

    @staticmethod
    def backward(ctx, dlogits, dfeats):
        x, W, rows = ctx.saved_tensors
        dW = torch.empty_like(W); db = torch.empty(W.shape[0], dtype=torch.float32, device=W.device)
        if dlogits is None:
            dlogits = torch.zeros((x.shape[0], W.shape[0]), dtype=torch.float32, device=x.device)
        dx = ops.rowdot_bwd(x, W, dlogits.contiguous().float(), dW=dW, db=db, need_dx=True)
        if dfeats is not None and rows.numel():
            dx.index_add_(0, rows, dfeats.to(dx.dtype))
        return dx, dW, db, None


class FusedHeadsFn(torch.autograd.Function):
    """the training-step loss of the wrapper in one forward and one backward pass of HIP kernels (csrc/heads.hip): classifier logits
    over every token (rowdot), token cross-entropy per half (anchor | augmented), CSSL InfoNCE over index lists, TSSP Linear + CE.
    Index lists live in ONE int64 device buffer (`idx`, uploaded once per step); `plan` holds offsets / counts (host ints).
    Returns (total loss, logits [M, C]); only the loss is differentiable."""

    @staticmethod
    def forward(ctx, x, Wc, bc, Wt, bt, labels_all, idx, class_w, plan):
        M, H = x.shape
        C_ = Wc.shape[0]
        logits = ops.rowdot_fwd(x, Wc, bc)
        dev = x.device
        gamma = float(plan.get("gamma", 0.0))
        out8 = torch.empty(16, dtype=torch.float32, device=dev)
        acc4 = torch.empty(32 * ((M + 255) // 256) + plan["n_anchor"] + plan["nt"] + 8, dtype=torch.float32, device=dev)
        unit = torch.empty(M, C_ * (2 if gamma != 0.0 else 1), dtype=torch.float32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        P = plan
        args = (x.data_ptr(), M, H, logits.data_ptr(), labels_all.data_ptr(), None if class_w is None else class_w.data_ptr(),
                C_, P["nseg"], unit.data_ptr(), out8.data_ptr(), acc4.data_ptr(), idx.data_ptr(), P["feat_off"],
                P["anchor_off"], P["lists_off"], P["n_anchor"], P["n_list"], P["pk"], P["temp"],
                None if Wt is None else Wt.data_ptr(), None if bt is None else bt.data_ptr(), P["t_rows_off"],
                P["t_labels_off"], P["nt"], 0 if Wt is None else Wt.shape[0], P["w_ts"], P["w_cl"], P["w_tssp2"])
        if gamma != 0.0:
            L.check(L.load().amdseg_heads_fwd_focal(*args, gamma, s), "amdseg_heads_fwd_focal")
        else:
            L.check(L.load().amdseg_heads_fwd(*args, s), "amdseg_heads_fwd")
        ctx.save_for_backward(x, Wc, Wt if Wt is not None else x.new_empty(0), bt if bt is not None else x.new_empty(0), idx, unit, out8)
        ctx.plan, ctx.has_tssp = P, Wt is not None
        ctx.mark_non_differentiable(logits)
        return out8[4], logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        x, Wc, Wt, bt, idx, unit, out8 = ctx.saved_tensors
        P = ctx.plan
        M, H = x.shape
        C_ = Wc.shape[0]
        lib = L.load()
        s = torch.cuda.current_stream().cuda_stream
        g = gloss.reshape(1).float().contiguous()
        dlogits = torch.empty(M, C_, dtype=torch.float32, device=x.device)
        gamma = float(P.get("gamma", 0.0))
        if gamma != 0.0:
            L.check(lib.amdseg_heads_bwd_ce_focal(g.data_ptr(), M, C_, P["nseg"], unit.data_ptr(), out8.data_ptr(), P["w_ts"], gamma,
                                                  dlogits.data_ptr(), s), "amdseg_heads_bwd_ce_focal")
        else:
            L.check(lib.amdseg_heads_bwd_ce(g.data_ptr(), M, C_, P["nseg"], unit.data_ptr(), out8.data_ptr(), P["w_ts"], dlogits.data_ptr(), s),
                    "amdseg_heads_bwd_ce")
        # the heads' parameter gradients: straight into the parameters' .grad views of the flat gradient buffer (accumulating, as autograd
        # would) when the caller handed them over (`plan["direct"]`: native mode, every .grad attached) -- otherwise returned to autograd, which
        # adds them with one elementwise kernel per parameter
        direct = P.get("direct")
        if direct is not None and any(t is not None and t.grad is None for t in direct):
            direct = None
        if direct is not None:
            dWc, dbc = direct[0].grad, direct[1].grad
            dx = ops.rowdot_bwd(x, Wc, dlogits, dW=dWc, db=dbc, need_dx=True, accumulate=True)
        else:
            dWc = torch.empty_like(Wc); dbc = torch.empty(C_, dtype=torch.float32, device=x.device)
            dx = ops.rowdot_bwd(x, Wc, dlogits, dW=dWc, db=dbc, need_dx=True)
        dWt = dbt = None
        if ctx.has_tssp:
            if direct is not None:
                dWt, dbt = direct[2].grad, direct[3].grad       # (amdseg_heads_bwd_rows ADDS its fixed-point sums to them)
            else:
                dWt = torch.zeros_like(Wt); dbt = torch.zeros_like(bt)
        if P["n_anchor"] > 0 or P["nt"] > 0:
            # scatter sums as 64-bit fixed point (order independent: the step stays bit-reproducible), zeroed by the call
            Ct = Wt.shape[0] if ctx.has_tssp else 0
            fix = torch.empty((P["n_feat"] + P["nt"]) * H + Ct * H + Ct, dtype=torch.int64, device=x.device)
            L.check(lib.amdseg_heads_bwd_rows(g.data_ptr(), x.data_ptr(), M, H, dx.data_ptr(), idx.data_ptr(), P["feat_off"], P["anchor_off"],
                                              P["lists_off"], P["n_anchor"], P["n_list"], P["pk"], P["temp"],
                                              Wt.data_ptr() if ctx.has_tssp else None, bt.data_ptr() if ctx.has_tssp else None,
                                              P["t_rows_off"], P["t_labels_off"], P["nt"], Wt.shape[0] if ctx.has_tssp else 0,
                                              None if dWt is None else dWt.data_ptr(), None if dbt is None else dbt.data_ptr(),
                                              P["w_cl"], P["w_tssp2"], P["n_feat"], fix.data_ptr(), fix.numel() * 8, s),
                    "amdseg_heads_bwd_rows")
        if direct is not None:
            return dx, None, None, None, None, None, None, None, None
        return dx, dWc, dbc, dWt, dbt, None, None, None, None
