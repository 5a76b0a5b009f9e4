"""The fast path behind the reference's training surface.

The reference trains with `transformers.Trainer` (ts_sentence_seq_labeling.py:43,1077-1094; run_finetune.sh:29-31,61,73-76:
AdamW lr 5e-5 linear decay, clip 1.0, gradient_accumulation_steps 2, torch DDP across the GPUs of the node).  With the stock
Trainer the drop-in model classes work, but the optimiser is torch's (a per-tensor / foreach pass over ~200 parameters plus a
separate clip pass) and multi-GPU goes through torch DDP (`engine.ddp_compat`: a snapshot of the flat gradient per backward).

`Trainer` below is a subclass with the same constructor; swapping the import line in the driver
    from transformers import Trainer      ->      from spokennlp_amd.trainer import Trainer
puts the engine's own pieces behind the same loop, nothing else changes (callbacks, checkpoints, evaluation, lr scheduler):
  * `create_optimizer`  -> `AmdsegFusedAdamW`: ONE HIP pass over the flat fp32 parameter / gradient / moment buffers with the clip
    coefficient applied inside (no separate scaling pass), HF's decay / no-decay parameter groups as a per-chunk flag byte, gradient
    zeroing fused, bf16 weight copies refreshed;
  * `_clip_grad_norm`   -> the HIP sum-of-squares reduction (device scalar, no host sync); the optimiser consumes the coefficient;
  * multi-GPU           -> the engine's per-layer RCCL all-reduce launched from inside backward (dp.GradBuckets) instead of torch DDP;
    `gradient_accumulation_steps` > 1 runs the non-final micro-steps under `no_sync()` so every bucket is reduced exactly once.
"""
import contextlib
import os

import torch
import transformers

# multi-GPU launches: the ring of the engine's overlapped gradient exchange gets at most 32 channels unless the launcher chose otherwise (every RCCL
# channel holds a CU for the length of the exchange and the GEMM grids of backward are sized by rounds over the CUs: dp.init_from_env,
# profiles/r04_cu_budget_probe.md).  Has to be in the environment before the process group exists, i.e. before TrainingArguments is built.
# (It therefore also applies to a run that later asks for `Trainer(amdseg_native=False)` -- torch DDP, no CU budget --: AMDSEG_NCCL_CHANNEL_CAP=0 in
# the environment leaves NCCL_MAX_NCHANNELS alone; any other value is used as the cap.  The engine logs the budget it derived from it once.)
if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 and os.environ.get("AMDSEG_NCCL_CHANNEL_CAP", "32") not in ("0", ""):
    os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("AMDSEG_NCCL_CHANNEL_CAP", "32"))

from . import lib as L


class AmdsegFusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps added to sqrt(v_hat)) executed by `amdseg_adamw` over the
    engine's flat buffers.  One param group holding every parameter of the model (lr schedulers write `param_groups[0]["lr"]`);
    `decay_names` selects the parameters weight decay applies to."""
    amdseg_fused = True

    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decay_names=None, max_grad_norm=0.0,
                 param_groups=None):
        """`param_groups` (optional): torch-style groups over THIS model's parameters, as HF's `create_optimizer` builds them (decay /
        no-decay).  The one HIP pass has ONE learning rate, (beta1, beta2) and eps: groups that differ in any of them (layer-wise lr decay,
        a head with its own lr) are refused loudly instead of silently collapsing to the first group's values; groups that differ only in
        `weight_decay` (some value / 0) become the per-chunk decay flag."""
        if not hasattr(model, "engine"):
            raise L.AmdsegError("AmdsegFusedAdamW needs one of the spokennlp_amd model classes (it steps the engine's flat buffers)")
        self.model = model
        if param_groups is not None:
            lr, betas, eps, weight_decay, decay_names = self._collapse_groups(model, param_groups, dict(lr=lr, betas=betas, eps=eps,
                                                                                                        weight_decay=weight_decay))
        self.decay_names = None if decay_names is None else set(decay_names)
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self._coef = None
        self._engine_ref = None
        self._flag_key = None
        self._pending_state = None
        super().__init__([p for p in model.parameters()], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @staticmethod
    def _collapse_groups(model, groups, defaults):
        names = {id(p): n for n, p in model.named_parameters()}
        groups = [dict(defaults, **g) for g in groups]
        for k in ("lr", "betas", "eps"):
            vals = {tuple(g[k]) if isinstance(g[k], (list, tuple)) else g[k] for g in groups}
            if len(vals) > 1:
                raise L.AmdsegError(f"AmdsegFusedAdamW runs one pass with ONE {k}; the parameter groups ask for {sorted(map(str, vals))}. "
                                    f"Use a torch optimiser (Trainer(amdseg_native=False)) for per-group {k}.")
        wds = sorted({float(g["weight_decay"]) for g in groups})
        nonzero = [w for w in wds if w != 0.0]
        if len(nonzero) > 1:
            raise L.AmdsegError(f"AmdsegFusedAdamW supports one weight-decay value plus 0 (decay flag per parameter); got {wds}")
        decay = set()
        for g in groups:
            for p_ in g["params"]:
                if id(p_) not in names:
                    raise L.AmdsegError("AmdsegFusedAdamW: a parameter group holds a tensor that is not a parameter of this model")
                if float(g["weight_decay"]) != 0.0:
                    decay.add(names[id(p_)])
        g0 = groups[0]
        return g0["lr"], tuple(g0["betas"]), g0["eps"], (nonzero[0] if nonzero else 0.0), (decay if nonzero else None)

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):            # torch.optim.Optimizer.__init__ adds the first (only) group through here
            raise L.AmdsegError("AmdsegFusedAdamW holds every parameter of the model in one group (one lr / betas / eps for the flat pass); "
                                "a second group would be ignored by the kernel")
        super().add_param_group(param_group)

    # ---- engine plumbing
    def _engine(self):
        eng = self.model.engine()
        old = self._engine_ref() if self._engine_ref is not None else None
        # frozen / decay flags follow the CURRENT requires_grad pattern (a head unfrozen mid-training must start to move)
        key = (self.param_groups[0]["weight_decay"] != 0, tuple(p.requires_grad for p in eng.fp.params.values()))
        if eng is not old:                         # first use, or the model rebuilt its engine (parameters were re-materialised)
            last = getattr(self, "_last", None)    # moments of the engine stepped last (kept alive here: the old engine may be gone already)
            if last is not None and eng.adam_m is None:
                # resize_token_embeddings / model.to / `p.data = new` mid-training: keep the moments and the step count when the flat
                # layout is unchanged; otherwise say so instead of silently restarting the bias correction from zero
                if last["layout"] == (eng.fp.numel, tuple(eng.fp.offsets.items())):
                    eng.adam_m = last["m"].to(eng.device); eng.adam_v = last["v"].to(eng.device); eng.opt_step = last["step"]
                else:
                    import warnings
                    warnings.warn("spokennlp_amd: the model rebuilt its parameter storage with a different layout (resized embeddings?); "
                                  "the AdamW moments and step count restart from zero", RuntimeWarning, stacklevel=3)
            import weakref
            self._engine_ref = weakref.ref(eng)
            self._flag_key = None
            if self._pending_state is not None:
                self._install_state(eng, self._pending_state)
                self._pending_state = None
        if key != self._flag_key:
            eng.set_param_flags(self.decay_names if key[0] else None)
            self._flag_key = key
        return eng

    def _live_engine(self):
        """the engine this optimiser last stepped, if the model still runs on it"""
        eng = self._engine_ref() if self._engine_ref is not None else None
        return eng if eng is not None and getattr(self.model, "_engine", None) is eng else None

    @staticmethod
    def _world():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def grad_norm(self, max_norm=None):
        """total gradient norm as a device scalar (torch.nn.utils.clip_grad_norm_'s return value); keeps the clip coefficient for the
        coming step().  Under the engine's data parallelism this first waits for the outstanding bucket reductions."""
        eng = self._engine()
        eng.finish_grad_sync()
        scale = 1.0 / self._world() if eng.buckets is not None else 1.0
        mx = self.max_grad_norm if max_norm is None else float(max_norm)
        norm, coef = eng.grad_norm_and_clip_coef(mx if mx != float("inf") else 0.0, scale)
        self._coef = coef
        return norm

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._engine()
        g = self.param_groups[0]
        if self._coef is None:
            self.grad_norm()
        eng.adamw_step(float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]), weight_decay=float(g["weight_decay"]),
                       coef=self._coef, zero_grad=True)
        self._coef = None
        self._last = dict(m=eng.adam_m, v=eng.adam_v, step=eng.opt_step, layout=(eng.fp.numel, tuple(eng.fp.offsets.items())))
        return loss

    def zero_grad(self, set_to_none=True):
        # step() already zeroed the flat gradient buffer inside the AdamW pass; the views stay attached
        eng = self._engine()
        if not eng.fp.grad_is_zero:
            eng.zero_grad()
        elif not set_to_none:
            # an explicit "make the gradients literally zero": the slice the fused AdamW left for the next backward to overwrite is zeroed now
            # (a reader of p.grad between step() and the next backward -- logging callbacks, custom clipping -- saw stale values)
            eng.fp.flush_stale()

    # ---- checkpointing (Trainer saves optimizer.state_dict() next to the model): the state is three flat tensors
    def state_dict(self):
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = list(range(len(self.param_groups[0]["params"])))
        eng = self._live_engine()
        state = {}
        if eng is not None and eng.adam_m is not None:
            state = {"amdseg_flat": {"step": torch.tensor(float(eng.opt_step)), "exp_avg": eng.adam_m, "exp_avg_sq": eng.adam_v,
                                     "numel": torch.tensor(float(eng.fp.numel))}}
        elif self._pending_state is not None:
            state = {"amdseg_flat": self._pending_state}
        return {"state": state, "param_groups": groups}

    @staticmethod
    def _install_state(eng, st):
        if int(st["numel"]) != eng.fp.numel:
            raise L.AmdsegError("optimizer checkpoint does not match this model's flat parameter layout")
        eng.adam_m = st["exp_avg"].to(device=eng.device, dtype=torch.float32).clone()
        eng.adam_v = st["exp_avg_sq"].to(device=eng.device, dtype=torch.float32).clone()
        eng.opt_step = int(st["step"])

    def load_state_dict(self, state_dict):
        for g, new in zip(self.param_groups, state_dict["param_groups"]):
            for k, v in new.items():
                if k != "params":
                    g[k] = v
        st = state_dict.get("state", {}).get("amdseg_flat")
        if st is not None:
            eng = self._live_engine()
            if eng is not None:
                self._install_state(eng, st)
            else:
                self._pending_state = st


class NativeDataParallel(torch.nn.Module):
    """what `Trainer._wrap_model` returns for multi-GPU training INSTEAD of torch DDP: the same module, with the engine's bucketed
    gradient exchange switched on and DDP's `no_sync()` contract (accelerate looks it up for gradient accumulation)."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        if not module.engine().enable_data_parallel():
            raise L.AmdsegError("NativeDataParallel needs an initialised torch.distributed world of size > 1")

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        return self.module.no_sync()


import inspect as _inspect
_TRAINING_STEP_TAKES_COUNT = len(_inspect.signature(transformers.Trainer.training_step).parameters) >= 4


class Trainer(transformers.Trainer):
    """`transformers.Trainer` with the fused optimiser, HIP gradient norm and native data parallelism (see the module docstring).
    `amdseg_native=False` keeps the stock behaviour (torch optimiser, torch DDP)."""

    # the private hooks of transformers.Trainer this subclass overrides; a release that lacks one of them would run the stock clip
    # (`accelerator.clip_grad_norm_` on the flat-buffer views) BEFORE the engine's tail bucket is reduced -- silently wrong gradients
    _NATIVE_HOOKS = ("_clip_grad_norm", "_get_grad_norm", "get_decay_parameter_names", "_wrap_model", "_prepare_inputs", "create_optimizer")

    def __init__(self, *args, amdseg_native=True, **kwargs):
        self.amdseg_native = bool(amdseg_native)
        if self.amdseg_native:
            missing = [h for h in self._NATIVE_HOOKS if not hasattr(transformers.Trainer, h)]
            if missing:
                raise L.AmdsegError(f"spokennlp_amd.trainer.Trainer(amdseg_native=True) overrides transformers.Trainer.{', '.join(missing)}, which "
                                    f"transformers {transformers.__version__} does not have (tested with 5.x); pass amdseg_native=False to train "
                                    f"with the stock optimiser / torch DDP on the same model classes")
        targs = kwargs.get("args") or next((a for a in args if isinstance(a, transformers.TrainingArguments)), None)
        if self.amdseg_native and targs is not None and targs.dataloader_pin_memory:
            # batches arrive in pinned host memory; a blocking `.to(device)` (accelerate's default) makes the host wait for every
            # kernel queued so far once per step, so the next step's launches start on an idle GPU (measured: 19.1 vs 15.3 ms per
            # bert-base step).  Asynchronous copies from pinned memory are ordered on the compute stream and safe.
            ac = getattr(targs, "accelerator_config", None)
            if ac is not None and hasattr(ac, "non_blocking") and not ac.non_blocking:
                ac.non_blocking = True
        super().__init__(*args, **kwargs)
        self.amdseg_native = self.amdseg_native and hasattr(self.model, "engine")

    def training_step(self, model, inputs, num_items_in_batch=None):
        """`logging_nan_inf_filter` (a TrainingArguments default) makes the stock loop evaluate `torch.isnan(loss) or torch.isinf(loss)` in
        Python once per step ([hf] trainer.py, `_inner_training_loop`): a host read of a device scalar that drains the launch queue
        (bert-base: 19.0 vs 16.0 ms per step).  The same substitution -- a non-finite step loss is replaced, for the LOGGED running loss
        only, by the average since the last log -- is done here on the device and the host-side check is switched off for this run."""
        if _TRAINING_STEP_TAKES_COUNT:
            loss = super().training_step(model, inputs, num_items_in_batch)
        else:                                                   # transformers < 4.46: two arguments
            loss = super().training_step(model, inputs)
        if self.amdseg_native and hasattr(self, "_tr_loss") and hasattr(self, "_globalstep_last_logged"):
            if self.args.logging_nan_inf_filter:
                self.args.logging_nan_inf_filter = False
                self._amdseg_nan_filter = True
            if getattr(self, "_amdseg_nan_filter", False):
                loss = filter_nonfinite(loss, self._tr_loss, 1 + self.state.global_step - self._globalstep_last_logged)
        return loss

    # ---- batches: pinned host memory -> device on a copy stream
    def get_train_dataloader(self):
        """accelerate's DataLoaderShard copies every batch to the device on the COMPUTE stream: nine small H2D transfers that sit between
        the optimiser of step i and the embedding kernel of step i + 1 with the compute units idle (~0.8 ms of a 14.4 ms bert-base step).
        Here the shard hands out the pinned host batch and `_prepare_inputs` copies it on a side stream -- the host runs ~9 ms ahead of
        the GPU, so the copy finishes under step i's kernels and the compute stream only waits on an event that has long fired."""
        return self._host_batches(super().get_train_dataloader())

    def get_eval_dataloader(self, eval_dataset=None):
        return self._host_batches(super().get_eval_dataloader(eval_dataset))

    def get_test_dataloader(self, test_dataset):
        return self._host_batches(super().get_test_dataloader(test_dataset))

    def _host_batches(self, dl):
        if (self.amdseg_native and self.args.dataloader_pin_memory and torch.cuda.is_available() and self.args.device.type == "cuda"
                and type(dl).__name__ == "DataLoaderShard" and getattr(dl, "device", None) is not None):
            dl.device = None
            self._amdseg_side_copy = True
            if getattr(self, "_amdseg_copy_stream", None) is None:
                self._amdseg_copy_stream = torch.cuda.Stream(device=self.args.device)
        return dl

    def _prepare_inputs(self, inputs):
        if (getattr(self, "_amdseg_side_copy", False) and isinstance(inputs, dict)
                and any(isinstance(v, torch.Tensor) and v.device.type == "cpu" for v in inputs.values())):
            dev = self.args.device
            main = torch.cuda.current_stream(dev)
            moved = []
            with torch.cuda.stream(self._amdseg_copy_stream):          # destination blocks belong to the copy stream's allocator pool
                out = {}
                for k, v in inputs.items():
                    if isinstance(v, torch.Tensor) and v.device.type == "cpu":
                        v = v.to(dev, non_blocking=True)               # pinned source: asynchronous, the host allocator defers its reuse
                        moved.append(v)
                    out[k] = v
                ev = torch.cuda.Event()
                ev.record(self._amdseg_copy_stream)
            main.wait_event(ev)
            for t in moved:
                t.record_stream(main)                                  # freed blocks wait for the compute stream's readers
            # the model's heads build their index lists on the host from the label-like tensors: hand it the host originals of this batch, so
            # it does not copy them back and WAIT for that copy (a wait for the previous step's GPU work: the host could never run more than one
            # forward ahead, and every host hiccup reached the GPU)
            m = self.model
            while hasattr(m, "module"):
                m = m.module
            if hasattr(m, "amdseg_set_host_twins"):
                m.amdseg_set_host_twins({k: (out[k], v) for k, v in inputs.items()
                                         if isinstance(v, torch.Tensor) and v.device.type == "cpu" and out[k] is not v})
            inputs = out
        return super()._prepare_inputs(inputs)

    def train(self, *args, **kwargs):
        try:
            return super().train(*args, **kwargs)
        finally:
            if getattr(self, "_amdseg_nan_filter", False):         # hand the caller's TrainingArguments back unchanged
                self.args.logging_nan_inf_filter = True
                self._amdseg_nan_filter = False

    def _fused(self):
        opt = self.optimizer
        while opt is not None and not isinstance(opt, AmdsegFusedAdamW) and hasattr(opt, "optimizer"):
            opt = opt.optimizer                 # accelerate's AcceleratedOptimizer wrapper
        return opt if isinstance(opt, AmdsegFusedAdamW) else None

    def create_optimizer(self, model=None):
        if not self.amdseg_native or self.optimizer is not None or getattr(self, "optimizer_cls_and_kwargs", None) is not None:
            return super().create_optimizer(model)
        a = self.args
        if "adamw" not in str(a.optim).lower():
            raise L.AmdsegError(f"the fused optimiser implements AdamW; --optim {a.optim} needs amdseg_native=False")
        self.optimizer = AmdsegFusedAdamW(self.model, lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon,
                                          weight_decay=a.weight_decay, decay_names=self.get_decay_parameter_names(self.model),
                                          max_grad_norm=a.max_grad_norm)
        return self.optimizer

    def _wrap_model(self, model, training=True, dataloader=None):
        import torch.distributed as dist
        if (self.amdseg_native and training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                and self.args.parallel_mode == transformers.training_args.ParallelMode.DISTRIBUTED):
            if isinstance(model, NativeDataParallel):
                return model
            return NativeDataParallel(self.model)
        return super()._wrap_model(model, training, dataloader)

    def _clip_grad_norm(self, model):
        opt = self._fused()
        if opt is None:
            return super()._clip_grad_norm(model)
        return opt.grad_norm(self.args.max_grad_norm)

    def _get_grad_norm(self, model, grad_norm=None):
        opt = self._fused()
        if opt is None or grad_norm is not None:
            return super()._get_grad_norm(model, grad_norm=grad_norm)
        return opt.grad_norm(float("inf"))


def filter_nonfinite(loss, tr_loss, steps_since_log):
    """device-side form of the stock loop's `logging_nan_inf_filter` branch: tr_loss += tr_loss / steps_since_log when the step loss
    is nan / inf, += loss otherwise -- returned as the value to add"""
    avg = (tr_loss / steps_since_log).to(device=loss.device, dtype=loss.dtype)
    return torch.where(torch.isfinite(loss), loss, avg)


@contextlib.contextmanager
def accumulate(model, sync):
    """hand-written loops: `with accumulate(model, sync=is_last_micro_step): loss.backward()`"""
    if sync:
        yield
    else:
        with model.no_sync():
            yield
