"""MI355X-native PoNet token-classification model:
    alimeeting4mug/src/models/modeling_ponet.py:34-109  PoNetForTokenClassification
        = PoNetModel(input_ids, attention_mask, token_type_ids, segment_ids) -> dropout -> Linear(H, num_labels)
          -> CrossEntropy with labels forced to -100 where attention_mask != 1 (:86-98).
The reference imports the encoder from `modelscope.models.nlp.ponet` (modelscope==1.1.0), which is NOT in the reference tree
and not installable here: this module follows oracle/ponet_oracle.py (the published PoNet algorithm) and its parity with the
original is UNPINNED.  Same forward signature and return convention as the reference class; ModelScope's
`from_pretrained(model_name_or_path=..., task=...)` hub loader is not reproduced (HF `from_pretrained` / `state_dict` work,
parameter names `ponet.*`, `classifier.*`).

Engine: the BERT engine with an EXTERNAL token mixer (amdseg_bert_cfg.mixer = 1): one fused 5H-wide projection GEMM
(Hq | Hk | Ho | Hl | Hs), then
  * global aggregation: mean query, one softmax row per (sequence, head) over all tokens, weighted key sum -- the three O(L)
    passes are the MFMA row kernels of csrc/longformer.hip on column blocks of the projection (strided `_ld` entry points),
  * segment / local max-pooling + fusion and their backward: csrc/ponet.hip (HBM-bound row kernels),
followed by the shared attention-output / FFN half of the composite layer call.
"""
import ctypes as C
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import BertConfig, PreTrainedModel
from transformers.modeling_outputs import TokenClassifierOutput

from . import lib as L
from . import ops
from .engine import BertEncoderEngine, RowDotFn

PONET_LAYER_ORDER = ["attention.self.dense_q.weight", "attention.self.dense_k.weight", "attention.self.dense_o.weight",
                     "attention.self.dense_local.weight", "attention.self.dense_segment.weight",
                     "attention.self.dense_q.bias", "attention.self.dense_k.bias", "attention.self.dense_o.bias",
                     "attention.self.dense_local.bias", "attention.self.dense_segment.bias",
                     "attention.output.dense.weight", "attention.output.dense.bias",
                     "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                     "intermediate.dense.weight", "intermediate.dense.bias",
                     "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]


class PoNetConfig(BertConfig):
    model_type = "ponet"


# ------------------------------------------------------------------------------------------------ parameter container
class _Named(nn.Module):
    pass


def _layer(cfg):
    H, I = cfg.hidden_size, cfg.intermediate_size
    lay = _Named()
    lay.attention = _Named()
    lay.attention.self = _Named()
    for n in ("dense_q", "dense_k", "dense_o", "dense_local", "dense_segment"):
        setattr(lay.attention.self, n, nn.Linear(H, H))
    lay.attention.output = _Named()
    lay.attention.output.dense = nn.Linear(H, H)
    lay.attention.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
    lay.intermediate = _Named()
    lay.intermediate.dense = nn.Linear(H, I)
    lay.output = _Named()
    lay.output.dense = nn.Linear(I, H)
    lay.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
    return lay


class PoNetModel(nn.Module):
    """parameter container with the `ponet.*` names; its torch forward does not exist -- the HIP engine runs it"""

    def __init__(self, cfg, add_pooling_layer=False):
        super().__init__()
        H = cfg.hidden_size
        self.embeddings = _Named()
        self.embeddings.word_embeddings = nn.Embedding(cfg.vocab_size, H, padding_idx=cfg.pad_token_id)
        self.embeddings.position_embeddings = nn.Embedding(cfg.max_position_embeddings, H)
        self.embeddings.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, H)
        self.embeddings.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        # the reference driver rewrites this buffer when it extends the position table (ponet_topic_segmentation.py:482)
        self.embeddings.register_buffer("position_ids", torch.arange(cfg.max_position_embeddings).expand((1, -1)), persistent=False)
        self.encoder = _Named()
        self.encoder.layer = nn.ModuleList([_layer(cfg) for _ in range(cfg.num_hidden_layers)])


# ------------------------------------------------------------------------------------------------ engine
class PoNetEncoderEngine(BertEncoderEngine):
    supports_parity = False
    def __init__(self, module, config, device, bert_attr="ponet"):
        super().__init__(module, config, device, bert_attr=bert_attr, layer_order=PONET_LAYER_ORDER, nproj=5)
        H, heads = self.H, self.heads
        hm = torch.zeros(heads, H, device=device)
        for h in range(heads):
            hm[h, h * 64:(h + 1) * 64] = 1.0
        self.headmask = hm
        self.attn_keepmask = False                          # no softmax attention in this encoder
        # the global aggregation branch as streaming passes of csrc/ponet_global.hip (4 launches forward, 5 backward); AMDSEG_PN_LF_CHAIN=1 keeps
        # the rounds-1/2 formulation on the Longformer global-row kernels (10 / 12 launches and torch glue; same dropout decisions)
        self.fused_global = self.H <= 1024                  # (engine.fused_global = False: the Longformer global-row chain instead, tests)
        self._seg = None
        # amdseg_bert_cfg.pad_guard holds for the pooling mixer too: a padded token n has dctx_n = 0, so dHo_n = 0; it is no valid neighbour /
        # run member, so no local or segment maximum routes a gradient to it; it is a masked key of the global aggregation (p = 0): dproj_n = 0

    def set_segments(self, segment_ids):
        self._seg = segment_ids

    def _arena(self, B, Lseq, train, fp32=False):
        if fp32:
            raise L.AmdsegError("the PoNet path has no fp32 parity mode (its reference arithmetic is not available to pin against)")
        A = super()._arena(B, Lseq, train, fp32)
        if "pn" not in A:
            M, H, dev = B * Lseq, self.H, self.device
            nsave = self.nlayers if train else 1
            # part: the [M, H] 32-bit run-maximum keys of a layer (value | argmax, csrc/ponet.hip), kept for backward; parg: unused since the
            # key layout (one dummy shared by all layers keeps the C-ABI call shape)
            dummy = torch.empty(8, H, dtype=torch.int16, device=dev)
            A["pn"] = dict(part=[torch.empty(2 * M, H, dtype=torch.bfloat16, device=dev) for _ in range(nsave)],
                           parg=[dummy for _ in range(nsave)],
                           lf_partials=torch.empty(B * (Lseq // 64) * self.heads * H, dtype=torch.float32, device=dev),
                           vt=torch.empty(B * H * 32, dtype=torch.bfloat16, device=dev),
                           gscratch=ops.ponet_global_scratch(B, Lseq, H, self.heads, dev) if self.fused_global else None,
                           dpd=torch.empty(B, self.heads, Lseq, dtype=torch.float32, device=dev) if (train and self.fused_global) else None)
            if train:
                A["pn"].update(E=torch.empty(M, H, dtype=torch.bfloat16, device=dev), psum=torch.empty(M, H, dtype=torch.float32, device=dev),
                               zeros=torch.zeros(B, 1, Lseq, dtype=torch.float32, device=dev))
        return A

    def forward(self, input_ids, attention_mask, token_type_ids, train, seed=0, p_out=0.0):
        if self._seg is None:
            raise L.AmdsegError("PoNet needs segment_ids (set_segments) before forward")
        B, Lseq = input_ids.shape
        seg = self._seg
        with torch.no_grad():
            pos = torch.arange(Lseq, device=seg.device).expand(B, Lseq)
            diff = seg[:, 1:] != seg[:, :-1]
            one = torch.ones(B, 1, dtype=torch.bool, device=seg.device)
            rs = torch.cummax(torch.where(torch.cat((one, diff), 1), pos, torch.zeros_like(pos)), dim=1).values
            re = torch.flip(torch.cummin(torch.flip(torch.where(torch.cat((diff, one), 1), pos, torch.full_like(pos, Lseq - 1)), (1,)), dim=1).values, (1,))
            valid = (attention_mask == 1).to(torch.float32)
            self._run = (rs.to(torch.int32).reshape(-1).contiguous(), re.to(torch.int32).reshape(-1).contiguous())
            from .engine import MASK_BIAS
            pool_valid = valid
            if not getattr(self.cfg, "ponet_special_tokens_mixing", True):
                # the other reading of the unavailable original (oracle/ponet_oracle.py `pool_valid`): [CLS] (position 0) and [SEP] (the last
                # valid token) enter no local / segment pooling window and get no mixing output; they stay keys of the global aggregation
                pool_valid = valid.clone()
                pool_valid[:, 0] = 0.0
                last = ((valid > 0).long() * torch.arange(1, Lseq + 1, device=valid.device)[None, :]).amax(1) - 1
                rows = torch.arange(B, device=valid.device)[last >= 0]
                pool_valid[rows, last[last >= 0]] = 0.0
            mb = ((1.0 - pool_valid) * MASK_BIAS).reshape(-1).contiguous()
            self._pool_mb = mb.view(B, Lseq)
            self._work = ops.ponet_plan(mb, self._run[0], B, Lseq)           # work lists of the pooling kernels, once per batch
            self._valid = valid.view(B, 1, Lseq).contiguous()
            self._coef_mean = (valid / valid.sum(1, keepdim=True).clamp(min=1.0)).view(B, 1, Lseq).contiguous()
        out, ctx = super().forward(input_ids, attention_mask, token_type_ids, train, seed, p_out)
        ctx["pn"] = dict(run=self._run, valid=self._valid, coef_mean=self._coef_mean, work=self._work, pool_mb=self._pool_mb)
        return out, ctx

    def _views(self, proj, H):
        return [proj[:, k * H:(k + 1) * H] for k in range(5)]           # Hq, Hk, Ho, Hl, Hs column blocks (row stride 5H)

    def _layer_forward(self, lib, cfg, lp, A, i, mb, s, train):
        B, Lseq, H, heads = cfg.B, cfg.L, self.H, self.heads
        cfg.nproj, cfg.mixer = 5, 1
        acts = A["acts_struct"][i]
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].1")
        li = i if train else 0
        la, pn = A["layers"][li], A["pn"]
        hq, hk = self._views(la["qkv"], H)[:2]
        rs, re = self._run
        seed = (int(cfg.seed) * 0x9E3779B1 + 104729 * (i + 1)) & 0x7FFFFFFFFFFFFFFF
        if self.fused_global and la["qkv"].dtype == torch.bfloat16:
            with torch.no_grad():
                g, vecq, scores, lse = ops.ponet_global_fwd(hq, hk, self._coef_mean, A["mask_bias"], B, Lseq, H, heads, cfg.p_attn, seed,
                                                            pn["gscratch"])
                ops.ponet_pool_fwd(la["qkv"], self._pool_mb, rs, re, self._work, g, pn["part"][li], pn["parg"][li], la["ctx"], B, Lseq, H)
            cfg.phase = 2
            L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].2")
            cfg.phase = 0
            return dict(vecq=vecq, scores=scores, lse=lse, g=g, seed=seed) if train else None
        with torch.no_grad():
            qbar = ops.lf_wsum(hq, self._coef_mean, H, pn["lf_partials"])                      # [B, 1, H]
            vecq = (qbar * self.headmask.unsqueeze(0) * 0.125).contiguous()                   # [B, heads, H], head-sliced, / sqrt(d)
            scores = ops.lf_rowvec_dot(hk, vecq, B, Lseq, add_tok=A["mask_bias"])
            p, pd, _sp = ops.lf_softmax_fwd(scores, cfg.p_attn, seed)
            y = ops.lf_wsum(hk, pd, H, pn["lf_partials"])                                      # [B, heads, H]
            g = (y * self.headmask.unsqueeze(0)).sum(1).contiguous()                           # [B, H]
            ops.ponet_pool_fwd(la["qkv"], self._pool_mb, rs, re, self._work, g, pn["part"][li], pn["parg"][li], la["ctx"], B, Lseq, H)
        cfg.phase = 2
        L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].2")
        cfg.phase = 0
        return dict(vecq=vecq, p=p, g=g, seed=seed) if train else None

    def _layer_backward(self, lib, cfg, A, i, mb, dy, other, s, saved):
        B, Lseq, H, heads = cfg.B, cfg.L, self.H, self.heads
        cfg.nproj, cfg.mixer = 5, 1
        args = (C.byref(cfg), C.byref(self.lparams[i]), C.byref(self.lgrads[i]), C.byref(A["acts_struct"][i]), C.byref(A["ws_struct"]),
                mb, dy.data_ptr(), other.data_ptr(), i, s)
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].1")
        la, pn, ws = A["layers"][i], A["pn"], A["ws"]
        proj, dproj = la["qkv"], ws["dqkv"]
        hk = self._views(proj, H)[1]
        dhq, dhk = self._views(dproj, H)[:2]
        rs, re = self._run
        vecq, g = saved["vecq"], saved["g"]
        if "lse" in saved:                                     # csrc/ponet_global.hip
            with torch.no_grad():
                dg = ops.ponet_pool_bwd(proj, self._pool_mb, rs, re, self._work, g, pn["part"][i], pn["parg"][i], ws["dctx"], dproj, pn["psum"],
                                        B, Lseq, H)            # [B, H]: sum of dctx * Ho over the valid tokens
                ops.ponet_global_bwd(hk, self._coef_mean, vecq, saved["scores"], saved["lse"], dg, dhq, dhk, B, Lseq, H, heads, cfg.p_attn,
                                     saved["seed"], pn["gscratch"], pn["dpd"])
            cfg.phase = 2
            L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].2")
            cfg.phase = 0
            return
        p = saved["p"]
        with torch.no_grad():
            dg = ops.ponet_pool_bwd(proj, self._pool_mb, rs, re, self._work, g, pn["part"][i], pn["parg"][i], ws["dctx"], dproj, pn["psum"],
                                    B, Lseq, H).view(B, 1, H)                                  # sum of dctx * Ho over the valid tokens
            dgh = (dg * self.headmask.unsqueeze(0)).contiguous()                              # [B, heads, H]
            dpd = ops.lf_rowvec_dot(hk, dgh, B, Lseq)
            ds, pd = ops.lf_softmax_bwd(p, dpd, cfg.p_attn, saved["seed"])
            ops.lf_dx_update(dhk, pd, dgh, ds, vecq, pn["vt"], assign=True)                    # dHk = pd (x) dg_h + ds (x) qbar_h / sqrt(d)
            t = ops.lf_wsum(hk, ds, H, pn["lf_partials"])
            dqbar = ((t * self.headmask.unsqueeze(0)).sum(1, keepdim=True) * 0.125).contiguous()   # [B, 1, H]
            ops.lf_dx_update(dhq, self._coef_mean, dqbar, pn["zeros"], dqbar, pn["vt"], assign=True)
        cfg.phase = 2
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].2")
        cfg.phase = 0


class _PoNetEncoderFn(torch.autograd.Function):
    """as engine.EncoderFn, plus segment_ids.  Any [B, L] is accepted (the reference model takes every shape): windows that do not
    fill the 64-token blocks / 128-row tiles are padded with masked tokens (their segment id continues the last one; masked tokens
    take part in no pooling branch) and fully masked sequences, and the output is cut back."""

    @staticmethod
    def forward(ctx, trigger, engine, input_ids, attention_mask, token_type_ids, segment_ids, train, seed, p_out, *params):
        from .engine import EncoderFn
        ctx.nparams = len(params)                            # > 0: DDP-compatible mode (engine.ddp_compat)
        B, Lq = input_ids.shape
        Bp, Lp = EncoderFn.aligned_shape(B, Lq)
        ctx.shapes = (B, Lq, Bp, Lp)
        if (Bp, Lp) != (B, Lq):
            pad_id = getattr(engine.cfg, "pad_token_id", None) or 0
            grow = (0, Lp - Lq, 0, Bp - B)
            input_ids = F.pad(input_ids, grow, value=pad_id)
            attention_mask = F.pad(attention_mask, grow, value=0)
            token_type_ids = F.pad(token_type_ids, grow, value=0)
            seg = F.pad(segment_ids, grow, value=0)
            if Lp != Lq:
                seg[:B, Lq:] = segment_ids[:, -1:] + 1
            segment_ids = seg
        engine.set_segments(segment_ids.contiguous())
        out, ectx = engine.forward(input_ids.contiguous(), attention_mask.contiguous(), token_type_ids.contiguous(), train, seed, p_out)
        ctx.engine, ctx.ectx = engine, ectx
        if train:
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
        eng, pn = ctx.engine, ctx.ectx["pn"]
        eng._run, eng._valid, eng._coef_mean, eng._work, eng._pool_mb = pn["run"], pn["valid"], pn["coef_mean"], pn["work"], pn["pool_mb"]
        if ctx.nparams:
            grads = eng.compat_backward(lambda: eng.backward(ctx.ectx, dseq, accumulate=True))
            return (torch.zeros(1, device=dseq.device),) + (None,) * 8 + grads
        eng.backward(ctx.ectx, dseq, accumulate=True)
        return (None,) * 9                              # (the trigger needs no gradient: None = nothing to accumulate, no fill / add kernel per step)


# ------------------------------------------------------------------------------------------------ model
class PoNetForTokenClassification(PreTrainedModel):
    config_class = PoNetConfig
    base_model_prefix = "ponet"
    _keys_to_ignore_on_load_unexpected = [r"pooler"]

    def __init__(self, config):
        super().__init__(config)
        self.num_labels = config.num_labels
        self.ponet = PoNetModel(config, add_pooling_layer=False)
        self.dropout_p = float(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.post_init()
        self._engine = None
        self._step_seed = 0
        self.amdseg_seed = 0

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_(); module.weight.data.fill_(1.0)

    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise L.AmdsegError("spokennlp_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        if self._engine is None or not self._engine.fp.intact() or self._engine.device != p.device:
            self._engine = PoNetEncoderEngine(self, self.config, p.device, bert_attr="ponet")
        return self._engine

    def no_sync(self):
        return self.engine().no_sync()

    # ---- the ModelScope-flavoured surface the reference driver / MyTrainer use (modeling_ponet.py:111-119, trainer.py:33-60,
    #      ponet_topic_segmentation.py:416-418): `from_pretrained(model_name_or_path=..., task=..., revision=...)`, `model.model_dir`,
    #      `save_pretrained(output_dir, state_dict)`.  Hub download is ModelScope's (no network here): a LOCAL directory is required.
    model_dir = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, *args, model_name_or_path=None, task=None, revision=None, **kwargs):
        path = pretrained_model_name_or_path if pretrained_model_name_or_path is not None else model_name_or_path
        if path is None:
            raise L.AmdsegError("from_pretrained needs a local checkpoint directory (model_name_or_path=...)")
        import os
        if not os.path.isdir(path):
            raise L.AmdsegError(f"{path!r} is not a local directory: ModelScope hub ids (e.g. damo/nlp_ponet_fill-mask_chinese-base) must be "
                                f"downloaded first, this build has no hub client")
        wfile = os.path.join(path, "pytorch_model.bin")
        if os.path.isfile(wfile) and not os.path.isfile(os.path.join(path, "model.safetensors")):
            # what `save_pretrained(output_dir, state_dict)` below (== the reference's) writes: a bare torch.save of the state dict
            cfg = kwargs.pop("config", None) or cls.config_class.from_pretrained(path)
            m = cls(cfg)
            sd = torch.load(wfile, map_location="cpu")
            missing, unexpected = m.load_state_dict(sd, strict=False)
            bad = [k for k in unexpected if "pooler" not in k and "position_ids" not in k]
            if bad:
                raise L.AmdsegError(f"unexpected keys in {wfile}: {bad[:5]}")
        else:
            m = super().from_pretrained(path, *args, **kwargs)
        m.model_dir = path
        return m

    def save_pretrained(self, output_dir, state_dict=None, **kwargs):
        import os
        import shutil
        if state_dict is None and kwargs:
            return super().save_pretrained(output_dir, **kwargs)          # plain HF call
        os.makedirs(output_dir, exist_ok=True)
        torch.save(state_dict if state_dict is not None else self.state_dict(), os.path.join(output_dir, "pytorch_model.bin"))
        self.config.to_json_file(os.path.join(output_dir, "config.json"))
        if self.model_dir and os.path.isfile(os.path.join(self.model_dir, "configuration.json")):      # ModelFile.CONFIGURATION
            shutil.copy(os.path.join(self.model_dir, "configuration.json"), os.path.join(output_dir, "configuration.json"))

    def extend_position_embeddings(self, max_pos):
        """ponet_topic_segmentation.py:466-482: tile the trained position table up to `max_pos` rows (512 -> 4096 in the script)"""
        w = self.ponet.embeddings.position_embeddings.weight
        cur, E = w.shape
        if max_pos <= cur:
            return
        new = w.new_empty(max_pos, E)
        k = 0
        while k < max_pos - 1:
            n = min(cur, max_pos - k)
            new[k:k + n] = w.data[:n]
            k += cur
        w.data = new
        self.ponet.embeddings.position_ids = torch.arange(max_pos, device=w.device).reshape(1, max_pos)
        self.config.max_position_embeddings = max_pos

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, segment_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, labels=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        if position_ids is not None or head_mask is not None or inputs_embeds is not None or output_attentions or output_hidden_states:
            raise L.AmdsegError("position_ids / head_mask / inputs_embeds / attention or hidden-state outputs are not supported by the HIP path")
        if segment_ids is None:
            raise L.AmdsegError("PoNet needs segment_ids (ponet_topic_segmentation.py:564-596)")
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        eng = self.engine()
        train = self.training and torch.is_grad_enabled()
        compat = train and eng.ddp_compat()                 # torch DDP around the model: gradients go through autograd (engine.py)
        if train and not compat:
            eng.attach_grads_if_needed()
        self._step_seed += 1
        seed = (int(self.amdseg_seed) * 1000003 + self._step_seed) & 0x7FFFFFFF
        seq = _PoNetEncoderFn.apply(eng._trigger, eng, input_ids.contiguous(), attention_mask.contiguous(), token_type_ids.contiguous(),
                                    segment_ids.contiguous(), train, seed, self.dropout_p, *(eng.compat_params() if compat else ()))
        logits = RowDotFn.apply(seq, self.classifier.weight, self.classifier.bias)
        loss = None
        if labels is not None:
            active = torch.where(attention_mask.view(-1) == 1, labels.view(-1), torch.full_like(labels.view(-1), -100))     # :90-96
            loss = F.cross_entropy(logits.view(-1, self.num_labels), active, ignore_index=-100)
        if not return_dict:
            return ((loss,) + (logits,)) if loss is not None else (logits,)
        return TokenClassifierOutput(loss=loss, logits=logits, hidden_states=None, attentions=None)
