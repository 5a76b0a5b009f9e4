"""Host-side engine of the MI355X Longformer encoder: the BERT engine (flat parameters, bf16 shadows,
arenas, composite layer calls) with
  * RoBERTa-style position ids ([hf] models/longformer/modeling_longformer.py:368-381),
  * band attention inside the composite layer (cfg.window / cfg.nglobal, csrc/attention.hip), and
  * the global [CLS] row ([hf]:964-1058) evaluated between the two phases of the composite layer call with the
    key_global / value_global projections folded onto the query side (csrc/longformer.hip): three HBM-bound passes over
    the layer input instead of two extra M x H x H GEMMs per layer.  The O(heads * H^2) algebra on [B, heads, H]
    vectors below is a handful of tiny rocBLAS calls through torch (plumbing); everything that touches O(L) data is HIP.
The reference wrapper makes [CLS] the only global token (longformer_for_ts.py:55-58); other global masks are rejected.
"""
import ctypes as C
import math

import torch

from . import lib as L
from . import ops
from .engine import BertEncoderEngine

GLOBAL_SUFFIXES = ("query_global", "key_global", "value_global")


class LongformerEncoderEngine(BertEncoderEngine):
    supports_parity = False
    def __init__(self, module, config, device, bert_attr="longformer"):
        super().__init__(module, config, device, bert_attr=bert_attr)
        aw = config.attention_window
        self.windows = [int(w) // 2 for w in (aw if isinstance(aw, (list, tuple)) else [aw] * self.nlayers)]
        if any(w <= 0 for w in self.windows):
            raise L.AmdsegError("attention_window must be >= 2")
        self.pad_id = int(config.pad_token_id)
        self.scale = 1.0 / math.sqrt(64.0)
        # False: no global token at all -- LongformerModel called with global_attention_mask=None (the mmvts text encoder,
        # mmvts/src/models/text_encoder/text_encoder.py:73-85 as driven by multi_modal_for_ts.py:173-176): pure band attention
        self.cls_global = True

    # ---- parameters of the global projections (fp32 masters; tiny algebra runs in fp32)
    def _gp(self, flat, i, which, kind):
        return self.fp.view(flat, f"{self.fp.encoder_prefix}{i}.attention.self.{which}.{kind}")

    def _position_ids(self, input_ids):
        mask = (input_ids != self.pad_id).to(torch.int64)
        return (torch.cumsum(mask, dim=1) * mask + self.pad_id).reshape(-1).contiguous()

    def _embed_backward_fixup(self, dpos, pad):
        dpos[self.pad_id].zero_()          # nn.Embedding(padding_idx=pad) of the position table ([hf]:394-396)

    # ---- forward
    def _layer_forward(self, lib, cfg, lp, A, i, mb, s, train):
        B, Lseq, H, heads = cfg.B, cfg.L, self.H, self.heads
        if not self.cls_global:
            cfg.window, cfg.nglobal, cfg.phase = self.windows[i], 0, 0
            return super()._layer_forward(lib, cfg, lp, A, i, mb, s, train)
        cfg.window, cfg.nglobal = self.windows[i], 1
        acts = A["acts_struct"][i]
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].1")
        la = A["layers"][i if train else 0]
        x_in = A["x"][i] if train else A["x"][i % 2]
        fp = self.fp.flat_p
        with torch.no_grad():
            Wq, bq = self._gp(fp, i, "query_global", "weight"), self._gp(fp, i, "query_global", "bias")
            Wk = self._gp(fp, i, "key_global", "weight").view(heads, 64, H)
            Wv, bv = self._gp(fp, i, "value_global", "weight").view(heads, 64, H), self._gp(fp, i, "value_global", "bias").view(heads, 64)
            x0 = x_in.view(B, Lseq, H)[:, 0, :].float()
            qg = torch.addmm(bq, x0, Wq.t()).mul_(self.scale).view(B, heads, 64)
            r = torch.einsum("bhe,hek->bhk", qg, Wk).contiguous()
            scores = ops.lf_rowvec_dot(x_in, r, B, Lseq, add_tok=A["mask_bias"])
            seed = (int(cfg.seed) * 0x9E3779B1 + 7919 * (i + 1)) & 0x7FFFFFFFFFFFFFFF
            p, pd, sp = ops.lf_softmax_fwd(scores, cfg.p_attn, seed)
            y = ops.lf_wsum(x_in, pd, H, A["lf_partials"])
            out = torch.einsum("bhk,hek->bhe", y, Wv) + bv.unsqueeze(0) * sp.unsqueeze(-1)
            la["ctx"].view(B, Lseq, H)[:, 0, :] = out.reshape(B, H).to(la["ctx"].dtype)
        cfg.phase = 2
        L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].2")
        cfg.phase = 0
        return dict(qg=qg, r=r, p=p, y=y, sp=sp, seed=seed) if train else None

    # ---- backward
    def _layer_backward(self, lib, cfg, A, i, mb, dy, other, s, saved):
        B, Lseq, H, heads = cfg.B, cfg.L, self.H, self.heads
        if not self.cls_global:
            cfg.window, cfg.nglobal, cfg.phase = self.windows[i], 0, 0
            return super()._layer_backward(lib, cfg, A, i, mb, dy, other, s, saved)
        cfg.window, cfg.nglobal = self.windows[i], 1
        args = (C.byref(cfg), C.byref(self.lparams[i]), C.byref(self.lgrads[i]), C.byref(A["acts_struct"][i]), C.byref(A["ws_struct"]),
                mb, dy.data_ptr(), other.data_ptr(), i, s)
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].1")
        x_in = A["x"][i]
        dctx = A["ws"]["dctx"].view(B, Lseq, H)
        fp, fg = self.fp.flat_p, self.fp.flat_g
        qg, r, p, y, sp = saved["qg"], saved["r"], saved["p"], saved["y"], saved["sp"]
        with torch.no_grad():
            Wq = self._gp(fp, i, "query_global", "weight")
            Wk = self._gp(fp, i, "key_global", "weight").view(heads, 64, H)
            Wv, bv = self._gp(fp, i, "value_global", "weight").view(heads, 64, H), self._gp(fp, i, "value_global", "bias").view(heads, 64)
            dout = dctx[:, 0, :].float().view(B, heads, 64)
            dctx[:, 0, :].zero_()                   # the band attention's own row 0 was overwritten in forward: no gradient
            self._gp(fg, i, "value_global", "weight").view(heads, 64, H).add_(torch.einsum("bhe,bhk->hek", dout, y))
            self._gp(fg, i, "value_global", "bias").view(heads, 64).add_((dout * sp.unsqueeze(-1)).sum(0))
            dyv = torch.einsum("bhe,hek->bhk", dout, Wv).contiguous()
            dsp = (dout * bv.unsqueeze(0)).sum(-1).contiguous()
            dpd = ops.lf_rowvec_dot(x_in, dyv, B, Lseq, add_bh=dsp)
            ds, pd = ops.lf_softmax_bwd(p, dpd, cfg.p_attn, saved["seed"])
            dr = ops.lf_wsum(x_in, ds, H, A["lf_partials"])
            self._gp(fg, i, "key_global", "weight").view(heads, 64, H).add_(torch.einsum("bhe,bhk->hek", qg, dr))
            dqg = torch.einsum("bhk,hek->bhe", dr, Wk).reshape(B, H).mul_(self.scale)
            x0 = x_in.view(B, Lseq, H)[:, 0, :].float()
            self._gp(fg, i, "query_global", "weight").addmm_(dqg.t(), x0)
            self._gp(fg, i, "query_global", "bias").add_(dqg.sum(0))
            dx0 = dqg @ Wq
        cfg.phase = 2
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].2")
        cfg.phase = 0
        with torch.no_grad():
            ops.lf_dx_update(other, pd, dyv, ds, r, A["lf_vt"])
            row0 = other.view(B, Lseq, H)[:, 0, :]
            row0.copy_((row0.float() + dx0).to(other.dtype))

    def _arena(self, B, Lseq, train, fp32=False):
        A = super()._arena(B, Lseq, train, fp32)
        if "lf_partials" not in A:
            A["lf_partials"] = torch.empty(B * (Lseq // 64) * self.heads * self.H, dtype=torch.float32, device=self.device)
            A["lf_vt"] = torch.empty(B * self.H * 32, dtype=torch.bfloat16, device=self.device)
        return A
