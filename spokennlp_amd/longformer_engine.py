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
import os

import torch

from . import lib as L
from . import ops
from .engine import BertEncoderEngine

GLOBAL_SUFFIXES = ("query_global", "key_global", "value_global")


class LongformerEncoderEngine(BertEncoderEngine):
    # "parity" precision (fp32 activations, split-bf16 contractions): the projections as in the BERT engine, the band attention on the
    # split-bf16 kernels of csrc/attention_split.hip, the global row in fp32 (its O(L) passes take fp32 rows)
    supports_parity = True
    parity_needs_split_attn = True
    def __init__(self, module, config, device, bert_attr="longformer"):
        super().__init__(module, config, device, bert_attr=bert_attr)
        aw = config.attention_window
        self.windows = [int(w) // 2 for w in (aw if isinstance(aw, (list, tuple)) else [aw] * self.nlayers)]
        if any(w <= 0 for w in self.windows):
            raise L.AmdsegError("attention_window must be >= 2")
        self.pad_id = int(config.pad_token_id)
        # band attention keeps the stateless dropout hash by default.  The keep-mask form exists (amdseg_attn_keepmask_band + the BAND / KM kernel
        # instantiations, engine.attn_keepmask = True before the first forward) and is slower end to end at longformer-base 8 x 4096: the three kernels gain 28 us per layer (109 / 146 /
        # 179 -> 102 / 144 / 165 us) and the generator costs more (288 vs 295 seq/s)
        self.attn_keepmask = False
        self.scale = 1.0 / math.sqrt(64.0)
        # False: no global token at all -- LongformerModel called with global_attention_mask=None (the mmvts text encoder,
        # mmvts/src/models/text_encoder/text_encoder.py:73-85 as driven by multi_modal_for_ts.py:173-176): pure band attention
        self.cls_global = True
        # the global-row chain (a dozen small, latency-bound launches per layer and direction plus three HBM passes) runs on a second
        # stream under the layer's big kernels: forward under the QKV GEMM + band attention, backward under the attention backward
        # and the weight-gradient GEMM.  engine.lf_overlap = False keeps everything on one stream.
        self.lf_overlap = device.type == "cuda"
        self._side_pending = False                          # weight gradients of the global projections still running on the second stream
        self._lf_side = torch.cuda.Stream(device=device, priority=0) if self.lf_overlap else None

    def _on_both(self, main, *tensors):
        """tensors created while one stream was current and read on the other: tell the caching allocator"""
        if self._lf_side is None:
            return
        for t in tensors:
            if t is not None:
                t.record_stream(main); t.record_stream(self._lf_side)

    # ---- parameters of the global projections (fp32 masters; tiny algebra runs in fp32)
    def _gp(self, flat, i, which, kind):
        return self.fp.view(flat, f"{self.fp.encoder_prefix}{i}.attention.self.{which}.{kind}")

    def _position_ids(self, input_ids):
        mask = (input_ids != self.pad_id).to(torch.int64)
        return (torch.cumsum(mask, dim=1) * mask + self.pad_id).reshape(-1).contiguous()

    def _embed_backward_fixup(self, dpos, pad):
        dpos[self.pad_id].zero_()          # nn.Embedding(padding_idx=pad) of the position table ([hf]:394-396)
        if self._side_pending:                         # end of backward: the global projections' weight gradients (side stream) are part of flat_g
            torch.cuda.current_stream().wait_stream(self._lf_side)
            self._side_pending = False

    # ---- forward
    def _layer_forward(self, lib, cfg, lp, A, i, mb, s, train):
        B, Lseq, H, heads = cfg.B, cfg.L, self.H, self.heads
        if not self.cls_global:
            cfg.window, cfg.nglobal, cfg.phase = self.windows[i], 0, 0
            return super()._layer_forward(lib, cfg, lp, A, i, mb, s, train)
        cfg.window, cfg.nglobal = self.windows[i], 1
        acts = A["acts_struct"][i]
        la = A["layers"][i if train else 0]
        x_in = A["x"][i] if train else A["x"][i % 2]
        fp = self.fp.flat_p
        dev = x_in.device
        dtx = L.F32 if x_in.dtype == torch.float32 else L.BF16
        main = torch.cuda.current_stream()
        side = self._lf_side if self.lf_overlap else main
        # bf16 path: forward phase 1 leaves the [CLS] rows of ctx unwritten (include/amdseg.h, amdseg_bert_cfg.phase), so the WHOLE global-row
        # chain -- it reads only x_in -- runs beside the projection + band attention and main waits once, for work that is long done.  (Waiting
        # for the attention, writing the row, and handing back cost two cross-queue hops = ~40 us per layer with the GPU idle.)
        early = side is not main and x_in.dtype == torch.bfloat16
        if side is not main:
            fork = torch.cuda.Event(); fork.record(main)                                  # x_in is complete
        if early:                                      # queue main's work first: the host needs ~8 launches for the chain
            cfg.phase = 1
            L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].1")
        if side is not main:
            side.wait_event(fork)
        with torch.no_grad(), torch.cuda.stream(side):
            ss = side.cuda_stream
            Wq, bq = self._gp(fp, i, "query_global", "weight"), self._gp(fp, i, "query_global", "bias")
            Wk = self._gp(fp, i, "key_global", "weight")
            Wv, bv = self._gp(fp, i, "value_global", "weight"), self._gp(fp, i, "value_global", "bias")
            qg = torch.empty(B, heads, 64, dtype=torch.float32, device=dev)
            r = torch.empty(B, heads, H, dtype=torch.float32, device=dev)
            # the O(heads * H^2) algebra of the global row runs in csrc/lf_global.hip (2 launches forward, 4 backward; round 1: ~25
            # rocBLAS / elementwise launches per layer and direction)
            L.check(lib.amdseg_lf_global_q(x_in.data_ptr(), dtx, Wq.data_ptr(), bq.data_ptr(), Wk.data_ptr(), qg.data_ptr(), r.data_ptr(),
                                           B, Lseq, H, heads, self.scale, ss), "amdseg_lf_global_q")
            scores = ops.lf_rowvec_dot(x_in, r, B, Lseq, add_tok=A["mask_bias"])
            seed = (int(cfg.seed) * 0x9E3779B1 + 7919 * (i + 1)) & 0x7FFFFFFFFFFFFFFF
            p, pd, sp = ops.lf_softmax_fwd(scores, cfg.p_attn, seed)
            y = ops.lf_wsum(x_in, pd, H, A["lf_partials"])
        if not early:
            # fp32 dtypes: the band attention writes its own row 0 of ctx; the global row then overwrites it
            cfg.phase = 1
            L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].1")
            if side is not main:
                attn_done = torch.cuda.Event(); attn_done.record(main); side.wait_event(attn_done)
        with torch.no_grad(), torch.cuda.stream(side):
            L.check(lib.amdseg_lf_global_out(Wv.data_ptr(), bv.data_ptr(), y.data_ptr(), sp.data_ptr(), la["ctx"].data_ptr(),
                                             L.F32 if la["ctx"].dtype == torch.float32 else L.BF16, B, Lseq, H, heads, side.cuda_stream),
                    "amdseg_lf_global_out")
        if side is not main:
            done = torch.cuda.Event(); done.record(side); main.wait_event(done)
            self._on_both(main, qg, r, scores, p, pd, sp, y)
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
        lp = self.lparams_parity[i] if cfg.dtype == L.F32S else self.lparams[i]
        args = (C.byref(cfg), C.byref(lp), C.byref(self.lgrads[i]), C.byref(A["acts_struct"][i]), C.byref(A["ws_struct"]),
                mb, dy.data_ptr(), other.data_ptr(), i, s)
        adt = L.F32 if cfg.dtype == L.F32S else L.BF16          # dtype of activations and activation gradients
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].1")
        x_in = A["x"][i]
        dctx = A["ws"]["dctx"]
        fp, fg = self.fp.flat_p, self.fp.flat_g
        qg, r, p, y, sp = saved["qg"], saved["r"], saved["p"], saved["y"], saved["sp"]
        dev = x_in.device
        f32 = dict(dtype=torch.float32, device=dev)
        main = torch.cuda.current_stream()
        side = self._lf_side if self.lf_overlap else main
        fast = side is not main and adt == L.BF16
        with torch.no_grad():
            Wq, Wk = self._gp(fp, i, "query_global", "weight"), self._gp(fp, i, "key_global", "weight")
            Wv, bv = self._gp(fp, i, "value_global", "weight"), self._gp(fp, i, "value_global", "bias")
            if not fast:
                dout, dyv, dsp = torch.empty(B, heads, 64, **f32), torch.empty(B, heads, H, **f32), torch.empty(B, heads, **f32)
                # consumes + zeroes dctx[:, 0] (the band attention's own row 0 was overwritten in forward: no gradient); on the main stream:
                # the attention backward below reads dctx
                L.check(lib.amdseg_lf_global_bwd_a(dctx.data_ptr(), adt, Wv.data_ptr(), bv.data_ptr(), dout.data_ptr(), dyv.data_ptr(),
                                                   dsp.data_ptr(), B, Lseq, H, heads, s), "amdseg_lf_global_bwd_a")
        if fast:
            # Two-stream order of the bf16 path.  Everything the global row adds to dx -- the rank-2*heads update and the [CLS] row's own
            # term -- depends only on dctx[:, 0] and saved tensors, so the side stream prepares it under the band attention backward + dx GEMM
            # (operand image vt, row vector trow) and main applies it in one read-modify-write pass right behind the dx GEMM; the weight
            # gradients of the three global projections follow on the side stream and are joined once, at the end of backward
            # (_embed_backward_fixup).  (The update + the row term + the weight gradients used to run under the weight-gradient GEMM, which
            # slowed them 3-5 x and left main waiting ~45 us per layer for the hand-back.)  Backward phase 6 takes the [CLS] rows of dctx as zero
            # itself (include/amdseg.h), so even the kernel that consumes them (amdseg_lf_global_bwd_a_ro) runs on the side stream.
            e1 = torch.cuda.Event(); e1.record(main)
            cfg.phase = 6                                      # attention backward + dx GEMM, queued before the host issues the chain
            L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].2")
            side.wait_event(e1)
            with torch.no_grad(), torch.cuda.stream(side):
                ss = side.cuda_stream
                dout, dyv, dsp = torch.empty(B, heads, 64, **f32), torch.empty(B, heads, H, **f32), torch.empty(B, heads, **f32)
                L.check(lib.amdseg_lf_global_bwd_a_ro(dctx.data_ptr(), adt, Wv.data_ptr(), bv.data_ptr(), dout.data_ptr(), dyv.data_ptr(),
                                                      dsp.data_ptr(), B, Lseq, H, heads, ss), "amdseg_lf_global_bwd_a_ro")
                dpd = ops.lf_rowvec_dot(x_in, dyv, B, Lseq, add_bh=dsp)
                ds, pd = ops.lf_softmax_bwd(p, dpd, cfg.p_attn, saved["seed"])
                dr = ops.lf_wsum(x_in, ds, H, A["lf_partials"])
                dqg, trow = torch.empty(B, H, **f32), torch.empty(B, H, **f32)
                L.check(lib.amdseg_lf_global_bwd_dx(Wq.data_ptr(), Wk.data_ptr(), dr.data_ptr(), dqg.data_ptr(), trow.data_ptr(), B, Lseq, H, heads,
                                                    self.scale, ss), "amdseg_lf_global_bwd_dx")
                L.check(lib.amdseg_lf_dx_prep(dyv.data_ptr(), r.data_ptr(), A["lf_vt"].data_ptr(), B, Lseq, H, heads, ss), "amdseg_lf_dx_prep")
                e2 = torch.cuda.Event(); e2.record(side)
                g = lambda which, kind: self._gp(fg, i, which, kind).data_ptr()          # noqa: E731
                L.check(lib.amdseg_lf_global_bwd_w(x_in.data_ptr(), adt, qg.data_ptr(), dout.data_ptr(), y.data_ptr(), sp.data_ptr(), dr.data_ptr(),
                                                   dqg.data_ptr(), g("query_global", "weight"), g("query_global", "bias"),
                                                   g("key_global", "weight"), g("value_global", "weight"), g("value_global", "bias"),
                                                   B, Lseq, H, heads, ss), "amdseg_lf_global_bwd_w")
                ew = torch.cuda.Event(); ew.record(side)
            main.wait_event(e2)
            L.check(lib.amdseg_lf_dx_apply(other.data_ptr(), H, pd.data_ptr(), ds.data_ptr(), A["lf_vt"].data_ptr(), trow.data_ptr(), B, Lseq, H,
                                           heads, s), "amdseg_lf_dx_apply")
            cfg.phase = 4
            L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].4")
            self._on_both(main, dout, dyv, dsp, dpd, ds, pd, dr, dqg, trow)
            self._side_pending = True
            if self.buckets is not None and self.grad_sync:   # data parallel: this layer's slice is reduced right after this call
                main.wait_event(ew)
            cfg.phase = 0
            return
        if side is not main:
            e1 = torch.cuda.Event(); e1.record(main); side.wait_event(e1)
        with torch.no_grad(), torch.cuda.stream(side):          # ... under the band attention backward
            dpd = ops.lf_rowvec_dot(x_in, dyv, B, Lseq, add_bh=dsp)
            ds, pd = ops.lf_softmax_bwd(p, dpd, cfg.p_attn, saved["seed"])
            dr = ops.lf_wsum(x_in, ds, H, A["lf_partials"])
        cfg.phase = 6 if side is not main else 2               # attention backward + dx GEMM (+ the weight gradients when single-stream)
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].2")
        if side is not main:
            e3 = torch.cuda.Event(); e3.record(main); side.wait_event(e3)      # `other` (dx of the layer) is complete
        with torch.no_grad(), torch.cuda.stream(side):         # ... under the grouped weight-gradient GEMM
            ops.lf_dx_update(other, pd, dyv, ds, r, A["lf_vt"])
            dqg = torch.empty(B, H, **f32)
            g = lambda which, kind: self._gp(fg, i, which, kind).data_ptr()          # noqa: E731
            L.check(lib.amdseg_lf_global_bwd_rest(x_in.data_ptr(), adt, other.data_ptr(), adt, Wq.data_ptr(), Wk.data_ptr(),
                                                  qg.data_ptr(), dout.data_ptr(), y.data_ptr(), sp.data_ptr(), dr.data_ptr(), dqg.data_ptr(),
                                                  g("query_global", "weight"), g("query_global", "bias"), g("key_global", "weight"),
                                                  g("value_global", "weight"), g("value_global", "bias"), B, Lseq, H, heads, self.scale,
                                                  side.cuda_stream),
                    "amdseg_lf_global_bwd_rest")
        if side is not main:
            e4 = torch.cuda.Event(); e4.record(side)
            cfg.phase = 4
            L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].4")
            main.wait_event(e4)
            self._on_both(main, dout, dyv, dsp, dpd, ds, pd, dr, dqg)
        cfg.phase = 0

    def _arena(self, B, Lseq, train, fp32=False):
        A = super()._arena(B, Lseq, train, fp32)
        if "lf_partials" not in A:
            A["lf_partials"] = torch.empty(B * (Lseq // 64) * self.heads * self.H, dtype=torch.float32, device=self.device)
            A["lf_vt"] = torch.empty(B * self.H * 32, dtype=torch.bfloat16, device=self.device)
        return A
