"""MI355X-native drop-in for the reference wrapper model
    emnlp2023-topic_segmentation/src/models/bert_for_ts.py:19-113  BertWithDAForSentenceLabelingTopicSegmentation
and its heads (modules/loss_calculator.py, modules/cssl.py, modules/tssp.py, modules/utils.py).

Same HuggingFace plug-in surface: a `BertPreTrainedModel` subclass with the reference's class name, parameter names
(`bert.*`, `loss_calculator.classifier.*`, `loss_calculator.tssp.classifier.*`), `forward(**batch)` argument names and
`(loss, logits (B,2,L,2), cos_sim (B,k))` return tuple, so `from_pretrained` / `save_pretrained` / `transformers.Trainer`
work unchanged (ts_sentence_seq_labeling.py:247-257,1077-1094).  Differences, all deliberate:
  * the encoder never runs torch ops: `self.bert` is only the parameter container; forward/backward go through
    libamdseg (engine.py).  The anchor and augmented passes are batched into ONE encoder pass of 2B sequences.
  * the reference's CPU-only in-place-on-leaf bug (`loss += ...`, loss_calculator.py:35,51) is not reproduced.
  * the host loops of EopPairCosineSimilarity / CSSL run on a host copy of `labels` fetched once, before the encoder
    kernels are queued, instead of B boolean-mask syncs per step.
There is no CPU fallback: a missing libamdseg.so or a CPU tensor raises.
"""
import random

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers.models.bert.modeling_bert import BertModel, BertPreTrainedModel

from . import lib as L
from .engine import BertEncoderEngine, EncoderFn, FusedHeadsFn, HeadInputsFn, RowDotFn  # noqa: F401

HEAD_DEFAULTS = dict(do_da_ts=False, do_cssl=False, do_tssp=False, ts_loss_weight=1.0, ts_score_predictor="lt",
                     ts_score_predictor_cos_temp=1, focal_loss_gamma=0.0, weight_label_zero=0.5, cl_loss_weight=0.0,
                     cl_temp=1, cl_anchor_level="eop_matrix", cl_positive_k=1, cl_negative_k=1, tssp_loss_weight=0.0,
                     tssp_ablation="none", num_tssp_labels=3)


class _TSSP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.classifier = nn.Linear(config.hidden_size, config.num_tssp_labels)


class _LossCalculator(nn.Module):
    """parameter container with the reference's names (loss_calculator.py:16-19)."""

    def __init__(self, config):
        super().__init__()
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.tssp = _TSSP(config)


class _RowRequests:
    """row lists (indices into the flattened [2B*L, H] encoder output) requested by the heads of one step; all of them are
    gathered by ONE kernel into one [n, H] matrix and each head works on its slice"""

    def __init__(self):
        self.rows = []

    def add(self, rows):
        s = len(self.rows)
        self.rows.extend(rows)
        return (s, len(rows))


class _IndexUploader:
    """collects the step's host-built index lists and uploads them with one pinned, asynchronous H2D copy."""

    def __init__(self, device):
        self.device, self.flat, self.dev = device, [], None

    def add(self, lst):
        off = len(self.flat)
        self.flat.extend(lst)
        return (off, len(lst))

    def flush(self):
        n = max(len(self.flat), 1)
        if self.device.type == "cuda":
            buf = torch.empty(n, dtype=torch.long, pin_memory=True)
            if self.flat:
                buf.copy_(torch.tensor(self.flat, dtype=torch.long))
            self.dev = buf.to(self.device, non_blocking=True)
        else:
            self.dev = torch.tensor(self.flat or [0], dtype=torch.long)

    def get(self, h):
        return self.dev[h[0]:h[0] + h[1]]


def _topic_segment_ids(label_rows):
    ids, seg = [], 0
    for ex in label_rows:
        if len(ex) == 0:
            continue
        for l in ex:
            ids.append(seg)
            if l == 0:
                seg += 1
        if ex[-1] == 1:
            seg += 1
    return ids


def _cos(x, y, temp):
    if temp == 0:
        return x.squeeze(1) @ y.squeeze(0).t()
    return F.cosine_similarity(x, y, dim=-1) / temp


class TopicSegHeadsMixin:
    """everything of the reference wrappers that is encoder-agnostic (bert_for_ts.py:35-113 == longformer_for_ts.py:34-129
    == electra_for_ts.py / bigbird_for_ts.py): the anchor/augmented split, LossCalculator, CSSL, TSSP, cos-sim output.
    The concrete class provides `engine()` (the HIP encoder engine over its HF parameter container)."""

    def amdseg_set_host_twins(self, pairs):
        """pairs: name -> (device tensor, the host tensor it was copied from).  A training loop that uploads the batch itself (spokennlp_amd.
        trainer.Trainer) leaves them here; forward() then reads label-like tensors from the host original instead of copying them back.
        Matched by identity of the device tensor, replaced every step."""
        import weakref
        self._amdseg_host_twins = {id(d): (weakref.ref(d), h) for d, h in pairs.values()}

    def _host_twin(self, t):
        tw = getattr(self, "_amdseg_host_twins", None)
        if not tw or t is None:
            return None
        e = tw.get(id(t))
        return e[1] if e is not None and e[0]() is t else None

    def _init_heads(self, config, classifier_dropout):
        self.classifier_dropout_p = float(classifier_dropout)
        self.loss_calculator = _LossCalculator(config)
        self._engine = None
        self._step_seed = 0
        self.amdseg_seed = 0

    @staticmethod
    def _fill_head_defaults(config):
        for k, v in HEAD_DEFAULTS.items():
            if not hasattr(config, k):
                setattr(config, k, v)

    def _next_seed(self):
        self._step_seed += 1
        return (int(self.amdseg_seed) * 1000003 + self._step_seed) & 0x7FFFFFFF

    def encode(self, input_ids, attention_mask, token_type_ids):
        """[N, L] int64 -> fp32 [N, L, H] sequence output (after the wrapper's dropout in training)."""
        eng = self.engine()
        train = self.training and torch.is_grad_enabled()
        return eng.encode(input_ids, attention_mask, token_type_ids, train, self._next_seed(), self.classifier_dropout_p)

    def no_sync(self):
        """gradient-accumulation context of the engine's own data parallelism (what `accelerate`/`Trainer` look up on a model that is
        not wrapped in torch DDP); see BertEncoderEngine.no_sync"""
        return self.engine().no_sync()

    # ------------------------------------------------------------------------------------------------ heads
    def _ts_loss(self, logits, labels):
        cfg = self.config
        weight = None
        if cfg.weight_label_zero != 0.5:
            weight = torch.tensor([cfg.weight_label_zero, 1 - cfg.weight_label_zero], dtype=torch.float32, device=logits.device)
        if cfg.focal_loss_gamma != 0:
            # the reference's FocalLoss ends up with reduction='mean' inside super().forward (modules/utils.py:145-168):
            # scalar mean CE times the per-row focal factor, averaged over ALL rows (ignored rows use target 0)
            # No labelled row at all: the reference sees a NaN mean and returns a constant 0 (:150-156) -- here the mean is written as
            # sum / max(denominator, tiny), which gives that 0 (and zero gradients) without reading anything back to the host.
            valid = labels != -100
            tgt = labels * valid.long()
            den = (weight[tgt] * valid).sum() if weight is not None else valid.sum().to(logits.dtype)
            ce = F.cross_entropy(logits, labels, weight=weight, ignore_index=-100, reduction="sum") / den.clamp(min=1e-30)
            pt = torch.gather(F.softmax(logits, 1), 1, tgt.unsqueeze(1))
            return torch.mean(torch.pow(1 - pt, cfg.focal_loss_gamma) * ce)
        return F.cross_entropy(logits, labels, weight=weight, ignore_index=-100)

    @staticmethod
    def _labelled_rows(labels_cpu):
        """host-side index lists: per example the positions with label != -100 and their labels."""
        pos, lab = [], []
        for row in labels_cpu:
            idx = (row != -100).nonzero(as_tuple=False).flatten().tolist()
            pos.append(idx); lab.append(row[idx].tolist())
        return pos, lab

    # ---- host planning: every index list of the step is built on the host from the (already fetched) label tensors,
    #      packed into ONE pinned buffer and uploaded with ONE async copy; the device math below never synchronises.
    def _plan_cos(self, up, req, pos, Lq, off):
        mx = max((len(p) for p in pos), default=0)
        rows_a, rows_b, dst = [], [], []
        for b, p in enumerate(pos):
            n = len(p)
            if n == 0:
                continue
            a = [off + q + b * Lq for q in p]
            rows_a += a
            rows_b += a[1:] + a[:1]
            dst += [b * mx + j for j in range(n)]
        return dict(mx=mx, n=len(rows_a), a=req.add(rows_a), b=req.add(rows_b), dst=up.add(dst))

    def _plan_cssl(self, up, req, pos, lab, Lq, off):
        """index lists of cssl.py:118-228 (same Python `random` call sequence as the reference)."""
        cfg = self.config
        rows = [off + q + b * Lq for b, p in enumerate(pos) for q in p]
        seg = _topic_segment_ids(lab)
        if not (len(seg) > 2 and seg[-1] > 0):
            return None
        n = len(seg)
        total_topic = seg[-1] + 1
        bot = [seg.index(i) for i in range(total_topic)]
        eot = [v - 1 for v in bot[1:]] + [n - 1]
        plan = dict(level=cfg.cl_anchor_level, n=n, rows=req.add(rows))
        if cfg.cl_anchor_level == "eop_matrix":
            plan["seg"] = up.add(seg)
            return plan
        pk, nk = cfg.cl_positive_k, cfg.cl_negative_k
        pos_i = [[] for _ in range(pk)]
        neg_i = [[] for _ in range(nk)]
        if cfg.cl_anchor_level == "eop_list":
            for idx, t in enumerate(seg):
                s_, e_ = bot[t], eot[t]
                choice = list(range(s_, e_)) or [e_]
                pid = idx
                for i in range(pk):
                    pid -= 1
                    if pid < s_:
                        pid = random.choice(choice)
                    pos_i[i].append(pid)
                choice = list(range(e_ + 1, eot[-1] + 1)) or list(range(bot[0], bot[1]))
                pid = e_
                for i in range(nk):
                    pid += 1
                    if pid >= n:
                        pid = random.choice(choice)
                    neg_i[i].append(pid)
            plan["anchors"] = None
        elif cfg.cl_anchor_level == "eot_list":
            for s_, e_ in zip(bot, eot):
                choice = list(range(s_, e_)) or [e_]
                pid = e_
                for i in range(pk):
                    pid -= 1
                    if pid < s_:
                        pid = random.choice(choice)
                    pos_i[i].append(pid)
            for e_ in eot:
                choice = list(range(e_ + 1, eot[-1] + 1)) or list(range(bot[0], bot[1]))
                pid = e_
                for i in range(nk):
                    pid += 1
                    if pid >= n:
                        pid = random.choice(choice)
                    neg_i[i].append(pid)
            plan["anchors"] = up.add(eot)
        else:
            raise ValueError("not supported cl_anchor_level %s " % cfg.cl_anchor_level)
        lists = pos_i + neg_i                                  # every list has one entry per anchor
        plan["nlists"], plan["lists"] = len(lists), up.add([v for ix in lists for v in ix])
        return plan

    def _plan_tssp(self, up, req, stm_cpu, spo_cpu, Lq, off):
        rows, labs = [], []
        for b in range(stm_cpu.shape[0]):
            idx = (stm_cpu[b] != -100).nonzero(as_tuple=False).flatten().tolist()
            rows += [off + q + b * Lq for q in idx]
            row = spo_cpu[b]
            labs += row[row != -100].tolist()
        return dict(rows=req.add(rows), labs=up.add(labs), n=len(rows))

    # ---- device math (no host synchronisation).  `feats` is the ONE gathered [n, H] matrix of the step (HeadInputsFn);
    #      a plan entry (start, n) is a row slice of it
    @staticmethod
    def _rows(feats, h):
        return feats[h[0]:h[0] + h[1]]

    def _cos_sim(self, feats, up, plan, temp, B):
        """utils.py:111-138: cos(row_i, row_{(i+1)%n}) / temp, padded with -100 to the batch max."""
        out = torch.full((B * max(plan["mx"], 1),), -100.0, dtype=feats.dtype, device=feats.device)
        if plan["n"]:
            xa, xb = self._rows(feats, plan["a"]), self._rows(feats, plan["b"])
            cs = (xa * xb).sum(-1) if temp == 0 else F.cosine_similarity(xa, xb, dim=-1) / temp
            out = out.index_put((up.get(plan["dst"]),), cs)
        return out.view(B, max(plan["mx"], 1))[:, :plan["mx"]] if plan["mx"] else out.view(B, 1)[:, :0]

    def _cssl(self, feats_all, up, plan):
        """cssl.py:230-274 with the degenerate amax pooling replaced by the equivalent row gather."""
        cfg = self.config
        if plan is None:
            return feats_all.new_zeros(())
        feats = self._rows(feats_all, plan["rows"])
        n = plan["n"]
        if plan["level"] == "eop_matrix":
            seg_t = up.get(plan["seg"])
            same = seg_t[:, None] == seg_t[None, :]
            num_mask = same & ~torch.eye(n, dtype=torch.bool, device=feats.device)
            # cos(x_i, x_j) / temp as ONE [n, H] x [H, n] product of the normalised rows: the broadcast form of nn.CosineSimilarity the reference
            # uses (cssl.py:56) materialises an [n, n, H] tensor -- 1.2 GB at 640 labelled rows -- for the same numbers
            if cfg.cl_temp == 0:
                sim = feats @ feats.t()
            else:
                xn = F.normalize(feats, dim=-1, eps=1e-8)
                sim = (xn @ xn.t()) / cfg.cl_temp
            e = torch.exp(sim)
            num = (num_mask * e).sum(0)
            den = num + ((~same) * e).sum(0)
            prob = num / den
            sel = prob != 0          # reference: mean of -log(prob) over prob != 0 (cssl.py:64-71), done without a host sync
            cnt = sel.sum().clamp(min=1)
            return (torch.where(sel, -torch.log(torch.where(sel, prob, torch.ones_like(prob))), torch.zeros_like(prob)).sum() / cnt)
        pk = cfg.cl_positive_k
        anchors = feats if plan["anchors"] is None else feats.index_select(0, up.get(plan["anchors"]))
        # all positive and negative lists in ONE gather + ONE cosine ([k, n, H] against [1, n, H]): the per-list form cost
        # ~8 tiny kernels per list forward and twice that backward, a visible slice of a 16 ms step
        other = feats.index_select(0, up.get(plan["lists"])).view(plan["nlists"], -1, feats.shape[-1])
        if cfg.cl_temp == 0:                                     # the matrix form of _cos: one list at a time, as before
            e = torch.exp(torch.cat([_cos(anchors, other[i], 0).unsqueeze(0) for i in range(plan["nlists"])]))
        else:
            e = torch.exp(_cos(anchors.unsqueeze(0), other, cfg.cl_temp))
        return (-torch.log(e[:pk].sum(0) / e.sum(0))).mean()

    def _tssp(self, feats_all, up, plan):
        """tssp.py:16-36 (returns w * CE; the caller multiplies by w again, loss_calculator.py:71)."""
        cfg = self.config
        feats = self._rows(feats_all, plan["rows"])
        logits = F.linear(feats, self.loss_calculator.tssp.classifier.weight, self.loss_calculator.tssp.classifier.bias)
        return cfg.tssp_loss_weight * F.cross_entropy(logits.reshape(-1, cfg.num_tssp_labels), up.get(plan["labs"]))

    def _plan_half(self, up, req, labels_cpu, Lq, off, da_example_flag, need_cos, stm_cpu=None, spo_cpu=None):
        cfg = self.config
        pos, lab = self._labelled_rows(labels_cpu)
        plan = dict(pos=pos, lab=lab, cos=None, cssl=None, tssp=None)
        if need_cos:
            plan["cos"] = self._plan_cos(up, req, pos, Lq, off)
        if not da_example_flag and cfg.cl_loss_weight != 0:
            plan["cssl"] = self._plan_cssl(up, req, pos, lab, Lq, off)
            plan["has_cssl"] = True
        if da_example_flag and cfg.tssp_loss_weight != 0:
            plan["tssp"] = self._plan_tssp(up, req, stm_cpu, spo_cpu, Lq, off)
        if cfg.ts_score_predictor == "cos":
            mx = plan["cos"]["mx"]
            flat = []
            for l_ in lab:
                flat += l_ + [-100] * (mx - len(l_))
            plan["cos_labels"] = up.add(flat)
        return plan

    def _loss_calculator(self, logits_half, feats, labels, up, plan, B, da_example_flag=False):
        """loss_calculator.py:25-73 on the pre-computed classifier logits of this half and the gathered head rows"""
        cfg = self.config
        cos = self._cos_sim(feats, up, plan["cos"], cfg.ts_score_predictor_cos_temp, B) if plan["cos"] is not None else None
        if cfg.ts_score_predictor == "lt":
            logits = logits_half
            ts = self._ts_loss(logits.reshape(-1, cfg.num_labels), labels.reshape(-1))
        elif cfg.ts_score_predictor == "cos":
            ts = F.binary_cross_entropy_with_logits(cos.reshape(-1), up.get(plan["cos_labels"]).float())
            logits = torch.sigmoid(cos)
        else:
            raise ValueError("not supported ts_score_predictor %s" % cfg.ts_score_predictor)
        loss = cfg.ts_loss_weight * ts
        if not da_example_flag and cfg.cl_loss_weight != 0:
            loss = loss + cfg.cl_loss_weight * self._cssl(feats, up, plan["cssl"])
        if da_example_flag and cfg.tssp_loss_weight != 0:
            loss = loss + cfg.tssp_loss_weight * self._tssp(feats, up, plan["tssp"])
        return loss, logits, cos

    # ------------------------------------------------------------------------------------------------ fused training heads
    def _fused_heads_ok(self, train):
        """the HIP heads (csrc/heads.hip) cover the training configurations of run_finetune.sh: linear token scores, plain or
        class-weighted CE or the reference's focal loss, CSSL in list form with a non-zero temperature, TSSP.  The cosine score predictor, the
        eop_matrix CSSL variant and evaluation (which also returns the cos-sim side output) stay on the torch formulation below."""
        cfg = self.config
        return (train and getattr(cfg, "amdseg_fused_heads", True) and cfg.ts_score_predictor == "lt"
                and (cfg.cl_loss_weight == 0 or (cfg.cl_anchor_level in ("eop_list", "eot_list") and cfg.cl_temp != 0
                                                 and cfg.cl_positive_k + cfg.cl_negative_k <= 16))
                and cfg.num_labels <= 4 and cfg.num_tssp_labels <= 4
                # the class-weight vector of the reference has two entries (loss_calculator.py: [w0, 1 - w0]); with more labels torch raises
                # a weight-size error -- leave that configuration to the torch heads so it still does
                and (cfg.num_labels == 2 or getattr(cfg, "weight_label_zero", 0.5) == 0.5))

    def _fused_heads(self, seq, labels, host, B, Lq, two_pass):
        cfg = self.config
        up, req = _IndexUploader(seq.device), _RowRequests()
        pos, lab = self._labelled_rows(host["labels"][:, 0])
        P = dict(nseg=2 if two_pass else 1, feat_off=0, anchor_off=-1, lists_off=0, n_anchor=0, n_list=0, pk=1, temp=float(cfg.cl_temp) or 1.0,
                 n_feat=0, t_rows_off=0, t_labels_off=0, nt=0, w_ts=float(cfg.ts_loss_weight), w_cl=float(cfg.cl_loss_weight),
                 gamma=float(cfg.focal_loss_gamma),
                 w_tssp2=float(cfg.tssp_loss_weight) ** 2)
        if cfg.cl_loss_weight != 0:
            cp = self._plan_cssl(up, req, pos, lab, Lq, 0)                  # same `random` call order as the reference (cssl.py:118-228)
            if cp is not None:
                s0, n = cp["rows"]
                P["feat_off"], P["n_feat"] = up.add(req.rows[s0:s0 + n])[0], n
                P["lists_off"] = cp["lists"][0]
                P["n_list"], P["pk"] = cp["nlists"], int(cfg.cl_positive_k)
                if cp["anchors"] is None:
                    P["n_anchor"] = n
                else:
                    P["anchor_off"], P["n_anchor"] = cp["anchors"]
        use_tssp = two_pass and cfg.tssp_loss_weight != 0
        if use_tssp:
            tp = self._plan_tssp(up, req, host["stm"], host["spo"], Lq, B * Lq)
            s0, n = tp["rows"]
            P["t_rows_off"], P["t_labels_off"], P["nt"] = up.add(req.rows[s0:s0 + n])[0], tp["labs"][0], n
        up.flush()
        labels_all = (torch.cat((labels[:, 0], labels[:, 1])) if two_pass else labels[:, 0]).reshape(-1).contiguous()
        class_w = None
        if cfg.weight_label_zero != 0.5:
            class_w = torch.tensor([cfg.weight_label_zero, 1 - cfg.weight_label_zero], dtype=torch.float32, device=seq.device)
        clf, tc = self.loss_calculator.classifier, self.loss_calculator.tssp.classifier
        eng = self.engine()
        if not eng.ddp_compat() and all(t.requires_grad for t in (clf.weight, clf.bias)) and (not use_tssp or all(t.requires_grad for t in (tc.weight, tc.bias))):
            # native mode: the heads' backward writes these gradients into their .grad views itself (torch DDP needs them from autograd instead)
            P["direct"] = (clf.weight, clf.bias, tc.weight if use_tssp else None, tc.bias if use_tssp else None)
        loss, logits_all = FusedHeadsFn.apply(seq.reshape(-1, seq.shape[-1]), clf.weight, clf.bias, tc.weight if use_tssp else None,
                                              tc.bias if use_tssp else None, labels_all, up.dev, class_w, P)
        C_ = logits_all.shape[-1]
        if two_pass:
            logits = logits_all.view(2, B, Lq, C_).transpose(0, 1)          # (B, 2, L, C): [anchor, augmented], no copy
        else:
            logits = logits_all.view(B, 1, Lq, C_).expand(B, 2, Lq, C_)     # bert_for_ts.py:96: logits[:, 1] duplicates logits[:, 0]
        return loss, logits

    # ------------------------------------------------------------------------------------------------ forward
    def forward(
        self,
        input_ids,
        attention_mask=None,
        head_mask=None,
        token_type_ids=None,
        position_ids=None,
        inputs_embeds=None,
        labels=None,
        output_attentions=None,
        output_hidden_states=None,
        return_dict=False,
        sent_level_labels=None,
        extract_eop_segment_ids=None,
        eop_index_for_aggregate_batch_eop_features=None,
        sent_pair_orders=None,
        sent_token_mask=None,
    ):
        if head_mask is not None or inputs_embeds is not None:
            raise L.AmdsegError("head_mask / inputs_embeds are not supported by the HIP encoder path")
        if output_attentions:
            raise L.AmdsegError("attention maps are never materialised by the fused HIP path (softmax lives in registers)")
        cfg = self.config
        B, two, Lq = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        # the host fetch of the step: label-like tensors are copied to pinned memory BEFORE the encoder kernels are queued
        # and awaited AFTER, so the copy only waits for the previous step's tail and all host-side index building below
        # overlaps the encoder running on the GPU
        two_pass = bool(cfg.do_da_ts or cfg.do_tssp)
        need_tssp = two_pass and labels is not None and cfg.tssp_loss_weight != 0
        host, ev = {}, None
        if labels is not None:
            fetch = {"labels": labels}
            if need_tssp:
                fetch["stm"] = sent_token_mask[:, 1]
                fetch["spo"] = sent_pair_orders[:, 1]
            # a caller that moved this batch to the device itself may have left the host originals (amdseg_set_host_twins): no copy back, no wait
            tw_l, tw_m, tw_o = self._host_twin(labels), self._host_twin(sent_token_mask), self._host_twin(sent_pair_orders)
            if tw_l is not None and (not need_tssp or (tw_m is not None and tw_o is not None)):
                fetch = {}
                host["labels"] = tw_l
                if need_tssp:
                    host["stm"], host["spo"] = tw_m[:, 1], tw_o[:, 1]
            for k, t in fetch.items():
                if t.is_cuda:
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    host[k] = h
                else:
                    host[k] = t
            if labels.is_cuda and fetch:
                ev = torch.cuda.Event()
                ev.record()
        if two_pass:   # anchor + augmented sequences in one encoder pass of 2B sequences
            ids = torch.cat((input_ids[:, 0], input_ids[:, 1])); am = torch.cat((attention_mask[:, 0], attention_mask[:, 1]))
            tt = torch.cat((token_type_ids[:, 0], token_type_ids[:, 1]))
        else:
            ids, am, tt = input_ids[:, 0].contiguous(), attention_mask[:, 0].contiguous(), token_type_ids[:, 0].contiguous()
        if position_ids is not None:                 # bert_for_ts.py:60,74: (B, 2, L) like the other columns, one slice per encoder pass
            if position_ids.shape != input_ids.shape:
                raise L.AmdsegError(f"position_ids must have the shape of input_ids {tuple(input_ids.shape)}, got {tuple(position_ids.shape)}")
            self.engine()._explicit_pos = (torch.cat((position_ids[:, 0], position_ids[:, 1])) if two_pass else position_ids[:, 0]).contiguous()
        try:
            return self._forward_encoded(ids, am, tt, labels, host, ev, B, Lq, two_pass, output_hidden_states)
        finally:
            if position_ids is not None:
                self.engine()._explicit_pos = None

    def _forward_encoded(self, ids, am, tt, labels, host, ev, B, Lq, two_pass, output_hidden_states):
        cfg = self.config
        hidden = None
        if output_hidden_states:
            # bert_for_ts.py:57-63,111-112: the flag goes to the ANCHOR encoder pass and its `hidden_states` tuple (embedding output + every
            # layer output, (B, L, H) each) is appended to the return.  Detached fp32 copies taken between the layer launches (no gradient
            # flows through them -- the reference's would, but no caller of the reference differentiates them).
            eng = self.engine()
            eng._hidden_sink = []
            try:
                seq = self.encode(ids, am, tt)
                hidden = tuple(h[:B] for h in eng._hidden_sink)
            finally:
                eng._hidden_sink = None
        else:
            seq = self.encode(ids, am, tt)
        if ev is not None:
            ev.synchronize()
        logits, cos = None, None
        loss = None
        if labels is not None and self._fused_heads_ok(self.training and torch.is_grad_enabled()):
            loss, logits = self._fused_heads(seq, labels, host, B, Lq, two_pass)
            return (loss, logits, torch.full((B, 1), -100.0, device=seq.device)) + ((hidden,) if hidden is not None else ())
        if labels is not None:
            train = self.training and torch.is_grad_enabled()
            need_cos = (not train) or cfg.ts_score_predictor == "cos"
            up, req = _IndexUploader(seq.device), _RowRequests()
            a_plan = self._plan_half(up, req, host["labels"][:, 0], Lq, 0, False, need_cos)
            d_plan = None
            if two_pass:
                d_plan = self._plan_half(up, req, host["labels"][:, 1], Lq, B * Lq, True, cfg.ts_score_predictor == "cos",
                                         stm_cpu=host.get("stm"), spo_cpu=host.get("spo"))
            rows_h = up.add(req.rows)
            up.flush()
            # ONE pass over the encoder output for all heads: classifier logits of every token + the gathered rows; its
            # backward returns ONE dense gradient (no per-head zero-fill / index_put / dense adds)
            clf = self.loss_calculator.classifier
            logits_all, feats = HeadInputsFn.apply(seq.reshape(-1, seq.shape[-1]), clf.weight, clf.bias, up.get(rows_h))
            logits_all = logits_all.view(seq.shape[0], Lq, -1)
            a_loss, a_logits, cos = self._loss_calculator(logits_all[:B], feats, labels[:, 0], up, a_plan, B)
            loss = a_loss
            logits = torch.cat((a_logits.unsqueeze(1), a_logits.unsqueeze(1)), dim=1)
            if two_pass:
                d_loss, d_logits, _ = self._loss_calculator(logits_all[B:], feats, labels[:, 1], up, d_plan, B, da_example_flag=True)
                loss = loss + d_loss
                logits = torch.cat((a_logits.unsqueeze(1), d_logits.unsqueeze(1)), dim=1)
            if cos is None:
                cos = torch.full((B, 1), -100.0, device=seq.device)
        output = (logits, cos) + ((hidden,) if hidden is not None else ())
        return ((loss,) + output) if loss is not None else output


class BertWithDAForSentenceLabelingTopicSegmentation(TopicSegHeadsMixin, BertPreTrainedModel):
    _keys_to_ignore_on_load_unexpected = [r"pooler"]

    def __init__(self, config):
        self._fill_head_defaults(config)
        super().__init__(config)
        self.config = config
        self.bert = BertModel(config)          # parameter container only (HF names); its torch forward is never called
        classifier_dropout = config.classifier_dropout if config.classifier_dropout is not None else config.hidden_dropout_prob
        self._init_heads(config, classifier_dropout)
        self.post_init()

    # ------------------------------------------------------------------------------------------------ engine plumbing
    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise L.AmdsegError("spokennlp_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        if self._engine is None or not self._engine.fp.intact() or self._engine.device != p.device:
            self._engine = BertEncoderEngine(self, self.config, p.device, bert_attr="bert")
        return self._engine

