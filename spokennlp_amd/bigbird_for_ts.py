"""MI355X-native drop-in for emnlp2023-topic_segmentation/src/models/bigbird_for_ts.py:19-113
(BigBirdWithDAForSentenceLabelingTopicSegmentation; selected by ts_sentence_seq_labeling.py:237-239 for model names
containing "bigbird").  Same class name, `forward(**batch)` arguments, `(loss, logits, cos_sim)` outputs and HF parameter
names (`bert.*`: the reference keeps a BigBirdModel under the attribute `bert`, bigbird_for_ts.py:27).

What differs from the BERT engine ([hf] models/big_bird/modeling_big_bird.py):
  * BigBirdEmbeddings: LayerNorm(dropout(word + type + position)) -- dropout BEFORE the LayerNorm;
  * hidden_act "gelu_new" (tanh form) in the FFN -- `amdseg_bert_cfg.act = 1`;
  * attention_type "block_sparse": block-list attention (`amdseg_attn_list_fwd/bwd`, lists from bigbird_plan.py), context rows
    of padded queries zeroed inside the kernels (`context_layer * from_mask`), additive key mask -10000 * (1 - mask); the random blocks are seeded
    with the layer index and are all block 0 in eval mode, as in the reference;
  * sequences of at most (5 + 2 * num_random_blocks) * block_size tokens switch the model to "original_full" attention FOR
    GOOD (`BigBirdModel.forward` calls `set_attention_type`), which is the BERT attention path;
  * inputs are padded to a multiple of block_size by the reference (`_pad_to_block_size`); this engine requires L % 64 == 0
    (the topic-segmentation driver pads every window to max_seq_length), and block_size == 64.
"""
import ctypes as C

import torch
from transformers.models.big_bird.modeling_big_bird import BigBirdModel, BigBirdPreTrainedModel

from . import bigbird_plan as plan
from . import lib as L
from . import ops
from .bert_for_ts import TopicSegHeadsMixin
from .engine import BertEncoderEngine


class BigBirdEncoderEngine(BertEncoderEngine):
    supports_parity = False
    def __init__(self, module, config, device, bert_attr="bert"):
        super().__init__(module, config, device, bert_attr=bert_attr)
        if getattr(config, "attention_type", "block_sparse") == "block_sparse" and config.block_size != plan.BLOCK:
            raise L.AmdsegError(f"block-sparse attention is implemented for block_size == {plan.BLOCK} (got {config.block_size})")
        if getattr(config, "rescale_embeddings", False):
            raise L.AmdsegError("rescale_embeddings is not implemented")
        if not getattr(config, "use_bias", True):
            raise L.AmdsegError("use_bias=False is not implemented")
        self.emb_dropout_pre_ln = True
        self.attention_type = getattr(config, "attention_type", "block_sparse")
        self.attn_keepmask = self.attn_keepmask and self.attention_type == "original_full"      # the block-list kernels have no dropout
        self._plans = {}
        self._cur = None

    def _plan(self, Lseq, train):
        key = (Lseq, bool(train))
        if key not in self._plans:
            cfg, dev = self.cfg, self.device
            per_layer = []
            for i in range(self.nlayers):                  # reference: BigBirdLayer(config, seed=layer_idx)
                t = plan.build(Lseq, self.heads, cfg.num_random_blocks, i, train, cfg.max_position_embeddings)
                per_layer.append(dict(klist=torch.from_numpy(t["klist"]).to(dev), kcnt=torch.from_numpy(t["kcnt"]).to(dev),
                                      qlist=torch.from_numpy(t["qlist"]).to(dev), qcnt=torch.from_numpy(t["qcnt"]).to(dev),
                                      korder=torch.from_numpy(t["korder"]).to(dev), qorder=torch.from_numpy(t["qorder"]).to(dev),
                                      stride=int(t["stride"])))
                if not train:                               # eval: no randomness -> every layer shares one plan
                    per_layer = per_layer * self.nlayers
                    break
            self._plans[key] = per_layer
        return self._plans[key]

    def forward(self, input_ids, attention_mask, token_type_ids, train, seed=0, p_out=0.0):
        B, Lseq = input_ids.shape
        if self.attention_type == "block_sparse" and Lseq <= plan.min_block_sparse_len(self.cfg.num_random_blocks):
            self.attention_type = "original_full"           # reference: permanent (set_attention_type), with a logged warning
        if self.attention_type == "block_sparse":
            fp32 = (not train) and getattr(self.cfg, "amdseg_precision", "bf16") == "fp32"
            valid = (attention_mask == 1)
            self._cur = dict(plan=self._plan(Lseq, train), fp32=fp32,
                             mb=((~valid).to(torch.float32) * -10000.0).reshape(-1).contiguous())
        else:
            self._cur = None
        return super().forward(input_ids, attention_mask, token_type_ids, train, seed, p_out)

    def _layer_forward(self, lib, cfg, lp, A, i, mb, s, train):
        if self._cur is None:
            return super()._layer_forward(lib, cfg, lp, A, i, mb, s, train)
        cur, pl = self._cur, self._cur["plan"][i]
        cfg.nproj, cfg.mixer = 3, 1
        acts = A["acts_struct"][i]
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].1")
        la = A["layers"][i if train else 0]
        with torch.no_grad():
            if cur["fp32"]:                                  # parity mode: plain fp32 kernel, same lists
                ops.attn_list_f32(la["qkv"], cur["mb"], cfg.B, cfg.L, self.heads, pl["klist"], pl["kcnt"], pl["stride"], ctx=la["ctx"])
            else:
                ops.attn_list_fwd(la["qkv"], cur["mb"], cfg.B, cfg.L, self.heads, pl["klist"], pl["kcnt"], pl["stride"], ctx=la["ctx"], lse=la["lse"], korder=pl["korder"])
            # rows of padded queries come out zero (reference: context_layer * from_mask) and carry no gradient: the kernels read it
            # off the key mask of the query's own position
        cfg.phase = 2
        L.check(lib.amdseg_bert_layer_fwd(C.byref(cfg), C.byref(lp), C.byref(acts), mb, i, s), f"amdseg_bert_layer_fwd[{i}].2")
        cfg.phase, cfg.mixer, cfg.nproj = 0, 0, 0
        return dict(cur=cur) if train else None

    def _layer_backward(self, lib, cfg, A, i, mb, dy, other, s, saved):
        if saved is None:
            return super()._layer_backward(lib, cfg, A, i, mb, dy, other, s, saved)
        cur = saved["cur"]
        pl = cur["plan"][i]
        cfg.nproj, cfg.mixer = 3, 1
        args = (C.byref(cfg), C.byref(self.lparams[i]), C.byref(self.lgrads[i]), C.byref(A["acts_struct"][i]), C.byref(A["ws_struct"]),
                mb, dy.data_ptr(), other.data_ptr(), i, s)
        cfg.phase = 1
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].1")
        la, ws = A["layers"][i], A["ws"]
        with torch.no_grad():
            ops.attn_list_bwd(la["qkv"], cur["mb"], la["ctx"], ws["dctx"], la["lse"], cfg.B, cfg.L, self.heads, pl["klist"], pl["kcnt"],
                              pl["qlist"], pl["qcnt"], pl["stride"], dqkv=ws["dqkv"], delta=ws["delta"], korder=pl["korder"], qorder=pl["qorder"])
        cfg.phase = 2
        L.check(lib.amdseg_bert_layer_bwd(*args), f"amdseg_bert_layer_bwd[{i}].2")
        cfg.phase, cfg.mixer, cfg.nproj = 0, 0, 0


class BigBirdWithDAForSentenceLabelingTopicSegmentation(TopicSegHeadsMixin, BigBirdPreTrainedModel):
    _keys_to_ignore_on_load_unexpected = [r"pooler"]       # bigbird_for_ts.py:20

    def __init__(self, config):
        self._fill_head_defaults(config)
        super().__init__(config)
        self.config = config
        self.bert = BigBirdModel(config)                    # parameter container only (bigbird_for_ts.py:27)
        dropout = config.classifier_dropout if config.classifier_dropout is not None else config.hidden_dropout_prob
        self._init_heads(config, dropout)                   # bigbird_for_ts.py:28-32
        self.post_init()

    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise L.AmdsegError("spokennlp_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        if self._engine is None or not self._engine.fp.intact() or self._engine.device != p.device:
            self._engine = BigBirdEncoderEngine(self, self.config, p.device, bert_attr="bert")
        return self._engine
