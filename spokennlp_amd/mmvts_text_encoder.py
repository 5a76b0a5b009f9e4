"""MI355X-native drop-in for mmvts/src/models/text_encoder/text_encoder.py:5-89 (`TextEncoder`), the text branch of the
multimodal video topic-segmentation model (`multi_modal_for_ts.py:27,173-184`; `run_finetune_text.sh` = fuse_type text_only
with a Chinese Longformer at 2048 / 4096 tokens, 8-GPU DDP).

Same surface: `TextEncoder(config)` holds the HF model under `self.text_encoder` (so `model.text_encoder.text_encoder.
resize_token_embeddings(...)`, main_text.py:291-295, `from_pretrained` initialisation and checkpoints keep working), picks BERT
when "bert" is in `config.text_encoder_name_or_path` and Longformer otherwise (:17,33), and `forward(input_ids,
attention_mask, global_attention_mask, ...)` returns `text_outputs[0]`: the [B, L, H] sequence features.  The HF module is a
parameter container only; the encoder runs on the libamdseg engines and is differentiable (EncoderFn), so the reference's
projector / predictor / loss layers on top of it run unchanged.

Longformer is called by the reference with `global_attention_mask=None` (nobody builds one in mmvts): pure sliding-window
attention, no global token -- `LongformerEncoderEngine.cls_global = False`.  A non-None global mask is rejected.  The other
keyword arguments of the reference forward (head_mask, position_ids, inputs_embeds, output_attentions, output_hidden_states)
are accepted and must be None / False, as in the reference's call.
"""
import torch
import torch.nn as nn

from . import lib as L
from .engine import BertEncoderEngine, EncoderFn
from .longformer_engine import LongformerEncoderEngine


class TextEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.text_encoder = self.get_encoder(config)
        self._engine = None
        self._step = 0
        self.amdseg_seed = 0

    def get_encoder(self, config):
        name = getattr(config, "text_encoder_name_or_path", "") or ""
        if "bert" in name:                                   # text_encoder.py:17
            self.encoder_type = "bert"
            from transformers.models.bert.modeling_bert import BertModel as Cls
        else:
            self.encoder_type = "lf"
            from transformers.models.longformer.modeling_longformer import LongformerModel as Cls
        if getattr(config, "init_model", False):             # text_encoder.py:22-28 / 38-44
            return Cls.from_pretrained(name, config=config, cache_dir=getattr(config, "cache_dir", None),
                                       revision=getattr(config, "model_revision", "main"))
        return Cls(config)

    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise L.AmdsegError("spokennlp_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        if self._engine is None or not self._engine.fp.intact() or self._engine.device != p.device:
            if self.encoder_type == "bert":
                self._engine = BertEncoderEngine(self, self.config, p.device, bert_attr="text_encoder")
            else:
                self._engine = LongformerEncoderEngine(self, self.config, p.device, bert_attr="text_encoder")
                self._engine.cls_global = False
        return self._engine

    def forward(self, input_ids, attention_mask=None, global_attention_mask=None, head_mask=None, token_type_ids=None,
                position_ids=None, inputs_embeds=None, labels=None, output_attentions=None, output_hidden_states=None,
                return_dict=False):
        if head_mask is not None or position_ids is not None or inputs_embeds is not None or output_attentions or output_hidden_states:
            raise L.AmdsegError("TextEncoder: head_mask / position_ids / inputs_embeds / output_attentions / output_hidden_states "
                                "are not used by the reference's call (multi_modal_for_ts.py:173-184) and are not implemented")
        if global_attention_mask is not None and bool((global_attention_mask != 0).any()):
            raise L.AmdsegError("TextEncoder: the reference passes global_attention_mask=None; global tokens are not implemented here")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        eng = self.engine()
        train = self.training and torch.is_grad_enabled()
        self._step += 1
        seed = (int(self.amdseg_seed) * 1000003 + self._step) & 0x7FFFFFFF
        return eng.encode(input_ids, attention_mask, token_type_ids, train, seed, 0.0)
