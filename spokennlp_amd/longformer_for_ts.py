"""MI355X-native drop-in for emnlp2023-topic_segmentation/src/models/longformer_for_ts.py:18-129
(LongformerWithDAForSentenceLabelingTopicSegmentation): same class name, HF parameter names (`longformer.*`,
`loss_calculator.*`), forward(**batch) signature incl. `global_attention_mask`, and (loss, logits, cos_sim) return.
`self.longformer` is only the parameter container; the encoder runs on libamdseg (longformer_engine.py).
"""
import torch
from transformers.models.longformer.modeling_longformer import LongformerModel, LongformerPreTrainedModel

from . import lib as L
from .bert_for_ts import TopicSegHeadsMixin
from .longformer_engine import LongformerEncoderEngine


class LongformerWithDAForSentenceLabelingTopicSegmentation(TopicSegHeadsMixin, LongformerPreTrainedModel):
    _keys_to_ignore_on_load_unexpected = [r"pooler"]

    def __init__(self, config):
        self._fill_head_defaults(config)
        super().__init__(config)
        self.config = config
        self.longformer = LongformerModel(config, add_pooling_layer=False)     # parameter container only
        self._init_heads(config, config.hidden_dropout_prob)                   # longformer_for_ts.py:27
        self.post_init()

    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise L.AmdsegError("spokennlp_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        if self._engine is None or not self._engine.fp.intact() or self._engine.device != p.device:
            self._engine = LongformerEncoderEngine(self, self.config, p.device, bert_attr="longformer")
        return self._engine

    def forward(
        self,
        input_ids,
        attention_mask=None,
        global_attention_mask=None,
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
        if global_attention_mask is not None:
            g = global_attention_mask
            ok = g.dim() == 2 and bool((g[:, 0] == 1).all()) and int(g.sum()) == g.shape[0]
            if not ok:
                raise L.AmdsegError("the HIP Longformer path supports the reference's global mask only: [CLS] (token 0) global")
        return TopicSegHeadsMixin.forward(
            self, input_ids, attention_mask=attention_mask, head_mask=head_mask, token_type_ids=token_type_ids,
            position_ids=position_ids, inputs_embeds=inputs_embeds, labels=labels, output_attentions=output_attentions,
            output_hidden_states=output_hidden_states, return_dict=return_dict, sent_level_labels=sent_level_labels,
            extract_eop_segment_ids=extract_eop_segment_ids,
            eop_index_for_aggregate_batch_eop_features=eop_index_for_aggregate_batch_eop_features,
            sent_pair_orders=sent_pair_orders, sent_token_mask=sent_token_mask)
