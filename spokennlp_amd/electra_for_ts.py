"""MI355X-native drop-in for emnlp2023-topic_segmentation/src/models/electra_for_ts.py:19-110
(ElectraWithDAForSentenceLabelingTopicSegmentation).  ELECTRA-base's encoder is layer-for-layer the BERT block
([hf] models/electra/modeling_electra.py: ElectraEmbeddings == BertEmbeddings with embedding_size, ElectraLayer ==
BertLayer), so the BERT engine runs it unchanged on the parameters found under `electra.*`.  Checkpoints whose
embedding_size differs from hidden_size (electra-small: 128 -> 256) carry `electra.embeddings_project`: the engine then runs the
embedding kernels at the embedding width and the projection as one NT GEMM (+ its weight / bias / input gradients), bf16 precision only.
"""
from transformers.models.electra.modeling_electra import ElectraModel, ElectraPreTrainedModel

from . import lib as L
from .bert_for_ts import TopicSegHeadsMixin
from .engine import BertEncoderEngine


class ElectraWithDAForSentenceLabelingTopicSegmentation(TopicSegHeadsMixin, ElectraPreTrainedModel):
    def __init__(self, config):
        self._fill_head_defaults(config)
        super().__init__(config)
        self.config = config
        self.electra = ElectraModel(config)        # parameter container only
        self._init_heads(config, config.hidden_dropout_prob)      # electra_for_ts.py:26
        self.post_init()

    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise L.AmdsegError("spokennlp_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        if self._engine is None or not self._engine.fp.intact() or self._engine.device != p.device:
            self._engine = BertEncoderEngine(self, self.config, p.device, bert_attr="electra")
        return self._engine
