"""The `AutoModelForTokenClassification` surface of the drop-in classes (BASELINE north_star; the reference driver imports it at
ts_sentence_seq_labeling.py:43 next to `AutoConfig` / `AutoTokenizer`, :188-224, and constructs the model classes directly, :250-257).

transformers ignores `AutoModelForTokenClassification.register(BertConfig, ...)` for its own config classes (the built-in mapping wins:
`from_config(BertConfig)` keeps returning `BertForTokenClassification`), so each family gets a config twin with its own `model_type`:

    amdseg-bert / amdseg-electra / amdseg-longformer / amdseg-bigbird

registered with `AutoConfig`, and a model class bound to that config registered with `AutoModelForTokenClassification`.  The bound
classes are SUBCLASSES of the drop-in classes with no body of their own (only `config_class`), so

    cfg = spokennlp_amd.auto.amdseg_config(AutoConfig.from_pretrained("bert-base-uncased"), do_da_ts=True, ...)
    model = AutoModelForTokenClassification.from_config(cfg)            # -> a BertWithDAForSentenceLabelingTopicSegmentation
    model.save_pretrained(out)                                          # config.json: model_type "amdseg-bert"
    AutoModelForTokenClassification.from_pretrained(out)                # -> the same class, same logits

while the direct-class path of the reference driver (`Cls.from_pretrained(path, config=config)`, stock `BertConfig`, `model_type
"bert"` in the checkpoint) is untouched.  Parameter names are identical on both paths: a checkpoint written by one loads into the other.
"""
from transformers import AutoConfig, AutoModelForTokenClassification
from transformers import BertConfig, BigBirdConfig, ElectraConfig, LongformerConfig

from .bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation
from .bigbird_for_ts import BigBirdWithDAForSentenceLabelingTopicSegmentation
from .electra_for_ts import ElectraWithDAForSentenceLabelingTopicSegmentation
from .longformer_for_ts import LongformerWithDAForSentenceLabelingTopicSegmentation


class AmdsegBertConfig(BertConfig):
    model_type = "amdseg-bert"


class AmdsegElectraConfig(ElectraConfig):
    model_type = "amdseg-electra"


class AmdsegLongformerConfig(LongformerConfig):
    model_type = "amdseg-longformer"


class AmdsegBigBirdConfig(BigBirdConfig):
    model_type = "amdseg-bigbird"


class AmdsegBertForTokenClassification(BertWithDAForSentenceLabelingTopicSegmentation):
    config_class = AmdsegBertConfig


class AmdsegElectraForTokenClassification(ElectraWithDAForSentenceLabelingTopicSegmentation):
    config_class = AmdsegElectraConfig


class AmdsegLongformerForTokenClassification(LongformerWithDAForSentenceLabelingTopicSegmentation):
    config_class = AmdsegLongformerConfig


class AmdsegBigBirdForTokenClassification(BigBirdWithDAForSentenceLabelingTopicSegmentation):
    config_class = AmdsegBigBirdConfig


FAMILIES = {                      # stock model_type -> (config twin, Auto-registered class, the drop-in class of the direct path)
    "bert": (AmdsegBertConfig, AmdsegBertForTokenClassification, BertWithDAForSentenceLabelingTopicSegmentation),
    "electra": (AmdsegElectraConfig, AmdsegElectraForTokenClassification, ElectraWithDAForSentenceLabelingTopicSegmentation),
    "longformer": (AmdsegLongformerConfig, AmdsegLongformerForTokenClassification, LongformerWithDAForSentenceLabelingTopicSegmentation),
    "big_bird": (AmdsegBigBirdConfig, AmdsegBigBirdForTokenClassification, BigBirdWithDAForSentenceLabelingTopicSegmentation),
}


def amdseg_config(config, **overrides):
    """the registered twin of a stock config (same fields, `model_type` "amdseg-<family>"); a twin is returned as is (updated)."""
    for ccls, _, _ in FAMILIES.values():
        if isinstance(config, ccls):
            for k, v in overrides.items():
                setattr(config, k, v)
            return config
    if config.model_type not in FAMILIES:
        raise ValueError(f"no MI355X drop-in for model_type {config.model_type!r} (have: {sorted(FAMILIES)})")
    ccls = FAMILIES[config.model_type][0]
    d = config.to_dict()
    for k in ("model_type", "transformers_version", "architectures"):
        d.pop(k, None)
    d.update(overrides)
    return ccls(**d)


_registered = False


def register():
    """idempotent; called at `import spokennlp_amd.auto` (and by `import spokennlp_amd` lazily through `spokennlp_amd.register_auto()`)"""
    global _registered
    if _registered:
        return
    for ccls, mcls, _ in FAMILIES.values():
        AutoConfig.register(ccls.model_type, ccls, exist_ok=True)
        AutoModelForTokenClassification.register(ccls, mcls, exist_ok=True)
    _registered = True


register()
