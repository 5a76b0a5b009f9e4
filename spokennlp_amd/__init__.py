"""spokennlp_amd -- MI355X-native (gfx950) BERT token-classification path for SpokenNLP topic segmentation.

Hand-written HIP kernels behind a C ABI (include/amdseg.h, spokennlp_amd/csrc) + the Python host mirror of the
reference's HuggingFace plug-in surface (emnlp2023-topic_segmentation/src/models/bert_for_ts.py).
"""
__version__ = "0.1.0"

# the Auto* surface (north_star: "keeping the HuggingFace AutoModelForTokenClassification + Trainer plug-in surface"): config twins with
# their own model_type + the drop-in classes, registered with AutoConfig / AutoModelForTokenClassification at package import
# (a ctypes-only user of `spokennlp_amd.lib` / `.build` must not depend on transformers being importable: the registration is best effort here and
#  `import spokennlp_amd.auto` raises the real error)
try:
    from . import auto  # noqa: E402,F401
    from .auto import amdseg_config  # noqa: E402,F401
except Exception as _e:                                      # noqa: BLE001 -- ImportError (transformers absent / too old) AND what transformers' lazy modules
    _auto_import_error = _e                                  # raise for a broken sub-module (RuntimeError "Failed to import ..."): the ctypes-only
                                                             # surface (`spokennlp_amd.lib`, `.build`) must import regardless

    def amdseg_config(*_a, **_k):
        raise ImportError(f"spokennlp_amd.auto could not be imported: {type(_auto_import_error).__name__}: {_auto_import_error}")
