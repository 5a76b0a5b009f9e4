"""spokennlp_amd -- MI355X-native (gfx950) BERT token-classification path for SpokenNLP topic segmentation.

Hand-written HIP kernels behind a C ABI (include/amdseg.h, spokennlp_amd/csrc) + the Python host mirror of the
reference's HuggingFace plug-in surface (emnlp2023-topic_segmentation/src/models/bert_for_ts.py).
"""
__version__ = "0.1.0"
