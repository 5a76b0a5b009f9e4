"""`run_inference.sh`-equivalent prediction loop on top of the drop-in model (ts_sentence_seq_labeling.py:1119-1210):
pre-tokenised documents -> feature builder -> batched model forward -> anchor-logit decode at the labelled positions ->
per-document merge -> prediction file + example-level metrics.  Host glue only; the encoder runs through the model class
(HIP path), the integer work through spokennlp_amd.preprocess / evaluate."""
import random

import numpy as np
import torch

from . import evaluate as E
from . import preprocess as P

MODEL_COLUMNS = ("input_ids", "attention_mask", "token_type_ids", "labels", "sent_level_labels", "extract_eop_segment_ids",
                 "eop_index_for_aggregate_batch_eop_features", "sent_pair_orders", "sent_token_mask")


def predict_documents(model, docs_sentence_ids, docs_labels, max_seq_length, bos_id, cls_id, pad_id, batch_size=8, device=None,
                      threshold=0.5, seed=42, tssp_ablation="none"):
    """returns (documents, metrics): documents = list of per-document dicts (labels / int_labels / predictions / predict_logits
    [/ eop_pair_cos_sim]) as written by the reference's predict branch; metrics = compute_metric_example_level(...)."""
    random.seed(seed)                                         # the feature builder draws the DA half from `random`
    cols = P.prepare_features(docs_sentence_ids, docs_labels, list(range(len(docs_sentence_ids))), max_seq_length, bos_id, cls_id,
                              pad_id, tssp_ablation=tssp_ablation)
    n = len(cols["input_ids"])
    # multi-GPU predict (run_inference.sh:35 launches 2 ranks): this rank's windows are r, r+W, ... (dp.shard_indices, the tail wrapped);
    # the rows come back in window order through dp.gather_sharded.  One process: mine == range(n), the gather is the identity.
    import torch.distributed as dist
    from . import dp
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    mine = dp.shard_indices(n, dist.get_rank(), world) if world > 1 else list(range(n))
    logits_all, cos_all = [], []
    model.eval()
    with torch.no_grad():
        for i in range(0, len(mine), batch_size):
            idx = mine[i:i + batch_size]
            pad_n = batch_size - len(idx)
            idx = idx + [idx[-1]] * pad_n                   # fixed batch shape (tile-aligned); the tail repeats the last sample
            batch = {k: torch.tensor([cols[k][j] for j in idx], dtype=torch.long, device=device) for k in MODEL_COLUMNS}
            _, logits, cos = model(**batch)[:3]
            keep = batch_size - pad_n
            logits_all.append(logits[:keep].float())
            cos_all.append(cos[:keep].float())
    logits_t = torch.cat(logits_all, 0)
    kmax = max(c.shape[1] for c in cos_all)
    if world > 1:                                             # cos_sim is (B, k) with k = the batch's own maximum: one width for the gather
        kt = torch.tensor([kmax], device=logits_t.device)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kmax = int(kt.item())
    cos_t = torch.cat([torch.nn.functional.pad(c, (0, kmax - c.shape[1]), value=-100.0) for c in cos_all], 0)
    logits_all = dp.gather_sharded(logits_t, n).cpu().numpy()
    cos_all = dp.gather_sharded(cos_t, n).cpu().numpy().tolist()
    labels = np.array(cols["labels"])
    decoded = P.decode_anchor_predictions(logits_all, labels)
    example_ids = [e[0] for e in cols["example_id"]]
    docs = P.merge_windows_to_documents(decoded, example_ids, len(docs_sentence_ids), eop_pair_cos_sim=cos_all)
    # (:1198-1206) documents with a single sentence have no labelled position: the reference drops them before scoring
    scored = [d for d in docs if len(d["int_labels"]) != 0]
    metrics = E.compute_metric_example_level([d["predict_logits"] for d in scored], [d["int_labels"] for d in scored], threshold=threshold)
    return docs, metrics


def write_predict_outputs(output_dir, docs, metrics, test_data_name, max_seq_length, ts_score_predictor="lt", threshold=0.5,
                          metric_key_prefix="predict"):
    """the three files the predict branch of the reference leaves in `output_dir` (ts_sentence_seq_labeling.py:1166-1222):
      <prefix>_<data>_max_seq<L>_ts_score_<p>.txt                                  one json line per document (:1173-1191)
      example_level_<prefix>_<data>_max_seq<L>_ts_score_<p>_results.json           `trainer.save_metrics` layout (:1214-1219)
      example_level_<...>_results_str_metric.txt                                   `convert_res_format` (utils.py:23-48; :1222)
    `threshold` is formatted with %s as the reference's key names are (0.5 -> "threshold_0.5_example_level_f1").  Returns the paths."""
    import json
    import os
    os.makedirs(output_dir, exist_ok=True)
    name = "_".join([metric_key_prefix, test_data_name, "max_seq%d" % max_seq_length, "ts_score_%s" % ts_score_predictor])
    pred_path = os.path.join(output_dir, name + ".txt")
    P.write_prediction_file(pred_path, docs)
    m = dict(metrics)
    m["%s_examples" % metric_key_prefix] = len(docs)
    res_path = os.path.join(output_dir, "example_level_" + name + "_results.json")
    with open(res_path, "w") as f:
        json.dump(m, f, indent=4, sort_keys=True)
    str_path = E.convert_res_format(res_path, threshold, out=lambda *_: None)
    return pred_path, res_path, str_path
