"""BASELINE config 1 (plumbing): 32 synthetic documents through the whole inference path -- feature builder -> drop-in
model (HIP encoder, fp32 parity mode) -> decode -> per-document merge -> prediction file -> example-level metrics --
against the same pipeline with the CPU oracle in place of the model: predicted boundary labels must be identical."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


class OracleModel:
    """the oracle behind the model-class call convention (test-side only)"""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def eval(self):
        return self

    def __call__(self, **batch):
        from oracle import bert_ts_oracle as O
        return O.model_forward(self.sd, self.cfg, batch)


@pytest.mark.parametrize("precision", ["fp32", "parity"])              # exact-fp32 MFMA | split-bf16 products: both must meet 1e-3 + bit-exact boundaries
@pytest.mark.parametrize("max_len,bs", [(128, 4), (100, 3)])          # (100, 3): neither a 64-token multiple nor a full 128-row tile
def test_32_documents_end_to_end(dev, tmp_path, max_len, bs, precision):
    from oracle import bert_ts_oracle as O
    from spokennlp_amd import data, inference, preprocess as P
    from tests.test_oracle_golden import load_case, flags_of
    from tests.test_gpu_model import build_model
    z, sd, _, arch = load_case("tiny_L128")
    flags = flags_of(z, "full_eval")
    docs = data.synth_docs(32, seed=2024, vocab=arch["vocab_size"], mean_sents=30, sd_sents=10, mean_boundaries=4, mu_tok=1.8, sigma_tok=0.5)
    bos = arch["vocab_size"] - 1
    sent_ids = [[s.tolist() for s in d["sentences"]] for d in docs]
    labels = [[0 if v == 1 else 1 for v in d["labels"]] for d in docs]           # jsonl label 1 (section end) -> "B-EOP" = id 0
    m = build_model(arch, dict(flags, amdseg_precision=precision), sd, dev)
    m.config.amdseg_precision = precision
    got_docs, got_metrics = inference.predict_documents(m, sent_ids, labels, max_len, bos, data.CLS_ID, data.PAD_ID, batch_size=bs, device=dev)
    ref = OracleModel(sd, O.make_cfg(num_labels=2, **arch, **flags))
    ref_docs, ref_metrics = inference.predict_documents(ref, sent_ids, labels, max_len, bos, data.CLS_ID, data.PAD_ID, batch_size=bs, device=None)
    assert len(got_docs) == 32
    worst = 0.0
    for g, r, lab in zip(got_docs, ref_docs, labels):
        assert g["int_labels"] == r["int_labels"] and g["labels"] == r["labels"]
        assert g["predictions"] == r["predictions"]                              # predicted boundary labels: bit-exact
        assert len(g["predictions"]) == len(lab) - 1                             # every sentence but the document's last is predicted once
        worst = max(worst, float(np.abs(np.array(g["predict_logits"]) - np.array(r["predict_logits"])).max()))
        assert np.allclose(g["eop_pair_cos_sim"], r["eop_pair_cos_sim"], atol=1e-4)
    assert worst < 1e-3, worst                                                   # north-star logits tolerance
    for k, v in ref_metrics.items():
        assert abs(got_metrics[k] - v) < 1e-9, k
    path = tmp_path / "predict_synth_max_seq128_ts_score_lt.txt"
    P.write_prediction_file(str(path), got_docs)
    lines = open(path).read().splitlines()
    assert len(lines) == 32 and json.loads(lines[5])["predictions"] == got_docs[5]["predictions"]
    # ... and the files run_inference.sh leaves behind the prediction file (ts_sentence_seq_labeling.py:1214-1222, utils.py:23-48):
    # documents -> HIP encoder -> decode -> example-level metrics -> *_results.json -> *_str_metric.txt
    pred_p, res_p, str_p = inference.write_predict_outputs(str(tmp_path / "out"), got_docs, got_metrics, "synth", max_len)
    assert os.path.basename(pred_p) == "predict_synth_max_seq%d_ts_score_lt.txt" % max_len and open(pred_p).read() == open(path).read()
    assert os.path.basename(str_p) == "example_level_predict_synth_max_seq%d_ts_score_lt_results_str_metric.txt" % max_len
    saved = json.load(open(res_p))
    assert saved["predict_examples"] == 32 and saved["threshold_0.5_example_level_f1"] == got_metrics["threshold_0.5_example_level_f1"]
    want = " / ".join("%.2f" % (ref_metrics["threshold_0.5_example_level_" + k] * 100) for k in ("precision", "recall", "f1", "pk", "wd"))
    assert open(str_p).read() == "p / r / f / pk / wd\nthreshold_0.5_example_level_metric\n" + want + "\n\n"
    print("e2e: max|dlogit|", worst, {k: got_metrics[k] for k in ("precision", "recall", "f1")})


def test_prefetcher_feeds_training_steps(dev):
    """jsonl-shaped documents -> features -> DevicePrefetcher (pinned, side-stream H2D) -> a few training steps"""
    from spokennlp_amd import data, loader as LD
    from tests.test_oracle_golden import load_case, flags_of
    from tests.test_gpu_model import build_model
    z, sd, _, arch = load_case("tiny_L128")
    docs = data.synth_docs(12, seed=3, vocab=arch["vocab_size"], mean_sents=30, sd_sents=8, mean_boundaries=4, mu_tok=1.8, sigma_tok=0.5)
    sent_ids = [[s.tolist() for s in d["sentences"]] for d in docs]
    labels = [[0 if v == 1 else 1 for v in d["labels"]] for d in docs]
    random.seed(1)
    feats = LD.build_features(sent_ids, labels, list(range(len(docs))), 128, arch["vocab_size"] - 1, data.CLS_ID, data.PAD_ID)
    batches = LD.batch_indices(len(feats["input_ids"]), 2)
    m = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    eng = m.engine()
    losses = []
    for step, batch in enumerate(LD.DevicePrefetcher(feats, batches[:6], dev)):
        assert batch["input_ids"].is_cuda and batch["input_ids"].shape == (2, 2, 128)
        random.seed(step)
        loss = m(**batch)[0]
        loss.backward()
        eng.adamw_step(5e-5)
        losses.append(loss.item())
    assert len(losses) == 6 and all(np.isfinite(losses))
    # the same with the host originals handed to the model (no copy back of the labels, no event wait in forward): same first loss
    m2 = build_model(arch, flags_of(z, "train_full"), sd, dev).train()
    waits = {"n": 0}
    real_sync = torch.cuda.Event.synchronize

    def counting_sync(self):
        waits["n"] += 1
        return real_sync(self)
    torch.cuda.Event.synchronize = counting_sync
    try:
        losses2 = []
        for step, batch in enumerate(LD.DevicePrefetcher(feats, batches[:6], dev, model=m2)):
            random.seed(step)
            loss = m2(**batch)[0]
            loss.backward()
            m2.engine().adamw_step(5e-5)
            losses2.append(loss.item())
    finally:
        torch.cuda.Event.synchronize = real_sync
    assert waits["n"] == 0 and losses2[0] == losses[0]
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(losses, losses2))


def _predict_worker(rank, world, port, out_dir, max_len, bs):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spokennlp_amd import data, inference
        from tests.test_oracle_golden import load_case, flags_of
        from tests.test_gpu_model import build_model
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        z, sd, _, arch = load_case("tiny_L128")
        docs = data.synth_docs(11, seed=77, vocab=arch["vocab_size"], mean_sents=30, sd_sents=10, mean_boundaries=4, mu_tok=1.8, sigma_tok=0.5)
        sent_ids = [[s.tolist() for s in d["sentences"]] for d in docs]
        labels = [[0 if v == 1 else 1 for v in d["labels"]] for d in docs]
        m = build_model(arch, flags_of(z, "full_eval"), sd, dev)
        got, metrics = inference.predict_documents(m, sent_ids, labels, max_len, arch["vocab_size"] - 1, data.CLS_ID, data.PAD_ID, batch_size=bs, device=dev)
        torch.save(dict(docs=got, metrics=metrics), os.path.join(out_dir, f"pred{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_predict_documents_three_ranks_equal_one_process(dev, tmp_path):
    """run_inference.sh:35 launches the predict branch on several ranks.  `inference.predict_documents` then shards the windows (rank r takes r, r + W, ...,
    the tail wrapped), and every rank gathers logits and cos-sim rows back in window order (dp.gather_sharded): documents and metrics on every rank are
    bit-identical to the single-process run -- 3 ranks (a window count that does not divide), sharing this box's GPU over gloo."""
    import socket
    import torch.multiprocessing as mp
    from spokennlp_amd import data, inference
    from tests.test_oracle_golden import load_case, flags_of
    from tests.test_gpu_model import build_model
    max_len, bs = 128, 2
    z, sd, _, arch = load_case("tiny_L128")
    docs = data.synth_docs(11, seed=77, vocab=arch["vocab_size"], mean_sents=30, sd_sents=10, mean_boundaries=4, mu_tok=1.8, sigma_tok=0.5)
    sent_ids = [[s.tolist() for s in d["sentences"]] for d in docs]
    labels = [[0 if v == 1 else 1 for v in d["labels"]] for d in docs]
    m = build_model(arch, flags_of(z, "full_eval"), sd, dev)
    ref_docs, ref_metrics = inference.predict_documents(m, sent_ids, labels, max_len, arch["vocab_size"] - 1, data.CLS_ID, data.PAD_ID, batch_size=bs, device=dev)
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    mp.spawn(_predict_worker, args=(3, port, str(tmp_path), max_len, bs), nprocs=3, join=True)
    for r in range(3):
        got = torch.load(tmp_path / f"pred{r}.pt", weights_only=False)
        assert len(got["docs"]) == len(ref_docs)
        for g, d in zip(got["docs"], ref_docs):
            assert g["predictions"] == d["predictions"] and g["int_labels"] == d["int_labels"]
            assert g["predict_logits"] == d["predict_logits"] and g["eop_pair_cos_sim"] == d["eop_pair_cos_sim"]
        assert got["metrics"] == ref_metrics
