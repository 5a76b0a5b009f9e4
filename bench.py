#!/usr/bin/env python
"""bench.py -- train throughput of the MI355X-native BERT topic-segmentation fine-tune step.

Metric (BASELINE.json): train sequences/s (512-token) for bert-base topic segmentation; a "step" = one optimiser step
over one batch of synthetic Wiki-727K-shaped windows: 2 encoder passes worth of sequences (anchor + augmented, the
reference's run_finetune.sh flags) forward + backward + grad-norm clip + AdamW, all inside the timed region, inputs
already resident in HBM.  N GPUs = pure data parallel (weak scaling: per-GPU batch fixed), RCCL all-reduce of the flat
gradient buffer overlapped with backward.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5        # starts the 8 ranks itself (torch.distributed.run, 127.0.0.1, a free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` (dominant kernel = gemm_nt MFMA projections,
timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores).
"""
import argparse
import json
import os
import random
import sys
import socket
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0       # gfx950 dense bf16 (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)


def flops_per_seq(L, H, I, layers, train=True, span=None, nproj=3):
    """algorithmic FLOPs: projections 2*(nproj*H^2 + H^2 + 2HI) + attention 2*(2*span*H) per token per layer
    (span = L keys for BERT, the 2w+1 band + 1 global key for Longformer, 0 for PoNet whose mixing is O(L) pooling)."""
    per_tok_layer = 2 * ((nproj + 1) * H * H + 2 * H * I) + 4 * (L if span is None else span) * H
    return per_tok_layer * layers * L * (3 if train else 1)


def build(args, device):
    from transformers import BertConfig
    if args.model == "ponet":           # BASELINE config 4: PoNet-base (12 x 768), chinese vocab 21128 + [EOS], L = 4096
        from spokennlp_amd.ponet import PoNetConfig, PoNetForTokenClassification
        cfg = PoNetConfig(vocab_size=21129, num_labels=2, max_position_embeddings=4096)
        torch.manual_seed(0)
        return PoNetForTokenClassification(cfg).to(device).train(), cfg
    if args.model == "longformer":      # BASELINE config 5: longformer-base-4096 (+[BOS]), window 512, CLS global
        from transformers import LongformerConfig
        from spokennlp_amd.longformer_for_ts import LongformerWithDAForSentenceLabelingTopicSegmentation as M
        cfg = LongformerConfig(vocab_size=50266, num_labels=2, max_position_embeddings=4098, type_vocab_size=1, pad_token_id=1,
                               attention_window=[512] * 12, layer_norm_eps=1e-5)
    elif args.model == "bigbird":       # google/bigbird-roberta-base (+[BOS]): 12 x 768, block 64, 3 random blocks, gelu_new
        from transformers import BigBirdConfig
        from spokennlp_amd.bigbird_for_ts import BigBirdWithDAForSentenceLabelingTopicSegmentation as M
        cfg = BigBirdConfig(vocab_size=50359, num_labels=2, max_position_embeddings=4096, attention_type="block_sparse",
                            block_size=64, num_random_blocks=3)
    else:
        from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
        cfg = BertConfig(vocab_size=30523, num_labels=2)          # bert-base-uncased + [BOS]
    flags = dict(do_da_ts=True, do_cssl=True, do_tssp=True, ts_loss_weight=1.0, cl_loss_weight=0.5, cl_temp=0.1,
                 cl_anchor_level="eop_list", cl_positive_k=1, cl_negative_k=3, tssp_loss_weight=1.0) if args.workload == "full_da" else {}
    for k, v in flags.items():
        setattr(cfg, k, v)
    cfg.amdseg_precision = getattr(args, "precision", "bf16")
    torch.manual_seed(0)
    m = M(cfg).to(device).train()
    return m, cfg


def make_ponet_batches(args, n, seed, device):
    """synthetic meeting-like documents through the restated PoNet feature builder (paragraph-level segment ids)"""
    import numpy as np
    from spokennlp_amd import preprocess as P
    rng = np.random.default_rng(1234 + seed)
    eos, cls, pad = 21128, 101, 0
    docs, labels = [], []
    for _ in range(max(8, args.seqs_per_gpu * n // 2)):
        ns = int(rng.integers(300, 600))
        docs.append([rng.integers(1000, 21000, int(np.clip(rng.lognormal(3.0, 0.5), 3, 100))).tolist() + [eos] for _ in range(ns)])
        lab = [(0 if rng.random() < 0.03 else (1 if rng.random() < 0.3 else -100)) for _ in range(ns)]     # paragraph ends carry labels
        lab[-1] = 0
        labels.append(lab)
    cols = P.ponet_prepare_features(docs, labels, list(range(len(docs))), args.seq_len, eos, cls, pad, use_paragraph_segment=True)
    keys = ("input_ids", "attention_mask", "token_type_ids", "segment_ids", "labels")
    nwin = len(cols["input_ids"])
    B = args.seqs_per_gpu
    out = []
    for i in range(n):
        idx = [(i * B + k) % nwin for k in range(B)]
        out.append({k: torch.tensor([cols[k][j] for j in idx], dtype=torch.long, device=device) for k in keys})
    return out, B


def make_batches(args, n, seed, device):
    if args.model == "ponet":
        return make_ponet_batches(args, n, seed, device)
    from spokennlp_amd import data
    pairs = args.seqs_per_gpu // 2 if args.workload == "full_da" else args.seqs_per_gpu
    if args.model in ("longformer", "bigbird"):
        docs = data.synth_docs(max(64, pairs * n * 3), seed=1234 + seed, vocab=50266, mean_sents=160, sd_sents=40)
    else:
        docs = data.synth_docs(max(64, pairs * n // 2), seed=1234 + seed)
    bs = data.batches_from_docs(docs, args.seq_len, pairs, seed=seed)
    while len(bs) < n:
        bs = bs + bs
    if args.model == "longformer":      # RoBERTa convention: pad id 1
        for b in bs:
            b["input_ids"] = torch.where(b["attention_mask"] == 0, torch.ones_like(b["input_ids"]), b["input_ids"])
    return [{k: v.to(device) for k, v in b.items()} for b in bs[:n]], pairs


PROF_CLASSES = (("gemm_nt_dp_kernel", 0), ("gemm_tn_dp_kernel", 1), ("attn_fwd_kernel", 2), ("attn_bwd_dq_kernel", 3), ("attn_bwd_dkv_kernel", 4))
# HBM-bound classes: the launch timer's `work` is algorithmic BYTES (include/amdseg.h AMDSEG_PROF_ADD_LN_FWD ..)
PROF_CLASSES_HBM = (("add_ln_fwd_kernel", 5), ("ln_bwd_kernel", 6), ("adamw_kernel", 7), ("attn_keepmask_kernel", 8))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s (6.3 TB/s measured copy)


class _V:                                                   # (the three numbers of one launch class, as the ctypes out-parameters used to hand them over)
    def __init__(self, v):
        self.value = v


def prof_read_all(nsteps, ctx):
    """every class of the launch timer of the engine's context (amdseg_ctx) after `nsteps` armed steps: MFMA classes in TFLOP/s against the bf16 peak,
    HBM classes in GB/s against 8 TB/s"""
    torch.cuda.synchronize()
    out = {}
    for table, hbm in ((PROF_CLASSES, False), (PROF_CLASSES_HBM, True)):
        for name, cls in table:
            us, work, n = (_V(x) for x in ctx.prof_read(cls))
            if not n.value:
                continue
            rec = dict(launches_per_step=round(n.value / nsteps, 2), avg_launch_us=round(us.value / n.value, 2), us_per_step=round(us.value / nsteps, 1))
            if hbm:
                gbs = work.value / us.value / 1e3
                rec.update(bound="hbm", mbytes_per_launch=round(work.value / n.value / 1e6, 2), achieved=round(gbs, 1), unit="GB/s",
                           frac=round(gbs / HBM_PEAK_GBS, 4))
            else:
                rec.update(gflop_per_launch=round(work.value / n.value / 1e9, 3), achieved=round(work.value / us.value / 1e6, 1),
                           frac=round(work.value / us.value / 1e6 / MFMA_PEAK_TFLOPS, 4))
            out[name] = rec
    return out


def _graphs_suspended(flag):
    """launches inside a replayed hipGraph (the engine's small-batch inference path) carry no events: the eager path runs while the launch timer is armed"""
    from spokennlp_amd import engine
    engine.GRAPHS_SUSPENDED = bool(flag)


def prof_arm(ctx):
    if ctx.prof_enable(1) < 0:
        return False
    ctx.prof_reset()
    _graphs_suspended(True)
    return True


def prof_collect(nsteps, ctx):
    out = prof_read_all(nsteps, ctx)
    ctx.prof_enable(0)
    _graphs_suspended(False)
    return out


def instep_roofline(step, first_step, nsteps, ctx):
    """per-kernel MFMA roofline measured INSIDE real training steps with HIP events: while armed, libamdseg launches the dominant
    kernels through hipExtLaunchKernelGGL with a start and a stop event, which are filled from the dispatch's own completion-signal
    timestamps (csrc/prof.h -- the span rocprofv3 --kernel-trace reports; a hipEventRecord pair AROUND a launch would add the ~5 us
    marker-to-marker gap).  achieved = sum of algorithmic FLOPs of the launches / sum of their spans, over `nsteps` extra steps."""
    if ctx.prof_enable(1) < 0:
        return None
    ctx.prof_reset()
    _graphs_suspended(True)
    for i in range(first_step, first_step + nsteps):
        step(i)
    out = prof_read_all(nsteps, ctx)
    ctx.prof_enable(0)
    _graphs_suspended(False)
    return out


def gemm_roofline(model, args, device):
    """(--standalone-gemm) every gemm_nt launch shape of one step timed back to back on warm operands with HIP events: the kernel's
    best case, NOT what the step sees (in-step numbers: instep_roofline)."""
    from spokennlp_amd import ops
    cfg = model.config
    H, I = cfg.hidden_size, cfg.intermediate_size
    M = args.seqs_per_gpu * args.seq_len
    shapes = [  # (N, K, epilogue, calls per layer per step)  forward 4 + dgrad 4
        (3 * H, H, ops.EPI_BIAS, 1), (H, H, ops.EPI_BIAS, 1), (I, H, ops.EPI_BIAS_GELU, 1), (H, I, ops.EPI_BIAS, 1),
        (I, H, ops.EPI_GELU_BWD, 1), (H, I, ops.EPI_ADD_RES, 1), (H, H, ops.EPI_NONE, 1), (H, 3 * H, ops.EPI_ADD_RES, 1)]
    tot_t, tot_f, launches = 0.0, 0.0, 0
    detail = []
    for N, K, epi, calls in shapes:
        A = torch.randn(M, K, device=device).bfloat16(); B = (torch.randn(N, K, device=device) * 0.05).bfloat16()
        bias = torch.randn(N, device=device); R = torch.randn(M, N, device=device).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=device); out2 = torch.empty_like(out)
        kw = dict(bias=bias if epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU) else None, R=R if epi in (ops.EPI_ADD_RES, ops.EPI_GELU_BWD) else None,
                  out=out, out2=out2 if epi == ops.EPI_BIAS_GELU else None)
        for _ in range(3):
            ops.gemm_nt(A, B, epi, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.gemm_nt(A, B, epi, **kw)
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        f = 2.0 * M * N * K
        detail.append(dict(N=N, K=K, epi=epi, us=round(t * 1e6, 1), tflops=round(f / t / 1e12, 1)))
        tot_t += t * calls; tot_f += f * calls; launches += calls
    return dict(achieved=round(tot_f / tot_t / 1e12, 1), frac=round(tot_f / tot_t / 1e12 / MFMA_PEAK_TFLOPS, 4),
                avg_launch_us=round(tot_t / launches * 1e6, 1), per_shape=detail)


def vendor_yardstick(args, device, H=768, I=3072):
    """the SAME eight projection / dgrad shapes of a layer as plain C = A B^T products (no epilogue work) through the vendor library
    (hipBLASLt behind torch.matmul) and through amdseg_gemm_nt (EPI_NONE), back to back on warm operands, HIP events on the launch
    stream.  A yardstick for `roofline.frac`, not a product path: what a hand-tuned library kernel reaches on these K = 768 .. 3072 shapes
    on this board (power / clock limited well below the nominal 2.5 PFLOP/s) -- and the weight-gradient shapes, where torch has no grouped
    launch (four separate TN products against one grouped amdseg_gemm_tn_grouped launch)."""
    from spokennlp_amd import ops
    M = args.seqs_per_gpu * args.seq_len

    def timeit(fn, reps=10, warm=3):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    shapes = [(3 * H, H), (H, H), (I, H), (H, I), (I, H), (H, I), (H, H), (H, 3 * H)]       # forward 4 + dgrad 4 (N, K)
    rows, tv, to, fl = [], 0.0, 0.0, 0.0
    seen = {}
    for N, K in shapes:
        if (N, K) not in seen:
            A = torch.randn(M, K, device=device).bfloat16(); B = (torch.randn(N, K, device=device) * 0.05).bfloat16()
            out = torch.empty(M, N, dtype=torch.bfloat16, device=device)
            t_v = timeit(lambda: torch.matmul(A, B.t(), out=out))
            t_o = timeit(lambda: ops.gemm_nt(A, B, ops.EPI_NONE, out=out))
            seen[(N, K)] = (t_v, t_o)
            rows.append(dict(N=N, K=K, hipblaslt_us=round(t_v * 1e6, 1), hipblaslt_tflops=round(2.0 * M * N * K / t_v / 1e12, 1),
                             amdseg_us=round(t_o * 1e6, 1), amdseg_tflops=round(2.0 * M * N * K / t_o / 1e12, 1)))
        t_v, t_o = seen[(N, K)]
        tv += t_v; to += t_o; fl += 2.0 * M * N * K
    wg = [(H, I), (I, H), (H, H), (3 * H, H)]
    As = [torch.randn(M, n, device=device).bfloat16() for n, _ in wg]
    Bs = [torch.randn(M, k, device=device).bfloat16() for _, k in wg]
    Cs = [torch.zeros(n, k, device=device) for n, k in wg]
    t_tn_o = timeit(lambda: ops.gemm_tn_grouped(As, Bs, Cs, accumulate=True), reps=6)
    t_tn_v = timeit(lambda: [torch.matmul(a.t(), b) for a, b in zip(As, Bs)], reps=6)
    fl_tn = sum(2.0 * M * n * k for n, k in wg)
    return dict(what="plain GEMMs (no epilogue) on the 8 NT shapes of a layer, M = %d, stand-alone back to back; hipblaslt = torch.matmul" % M,
                nt_layer_avg=dict(hipblaslt_tflops=round(fl / tv / 1e12, 1), hipblaslt_frac=round(fl / tv / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                  amdseg_tflops=round(fl / to / 1e12, 1), amdseg_frac=round(fl / to / 1e12 / MFMA_PEAK_TFLOPS, 4)),
                nt_shapes=rows,
                tn_layer=dict(hipblaslt_4_launches_us=round(t_tn_v * 1e6, 1), hipblaslt_tflops=round(fl_tn / t_tn_v / 1e12, 1),
                              amdseg_grouped_us=round(t_tn_o * 1e6, 1), amdseg_tflops=round(fl_tn / t_tn_o / 1e12, 1)))


def pool_roofline(model, args, device):
    """PoNet config 4: the segment/local max-pool + fusion kernel pair (csrc/ponet.hip) against the HBM roofline;
    algorithmic bytes = read Hs, Ho, Hl + write ctx = 4 * M * H * 2 B per launch."""
    from spokennlp_amd import ops
    eng = model.engine()
    B, Lq, H = args.seqs_per_gpu, args.seq_len, model.config.hidden_size
    A = eng._arena(B, Lq, True)
    la, pn = A["layers"][0], A["pn"]
    rs, re = eng._run
    g = torch.zeros(B, H, device=device)
    f = lambda: ops.ponet_pool_fwd(la["qkv"], A["mask_bias"], rs, re, eng._work, g, pn["part"][0], pn["parg"][0], la["ctx"], B, Lq, H)   # noqa: E731
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    by = 4.0 * B * Lq * H * 2
    return dict(bound="hbm", achieved=round(by / t / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(by / t / 8e12, 4), traffic=None,
                kernel="pn_zero_kernel + pn_segmax_kernel + pn_combine_fwd_kernel", avg_launch_us=round(t * 1e6, 1), algorithmic_bytes=by)


def dp_record(eng, world, device, sync_marks, step, first_step, args):
    """what the N > 1 line says about its own gradient exchange (every rank runs this: it holds collectives; rank 0 prints it):
    the backend and world size as the collective library sees them (an all-reduce of ones), the buckets of one step, the EXPOSED
    communication per step (compute-stream time between the end of backward and the start of clip + AdamW, i.e. the tail bucket's
    all-reduce plus whatever of the per-layer buckets had not finished under backward), and the same step with the word-embedding
    gradient -- 94 MB of that tail -- exchanged in bf16 (AMDSEG_DP_BF16_EMBED=1, off by default)."""
    import torch.distributed as dist
    from spokennlp_amd.dp import GradBuckets
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    # the devices the ranks actually sit on (two ranks on one GPU over gloo are not two GPUs)
    props = torch.cuda.get_device_properties(device)
    # (identity = host + whatever the runtime offers: uuid, PCI location, device index.  Ranks isolated by HIP_VISIBLE_DEVICES all see index 0 and some
    #  runtimes report an all-zero uuid, so the identity is a REPORT -- `distinct_gpus` -- and the refusal below rests on RCCL's own sum: RCCL does not
    #  build a communicator with two ranks on one device)
    ident = [str(getattr(props, n, "")) for n in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id")] + [str(torch.cuda.current_device())]
    me = f"{socket.gethostname()}:" + ":".join(ident)
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    distinct = len(set(everyone))
    if dist.get_backend() == "nccl" and float(ones.item()) != float(world):
        # the line may say n_gpus = N only when RCCL itself summed N ones
        raise SystemExit(f"bench.py --gpus {world}: the nccl (RCCL) all-reduce of ones returned {ones.item()} (ranks on {distinct} distinct GPU identities); "
                         f"refusing to report n_gpus = {world}")
    b = eng.buckets
    sizes = [(hi - lo) * 4 for lo, hi in b.layer_slices] + [(b.rest_slice[1] - b.emb_slice[1]) * 4, (b.emb_slice[1] - b.emb_slice[0]) * 4]
    exposed = sorted(e0.elapsed_time(e1) for e0, e1 in sync_marks)
    rec = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), allreduce_of_ones=float(ones.item()), distinct_gpus=distinct,
               transport="amdseg_allreduce_* (C ABI, csrc/comm.hip)" if b.native is not None else "torch.distributed.all_reduce",
               rccl_comm_ranks=(b.native.world if b.native is not None else None), wire=b.wire,
               buckets_per_step=len(sizes), bytes_per_step=int(sum(sizes)), tail_bucket_bytes=int(sizes[-1]),
               layer_bucket_bytes=int(sizes[0]), bucket_order="encoder layers last to first, then the embedding tables, all from inside backward (side stream); pooler + loss heads from finish_grad_sync",
               exposed_comm_ms_per_step=round(sum(exposed) / max(len(exposed), 1), 3),
               exposed_comm_ms_median=round(exposed[len(exposed) // 2], 3) if exposed else None,
               exposed_comm_method="HIP events on the compute stream around finish_grad_sync() (tail bucket issue + wait for every bucket), "
                                   "timed region average")
    if dist.get_backend() != "nccl":
        rec["note"] = (f"AMDSEG_DIST_BACKEND={dist.get_backend()}: {world} ranks on {distinct} GPU(s) -- a functional run of the N > 1 path, not a scaling measurement")
    # the schedule of one step as the buckets logged it (first element, end, wire dtype) and its bytes on the wire
    last = b.log or b.last_log
    rec["schedule"] = dict(buckets=len(last), bytes_on_wire=sum((e - a) * (2 if w == "bf16" else 4) for a, e, w in last),
                           order=[[a, e, w] for a, e, w in last[:3]] + (["..."] if len(last) > 3 else []))
    rec["nccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
    rec["backward_cu_budget"] = getattr(eng, "_bwd_cu_budget", 0) or None     # CUs the tile-width rule of backward counts on (amdseg_ctx_set_cu_budget)
    try:
        if dist.get_backend() == "nccl":
            rec["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                   # noqa: BLE001 -- version probing must never cost the bench line
        rec["rccl_version"] = f"unavailable ({type(e).__name__})"
    # the same steps with the word-embedding gradient in bf16 on the wire
    nextra = max(4, min(10, args.steps))
    old = eng.buckets
    eng.buckets = GradBuckets(eng.fp, bf16_embeddings=True)
    torch.cuda.synchronize(); dist.barrier()
    n0 = len(sync_marks)
    t0 = time.perf_counter()
    for i in range(first_step, first_step + nextra):
        step(i)
    torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ex = [e0.elapsed_time(e1) for e0, e1 in sync_marks[n0:]]
    ws = eng.buckets.word_slice or (0, 0)
    word = ws[1] - ws[0]
    rec["bf16_embed"] = dict(steps=nextra, ms_per_step=round(dt.item() / nextra * 1e3, 3), exposed_comm_ms_per_step=round(sum(ex) / len(ex), 3),
                             tail_bucket_bytes_on_wire=int(sizes[-1] - 2 * word))
    # ... and with EVERY bucket in bf16 on the wire (AMDSEG_DP_WIRE=bf16: 217.8 MB instead of 435.6 MB per bert-base step)
    eng.buckets = GradBuckets(eng.fp, wire="bf16")
    torch.cuda.synchronize(); dist.barrier()
    n0 = len(sync_marks)
    t0 = time.perf_counter()
    for i in range(first_step + nextra, first_step + 2 * nextra):
        step(i)
    torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ex = [e0.elapsed_time(e1) for e0, e1 in sync_marks[n0:]]
    rec["bf16_wire"] = dict(steps=nextra, ms_per_step=round(dt.item() / nextra * 1e3, 3), exposed_comm_ms_per_step=round(sum(ex) / len(ex), 3),
                            bytes_per_step=int(sum(sizes) // 2))
    eng.buckets = old
    return rec


def exposed_split(sync_split):
    """the exposed part of the exchange by bucket class: from the end of backward on the compute stream (e0) to the
    side-stream events behind (a) the last per-layer bucket, (b) the embeddings bucket -- issued by engine.backward right behind the embedding
    backward --, (c) the rest (pooler + loss heads, issued by finish_grad_sync).  Each figure = how long that class kept running after the
    previous one (or after e0) was done; per step, averaged over the timed region; this rank."""
    if not sync_split:
        return None
    acc = dict(layer_buckets=0.0, embeddings_bucket=0.0, rest_bucket=0.0)
    n = 0
    for e0, m in sync_split:
        if not all(k in m for k in ("layers_done", "embeddings_done", "rest_done")):
            continue
        t_l = max(0.0, e0.elapsed_time(m["layers_done"]))
        t_e = max(t_l, e0.elapsed_time(m["embeddings_done"]))
        t_r = max(t_e, e0.elapsed_time(m["rest_done"]))
        acc["layer_buckets"] += t_l; acc["embeddings_bucket"] += t_e - t_l; acc["rest_bucket"] += t_r - t_e
        n += 1
    if not n:
        return None
    out = {k: round(v / n, 3) for k, v in acc.items()}
    out["note"] = ("ms per step after the end of backward: layer buckets still running / then the embeddings bucket (issued from inside backward) / "
                   "then pooler + heads (issued by finish_grad_sync)")
    return out


def host_cpu():
    """(model string, physical cores, logical CPUs) of the box this runs on"""
    model, phys = "unknown CPU", None
    try:
        cores = set()
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown CPU":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, phys or max(1, logical // 2), logical


def cpu_baseline(args):
    """the CPU oracle (validated against the reference's golden vectors; it is the restatement that runs here, not the reference's
    files) timed on the host cores: bert-base shape, B = 8 sequences of 512 tokens, fp32 stock PyTorch CPU ops.  Two legs:
    forward-only (eval, no_grad) and the training step (forward + backward + clip + AdamW).  The thread count is SWEPT first (more threads than
    ~one NUMA domain make B = 8 x 512 slower, round 3: 128 threads 0.64 seq/s, 8 threads 1.42): forward-only over {8, 16, 32}
    (or up to the physical cores of a smaller box), the training step at the two best of them (1 warm-up + best of 3 each); then the BASELINE.md section 4 protocol (3 warm-up
    + 10 timed steps, median) at the best count gives `value`; `cores` = the threads that ran it.  A reported baseline, not the target."""
    from oracle import bert_ts_oracle as O
    from tests.util import tiny_state_dict
    from spokennlp_amd import data
    model, phys, logical = host_cpu()
    arch = dict(vocab_size=30523, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=512, type_vocab_size=2)
    flags = dict(do_da_ts=True, do_cssl=True, do_tssp=True, cl_loss_weight=0.5, cl_temp=0.1, cl_anchor_level="eop_list",
                 cl_positive_k=1, cl_negative_k=3, tssp_loss_weight=1.0) if args.workload == "full_da" else {}
    sd = tiny_state_dict(arch, seed=0, std=0.02)
    params = {k: torch.nn.Parameter(v) for k, v in sd.items()}
    cfg = O.make_cfg(num_labels=2, **arch, **flags)
    nseq = 8
    pairs = nseq // 2 if args.workload == "full_da" else nseq
    docs = data.synth_docs(32, seed=99)
    batch = data.batches_from_docs(docs, args.seq_len, pairs, seed=1)[0]
    opt = torch.optim.AdamW(list(params.values()), lr=5e-5)

    def train_step():
        opt.zero_grad(set_to_none=True)
        random.seed(0)
        loss, _, _ = O.model_forward(params, cfg, batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()

    def fwd_step():
        random.seed(0)
        with torch.no_grad():
            O.model_forward(params, cfg, batch)

    def sweep(fn):                       # thread sweep: 1 warm-up + best of 3
        fn()
        ts = []
        for _ in range(3):
            t = time.time(); fn(); ts.append(time.time() - t)
        return min(ts)

    def protocol(fn, warm=3, n=10):      # 3 warm-up + 10 timed steps; median, and the spread so the figure can be trusted
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t = time.time(); fn(); ts.append(time.time() - t)
        ts.sort()
        return 0.5 * (ts[n // 2 - 1] + ts[n // 2]) if n % 2 == 0 else ts[n // 2], ts[0], ts[-1]

    t_start = time.time()
    # "all physical cores".  The sweep runs {16, 32, 64, <physical cores>} (plus 8 on a small box) with 1 warm-up + 1 timed step
    # each -- enough to pick -- forward-only AND the training step at every count (the same rule for both legs), then the protocol (3 warm-up + 10 timed,
    # median) at each leg's best count.  (Rounds 3-5 on the 128-core EPYC 9575F pair: 16 threads 8.7-13.4 seq/s forward, 64 / 128 threads 1-2 seq/s --
    # B = 8 x 512 does not feed two sockets; the all-cores figure is in `thread_sweep` either way.)  A count whose single step would take the leg past
    # its time budget is still measured once (it IS the protocol's named configuration); the budget only stops further repetitions.
    counts = sorted({c for c in (16, 32, 64, 128) if c <= phys} | {phys} | ({8} if phys < 16 else set()))
    budget_s = float(os.environ.get("AMDSEG_CPU_BASELINE_BUDGET_S", "150"))

    def one(fn):
        fn()
        t = time.time(); fn()
        return time.time() - t

    fwd, train = {}, {}
    for c in counts:
        torch.set_num_threads(c)
        fwd[c] = round(nseq / one(fwd_step), 3)
        train[c] = round(nseq / one(train_step), 3)
    best = max(train, key=lambda c: train[c])
    best_f = max(fwd, key=lambda c: fwd[c])
    left = max(budget_s - (time.time() - t_start), 20.0)
    # protocol repetitions sized to what is left (never fewer than 3 timed steps): the training leg gets two thirds
    n_t = int(max(3, min(10, (left * 0.66) / (nseq / train[best]) - 3)))
    n_f = int(max(3, min(10, (left * 0.34) / (nseq / fwd[best_f]) - 3)))
    torch.set_num_threads(best)
    med, lo, hi = protocol(train_step, n=n_t)
    torch.set_num_threads(best_f)
    fmed, flo, fhi = protocol(fwd_step, n=n_f)
    return dict(value=round(nseq / med, 3), unit="seq/s", cores=best, kind="port", cpu=model, physical_cores=phys, logical_cpus=logical,
                spread=dict(fastest_step_seq_per_s=round(nseq / lo, 3), slowest_step_seq_per_s=round(nseq / hi, 3)),
                forward_only=dict(value=round(nseq / fmed, 3), unit="seq/s", cores=best_f,
                                  spread=dict(fastest_step_seq_per_s=round(nseq / flo, 3), slowest_step_seq_per_s=round(nseq / fhi, 3))),
                thread_sweep=dict(forward_only_seq_per_s={str(c): fwd[c] for c in counts}, train_seq_per_s={str(c): train[c] for c in train}),
                seconds=round(time.time() - t_start, 1),
                protocol_steps=dict(train=dict(warmup=3, timed=n_t), forward_only=dict(warmup=3, timed=n_f)),
                sample=f"fp32 torch CPU oracle (= the restatement validated against the reference's golden vectors, not the reference's files), "
                       f"bert-base shape, {nseq} x {args.seq_len}-token sequences per step, {args.workload}.  Thread count swept first over {counts} threads "
                       f"(= up to ALL {phys} physical cores; forward-only and the training step at every count, 1 warm-up + 1 timed step each), then the "
                       f"BASELINE.md section 4 protocol at each leg's best count: 3 warm-up + {n_t} (train) / {n_f} (forward) timed steps, value = {nseq} / median "
                       f"step time (training step = fwd+bwd+clip+AdamW at {best} threads; forward-only = eval, no_grad at {best_f} threads); {model}, "
                       f"{phys} physical cores / {logical} logical CPUs")


def run_leg(args, device, mode, precision, steps, warmup, prof_steps, seed=7):
    """one secondary leg on rank 0 (world 1): the same workload at another (mode, precision), its own model and batches, wall clock
    around `steps` steps bracketed by synchronize, then `prof_steps` more with the launch timer armed (in-step roofline)."""
    import copy
    a = copy.copy(args)
    a.mode, a.precision = mode, precision
    model, cfg = build(a, device)
    eng = model.engine()
    batches, pairs = make_batches(a, 8, seed=seed, device=device)
    if mode == "infer":
        model.eval()
    total = steps + warmup + prof_steps

    def step(i):
        random.seed(i)
        if mode == "infer":
            with torch.no_grad():
                return model(**batches[i % len(batches)])[0]
        loss = model(**batches[i % len(batches)])[0]
        loss.backward()
        eng.finish_grad_sync()
        eng.adamw_step(5e-5 * max(0.0, (total - i) / total), max_grad_norm=1.0, grad_scale=1.0)
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = a.seqs_per_gpu * steps / dt
    fl = flops_per_seq(a.seq_len, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, mode == "train")
    out = dict(value=round(value, 1), unit="seq/s", mode=mode, precision=precision, steps=steps, warmup=warmup,
               ms_per_step=round(dt / steps * 1e3, 3), seqs_per_step=a.seqs_per_gpu, seq_len=a.seq_len,
               mfma_frac_whole_step=round(value * fl / (MFMA_PEAK_TFLOPS * 1e12), 4), final_loss=round(float(loss.detach()), 4))
    prof = instep_roofline(step, warmup + steps, prof_steps, eng.ctx) if prof_steps else None
    if prof:
        dom = "gemm_nt_dp_kernel" if "gemm_nt_dp_kernel" in prof else max(prof, key=lambda k: prof[k]["us_per_step"])
        d = prof[dom]
        hbm_dom = d.get("bound") == "hbm"
        rl = dict(bound="hbm" if hbm_dom else "mfma", kernel=dom, achieved=d["achieved"], peak=HBM_PEAK_GBS if hbm_dom else MFMA_PEAK_TFLOPS,
                  unit="GB/s" if hbm_dom else "TFLOP/s", frac=d["frac"],
                  avg_launch_us=d["avg_launch_us"], launches_per_step=d["launches_per_step"], gflop_per_launch=d.get("gflop_per_launch"),
                  kernels=prof)
        if precision == "parity":
            rl["note"] = ("flops are the EXECUTED split-bf16 products (K' = 3K: hi*hi + hi*lo + lo*hi); on the reference's fp32 flops the "
                          "fraction is frac / 3")
            rl["frac_reference_flops"] = round(d["frac"] / 3.0, 4)
        out["roofline"] = rl
    del model, eng, batches
    torch.cuda.empty_cache()
    return out


def pmc_traffic(args):
    """(profiles/pmc_traffic.json, stale) -- the file if it describes this workload, else None; stale = the kernel sources it was measured on
    (csrc_sha, spokennlp_amd.build.sources_sha) are not the sources shipped now: its figures are then NOT attached to the line"""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None, False
    w = tr.get("workload", {})
    mine = dict(model=args.model, mode=args.mode, seq_len=args.seq_len, seqs_per_gpu=args.seqs_per_gpu, workload=args.workload,
                precision=getattr(args, "precision", "bf16"))
    if not all(w.get(k) == v for k, v in mine.items()):
        return None, False
    from spokennlp_amd.build import sources_sha
    return tr, tr.get("csrc_sha") != sources_sha()


def parity_report(device):
    """SURVEY 8(d) "Parity report": the HIP path in the fast (bf16) and in the tolerance-meeting ("parity") precision against the
    reference's stored outputs; measured here so the driver-run line carries it (tests/parity_values.py; outside every timed region)."""
    from tests import parity_values as PV
    return PV.rounded(PV.measure(device))


def via_trainer(args, device, nsteps=30, nwarm=8, nan_filter=False):
    """the same workload driven by the reference's training surface: `spokennlp_amd.trainer.Trainer` (the transformers.Trainer subclass
    of the one-line import swap) with its default collator / dataloader (host batches -> device every step), linear lr schedule, clip
    1.0, fused AdamW.  Timed between step `nwarm` and the end with a callback; single process."""
    from transformers import TrainerCallback, TrainingArguments, default_data_collator
    from spokennlp_amd.trainer import Trainer
    import tempfile
    model, cfg = build(args, device)
    batches, pairs = make_batches(args, 8, seed=123, device=torch.device("cpu"))
    samples = [{k: v[i] for k, v in b.items()} for b in batches for i in range(pairs)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return pairs * (nsteps + nwarm)

        def __getitem__(self, i):
            return samples[i % len(samples)]

    mark = {}

    class Clock(TrainerCallback):
        def on_step_begin(self, a, state, control, **kw):
            if state.global_step == nwarm:
                torch.cuda.synchronize(); mark["t0"] = time.perf_counter()

        def on_train_end(self, a, state, control, **kw):
            torch.cuda.synchronize(); mark["t1"] = time.perf_counter()

    with tempfile.TemporaryDirectory() as tmp:
        targs = TrainingArguments(output_dir=tmp, per_device_train_batch_size=pairs, max_steps=nsteps + nwarm, learning_rate=5e-5,
                                  lr_scheduler_type="linear", max_grad_norm=1.0, report_to=[], save_strategy="no", logging_strategy="no",
                                  seed=0, dataloader_drop_last=True, dataloader_num_workers=2, dataloader_pin_memory=True,
                                  disable_tqdm=True, logging_nan_inf_filter=nan_filter)
        tr = Trainer(model=model, args=targs, train_dataset=DS(), data_collator=default_data_collator, callbacks=[Clock()])
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):       # the Trainer prints its own summary dict: stdout carries the ONE JSON line only
            tr.train()
    dt = mark["t1"] - mark["t0"]
    return dict(value=round(args.seqs_per_gpu * nsteps / dt, 1), unit="seq/s", ms_per_step=round(dt / nsteps * 1e3, 3), steps=nsteps,
                logging_nan_inf_filter=nan_filter,
                surface="spokennlp_amd.trainer.Trainer(transformers.Trainer): default collator + dataloader, fused AdamW, HIP grad norm; "
                        "logging_nan_inf_filter=True (the TrainingArguments default) is evaluated on the device by the subclass (the stock loop "
                        "reads the loss on the host every step)")


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks here -- the reference's own launch shape,
    `python -m torch.distributed.launch --nproc_per_node N` (run_finetune.sh:1-2,61; run_inference.sh:35) -- one process per GPU over RCCL,
    rendezvous on 127.0.0.1 and a free port, and pass rank 0's ONE JSON line through.  Fewer visible GPUs than ranks is refused loudly
    unless AMDSEG_DIST_BACKEND=gloo says the ranks are to share the GPUs that are there (tests, one-GPU boxes)."""
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and os.environ.get("AMDSEG_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible -- one rank per GPU over RCCL needs {n}; set AMDSEG_DIST_BACKEND=gloo "
                         f"to run the {n} ranks on the GPU(s) present (a functional check, not a scaling measurement)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")                     # run_finetune.sh:7
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="bert", choices=["bert", "longformer", "ponet", "bigbird"])
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--seqs-per-gpu", type=int, default=None)
    ap.add_argument("--workload", default="full_da", choices=["full_da", "plain"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"], help="infer = forward-only (eval, no_grad) sequences/s")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "parity", "fp32"],
                    help="bf16 = the fast path (default); parity = fp32 activations + split-bf16 products (forward and backward, "
                         "reference-grade numerics); fp32 = exact fp32 MFMA (inference only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--standalone-gemm", action="store_true", help="also time the gemm_nt shapes back to back on warm operands")
    ap.add_argument("--via-trainer", action="store_true", help="force the transformers.Trainer leg (default: on for bert, 1 GPU, train)")
    ap.add_argument("--no-via-trainer", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip parity_report / parity_leg / infer_leg (default: on for the default workload: bert, 1 GPU, train, bf16)")
    ap.add_argument("--prof-steps", type=int, default=10, help="extra steps with the launch timer armed (roofline) when it is not armed "
                                                               "over the timed region itself")
    ap.add_argument("--prof-in-timed", action="store_true", help="arm the launch timer over the timed region itself instead of over "
                                                                 "--prof-steps extra steps after it (measured: the event-carrying launches cost "
                                                                 "5 % of the step, 15.3 vs 14.5 ms, so `value` would be perturbed; the per-kernel "
                                                                 "averages are the same either way, 65.1 vs 65.4 us)")
    args = ap.parse_args()
    if args.seq_len is None:
        args.seq_len = 512 if args.model == "bert" else 4096
    if args.seqs_per_gpu is None:
        args.seqs_per_gpu = 32 if args.model == "bert" else 8
    if args.model != "bert":
        args.no_cpu_baseline = True       # the cpu_baseline leg times the BERT oracle (headline metric) only
    if args.precision != "bf16":
        args.no_via_trainer = True

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:                    # not under torchrun: become the launcher of the N ranks
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; they must agree "
                         f"(n_gpus in the JSON line is the number of ranks that ran)")

    from spokennlp_amd import dp
    rank, world, local = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    model, cfg = build(args, device)
    eng = model.engine()
    eng.enable_data_parallel()
    batches, pairs = make_batches(args, 8, seed=rank, device=device)
    total_steps = args.steps + args.warmup
    lr0 = 5e-5

    if args.mode == "infer":
        model.eval()
        args.no_cpu_baseline = True

    # world > 1: the exposed part of the gradient exchange = what the compute stream spends between the end of backward and the start of
    # clip + AdamW (the tail bucket -- embeddings + heads, produced last -- is issued there, then every outstanding bucket is waited for)
    sync_marks = []
    sync_split = []
    if world > 1 and eng.buckets is not None:
        eng.buckets.timing = True               # side-stream events behind the last layer bucket / the embeddings bucket / the rest

    def step(i):
        random.seed(i)
        if args.mode == "infer":
            with torch.no_grad():
                return model(**batches[i % len(batches)])[0]
        loss = model(**batches[i % len(batches)])[0]
        loss.backward()
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.finish_grad_sync()
            e1.record()
            sync_marks.append((e0, e1))
            if eng.buckets is not None and eng.buckets.timing:
                sync_split.append((e0, dict(eng.buckets.marks)))
        else:
            eng.finish_grad_sync()
        lr = lr0 * max(0.0, (total_steps - i) / total_steps)        # linear decay, no warm-up (run_finetune.sh:73)
        eng.adamw_step(lr, max_grad_norm=1.0, grad_scale=1.0 / world)
        return loss

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    sync_marks.clear()
    sync_split.clear()
    prof_timed = (not args.no_roofline) and args.prof_in_timed and prof_arm(eng.ctx)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.warmup, total_steps):
        loss = step(i)
        marks[i - args.warmup + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    seqs = args.seqs_per_gpu * world * args.steps
    value = seqs / dt
    fl = flops_per_seq(args.seq_len, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, args.mode == "train",
                       span={"bert": None, "longformer": 514, "ponet": 0, "bigbird": 8 * 64}[args.model], nproj=5 if args.model == "ponet" else 3)
    name = {"bert": "bert-base-uncased(+[BOS])", "longformer": "longformer-base-4096(+[BOS], window 512, CLS global)",
            "ponet": "PoNet-base(+[EOS], paragraph segment ids)",
            "bigbird": "bigbird-roberta-base(+[BOS], block-sparse: block 64, 3 random blocks)"}[args.model]
    out = dict(metric=f"{args.mode} seq/s ({args.seq_len}-tok) " + {"bert": "bert-base", "longformer": "longformer-base", "ponet": "PoNet-base",
                                                                 "bigbird": "bigbird-base"}[args.model] + " topic-seg",
               value=round(value, 2), unit="seq/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
               scaling="weak", vs_baseline=None,
               dtype={"bf16": "bf16", "parity": "f32 (activations fp32, products as split-bf16 MFMA, fp32 accumulate)", "fp32": "f32"}[args.precision],
               data="synthetic",
               config=dict(workload=f"{name} topic-seg fine-tune, {args.workload}, seq_len={args.seq_len}, "
                                    f"{args.seqs_per_gpu} seqs/GPU/step ({pairs} samples), "
                                    + ("fwd+bwd+clip+AdamW, dropout 0.1" if args.mode == "train" else "inference forward only (eval, no_grad)"),
                           global_batch=args.seqs_per_gpu * world, seq_len=args.seq_len, parallelism=f"dp{world}"),
               mfma_frac_whole_step=round(value / world * fl / (MFMA_PEAK_TFLOPS * 1e12), 4),
               final_loss=round(float(loss.detach()), 4))
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    out["ms_per_step_median"] = round(per_step[len(per_step) // 2], 3)
    if world > 1 and args.mode == "train":
        split = exposed_split(sync_split)
        out["dp"] = dp_record(eng, world, device, sync_marks, step, total_steps, args)
        if split:
            out["dp"]["exposed_comm_split_ms"] = split
    prof = None
    prof_n = args.steps
    if prof_timed:                                          # start / stop events of every launch of the timed region itself
        prof = prof_collect(args.steps, eng.ctx)
    elif not args.no_roofline:                              # every rank runs the extra steps (the exchange is collective); rank 0 reports
        prof = instep_roofline(step, total_steps, args.prof_steps, eng.ctx)
        prof_n = args.prof_steps
    if rank == 0:
        if prof and "gemm_tn_dp_kernel" in prof and args.model in ("bert", "longformer") and args.precision == "bf16":
            # the weight-gradient GEMM's flops above are the reference's arithmetic (every token row); the rows of trailing padding are exact
            # zeros in dY and their 64-token tiles are not multiplied (amdseg_bert_cfg.pad_runs): say how much of the work was executed
            eng = (model.module if hasattr(model, "module") else model).engine()
            if getattr(eng, "skip_padded_rows_bwd", False):
                am = torch.cat([b["attention_mask"].reshape(-1, args.seq_len) for b in batches]).cpu()
                kend = ((am != 0).long() * torch.arange(1, args.seq_len + 1)[None, :]).amax(dim=1)
                f = float(((kend + 63) // 64).sum()) / (am.shape[0] * (args.seq_len // 64))
                # ... and the input-gradient half of the NT launches (du, dx1, dctx, dx_in: `ZPAD` in csrc/api.hip) skips every 256-row tile that starts
                # at or past its sequence's kend (DP_ZTILE, csrc/gemm_dp.hip); the forward half walks every tile.  Per layer the forward and the
                # input-gradient GEMMs carry the same flops, so the NT kernel executed 1 - skipped / 2 of what it is credited with
                if args.seq_len % 256 == 0 and "gemm_nt_dp_kernel" in prof:
                    tps = args.seq_len // 256
                    walked = torch.clamp((kend + 255) // 256, max=tps).sum().item()
                    f_dgrad = float(walked) / (am.shape[0] * tps)
                    n_ = prof["gemm_nt_dp_kernel"]
                    n_["dgrad_row_tiles_walked_frac"] = round(f_dgrad, 4)
                    n_["nt_tiles_walked_frac"] = round(0.5 + 0.5 * f_dgrad, 4)
                    n_["achieved_executed"] = round(n_["achieved"] * n_["nt_tiles_walked_frac"], 1)
                    n_["frac_executed"] = round(n_["frac"] * n_["nt_tiles_walked_frac"], 4)
                    n_["note"] = ("achieved / frac count the reference's flops; the input-gradient launches (half of this kernel's flops) skip the 256-row tiles "
                                  "made of trailing padding only (exact-zero rows) -- *_executed count only the tiles that were multiplied")
                t = prof["gemm_tn_dp_kernel"]
                t["token_tiles_walked_frac"] = round(f, 4)
                t["achieved_executed"] = round(t["achieved"] * f, 1)
                t["frac_executed"] = round(t["frac"] * f, 4)
                t["note"] = ("achieved / frac count the reference's flops (all token rows); the 64-token tiles of trailing padding hold exact-zero "
                             "dY rows and are skipped -- *_executed count only the tiles that were multiplied")
        if prof:
            dom = max(prof, key=lambda k: prof[k]["us_per_step"]) if "gemm_nt_dp_kernel" not in prof else "gemm_nt_dp_kernel"
            d = prof[dom]
            hbm_dom = d.get("bound") == "hbm"
            out["roofline"] = dict(bound="hbm" if hbm_dom else "mfma", achieved=d["achieved"], peak=HBM_PEAK_GBS if hbm_dom else MFMA_PEAK_TFLOPS,
                                   unit="GB/s" if hbm_dom else "TFLOP/s", frac=d["frac"], traffic=None,
                                   kernel=dom, avg_launch_us=d["avg_launch_us"], launches_per_step=d["launches_per_step"],
                                   gflop_per_launch=d.get("gflop_per_launch"),
                                   method=f"in-step: HIP start/stop events of every launch (hipExtLaunchKernelGGL, csrc/prof.h) over "
                                          + (f"the {prof_n} steps of the timed region" if prof_timed else f"{prof_n} real steps right after the "
                                             f"timed region") + "; HBM traffic needs separate rocprofv3 --pmc passes: see profiles/",
                                   kernels=prof)
            if "frac_executed" in d:                        # the dominant kernel on executed flops only (its skipped all-padding tiles not credited)
                out["roofline"]["frac_executed"] = d["frac_executed"]
                out["roofline"]["achieved_executed"] = d["achieved_executed"]
                out["roofline"]["tiles_walked_frac"] = d.get("nt_tiles_walked_frac", d.get("token_tiles_walked_frac"))
            # BASELINE.json's target is stated on "the encoder GEMMs": projection + input-gradient GEMMs (NT) and the grouped weight-gradient GEMM (TN)
            # together, the reference's flops over the time both kernels take in a step (and the same on the flops actually executed)
            nt_, tn_ = prof.get("gemm_nt_dp_kernel"), prof.get("gemm_tn_dp_kernel")
            if nt_ and tn_ and nt_.get("gflop_per_launch") and tn_.get("gflop_per_launch"):
                gf = nt_["gflop_per_launch"] * nt_["launches_per_step"] + tn_["gflop_per_launch"] * tn_["launches_per_step"]
                us = nt_["us_per_step"] + tn_["us_per_step"]
                gfx = (nt_["gflop_per_launch"] * nt_["launches_per_step"] * nt_.get("nt_tiles_walked_frac", 1.0)
                       + tn_["gflop_per_launch"] * tn_["launches_per_step"] * tn_.get("token_tiles_walked_frac", 1.0))
                out["roofline"]["encoder_gemms"] = dict(gflop_per_step=round(gf, 1), us_per_step=round(us, 1), achieved=round(gf / us * 1e3, 1),
                                                        frac=round(gf / us * 1e3 / MFMA_PEAK_TFLOPS, 4),
                                                        frac_executed=round(gfx / us * 1e3 / MFMA_PEAK_TFLOPS, 4), unit="TFLOP/s",
                                                        nt_tiles_walked_frac=nt_.get("nt_tiles_walked_frac", 1.0),
                                                        tn_token_tiles_walked_frac=tn_.get("token_tiles_walked_frac", 1.0),
                                                        note="gemm_nt_dp_kernel + gemm_tn_dp_kernel of one step together, reference flops / their time; "
                                                             "frac_executed (only the tiles that were multiplied) is the figure to hold against BASELINE.json's "
                                                             "40 % target, frac (the reference's flops, padding tiles credited) is the note")
                # the headline fraction for the target "≥ 40 % MFMA peak on the encoder GEMMs": executed flops only
                out["roofline"]["encoder_gemms_frac_executed"] = out["roofline"]["encoder_gemms"]["frac_executed"]
            # HBM bytes per launch: PMC counters cannot be read from inside the process, so these are the figures of the committed separate
            # rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by tools/pmc_to_json.py) -- attached only when the file's workload
            # is THIS workload, with the commit the passes ran on; null otherwise
            tr, stale = pmc_traffic(args)
            if tr and stale:
                out["roofline"]["traffic_stale"] = True
                out["roofline"]["traffic_note"] = (f"profiles/pmc_traffic.json was measured on kernel sources {tr.get('csrc_sha')} @ {tr.get('git')}; the shipped "
                                                   "sources hash differently, so its figures are not attached (re-run tools/run_pmc_instep.sh + tools/pmc_to_json.py)")
            elif tr and dom in tr["kernels"]:
                out["roofline"]["traffic"] = tr["kernels"][dom]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "B per launch"
                out["roofline"]["traffic_algorithmic"] = tr["kernels"][dom].get("algorithmic_bytes_per_launch")
                out["roofline"]["traffic_source"] = (f"{tr['source']} @ {tr['git']}: separate rocprofv3 --pmc passes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 "
                                                     f"correction of MI355X_MICROARCH.md) over `{tr['command']}`; read from profiles/pmc_traffic.json, "
                                                     "not re-measured by this run")
                out["roofline"]["traffic_all_kernels"] = tr["kernels"]
            if args.standalone_gemm:
                out["roofline"]["standalone"] = gemm_roofline(model, args, device)
        if not args.no_roofline and args.model == "ponet":
            out["pool_roofline"] = pool_roofline(model, args, device)
        want_tr = args.via_trainer or (args.model == "bert" and world == 1 and args.mode == "train" and not args.no_via_trainer)
        if want_tr and world == 1:
            try:
                out["via_trainer"] = via_trainer(args, device)
                out["via_trainer"]["default_args"] = {k: v for k, v in via_trainer(args, device, nan_filter=True).items()
                                                      if k in ("value", "ms_per_step", "logging_nan_inf_filter")}
            except Exception as e:                         # the contract line must still be printed
                out["via_trainer"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        default_workload = (args.model == "bert" and world == 1 and args.mode == "train" and args.precision == "bf16"
                            and args.seq_len == 512 and args.seqs_per_gpu == 32)
        if default_workload and not args.no_extra_legs:
            # the legs the headline does not cover, each bounded to seconds, all outside the timed region above
            def overlap_leg():
                # the same workload with the weight-gradient GEMM of layer i on a second stream under layer i - 1's backward (AMDSEG_OVERLAP_WGRAD=1): opt-in
                # because per-kernel spans then describe co-running kernels (profiles/r05_default_switches.md) -- its throughput is reported here
                old = os.environ.get("AMDSEG_OVERLAP_WGRAD")
                os.environ["AMDSEG_OVERLAP_WGRAD"] = "1"
                try:
                    r = run_leg(args, device, "train", "bf16", 40, 10, 0)
                finally:
                    if old is None:
                        os.environ.pop("AMDSEG_OVERLAP_WGRAD", None)
                    else:
                        os.environ["AMDSEG_OVERLAP_WGRAD"] = old
                r["note"] = "AMDSEG_OVERLAP_WGRAD=1 (opt-in); the headline `value` is measured without it"
                return r

            for key, fn in (("vendor_yardstick", lambda: vendor_yardstick(args, device)),
                            ("overlap_wgrad_leg", overlap_leg),
                            ("parity_report", lambda: parity_report(device)),
                            ("parity_leg", lambda: run_leg(args, device, "train", "parity", 10, 3, 4)),
                            ("infer_leg", lambda: dict(bf16=run_leg(args, device, "infer", "bf16", 40, 8, 6),
                                                       parity=run_leg(args, device, "infer", "parity", 20, 4, 4)))):
                try:
                    out[key] = fn()
                except Exception as e:                     # the contract line must still be printed
                    out[key] = dict(error=f"{type(e).__name__}: {e}"[:300])
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():     # (also the group accelerate opens for the Trainer leg under torchrun)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
