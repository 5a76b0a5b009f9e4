"""Data-parallel gradient exchange, world_size 2 on CPU with the gloo backend (the N>1 path of bench.py uses the same
GradBuckets over RCCL).  Checks: bucket slices tile the flat buffer exactly, per-layer async all-reduce + rest + wait
gives the elementwise SUM on every rank, and sample sharding is a partition."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from transformers import BertConfig
    from spokennlp_amd import dp
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    from spokennlp_amd.engine import FlatParams
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=50, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, num_labels=2)
    fp = FlatParams(M(cfg), torch.device("cpu"))
    b = dp.GradBuckets(fp)
    # slices tile [0, numel) without gaps or overlap
    sl = sorted([b.rest_slice] + b.layer_slices)
    assert sl[0][0] == 0 and sl[-1][1] == fp.numel and all(sl[i][1] == sl[i + 1][0] for i in range(len(sl) - 1))
    g = torch.Generator().manual_seed(100 + rank)
    fp.flat_g.copy_(torch.randn(fp.numel, generator=g))
    mine = fp.flat_g.clone()
    for li in reversed(range(fp.nlayers)):
        b.reduce_layer(li)
    # the embeddings part of the rest goes first (engine.backward issues it right behind the embedding backward), the remainder (pooler, loss
    # heads) with finish_grad_sync: together they must cover the rest exactly once
    emb_names = [n for n in fp.offsets if ".embeddings." in "." + n]
    assert b.emb_slice[0] == 0
    assert b.emb_slice[1] <= b.rest_slice[1] and b.emb_slice[1] >= fp.offsets[emb_names[-1]] + fp.params[emb_names[-1]].numel()
    if rank == 0:
        b.reduce_embeddings()
        b.reduce_embeddings()              # idempotent within a step
    else:
        b.reduce_embeddings()
    b.reduce_rest()
    b.wait()
    expect = sum(torch.randn(fp.numel, generator=torch.Generator().manual_seed(100 + k)) for k in range(world))
    ok = torch.allclose(fp.flat_g, expect, atol=1e-6) and not torch.equal(fp.flat_g, mine)
    # the schedule of the step as logged: layers last to first, the embeddings part, the remainder of the rest -- fp32 on the wire, every element once
    sched = list(b.log)
    want = [(a, e, "fp32") for a, e in reversed(b.layer_slices)] + [(b.emb_slice[0], b.emb_slice[1], "fp32"), (b.emb_slice[1], b.rest_slice[1], "fp32")]
    ok = ok and sched == want and b.bytes_on_wire() == 4 * fp.numel
    # wire = "bf16" (AMDSEG_DP_WIRE=bf16): the SAME schedule, every bucket cast / summed / cast back -- half the bytes; the sums carry bf16 rounding
    b2 = dp.GradBuckets(fp, wire="bf16")
    fp.flat_g.copy_(torch.randn(fp.numel, generator=torch.Generator().manual_seed(100 + rank)))
    for li in reversed(range(fp.nlayers)):
        b2.reduce_layer(li)
    b2.reduce_embeddings(); b2.reduce_rest(); b2.wait()
    ok = ok and [(a, e) for a, e, _ in b2.log] == [(a, e) for a, e, _ in want] and all(w == "bf16" for _, _, w in b2.log)
    ok = ok and b2.bytes_on_wire() == 2 * fp.numel
    want_bf = sum(torch.randn(fp.numel, generator=torch.Generator().manual_seed(100 + k)).bfloat16() for k in range(world)).float()
    ok = ok and torch.allclose(fp.flat_g, want_bf, atol=2e-2) and (fp.flat_g - expect).abs().max().item() < 0.05
    # ... and the embeddings-only form touches exactly the word-embedding table
    b3 = dp.GradBuckets(fp, wire="bf16_embed")
    for li in reversed(range(fp.nlayers)):
        b3.reduce_layer(li)
    b3.reduce_embeddings(); b3.reduce_rest(); b3.wait()
    bf = [(a, e) for a, e, w in b3.log if w == "bf16"]
    ok = ok and bf == [b3.word_slice] and sum(e - a for a, e, _ in b3.log) == fp.numel
    try:
        dp.GradBuckets(fp, native=True)                # the C-ABI transport is RCCL on device buffers: refused loudly on a CPU buffer
        ok = False
    except RuntimeError:
        pass
    # parameter .grad views see the reduced values
    n = "bert.encoder.layer.2.output.dense.weight"
    ok = ok and torch.equal(fp.params[n].grad.flatten(), fp.flat_g[fp.offsets[n]:fp.offsets[n] + fp.params[n].numel()])
    idx = dp.shard_indices(11, rank, world)
    q.put((rank, bool(ok), idx))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_allreduce_gloo_world2():
    world = 2
    port = 29500 + (os.getpid() % 500)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    idx = sorted(i for _, _, ix in res for i in ix)
    assert set(idx) == set(range(11)) and len(idx) == 12       # DistributedSampler-style partition with one wrapped pad


def _worker8(rank, world, port, q):
    """world 8: the bucket walk on a Longformer layout (global projections live in the `rest` part but carry the encoder prefix -- the r04
    advisor finding: prefix-derived slices dropped layer N-1's LayerNorm) with a layer count that does not divide the name count evenly,
    and the predict shard / gather on uneven tails and on fewer items than ranks"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from transformers import LongformerConfig
    from spokennlp_amd import dp
    from spokennlp_amd.longformer_for_ts import LongformerWithDAForSentenceLabelingTopicSegmentation as M
    from spokennlp_amd.engine import FlatParams, LAYER_ORDER
    dp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    cfg = LongformerConfig(vocab_size=60, hidden_size=64, num_hidden_layers=5, num_attention_heads=1, intermediate_size=128, num_labels=2,
                           max_position_embeddings=70, type_vocab_size=1, pad_token_id=1, attention_window=[8] * 5)
    fp = FlatParams(M(cfg), torch.device("cpu"), encoder_prefix="longformer.encoder.layer.", layer_order=LAYER_ORDER)
    b = dp.GradBuckets(fp)
    ok = fp.nlayers == 5 and any("query_global" in n for n in list(fp.offsets)[:fp.n_rest])
    # every layer's slice is exactly that layer's own names, the last one ends at numel (its output LayerNorm is inside)
    for li, (a, e) in enumerate(b.layer_slices):
        first, last = fp.lp(li, LAYER_ORDER[0]), fp.lp(li, LAYER_ORDER[-1])
        ok = ok and a == fp.offsets[first] and e >= fp.offsets[last] + fp.params[last].numel()
    ok = ok and b.layer_slices[-1][1] == fp.numel and b.rest_slice == (0, b.layer_slices[0][0])
    g = torch.Generator().manual_seed(100 + rank)
    fp.flat_g.copy_(torch.randn(fp.numel, generator=g))
    for li in reversed(range(fp.nlayers)):
        b.reduce_layer(li)
    b.reduce_embeddings()
    b.reduce_rest()
    b.wait()
    expect = sum(torch.randn(fp.numel, generator=torch.Generator().manual_seed(100 + k)) for k in range(world))
    ok = ok and torch.allclose(fp.flat_g, expect, atol=1e-5)
    # predict: shard -> "compute" -> gather, for 8 ranks x 3 samples, an uneven tail, and more ranks than samples
    shards = {}
    for n_items in (24, 27, 29, 8, 5, 1):
        idx = dp.shard_indices(n_items, rank, world)
        rows = torch.tensor([[float(i), 10.0 * i] for i in idx])          # the row of item i is a function of i only
        got = dp.gather_sharded(rows, n_items)
        ok = ok and got.shape == (n_items, 2) and torch.equal(got[:, 0], torch.arange(n_items, dtype=torch.float32)) \
            and torch.equal(got[:, 1], 10.0 * torch.arange(n_items, dtype=torch.float32))
        shards[n_items] = idx
    q.put((rank, bool(ok), shards))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_buckets_and_predict_gather_gloo_world8():
    world = 8
    port = 30100 + (os.getpid() % 500)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    for n_items in (24, 27, 29, 8, 5, 1):
        per = (n_items + world - 1) // world
        allidx = [i for _, _, sh in res for i in sh[n_items]]
        assert all(len(sh[n_items]) == per for _, _, sh in res)
        assert set(allidx) == set(range(n_items)) and len(allidx) == per * world


def test_gather_sharded_single_process_is_identity():
    from spokennlp_amd import dp
    x = torch.arange(12.0).reshape(6, 2)
    assert torch.equal(dp.gather_sharded(x, 6), x) and torch.equal(dp.gather_sharded(x, 4), x[:4])
