"""Attention-probability dropout decided once per layer (amdseg_attn_keepmask, include/amdseg.h ABI 6): the two lane-mask layouts hold the
SAME Bernoulli matrix, its statistics are those of the stateless hash path (same realised rate, independent neighbours), and the three
attention kernels reading it compute what a torch reference computes with that mask.  Replaces, on the training path, the per-element
hash of [hf] models/bert/modeling_bert.py:111-136's `nn.functional.dropout(attn_weights)`."""
import numpy as np
import pytest
import torch

from tests.test_gpu_kernels import attn_ref, make_qkv, rel_err

pytestmark = pytest.mark.gpu


def _ops():
    from spokennlp_amd import ops
    return ops


def unpack_keep(keep, B, L, heads):
    """uint8 device buffer -> (keep from layout A, keep from layout B), both bool [B*heads, L(query), L(key)] (the layouts are documented in
    csrc/attention.hip, 'dropout keep masks')"""
    w = keep.cpu().numpy().view(np.uint64)
    BH = B * heads
    n = BH * L * L // 64
    bits = ((w[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    # A: word [bh][q/16][key/64][fc][r], bit [g][i16]  <->  q = 16 (q/16) + i16, key = 64 (key/64) + 16 fc + 4 g + r
    a = bits[:n].reshape(BH, L // 16, L // 64, 4, 4, 4, 16)             # bh, R16, C, fc, r, g, i16
    a = a.transpose(0, 1, 6, 2, 3, 5, 4).reshape(BH, L, L)              # bh, (R16, i16), (C, fc, g, r)
    # B: word [bh][key/16][q/64][qf][r], bit [g][i16]  <->  key = 16 (key/16) + i16, q = 64 (q/64) + 16 qf + 4 g + r
    b = bits[n:].reshape(BH, L // 16, L // 64, 4, 4, 4, 16)             # bh, K16, QC, qf, r, g, i16
    b = b.transpose(0, 2, 3, 5, 4, 1, 6).reshape(BH, L, L)              # bh, (QC, qf, g, r), (K16, i16)
    return torch.from_numpy(a.copy()), torch.from_numpy(b.copy())


@pytest.mark.parametrize("B,L,heads,p", [(2, 128, 2, 0.1), (1, 512, 3, 0.1), (3, 192, 1, 0.25), (1, 64, 1, 0.5)])
def test_keepmask_layouts_agree_and_statistics(dev, B, L, heads, p):
    ops = _ops()
    keep = ops.attn_keepmask(B, L, heads, p, 1234, dev)
    ka, kb = unpack_keep(keep, B, L, heads)
    assert torch.equal(ka, kb)                                          # layout B is the transpose of layout A, bit for bit
    k = ka.float()
    th = round(p * 65536)
    kq = 1.0 - th / 65536.0                                             # the realised keep rate of the hash path as well
    n = k.numel()
    assert abs(k.mean().item() - kq) < 4 * (kq * (1 - kq) / n) ** 0.5 + 1e-4
    # neighbours along the key axis, the query axis, the two diagonals and between heads are independent
    pairs = [(k[..., :-1], k[..., 1:]), (k[..., :-2], k[..., 2:]), (k[:, :-1, :], k[:, 1:, :]), (k[:, :-16, :], k[:, 16:, :]),
             (k[:, :-1, :-1], k[:, 1:, 1:]), (k[..., :-4], k[..., 4:]), (k[..., :-16], k[..., 16:])]
    if B * heads > 1:
        pairs.append((k[:1], k[1:2]))
    if L > 64:
        pairs.append((k[..., :-64], k[..., 64:]))
        pairs.append((k[:, :-64, :], k[:, 64:, :]))
    for a_, b_ in pairs:
        joint = (a_ * b_).mean().item()
        assert abs(joint - kq * kq) < 5 * (kq * kq * (1 - kq * kq) / a_.numel()) ** 0.5 + 2e-4, joint
    # every key column and every query row sees the rate
    sig = (kq * (1 - kq) / (B * heads * L)) ** 0.5
    assert (k.mean(dim=(0, 1)) - kq).abs().max().item() < 5.5 * sig
    assert (k.mean(dim=(0, 2)) - kq).abs().max().item() < 5.5 * sig
    # the same seed gives the same masks, another seed other ones
    assert torch.equal(ops.attn_keepmask(B, L, heads, p, 1234, dev), keep)
    k2, _ = unpack_keep(ops.attn_keepmask(B, L, heads, p, 1235, dev), B, L, heads)
    assert abs((k2.float() * k).mean().item() - kq * kq) < 5 * (kq * kq * (1 - kq * kq) / n) ** 0.5 + 2e-4


def test_keepmask_forward_applies_exactly_these_bits(dev):
    """q = k = 0 makes the probabilities uniform and one-hot V rows make the output row q the dropped-out probability row: the mask the
    forward kernel APPLIES is the unpacked layout A (extract_keep_mask of test_gpu_kernels.py, on the _keep entry point)."""
    ops = _ops()
    B, L, heads, p = 2, 256, 2, 0.1
    keep = ops.attn_keepmask(B, L, heads, p, 99, dev)
    ka, _ = unpack_keep(keep, B, L, heads)
    mb = torch.zeros(B, L, device=dev)
    seen = torch.zeros(B, heads, L, L)
    for blk in range(L // 64):
        qkv = torch.zeros(B, L, 3, heads, 64, device=dev)
        for j in range(64):
            qkv[:, blk * 64 + j, 2, :, j] = 1.0
        ctx, _ = ops.attn_fwd_keep(qkv.view(B * L, -1).bfloat16(), mb, B, L, heads, p, keep)
        o = ctx.float().view(B, L, heads, 64).permute(0, 2, 1, 3)
        seen[:, :, :, blk * 64:(blk + 1) * 64] = (o > 0).float().cpu()
    assert torch.equal(seen.view(B * heads, L, L).bool(), ka)


@pytest.mark.parametrize("B,L,heads,pad", [(2, 128, 2, False), (2, 512, 3, True), (1, 192, 1, False)])
def test_keepmask_forward_backward_vs_torch(dev, B, L, heads, pad):
    ops = _ops()
    p, seed = 0.1, 4321
    keep = ops.attn_keepmask(B, L, heads, p, seed, dev)
    ka, _ = unpack_keep(keep, B, L, heads)
    km = ka.view(B, heads, L, L).float().to(dev)
    th = round(p * 65536)
    inv_keep = 65536.0 / (65536 - th)
    qkv, mb = make_qkv(dev, B, L, heads, 21, pad=pad)
    g = torch.Generator(device="cpu").manual_seed(22)
    dctx = torch.randn(B * L, heads * 64, generator=g).to(dev).bfloat16()
    ctx, lse = ops.attn_fwd_keep(qkv, mb, B, L, heads, p, keep)
    q32 = qkv.float().requires_grad_(True)
    ref, _ = attn_ref(q32, mb, B, L, heads, keep=km, inv_keep=inv_keep)
    assert rel_err(ctx, ref) < 8e-3
    dqkv = ops.attn_bwd_keep(qkv, mb, ctx, dctx, lse, B, L, heads, p, keep)
    ref.backward(dctx.float())
    H = heads * 64
    for i, name in enumerate(["dq", "dk", "dv"]):                       # per section: dq reads layout A, dk / dv layout B
        assert rel_err(dqkv[:, i * H:(i + 1) * H], q32.grad[:, i * H:(i + 1) * H]) < 2e-2, name


def test_keepmask_p0_and_null_fall_back_to_the_hash_path(dev):
    ops = _ops()
    B, L, heads = 2, 128, 2
    qkv, mb = make_qkv(dev, B, L, heads, 5)
    ctx0, lse0 = ops.attn_fwd(qkv, mb, B, L, heads, p=0.0)
    keep = ops.attn_keepmask(B, L, heads, 0.1, 7, dev)
    ctx1, lse1 = ops.attn_fwd_keep(qkv, mb, B, L, heads, 0.0, keep)    # p = 0: no dropout whatever the buffer holds
    assert torch.equal(ctx0, ctx1) and torch.equal(lse0, lse1)
    ctx2, _ = ops.attn_fwd_keep(qkv, mb, B, L, heads, 0.1, None)        # no buffer: hash path with seed 0
    ctx3, _ = ops.attn_fwd(qkv, mb, B, L, heads, p=0.1, seed=0)
    assert torch.equal(ctx2, ctx3)


def test_keepmask_with_kend_skips_only_unread_chunks(dev):
    """chunks past a sequence's last unmasked key are not generated (the kernels never read them); everything in front of kend is the same
    stream of bits whether or not kend is given -- per (wave, chunk group), so compare through the kernels, not bit by bit"""
    ops = _ops()
    B, L, heads, p = 2, 256, 2, 0.1
    qkv, mb = make_qkv(dev, B, L, heads, 5, pad=True)
    kend = torch.tensor([int((mb[b] == 0).nonzero().max()) + 1 for b in range(B)], dtype=torch.int32, device=dev)
    keep = torch.zeros(_ops().attn_keepmask(B, L, heads, p, 3, dev).numel(), dtype=torch.uint8, device=dev)
    from spokennlp_amd import lib as Lb
    Lb.check(Lb.load().amdseg_attn_keepmask(keep.data_ptr(), B, L, heads, p, 3, kend.data_ptr(), torch.cuda.current_stream().cuda_stream), "keepmask")
    ka, kb = unpack_keep(keep, B, L, heads)
    ka = ka.view(B, heads, L, L); kb = kb.view(B, heads, L, L)
    for b in range(B):
        nvis = -(-int(kend[b]) // 64) * 64
        assert torch.equal(ka[b, :, :, :nvis], kb[b, :, :, :nvis])
        assert abs(ka[b, :, :, :nvis].float().mean().item() - 0.9) < 0.01
        assert not ka[b, :, :, nvis:].any() and not kb[b, :, :, nvis:].any()        # untouched (the buffer was zeroed)


@pytest.mark.parametrize("B,L,heads,w,G", [(2, 256, 2, 64, 1), (1, 512, 3, 128, 1), (2, 384, 1, 32, 0)])
def test_band_keepmask_forward_backward_vs_torch(dev, B, L, heads, w, G):
    """the band (Longformer) kernels on keep masks generated for the band's cells: forward + dq / dk / dv against a torch band attention that
    applies the unpacked bits; the two layouts agree on every cell inside the band (cells outside it are never written or read)"""
    ops = _ops()
    p, seed = 0.1, 777
    keep = ops.attn_keepmask_band(B, L, heads, p, seed, w, G, dev)
    ka, kb = unpack_keep(keep, B, L, heads)
    i = torch.arange(L)
    inband = ((i[:, None] - i[None, :]).abs() <= w) | (i[None, :] < G)
    assert torch.equal(ka[:, inband], kb[:, inband])
    rate = ka[:, inband].float().mean().item()
    assert abs(rate - 0.9) < 0.01
    km = ka.view(B, heads, L, L).float().to(dev)
    inv_keep = 65536.0 / (65536 - round(p * 65536))
    qkv, mb = make_qkv(dev, B, L, heads, 31, pad=True)
    g = torch.Generator(device="cpu").manual_seed(32)
    dctx = torch.randn(B * L, heads * 64, generator=g).to(dev).bfloat16()
    ctx, lse = ops.attn_band_fwd_keep(qkv, mb, B, L, heads, w, G, p, keep)
    H = heads * 64
    q32 = qkv.float().requires_grad_(True)
    q, k, v = [t.view(B, L, heads, 64).transpose(1, 2) for t in q32.view(B, L, 3 * H).split(H, dim=-1)]
    s = q @ k.transpose(-1, -2) * 0.125 + mb.view(B, 1, 1, L)
    s = s.masked_fill(~inband.to(dev), float("-inf"))
    pr = torch.softmax(s, -1) * (mb.view(B, 1, L, 1) >= 0)
    ref = ((pr * km * inv_keep) @ v).transpose(1, 2).reshape(B * L, H)
    assert rel_err(ctx, ref) < 8e-3
    dqkv = ops.attn_band_bwd_keep(qkv, mb, ctx, dctx, lse, B, L, heads, w, G, p, keep)
    ref.backward(dctx.float())
    for j, name in enumerate(["dq", "dk", "dv"]):
        assert rel_err(dqkv[:, j * H:(j + 1) * H], q32.grad[:, j * H:(j + 1) * H]) < 2e-2, name




@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,L,heads,H,window,G,pad", [(4, 512, 12, 768, 0, 0, False), (2, 256, 2, 128, 0, 0, True), (1, 1024, 2, 1536, 256, 1, False),
                                                      (3, 64, 1, 64, 0, 0, False)])
def test_add_ln_fwd_keepmask_is_the_two_calls_bit_for_bit(dev, dtype, B, L, heads, H, window, G, pad):
    """amdseg_add_ln_fwd_keepmask (ABI 14): the LayerNorm rows and the keep masks of an attention layer as interleaved workgroups of ONE launch;
    every output -- z, out, mean, rstd, the kept hidden-dropout bits, both mask layouts -- equals what amdseg_add_ln_fwd and
    amdseg_attn_keepmask / _band write alone"""
    ops = _ops()
    from spokennlp_amd import lib as Lb
    M = B * L
    g = torch.Generator(device="cpu").manual_seed(5)
    y0 = torch.randn(M, H, generator=g).to(dev, dtype)
    res = torch.randn(M, H, generator=g).to(dev, dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    kend = torch.tensor([L - 70 * (b % 3) for b in range(B)], dtype=torch.int32, device=dev) if pad else None
    # the two launches
    y1 = y0.clone()
    out1, mean1, rstd1 = ops.add_ln_fwd(y1, res, gamma, beta, 1e-12, 0.1, 77)
    keep1 = ops.attn_keepmask_band(B, L, heads, 0.1, 991, window, G, dev) if window else ops.attn_keepmask(B, L, heads, 0.1, 991, dev, kend=kend)
    # the one launch
    y2 = y0.clone()
    bits = torch.zeros(M * H // 8, dtype=torch.uint8, device=dev)
    out2, mean2, rstd2, keep2 = ops.add_ln_fwd_keepmask(y2, res, gamma, beta, 1e-12, 0.1, 77, B, L, heads, 0.1, 991, kend=kend, window=window, nglobal=G,
                                                        drop_bits=bits)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(out1, out2) and torch.equal(mean1, mean2) and torch.equal(rstd1, rstd2)
    if pad:                                                # chunks past kend are written by neither form: compare what the consumers read
        ka1, kb1 = unpack_keep(keep1, B, L, heads); ka2, kb2 = unpack_keep(keep2, B, L, heads)
        for b in range(B):
            nv, hs = (int(kend[b]) + 63) // 64 * 64, slice(b * heads, (b + 1) * heads)
            assert torch.equal(ka1[hs, :, :nv], ka2[hs, :, :nv]) and torch.equal(kb1[hs, :, :nv], kb2[hs, :, :nv])
    else:
        assert torch.equal(keep1, keep2)
    assert int(bits.count_nonzero()) > 0                   # the hidden-dropout decisions were kept as well ...
    y3 = y0.clone()                                        # ... and are the hash's: the row kernel without them gives the same z
    ops.add_ln_fwd(y3, res, gamma, beta, 1e-12, 0.1, 77)
    assert torch.equal(y3, y2)
    # errors: no mask buffer, a sequence length the generator does not tile
    lib = Lb.load()
    s = torch.cuda.current_stream().cuda_stream
    args = lambda keep_ptr, L_: (y2.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(),
                                 M, H, 1e-12, 0.1, 77, Lb.BF16 if dtype == torch.bfloat16 else Lb.F32, None, 1, keep_ptr, B, L_, heads, 0.1, 991, None, 0, 0, s)
    assert lib.amdseg_add_ln_fwd_keepmask(*args(None, L)) != 0
    assert lib.amdseg_add_ln_fwd_keepmask(*args(keep2.data_ptr(), L + 8)) != 0
