"""Per-kernel numerics tests of libamdseg on a real MI355X: each HIP kernel (called through the C ABI) against a plain
fp32 torch reference of the same op on the same seeded inputs.  Tolerances are stated per test: bf16 outputs carry a
2^-9 relative rounding, MFMA accumulates in fp32."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from spokennlp_amd import ops
    return ops


def rel_err(a, b):
    a = a.float(); b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


# ----------------------------------------------------------------------------------------------------------- GEMM NT
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (512, 768, 768), (1024, 2304, 768), (256, 768, 3072)])
def test_gemm_nt_plain(dev, M, N, K):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    B = torch.randn(N, K, generator=g).to(dev).bfloat16()
    ref = A.float() @ B.float().t()
    C32 = ops.gemm_nt(A, B, ops.EPI_NONE, out_dtype=torch.float32)
    # fp32 accumulation of exact bf16 products: only summation-order differences
    assert rel_err(C32, ref) < 2e-6
    assert (C32 - ref).abs().max().item() < 1e-3 * math.sqrt(K / 64)
    Cb = ops.gemm_nt(A, B, ops.EPI_NONE)
    assert torch.equal(Cb, C32.bfloat16()) or rel_err(Cb, ref) < 4e-3   # one bf16 rounding


def test_gemm_nt_transpose_detect(dev):
    """asymmetric operands: catches row/col swaps of the accumulator layout."""
    ops = _ops()
    M, N, K = 128, 256, 64
    A = torch.zeros(M, K, device=dev); B = torch.zeros(N, K, device=dev)
    A[:, 0] = torch.arange(M, device=dev) % 32          # small ints, exact in bf16
    B[:, 0] = 1.0
    B[:, 1] = torch.arange(N, device=dev) % 16
    A[:, 1] = 1.0
    C = ops.gemm_nt(A.bfloat16(), B.bfloat16(), ops.EPI_NONE, out_dtype=torch.float32)
    ref = A @ B.t()
    assert torch.equal(C, ref)


def test_gemm_nt_epilogues(dev):
    ops = _ops()
    M, N, K = 256, 512, 256
    g = torch.Generator(device="cpu").manual_seed(7)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
    B = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev).bfloat16()
    base = A.float() @ B.float().t()
    C = ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias, out_dtype=torch.float32)
    assert rel_err(C, base + bias) < 2e-6
    H, U = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU, bias=bias)
    assert rel_err(U, base + bias) < 4e-3
    assert rel_err(H, gelu(base + bias)) < 4e-3
    C = ops.gemm_nt(A, B, ops.EPI_ADD_RES, R=R, out_dtype=torch.float32)
    assert rel_err(C, base + R.float()) < 2e-6
    C = ops.gemm_nt(A, B, ops.EPI_GELU_BWD, R=R)
    assert rel_err(C, base * gelu_grad(R.float())) < 4e-3


@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (512, 1024, 1536), (512, 2304, 768), (256, 192, 1024), (768, 3072, 768), (256, 576, 832), (512, 960, 768)])
def test_gemm_nt_deep_pipeline_kernel_epilogues(dev, M, N, K):
    """K >= 768, M % 256 == 0 and N % 256 == 0 or N % 192 == 0: the deep-pipeline kernel, 256 x 256 or 256 x 192 tiles (whichever
    when N is no multiple of 256; K tiles odd and even); all epilogues, fp32 output, exact integers, strided views"""
    _check_all_epilogues(dev, M, N, K)


@pytest.mark.parametrize("M,N,K", [(2304, 1920, 192), (4096, 768, 3072), (128, 128, 64), (384, 640, 128)])
def test_gemm_nt_small_kernel_both_ring_depths(dev, M, N, K):
    """the 128 x 128 kernel (forced through the small-tile hook of an explicit context where the shape would take a wider tile): 270 tiles = more than
    one per CU -> two LDS stages, two workgroups per CU; 192 / 1 / 15 tiles -> the ring of four stages (round 6: the small-M form), with 3 (K = 192),
    48, 1 and 2 K tiles (ring not filled, wrap-around, tail waits).  All epilogues + the bit-identity with the unforced kernels of the same shapes."""
    ops = _ops()
    ctx = ops.L.Ctx()
    ctx.force_small_tile(1)
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16(); B = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev); R = torch.randn(M, N, generator=g).to(dev).bfloat16()
    base = A.float() @ B.float().t()
    with ctx.bound():
        f0 = ops.gemm_nt(A, B, ops.EPI_NONE, out_dtype=torch.float32)
        f1 = ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias)
        f3 = ops.gemm_nt(A, B, ops.EPI_ADD_RES, R=R)
        h, u = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU, bias=bias)
        gb = ops.gemm_nt(A, B, ops.EPI_GELU_BWD, R=R)
    assert rel_err(f0, base) < 2e-6 and rel_err(f1, base + bias) < 4e-3 and rel_err(f3, base + R.float()) < 4e-3
    assert rel_err(u, base + bias) < 4e-3 and rel_err(h, gelu(base + bias)) < 4e-3 and rel_err(gb, base * gelu_grad(R.float())) < 4e-3
    # whatever kernel the unforced call takes for this shape: the same fp32 sums, hence the same bits
    assert torch.equal(f0, ops.gemm_nt(A, B, ops.EPI_NONE, out_dtype=torch.float32))
    assert torch.equal(f1, ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias)) and torch.equal(f3, ops.gemm_nt(A, B, ops.EPI_ADD_RES, R=R))
    ctx.close()


@pytest.mark.parametrize("M,N,K", [(256, 192, 64), (512, 384, 320), (768, 960, 128)])
def test_gemm_nt_pingpong_kernel_epilogues(dev, M, N, K):
    """shapes with M % 256 == 0 and N % 192 == 0 (K < 768) take the 256x192 ping-pong kernel; all epilogues + fp32 output."""
    _check_all_epilogues(dev, M, N, K)


def _check_all_epilogues(dev, M, N, K):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M * 3 + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
    B = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev).bfloat16()
    base = A.float() @ B.float().t()
    assert rel_err(ops.gemm_nt(A, B, ops.EPI_NONE, out_dtype=torch.float32), base) < 2e-6
    assert rel_err(ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias, out_dtype=torch.float32), base + bias) < 2e-6
    Cb = ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias)
    assert torch.equal(Cb, ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias, out_dtype=torch.float32).bfloat16())
    H, U = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU, bias=bias)
    assert rel_err(U, base + bias) < 4e-3 and rel_err(H, gelu(base + bias)) < 4e-3
    assert rel_err(ops.gemm_nt(A, B, ops.EPI_ADD_RES, R=R, out_dtype=torch.float32), base + R.float()) < 2e-6
    assert rel_err(ops.gemm_nt(A, B, ops.EPI_GELU_BWD, R=R), base * gelu_grad(R.float())) < 4e-3
    if K >= 128 and M % 256 == 0:
        # AMDSEG_EPI_KEEP_DERIV (deep-pipeline kernel): the forward keeps gelu'(pre-activation) as its second output, the backward multiplies by it
        from spokennlp_amd import lib as L
        Hd, D = ops.gemm_nt(A, B, ops.EPI_BIAS_GELU | L.EPI_KEEP_DERIV, bias=bias)
        assert rel_err(Hd, gelu(base + bias)) < 4e-3 and rel_err(D, gelu_grad(base + bias)) < 4e-3
        if K >= 768:
            assert torch.equal(Hd, H)                        # the same kernel, the same activation bits as the pre-activation form
        assert rel_err(ops.gemm_nt(A, B, ops.EPI_GELU_BWD | L.EPI_KEEP_DERIV, R=D), base * D.float()) < 4e-3
        # ... and the pair is the pair it replaces, to bf16 rounding of the kept tensor
        assert rel_err(ops.gemm_nt(A, B, ops.EPI_GELU_BWD | L.EPI_KEEP_DERIV, R=D).float(), ops.gemm_nt(A, B, ops.EPI_GELU_BWD, R=U).float()) < 8e-3
    if K >= 128 and M % 256 == 0 and N % 256 == 0:
        # ... and the derivative as one byte per element (AMDSEG_EPI_DERIV_U8): q = round((g' + 0.135) * 200), |error| <= 0.0025 (+ the bf16-level
        # noise of the accumulators under it)
        from spokennlp_amd import lib as L
        fl = L.EPI_KEEP_DERIV | L.EPI_DERIV_U8
        Q = torch.zeros(M, N, dtype=torch.uint8, device=dev)
        H8 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, H8.data_ptr(), N, M, N, K, ops.EPI_BIAS_GELU | fl, bias.data_ptr(), None, 0,
                                     Q.data_ptr(), N, 0, st)
        assert rc == 0
        assert rel_err(H8, gelu(base + bias)) < 4e-3
        want = gelu_grad(base + bias)
        got = Q.float() * 0.005 - 0.135
        assert (got - want).abs().max().item() < 0.0025 + 2e-3, (got - want).abs().max().item()
        dU = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, dU.data_ptr(), N, M, N, K, ops.EPI_GELU_BWD | fl, None, Q.data_ptr(), N,
                                     None, 0, 0, st)
        assert rc == 0
        assert rel_err(dU, base * got) < 4e-3
        # the same for gelu_new (AMDSEG_EPI_ACT_TANH: BigBird's hidden_act)
        x = (base + bias).double()
        inner = 0.7978845608028654 * (x + 0.044715 * x ** 3)
        th = torch.tanh(inner)
        want_h = (0.5 * x * (1 + th)).float()
        want_d = (0.5 * (1 + th) + 0.5 * x * (1 - th * th) * 0.7978845608028654 * (1 + 3 * 0.044715 * x * x)).float()
        rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, H8.data_ptr(), N, M, N, K, ops.EPI_BIAS_GELU | fl | L.EPI_ACT_TANH, bias.data_ptr(),
                                     None, 0, Q.data_ptr(), N, 0, st)
        assert rc == 0
        assert rel_err(H8, want_h) < 4e-3
        assert ((Q.float() * 0.005 - 0.135) - want_d).abs().max().item() < 0.0025 + 2e-3
    # asymmetric small-integer operands: exact, catches any row/col or fragment swap
    A2 = torch.zeros(M, K, device=dev); B2 = torch.zeros(N, K, device=dev)
    A2[:, 0] = torch.arange(M, device=dev) % 61; B2[:, 0] = 1.0
    B2[:, 1] = torch.arange(N, device=dev) % 53; A2[:, 1] = 1.0
    assert torch.equal(ops.gemm_nt(A2.bfloat16(), B2.bfloat16(), ops.EPI_NONE, out_dtype=torch.float32), A2 @ B2.t())
    # strided operands / output inside wider buffers
    Abig = torch.randn(M, 2 * K, generator=g).to(dev).bfloat16()
    out = torch.zeros(M, 2 * N, dtype=torch.float32, device=dev)
    from spokennlp_amd import lib as L
    rc = L.load().amdseg_gemm_nt(Abig[:, K:].data_ptr(), 2 * K, B.data_ptr(), K, out[:, N:].data_ptr(), 2 * N, M, N, K, 0, None, None, 0,
                                 None, 0, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert rel_err(out[:, N:], Abig[:, K:].float() @ B.float().t()) < 2e-6 and out[:, :N].abs().max().item() == 0


def test_gemm_nt_strided_views(dev):
    """operands / outputs that are column slices of wider buffers (ld != width)."""
    ops = _ops()
    M, N, K = 128, 128, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    Abig = torch.randn(M, 3 * K, generator=g).to(dev).bfloat16()
    B = torch.randn(N, K, generator=g).to(dev).bfloat16()
    A = Abig[:, K:2 * K]
    from spokennlp_amd import lib as L
    out = torch.zeros(M, 2 * N, dtype=torch.float32, device=dev)
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), 3 * K, B.data_ptr(), K, out[:, N:].data_ptr(), 2 * N, M, N, K, 0, None, None, 0,
                                 None, 0, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = A.float() @ B.float().t()
    assert rel_err(out[:, N:], ref) < 2e-6
    assert out[:, :N].abs().max().item() == 0


def test_gemm_nt_rejects_bad_shapes(dev):
    from spokennlp_amd import lib as L
    a = torch.zeros(128, 64, dtype=torch.bfloat16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert L.load().amdseg_gemm_nt(a.data_ptr(), 64, a.data_ptr(), 64, a.data_ptr(), 64, 100, 128, 64, 0, None, None, 0, None, 0, 0, s) == 1001
    assert L.load().amdseg_gemm_nt(None, 64, a.data_ptr(), 64, a.data_ptr(), 64, 128, 128, 64, 0, None, None, 0, None, 0, 0, s) == 1002


# ----------------------------------------------------------------------------------------------------------- GEMM TN
def test_gemm_tn_grouped(dev):
    ops = _ops()
    M = 512
    shapes = [(128, 256), (256, 128), (384, 128), (128, 128)]
    g = torch.Generator(device="cpu").manual_seed(11)
    As = [torch.randn(M, n, generator=g).to(dev).bfloat16() for n, _ in shapes]
    Bs = [torch.randn(M, k, generator=g).to(dev).bfloat16() for _, k in shapes]
    Cs = [torch.full((n, k), 7.0, dtype=torch.float32, device=dev) for n, k in shapes]
    ops.gemm_tn_grouped(As, Bs, Cs, accumulate=False)
    for a, b, c in zip(As, Bs, Cs):
        ref = a.float().t() @ b.float()
        assert rel_err(c, ref) < 2e-6
    ops.gemm_tn_grouped(As, Bs, Cs, accumulate=True)
    for a, b, c in zip(As, Bs, Cs):
        ref = 2 * (a.float().t() @ b.float())
        assert rel_err(c, ref) < 2e-6


@pytest.mark.parametrize("M", [128, 192, 1024])
def test_gemm_tn_grouped_dp(dev, M):
    """256 x 128 deep-pipeline weight-gradient kernel (every N % 256 == 0): ragged K-tile counts (M/64 = 2, 3, 16), strided A
    (a column block of a wider matrix, like the dQKV slices), overwrite and accumulate, and agreement with the 128 x 128 kernel."""
    ops = _ops()
    shapes = [(256, 384), (512, 128), (768, 256), (256, 128)]
    g = torch.Generator(device="cpu").manual_seed(12)
    wide = torch.randn(M, 1024, generator=g).to(dev).bfloat16()
    As = [wide[:, 128:128 + shapes[0][0]]] + [torch.randn(M, n, generator=g).to(dev).bfloat16() for n, _ in shapes[1:]]
    Bs = [torch.randn(M, k, generator=g).to(dev).bfloat16() for _, k in shapes]
    Cs = [torch.full((n, k), 7.0, dtype=torch.float32, device=dev) for n, k in shapes]
    ops.gemm_tn_grouped(As, Bs, Cs, accumulate=False)
    for a, b, c in zip(As, Bs, Cs):
        ref = a.float().t() @ b.float()
        assert rel_err(c, ref) < 2e-6
    ops.gemm_tn_grouped(As, Bs, Cs, accumulate=True)
    for a, b, c in zip(As, Bs, Cs):
        ref = 2 * (a.float().t() @ b.float())
        assert rel_err(c, ref) < 2e-6
    # fused bias gradients (column sums of A from the GEMM's own fragments), two of the four problems, overwrite then accumulate
    cs = [None, torch.full((shapes[1][0],), 3.0, device=dev), None, torch.full((shapes[3][0],), 3.0, device=dev)]
    C3 = [torch.empty_like(c) for c in Cs]
    ops.gemm_tn_grouped(As, Bs, C3, accumulate=False, colsums=cs)
    for i in (1, 3):
        ref = As[i].float().sum(0)
        assert (cs[i] - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item()), i
    ops.gemm_tn_grouped(As, Bs, C3, accumulate=True, colsums=cs)
    for i in (1, 3):
        ref = 2 * As[i].float().sum(0)
        assert (cs[i] - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item()), i
    for a, b, c in zip(As, Bs, C3):
        assert rel_err(c, 2 * (a.float().t() @ b.float())) < 2e-6
    ctx = ops.L.Ctx()                                       # the small-tile hook lives in an explicit context, bound for the cfg-less call
    assert ctx.force_small_tile(1) == 0
    with ctx.bound():
        C2 = [torch.empty_like(c) for c in Cs]
        ops.gemm_tn_grouped(As, Bs, C2, accumulate=False)
    C4 = [torch.empty_like(c) for c in Cs]
    ops.gemm_tn_grouped(As, Bs, C4, accumulate=False)        # unbound again: the deep-pipeline kernel
    for c2, c4 in zip(C2, C4):
        assert rel_err(c2, c4) < 2e-6
    ctx.close()
    for a, b, c in zip(As, Bs, C2):
        assert rel_err(c, a.float().t() @ b.float()) < 2e-6


def test_gemm_tn_transpose_detect(dev):
    ops = _ops()
    M, N, K = 64, 128, 256
    A = torch.zeros(M, N, device=dev); B = torch.zeros(M, K, device=dev)
    A[0] = torch.arange(N, device=dev) % 32
    B[0] = 1.0
    A[1] = 1.0
    B[1] = torch.arange(K, device=dev) % 16
    C = torch.empty(N, K, dtype=torch.float32, device=dev)
    ops.gemm_tn_grouped([A.bfloat16()], [B.bfloat16()], [C])
    assert torch.equal(C, A.t() @ B)


# ----------------------------------------------------------------------------------------------------------- attention
def attn_ref(qkv, mask_bias, B, L, heads, keep=None, inv_keep=1.0):
    """fp32 reference on the bf16 inputs. keep: optional [B,heads,L,L] 0/1 mask of kept probabilities."""
    H = heads * 64
    x = qkv.float().view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)   # [3,B,h,L,64]
    q, k, v = x[0], x[1], x[2]
    s = q @ k.transpose(-1, -2) * 0.125 + mask_bias.view(B, 1, 1, L)
    p = torch.softmax(s, dim=-1)
    pd = p if keep is None else p * keep * inv_keep
    o = pd @ v
    return o.permute(0, 2, 1, 3).reshape(B * L, H), p


def make_qkv(dev, B, L, heads, seed, pad=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = (torch.randn(B * L, 3 * heads * 64, generator=g) * 1.0).to(dev).bfloat16()
    mask = torch.ones(B, L)
    if pad:
        for b in range(B):
            n = L - (b * 37) % (L // 2)
            mask[b, n:] = 0
    mask_bias = ((1.0 - mask) * -30000.0).to(dev)
    return qkv, mask_bias


@pytest.mark.parametrize("B,L,heads", [(1, 64, 1), (2, 128, 2), (2, 512, 3)])
def test_attn_fwd(dev, B, L, heads):
    ops = _ops()
    qkv, mb = make_qkv(dev, B, L, heads, 5)
    ctx, lse = ops.attn_fwd(qkv, mb, B, L, heads)
    ref, p = attn_ref(qkv, mb, B, L, heads)
    # P is rounded to bf16 before P.V (relative 2^-9), output rounded to bf16
    assert (ctx.float() - ref).abs().max().item() < 2e-2
    assert rel_err(ctx, ref) < 6e-3
    x = qkv.float().view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = x[0] @ x[1].transpose(-1, -2) * 0.125 + mb.view(B, 1, 1, L)
    lse_ref = torch.logsumexp(s, dim=-1).reshape(-1)
    assert (lse - lse_ref).abs().max().item() < 1e-3


@pytest.mark.parametrize("B,L,heads", [(1, 64, 1), (2, 128, 2), (1, 512, 2)])
def test_attn_bwd(dev, B, L, heads):
    ops = _ops()
    qkv, mb = make_qkv(dev, B, L, heads, 9)
    g = torch.Generator(device="cpu").manual_seed(10)
    dctx = torch.randn(B * L, heads * 64, generator=g).to(dev).bfloat16()
    ctx, lse = ops.attn_fwd(qkv, mb, B, L, heads)
    dqkv = ops.attn_bwd(qkv, mb, ctx, dctx, lse, B, L, heads)
    q32 = qkv.float().requires_grad_(True)
    ref, _ = attn_ref(q32, mb, B, L, heads)
    ref.backward(dctx.float())
    assert rel_err(dqkv, q32.grad) < 1.5e-2
    # per-section (dq, dk, dv) so a wrong section cannot hide
    H = heads * 64
    for i, name in enumerate(["dq", "dk", "dv"]):
        assert rel_err(dqkv[:, i * H:(i + 1) * H], q32.grad[:, i * H:(i + 1) * H]) < 1.5e-2, name


def extract_keep_mask(ops, dev, B, L, heads, p, seed):
    """recover the kernel's dropout mask on attention probabilities: with q=k=0 the probabilities are uniform 1/L and
    with V = one-hot rows of a 64-key block the output row q is P_drop[q, block]."""
    keep = torch.zeros(B, heads, L, L, device=dev)
    mb = torch.zeros(B, L, device=dev)
    for blk in range(L // 64):
        qkv = torch.zeros(B, L, 3, heads, 64, device=dev)
        for j in range(64):
            qkv[:, blk * 64 + j, 2, :, j] = 1.0
        ctx, _ = ops.attn_fwd(qkv.view(B * L, -1).bfloat16(), mb, B, L, heads, p=p, seed=seed)
        o = ctx.float().view(B, L, heads, 64).permute(0, 2, 1, 3)      # [B,h,L(q),64(key in block)]
        keep[:, :, :, blk * 64:(blk + 1) * 64] = (o > 0).float()
    return keep


def test_attn_dropout_fwd_bwd(dev):
    ops = _ops()
    B, L, heads, p, seed = 2, 128, 2, 0.1, 1234
    keep = extract_keep_mask(ops, dev, B, L, heads, p, seed)
    frac = 1.0 - keep.mean().item()
    assert abs(frac - p) < 0.01, frac
    # mask quality: neighbours along the key axis (same hash word / next word) and along the query axis are independent
    kq = 1.0 - round(p * 65536) / 65536
    for a_, b_ in ((keep[..., :-1], keep[..., 1:]), (keep[..., :-2], keep[..., 2:]), (keep[..., :-1, :], keep[..., 1:, :]),
                   (keep[:, :1], keep[:, 1:])):
        joint = (a_ * b_).mean().item()
        assert abs(joint - kq * kq) < 4 * (0.09 / a_.numel()) ** 0.5 + 1e-3, joint
    sig = (0.09 / (B * heads * L)) ** 0.5                              # per key column / query row: B * heads * L samples each
    assert (keep.mean(dim=(0, 1, 2)) - kq).abs().max().item() < 5 * sig and (keep.mean(dim=(0, 1, 3)) - kq).abs().max().item() < 5 * sig
    th = round(p * 65536)
    inv_keep = 65536.0 / (65536 - th)
    qkv, mb = make_qkv(dev, B, L, heads, 21, pad=False)
    g = torch.Generator(device="cpu").manual_seed(22)
    dctx = torch.randn(B * L, heads * 64, generator=g).to(dev).bfloat16()
    ctx, lse = ops.attn_fwd(qkv, mb, B, L, heads, p=p, seed=seed)
    q32 = qkv.float().requires_grad_(True)
    ref, _ = attn_ref(q32, mb, B, L, heads, keep=keep, inv_keep=inv_keep)
    assert rel_err(ctx, ref) < 8e-3
    dqkv = ops.attn_bwd(qkv, mb, ctx, dctx, lse, B, L, heads, p=p, seed=seed)
    ref.backward(dctx.float())
    assert rel_err(dqkv, q32.grad) < 2e-2


# ----------------------------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_embed_ln_fwd_bwd(dev, dtype):
    ops = _ops()
    B, L, H, V = 2, 64, 128, 100
    g = torch.Generator(device="cpu").manual_seed(1)
    word = torch.randn(V, H, generator=g).to(dev); pos = torch.randn(L, H, generator=g).to(dev); typ = torch.randn(2, H, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev); beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    ids = torch.randint(0, V, (B * L,), generator=g).to(dev); tt = torch.randint(0, 2, (B * L,), generator=g).to(dev)
    out, z, mean, rstd = ops.embed_ln_fwd(ids, tt, None, word, pos, typ, gamma, beta, L, 1e-12, dtype=dtype)
    zr = word[ids] + typ[tt] + pos[torch.arange(B * L, device=dev) % L]
    ref = torch.nn.functional.layer_norm(zr, (H,), gamma, beta, 1e-12)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert (out.float() - ref).abs().max().item() < tol * 4
    assert (mean - zr.mean(-1)).abs().max().item() < 1e-5
    dz = torch.randn(B * L, H, generator=g).to(dev).to(dtype)
    dword = torch.zeros_like(word); dpos = torch.zeros_like(pos); dtyp = torch.zeros_like(typ)
    ops.embed_bwd(dz, ids, tt, None, dword, dpos, dtyp, L, pad_id=0)
    rw = torch.zeros_like(word).index_add_(0, ids, dz.float()); rw[0] = 0
    rp = torch.zeros_like(pos).index_add_(0, torch.arange(B * L, device=dev) % L, dz.float())
    rt = torch.zeros_like(typ).index_add_(0, tt, dz.float())
    assert (dword - rw).abs().max().item() < 1e-4
    assert (dpos - rp).abs().max().item() < 1e-4
    assert (dtyp - rt).abs().max().item() < 1e-3
    # type_vocab < 0: row 0 arrives holding colsum(dz) (the embedding LayerNorm backward's third column sum); rows of other types move out of it.
    # Explicit position ids (the atomic path) and rows that are exact zeros (they add nothing anywhere)
    dz2 = dz.clone(); dz2[5] = 0; dz2[70] = 0
    pos_ids = torch.randint(0, L, (B * L,), generator=g).to(dev)
    for tt2 in (tt, torch.zeros_like(tt)):
        dword2 = torch.zeros_like(word); dpos2 = torch.zeros_like(pos); dtyp2 = torch.zeros_like(typ)
        dtyp2[0] = dz2.float().sum(0)
        ops.embed_bwd(dz2, ids, tt2, pos_ids, dword2, dpos2, dtyp2, L, pad_id=0, type0_holds_colsum=True)
        rw2 = torch.zeros_like(word).index_add_(0, ids, dz2.float()); rw2[0] = 0
        assert (dword2 - rw2).abs().max().item() < 1e-4
        assert (dpos2 - torch.zeros_like(pos).index_add_(0, pos_ids, dz2.float())).abs().max().item() < 1e-4
        assert (dtyp2 - torch.zeros_like(typ).index_add_(0, tt2, dz2.float())).abs().max().item() < 1e-3


@pytest.mark.parametrize("dtype,H", [(torch.bfloat16, 768), (torch.float32, 768), (torch.bfloat16, 128), (torch.float32, 1024)])
def test_add_ln_fwd_bwd(dev, dtype, H):
    ops = _ops()
    M = 256
    g = torch.Generator(device="cpu").manual_seed(2)
    y = torch.randn(M, H, generator=g).to(dev).to(dtype); x = torch.randn(M, H, generator=g).to(dev).to(dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev); beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    dy = torch.randn(M, H, generator=g).to(dev).to(dtype)
    zr = (x.float() + y.float()).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(zr, (H,), gamma.clone().requires_grad_(True), beta, 1e-12)
    ybuf = y.clone()
    out, mean, rstd = ops.add_ln_fwd(ybuf, x, gamma, beta, 1e-12)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert (out.float() - ref).abs().max().item() < tol
    assert (ybuf.float() - zr).abs().max().item() < (1e-6 if dtype == torch.float32 else 2e-2)
    # backward against autograd on the *stored* z (what the kernel sees)
    z32 = ybuf.float().requires_grad_(True)
    g32 = gamma.clone().requires_grad_(True); b32 = beta.clone().requires_grad_(True)
    r2 = torch.nn.functional.layer_norm(z32, (H,), g32, b32, 1e-12)
    r2.backward(dy.float())
    dgamma = torch.zeros(H, device=dev); dbeta = torch.zeros(H, device=dev); dbias = torch.zeros(H, device=dev)
    dz, dbr = ops.ln_bwd(dy, ybuf, mean, rstd, gamma, dgamma=dgamma, dbeta=dbeta, dbias=dbias)
    rt = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(dz, z32.grad) < rt
    assert rel_err(dgamma, g32.grad) < rt
    assert rel_err(dbeta, b32.grad) < rt
    # dbias is summed from the unrounded fp32 dz inside the kernel; the stored dz carries one bf16 rounding
    assert rel_err(dbias, dz.float().sum(0)) < (1e-5 if dtype == torch.float32 else 5e-3)




def test_hidden_dropout_consistency(dev):
    """the dropout mask of add_ln_fwd (seed, element) is the one ln_bwd re-creates; drop rate ~ p; scaling unbiased."""
    ops = _ops()
    M, H, p, seed = 512, 768, 0.1, 99
    y = torch.ones(M, H, device=dev, dtype=torch.float32); x = torch.zeros(M, H, device=dev, dtype=torch.float32)
    gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
    ybuf = y.clone()
    out, mean, rstd = ops.add_ln_fwd(ybuf, x, gamma, beta, 1e-12, p=p, seed=seed)
    keep = (ybuf > 0).float()
    assert abs(1 - keep.mean().item() - p) < 5e-3
    assert abs(ybuf.mean().item() - 1.0) < 5e-3
    dy = torch.randn(M, H, device=dev)
    dz, dbr = ops.ln_bwd(dy, ybuf, mean, rstd, gamma, p=p, seed=seed)
    q = round(p * 65536) / 65536                                  # the realised rate: 16-bit threshold (common.h drop8_apply)
    assert torch.allclose(dbr, dz * keep / (1 - q), rtol=1e-6, atol=1e-6)
    # mask quality: the 8 elements of a 16-B chunk come from one hash -- no element position may be favoured, and neighbours inside a
    # pair / a chunk / across chunks must be independent (joint keep rate = (1-p)^2 within 4 sigma)
    big = torch.ones(4096, H, device=dev); zb = torch.zeros_like(big)
    ops.add_ln_fwd(big, zb, gamma, beta, 1e-12, p=p, seed=7)
    k = (big > 0).float()
    n = k.shape[0] * (H // 8)
    per_pos = k.view(-1, H // 8, 8).mean(dim=(0, 1))
    assert (per_pos - (1 - q)).abs().max().item() < 4 * (q * (1 - q) / n) ** 0.5 + 1e-4
    for shift in (1, 2, 8, H):                                    # same pair, same chunk, next chunk, next row
        a, b = k.flatten()[:-shift], k.flatten()[shift:]
        joint = (a * b).mean().item()
        assert abs(joint - (1 - q) ** 2) < 4 * (0.09 / a.numel()) ** 0.5 + 2e-4, (shift, joint)
    # a different seed gives a different mask
    yb2 = y.clone(); ops.add_ln_fwd(yb2, x, gamma, beta, 1e-12, p=p, seed=seed + 1)
    assert ((yb2 > 0).float() != keep).float().mean().item() > 0.05


def test_colsum_dropout_cast_transpose(dev):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(1000, 384, generator=g).to(dev)
    assert rel_err(ops.colsum(x), x.sum(0)) < 1e-5
    xb = x.bfloat16()
    assert rel_err(ops.colsum(xb), xb.float().sum(0)) < 1e-5
    wide = torch.randn(256, 512, generator=g).to(dev).bfloat16()
    assert rel_err(ops.colsum(wide[:, 128:256]), wide[:, 128:256].float().sum(0)) < 1e-5
    assert torch.equal(ops.cast(x, torch.bfloat16), xb)
    assert torch.equal(ops.cast(xb, torch.float32), xb.float())
    d = ops.dropout(x, 0.25, 5)
    kept = d != 0
    assert abs(1 - kept.float().mean().item() - 0.25) < 1e-2
    assert torch.allclose(d[kept], x[kept] / 0.75, rtol=1e-5)
    assert torch.equal(ops.dropout(x, 0.25, 5), d)
    W = torch.randn(192, 320, generator=g).to(dev)
    Wb = torch.empty(192, 320, dtype=torch.bfloat16, device=dev); Wt = torch.empty(320, 192, dtype=torch.bfloat16, device=dev)
    ops.cast_transpose(W, Wb, Wt)
    assert torch.equal(Wb, W.bfloat16()) and torch.equal(Wt, W.bfloat16().t().contiguous())
    # batched form; with W == NULL the bf16 copies are the input (they rode the AdamW pass) and only the transposes are written
    import ctypes as C
    from spokennlp_amd import lib as Lb
    lib = Lb.load()
    Ws = [torch.randn(128, 64, generator=g).to(dev), torch.randn(64, 256, generator=g).to(dev)]
    Wbs = [torch.empty(w.shape, dtype=torch.bfloat16, device=dev) for w in Ws]
    Wts = [torch.empty(w.shape[1], w.shape[0], dtype=torch.bfloat16, device=dev) for w in Ws]
    tab = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])        # noqa: E731
    Ns, Ks = (C.c_int * 2)(128, 64), (C.c_int * 2)(64, 256)
    s = torch.cuda.current_stream().cuda_stream
    Lb.check(lib.amdseg_cast_transpose_batched(2, tab(Ws), tab(Wbs), tab(Wts), Ns, Ks, s), "amdseg_cast_transpose_batched")
    for w, wb, wt in zip(Ws, Wbs, Wts):
        assert torch.equal(wb, w.bfloat16()) and torch.equal(wt, w.bfloat16().t().contiguous())
    for wt in Wts:
        wt.zero_()
    Lb.check(lib.amdseg_cast_transpose_batched(2, None, tab(Wbs), tab(Wts), Ns, Ks, s), "amdseg_cast_transpose_batched(copies in)")
    for wb, wt in zip(Wbs, Wts):
        assert torch.equal(wt, wb.t().contiguous())
    assert lib.amdseg_cast_transpose_batched(2, None, None, tab(Wts), Ns, Ks, s) != 0


@pytest.mark.parametrize("dtype,C", [(torch.float32, 2), (torch.bfloat16, 3)])
def test_rowdot(dev, dtype, C):
    ops = _ops()
    M, H = 300, 768
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn(M, H, generator=g).to(dev).to(dtype)
    W = (0.3 * torch.randn(C, H, generator=g)).to(dev); b = torch.randn(C, generator=g).to(dev)
    out = ops.rowdot_fwd(x, W, b)
    ref = x.float() @ W.t() + b
    assert (out - ref).abs().max().item() < 1e-3
    dl = torch.randn(M, C, generator=g).to(dev)
    dW = torch.zeros_like(W); db = torch.zeros_like(b)
    dx = ops.rowdot_bwd(x, W, dl, dW=dW, db=db)
    assert rel_err(dx, dl @ W) < (1e-5 if dtype == torch.float32 else 5e-3)
    assert rel_err(dW, dl.t() @ x.float()) < 1e-5
    assert rel_err(db, dl.sum(0)) < 1e-5


# ----------------------------------------------------------------------------------------------------------- optimiser
def test_adamw_matches_torch(dev):
    ops = _ops()
    n = 4096 * 3
    g = torch.Generator(device="cpu").manual_seed(8)
    p0 = torch.randn(n, generator=g)
    p = p0.clone().to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone(); opt.step()
        gd = gr.to(dev)
        ops.adamw(p, gd, m, v, shadow, 5e-3, 0.9, 0.999, 1e-8, 0.01, step, zero_grad=True)
        assert gd.abs().max().item() == 0
    assert (p.cpu() - pr.data).abs().max().item() < 2e-6
    assert torch.equal(shadow, p.bfloat16())


def test_gradnorm_clip(dev):
    ops = _ops()
    n = 1 << 20
    x = torch.randn(n, device=dev)
    out = torch.zeros(1, device=dev); partials = torch.empty(1024, device=dev)
    ops.sumsq(x, out, partials)
    assert abs(out.item() - (x.double() ** 2).sum().item()) / n < 1e-5
    coef = torch.zeros(1, device=dev); norm = torch.zeros(1, device=dev)
    ops.clip_coef(out, 1.0, 1.0, coef, norm)
    nr = x.norm().item()
    assert abs(norm.item() - nr) / nr < 1e-5
    assert abs(coef.item() - 1.0 / (nr + 1e-6)) / coef.item() < 1e-5
    y = x.clone(); ops.scale_(y, coef)
    assert abs(y.norm().item() - 1.0) < 1e-4


# ----------------------------------------------------------------------------------------------------------- fp32 parity mode
@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 32, 0), (256, 384, 160, 1), (512, 768, 768, 2), (256, 768, 3072, 1)])
def test_gemm_f32(dev, M, N, K, epi):
    from spokennlp_amd import lib as L
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(dev); B = (torch.randn(N, K, generator=g) * 0.1).to(dev); bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev)
    rc = L.load().amdseg_gemm_f32_nt(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, epi, bias.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = A.double() @ B.double().t()
    if epi >= 1:
        ref = ref + bias.double()
    if epi == 2:
        ref = gelu(ref)
    # exact fp32 fma chain: error is fp32 round-off of a K-long sum
    assert (C.double() - ref).abs().max().item() < 2e-5 * math.sqrt(K / 32)
    assert rel_err(C, ref.float()) < 1e-6


@pytest.mark.parametrize("B,L,heads", [(1, 64, 1), (2, 256, 2)])
def test_attn_f32(dev, B, L, heads):
    from spokennlp_amd import lib as Lb
    qkv, mb = make_qkv(dev, B, L, heads, 31)
    q32 = qkv.float()
    ctx = torch.empty(B * L, heads * 64, device=dev)
    rc = Lb.load().amdseg_attn_f32(q32.data_ptr(), mb.data_ptr(), ctx.data_ptr(), B, L, heads, 0.125, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref, _ = attn_ref(q32, mb, B, L, heads)
    assert (ctx - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,L", [(1, 64), (5, 128), (32, 512), (100, 192), (4, 4096)])
def test_pad_plan_and_guard(dev, B, L):
    """amdseg_pad_plan (kend / seq_order / pad_runs / pad_counts / additive mask from the int64 attention mask) against the same quantities
    written out in torch, with interior zeros, fully padded and full sequences; amdseg_pad_rows_guard against a host loop"""
    from spokennlp_amd import lib as Lb
    lib = Lb.load()
    g = torch.Generator().manual_seed(B * 1000 + L)
    lens = torch.randint(0, L + 1, (B,), generator=g)
    lens[0] = L
    if B > 2:
        lens[1] = 0
        lens[2] = 1
    am = (torch.arange(L)[None, :] < lens[:, None]).long()
    if B > 3:
        am[3, : int(lens[3]) // 2] = 0                        # interior zeros do not move kend
        am[3, 0] = 1 if lens[3] > 0 else 0
    am_d = am.to(dev)
    s = torch.cuda.current_stream().cuda_stream
    kend = torch.full((B,), -1, dtype=torch.int32, device=dev)
    order = torch.full((B,), -1, dtype=torch.int32, device=dev)
    runs = torch.full((B, 2), -1, dtype=torch.int32, device=dev)
    counts = torch.full((2,), -1, dtype=torch.int32, device=dev)
    mb = torch.full((B, L), 7.0, device=dev)
    Lb.check(lib.amdseg_pad_plan(am_d.data_ptr(), B, L, kend.data_ptr(), order.data_ptr(), runs.data_ptr(), counts.data_ptr(), mb.data_ptr(),
                                 -30000.0, s), "amdseg_pad_plan")
    ke = ((am != 0).long() * torch.arange(1, L + 1)[None, :]).amax(dim=1)
    assert torch.equal(kend.cpu().long(), ke)
    assert torch.equal(order.cpu().long(), torch.argsort(ke, descending=True, stable=True))
    nv = (ke + 63) // 64
    live = [b for b in range(B) if ke[b] > 0]
    assert counts.cpu().tolist() == [int(nv.sum()), len(live)]
    exp_runs = [[b * (L // 64), b * (L // 64) + int(nv[b])] for b in live]
    assert runs.cpu()[: len(live)].tolist() == exp_runs
    assert torch.equal(mb.cpu(), (1.0 - am.float()) * -30000.0)
    assert lib.amdseg_pad_plan(am_d.data_ptr(), B, L + 1, kend.data_ptr(), order.data_ptr(), runs.data_ptr(), counts.data_ptr(), None, 0.0, s) != 0
    # guard: exact zeros on the rows at positions >= kend <=> 0
    H = 64
    x = torch.randn(B, L, H, generator=g).to(dev)
    pad = (torch.arange(L, device=dev)[None, :] >= kend[:, None].long())
    x[pad] = 0
    guard = torch.full((1,), 5, dtype=torch.int32, device=dev)
    Lb.check(lib.amdseg_pad_rows_guard(x.data_ptr(), kend.data_ptr(), B, L, H, guard.data_ptr(), s), "amdseg_pad_rows_guard")
    assert int(guard.item()) == 0
    if int(pad.sum()) > 0:
        bi, pi = [int(v[0]) for v in torch.nonzero(pad, as_tuple=True)]
        for val in (1e-30, float("nan"), -0.0):
            x2 = x.clone()
            x2[bi, pi, H - 1] = val
            Lb.check(lib.amdseg_pad_rows_guard(x2.data_ptr(), kend.data_ptr(), B, L, H, guard.data_ptr(), s), "amdseg_pad_rows_guard")
            assert int(guard.item()) == (0 if val == 0 else 1), val


def test_add_ln_fwd_layernorm_only_mode(dev):
    """amdseg_add_ln_fwd with resid == NULL: LayerNorm of the buffer as it stands, which is not rewritten"""
    ops = _ops()
    M, N = 512, 768
    g = torch.Generator(device="cpu").manual_seed(5)
    z = torch.randn(M, N, generator=g).to(dev).bfloat16()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).to(dev); beta = (0.1 * torch.randn(N, generator=g)).to(dev)
    zb = z.clone()
    out_f, mean_f, rstd_f = ops.add_ln_fwd(zb, None, gamma, beta, 1e-12)
    assert torch.equal(zb, z)
    refln = torch.nn.functional.layer_norm(z.float(), (N,), gamma, beta, 1e-12)
    assert (out_f.float() - refln).abs().max().item() < 3e-2
    assert (mean_f - z.float().mean(1)).abs().max().item() < 1e-4


def test_c_abi_binder_without_python_or_torch(dev):
    """the drop-in boundary is a C ABI -- plain pointers, a stream, int codes, caller-owned buffers, no torch types.  tools/cabi_demo.cpp
    binds include/amdseg.h from C++ with nothing but the HIP runtime (hipMalloc'ed buffers, its own stream): one projection GEMM with the bias epilogue
    against a double-precision CPU product, the error path, and amdseg_allreduce_* with a world of one rank.  Built by __graft_entry__.build()."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "cabi_demo.bin")
    # a boundary test fails, it does not skip: on a GPU box the binder must have travelled with the snapshot
    assert os.path.exists(exe), "tools/cabi_demo.bin not built: run __graft_entry__.build() where hipcc is before going to the GPU box"
    env = dict(os.environ)
    import torch as _t                                       # the process needs A HIP runtime on its path: torch's own copy will do on a box without /opt/rocm/lib
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(os.path.dirname(_t.__file__), "lib"), "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "CABI_DEMO_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])
    assert f"amdseg ABI {_ops().L.ABI_VERSION}" in r.stdout
