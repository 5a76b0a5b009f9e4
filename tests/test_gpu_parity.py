""""parity" precision (config.amdseg_precision = "parity"; csrc/parity.hip): fp32 activations, every contraction as one bf16 MFMA
GEMM over the split images (hi + lo, three products), fp32 attention forward / backward.  The reference computes in fp32
(run_finetune.sh:61-96), the north star asks for logits within 1e-3 and bit-exact boundary decisions: asserted here for inference AND
for a training step (loss, every gradient) against the reference's golden vectors, tiny and at bert-base size."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from tests.test_oracle_golden import load_case, flags_of  # noqa: E402
from tests.test_gpu_model import build_model, to_dev  # noqa: E402


def test_split_images_reconstruct_fp32(dev):
    from spokennlp_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(256, 384, device=dev) * torch.logspace(-3, 3, 384, device=dev)).contiguous()
    for order in (0, 1):
        out = ops.split3(x, torch.empty(256, 3 * 384, dtype=torch.bfloat16, device=dev), order=order).float()
        hi = out[:, :384]
        lo = out[:, 2 * 384:] if order == 0 else out[:, 384:2 * 384]
        dup = out[:, 384:2 * 384] if order == 0 else out[:, 2 * 384:]
        assert torch.equal(dup, hi)
        assert torch.equal(hi, x.bfloat16().float())
        rel = ((hi + lo) - x).abs() / x.abs().clamp(min=1e-30)
        assert rel.max().item() < 2 ** -16
    W = torch.randn(128, 96, device=dev)
    t = ops.split3_transpose(W, torch.empty(96, 3 * 128, dtype=torch.bfloat16, device=dev)).float()
    assert torch.equal(t[:, :128], W.t().bfloat16().float()) and torch.equal(t[:, 256:], t[:, :128])
    assert ((t[:, :128] + t[:, 128:256]) - W.t()).abs().max().item() < 2 ** -16 * W.abs().max().item()


def test_split_gemm_is_fp32_grade(dev):
    """A . B^T through the K' = 3K images on the bf16 MFMA kernel vs an fp64 product: error at the fp32 level, not the bf16 level"""
    from spokennlp_amd import ops
    torch.manual_seed(1)
    M, N, K = 512, 768, 768
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05
    As = ops.split3(A, torch.empty(M, 3 * K, dtype=torch.bfloat16, device=dev), order=0)
    Bs = ops.split3(B, torch.empty(N, 3 * K, dtype=torch.bfloat16, device=dev), order=1)
    C = ops.gemm_nt(As, Bs, ops.EPI_NONE, out=torch.empty(M, N, dtype=torch.float32, device=dev))
    ref = A.double() @ B.double().t()
    err = (C.double() - ref).abs().max().item()
    bf = (A.bfloat16().float() @ B.bfloat16().float().t()).double()
    err_bf = (bf - ref).abs().max().item()
    print(f"split-bf16 GEMM max err {err:.2e} (plain bf16 operands: {err_bf:.2e}; fp32 torch: {(A @ B.t()).double().sub(ref).abs().max().item():.2e})")
    assert err < 2e-4 * ref.abs().max().item() / 10 and err < err_bf / 50


def test_gemm_bias_split_epilogue_writes_the_split_image(dev):
    """AMDSEG_EPI_BIAS_SPLIT: C <- bf16 hi, C2 <- bf16 lo of (A B^T + bias) -- what amdseg_split3 would make of the fp32 result, without the
    fp32 round trip; the shapes the 256 x 256 kernel does not tile are refused (the caller then runs GEMM + split3)"""
    from spokennlp_amd import lib as L, ops
    torch.manual_seed(2)
    M, N, K = 512, 768, 384
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    img = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
    s_ = torch.cuda.current_stream().cuda_stream
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, img.data_ptr(), 3 * N, M, N, K, 5, bias.data_ptr(), None, 0,
                                 img[:, 2 * N:].data_ptr(), 3 * N, 0, s_)
    assert rc == 0
    ref32 = ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias, out_dtype=torch.float32)
    want = ops.split3(ref32, torch.empty(M, 3 * N, dtype=torch.bfloat16, device=dev), order=0)
    assert torch.equal(img[:, :N], want[:, :N]) and torch.equal(img[:, 2 * N:], want[:, 2 * N:]) and float(img[:, N:2 * N].abs().max()) == 0.0
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, img.data_ptr(), 3 * N, 384, 128, K, 5, bias.data_ptr(), None, 0,
                                 img[:, 2 * N:].data_ptr(), 3 * N, 0, s_)
    assert rc == 1001                                       # AMDSEG_ERR_SHAPE


def test_gemm_gelu_bwd_split_epilogue_writes_the_image(dev):
    """AMDSEG_EPI_GELU_BWD_SPLIT: the image [hi | hi | lo] of (A B^T) * gelu_erf'(u), u fp32 -- what GEMM (fp32 out) + torch's exact GELU derivative
    + amdseg_split3 produce, in one launch"""
    from spokennlp_amd import lib as L, ops
    torch.manual_seed(4)
    M, N, K = 512, 768, 384
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    u = torch.randn(M, N, device=dev) * 1.5
    img = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
    s_ = torch.cuda.current_stream().cuda_stream
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, img.data_ptr(), 3 * N, M, N, K, 6, None, u.data_ptr(), N, None, 0, 0, s_)
    assert rc == 0
    y = ops.gemm_nt(A, B, ops.EPI_NONE, out_dtype=torch.float32)
    ud = u.double()
    grad = 0.5 * (1 + torch.erf(ud / 2 ** 0.5)) + ud * torch.exp(-0.5 * ud * ud) / (2 * torch.pi) ** 0.5
    want = y.double() * grad
    assert torch.equal(img[:, :N], img[:, N:2 * N])
    got = img[:, :N].double() + img[:, 2 * N:].double()
    assert (got - want).abs().max().item() < 3e-5 * want.abs().max().item()        # hi + lo carries 16 mantissa bits; erff / expf in fp32
    assert (img[:, 2 * N:].float().abs() <= img[:, :N].float().abs() * 2.0 ** -7 + 1e-30).all()        # lo is the rounding residue of hi
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, img.data_ptr(), 3 * N, 384, 128, K, 6, None, u.data_ptr(), N, None, 0, 0, s_)
    assert rc == 1001


def test_gemm_bias_gelu_split_epilogue_and_biasless_split(dev):
    """AMDSEG_EPI_BIAS_GELU_SPLIT: C = fp32 pre-activation, C2 = image of gelu_erf(C);  AMDSEG_EPI_BIAS_SPLIT with bias == NULL: the plain product"""
    from spokennlp_amd import lib as L, ops
    torch.manual_seed(6)
    M, N, K = 512, 768, 384
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.1).bfloat16(); bias = torch.randn(N, device=dev)
    u = torch.empty(M, N, device=dev)
    img = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
    s_ = torch.cuda.current_stream().cuda_stream
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, u.data_ptr(), N, M, N, K, 7, bias.data_ptr(), None, 0, img.data_ptr(), 3 * N, 1, s_)
    assert rc == 0
    ref = ops.gemm_nt(A, B, ops.EPI_BIAS, bias=bias, out_dtype=torch.float32)
    assert torch.equal(u, ref)
    want = torch.nn.functional.gelu(ref.double())
    assert torch.equal(img[:, :N], img[:, N:2 * N])
    got = img[:, :N].double() + img[:, 2 * N:].double()
    assert (got - want).abs().max().item() < 3e-5 * want.abs().max().item()
    img2 = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
    rc = L.load().amdseg_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, img2.data_ptr(), 3 * N, M, N, K, 5, None, None, 0, img2[:, 2 * N:].data_ptr(), 3 * N, 0, s_)
    assert rc == 0
    plain = ops.gemm_nt(A, B, ops.EPI_NONE, out_dtype=torch.float32)
    w3 = ops.split3(plain, torch.empty(M, 3 * N, dtype=torch.bfloat16, device=dev), order=0)
    assert torch.equal(img2[:, :N], w3[:, :N]) and torch.equal(img2[:, 2 * N:], w3[:, 2 * N:])


def test_split3_weights_batched_equals_the_single_matrix_calls(dev):
    from spokennlp_amd import lib as L, ops
    import ctypes as C
    torch.manual_seed(8)
    shapes = [(192, 64), (64, 256), (128, 128)]
    Ws = [torch.randn(n, k, device=dev) for n, k in shapes]
    outs = [torch.zeros(n, 3 * k, dtype=torch.bfloat16, device=dev) for n, k in shapes]
    outts = [torch.zeros(k, 3 * n, dtype=torch.bfloat16, device=dev) for n, k in shapes]
    n = len(Ws)
    vp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])          # noqa: E731
    rc = L.load().amdseg_split3_weights_batched(n, vp(Ws), vp(outs), vp(outts), (C.c_int * n)(*[s[0] for s in shapes]), (C.c_int * n)(*[s[1] for s in shapes]),
                                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for W, o, ot in zip(Ws, outs, outts):
        assert torch.equal(o, ops.split3(W, torch.empty_like(o), order=1))
        assert torch.equal(ot, ops.split3_transpose(W, torch.empty_like(ot)))


def _attn_ref(qkv, mask_bias, B, Lq, heads):
    H = heads * 64
    q, k, v = [t.view(B, Lq, heads, 64).transpose(1, 2) for t in qkv.view(B, Lq, 3 * H).split(H, -1)]
    s = q @ k.transpose(-1, -2) / 8.0 + mask_bias.view(B, 1, 1, Lq)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * Lq, H)


@pytest.mark.parametrize("B,Lq,heads", [(2, 128, 2), (1, 512, 3)])
def test_fp32_attention_forward_backward_vs_autograd(dev, B, Lq, heads):
    from spokennlp_amd import ops
    torch.manual_seed(Lq)
    H = heads * 64
    qkv = torch.randn(B * Lq, 3 * H, device=dev, dtype=torch.float64).requires_grad_(True)
    am = torch.ones(B, Lq, device=dev); am[-1, Lq - 37:] = 0
    mb = ((1 - am) * -1e30)
    dctx = torch.randn(B * Lq, H, device=dev)
    ref = _attn_ref(qkv, mb.double().clamp(min=-1e9), B, Lq, heads)
    ref.backward(dctx.double())
    q32 = qkv.detach().float().contiguous()
    ctx, lse = ops.pattn_fwd(q32, mb, B, Lq, heads)
    assert (ctx.double() - ref.detach()).abs().max().item() < 2e-5
    dqkv = ops.pattn_bwd(q32, mb, ctx, dctx, lse, B, Lq, heads)
    assert (dqkv.double() - qkv.grad).abs().max().item() < 5e-5 * max(1.0, qkv.grad.abs().max().item())


@pytest.mark.parametrize("B,Lq,heads", [(2, 128, 2), (1, 512, 3), (3, 192, 1)])
def test_split_bf16_attention_forward_backward_vs_autograd(dev, B, Lq, heads):
    """amdseg_sattn_fwd / _bwd (csrc/attention_split.hip: every contraction as hi.hi + hi.lo + lo.hi on the bf16 matrix cores) against fp64
    autograd: errors at the fp32 level, same bounds as the fp32-MFMA kernels they replace in the "parity" training step"""
    from spokennlp_amd import ops
    torch.manual_seed(Lq)
    H = heads * 64
    qkv = torch.randn(B * Lq, 3 * H, device=dev, dtype=torch.float64).requires_grad_(True)
    am = torch.ones(B, Lq, device=dev); am[-1, Lq - 37:] = 0
    mb = ((1 - am) * -30000.0)
    dctx = torch.randn(B * Lq, H, device=dev)
    ref = _attn_ref(qkv, mb.double(), B, Lq, heads)
    ref.backward(dctx.double())
    q32 = qkv.detach().float().contiguous()
    ctx, lse, qs = ops.sattn_fwd(q32, mb, B, Lq, heads)
    err = (ctx.double() - ref.detach()).abs().max().item()
    ctx_f32, lse_f32 = ops.pattn_fwd(q32, mb, B, Lq, heads)
    print(f"split-bf16 attention fwd max err {err:.2e} (fp32-MFMA kernel: {(ctx_f32.double() - ref.detach()).abs().max().item():.2e})")
    assert err < 5e-5
    assert (lse - lse_f32).abs().max().item() < 1e-4
    dqkv = ops.sattn_bwd(qs, mb, ctx, dctx, lse, B, Lq, heads)
    scale = max(1.0, qkv.grad.abs().max().item())
    for i, name in enumerate(("dq", "dk", "dv")):
        e = (dqkv[:, i * H:(i + 1) * H].double() - qkv.grad[:, i * H:(i + 1) * H]).abs().max().item()
        assert e < 1e-4 * scale, (name, e)


def test_split_bf16_attention_with_keep_masks(dev):
    """dropout read from the layer's keep masks (amdseg_attn_keepmask): forward and all three gradients against autograd with the unpacked mask"""
    from spokennlp_amd import ops
    from tests.test_gpu_keepmask import unpack_keep
    torch.manual_seed(5)
    B, Lq, heads, p = 2, 256, 2, 0.1
    H = heads * 64
    keep = ops.attn_keepmask(B, Lq, heads, p, 99, dev)
    ka, _ = unpack_keep(keep, B, Lq, heads)
    km = ka.view(B, heads, Lq, Lq).double().to(dev)
    inv_keep = 65536.0 / (65536 - round(p * 65536))
    qkv = torch.randn(B * Lq, 3 * H, device=dev, dtype=torch.float64).requires_grad_(True)
    mb = torch.zeros(B, Lq, device=dev)
    q, k, v = [t.view(B, Lq, heads, 64).transpose(1, 2) for t in qkv.view(B, Lq, 3 * H).split(H, -1)]
    pr = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) * km * inv_keep
    ref = (pr @ v).transpose(1, 2).reshape(B * Lq, H)
    dctx = torch.randn(B * Lq, H, device=dev)
    ref.backward(dctx.double())
    q32 = qkv.detach().float().contiguous()
    ctx, lse, qs = ops.sattn_fwd(q32, mb, B, Lq, heads, p=p, keep=keep)
    assert (ctx.double() - ref.detach()).abs().max().item() < 5e-5
    dqkv = ops.sattn_bwd(qs, mb, ctx, dctx, lse, B, Lq, heads, p=p, keep=keep)
    assert (dqkv.double() - qkv.grad).abs().max().item() < 1e-4 * max(1.0, qkv.grad.abs().max().item())
    from spokennlp_amd import lib as Lb
    with pytest.raises(Lb.AmdsegError):                    # dropout without the masks: refused, not silently undropped
        ops.sattn_fwd(q32, mb, B, Lq, heads, p=p, keep=None)


@pytest.mark.parametrize("Lq,window,ng", [(256, 32, 1), (512, 64, 1), (256, 48, 0)])
def test_split_bf16_band_attention_vs_autograd(dev, Lq, window, ng):
    """the Longformer band (|i - j| <= window or j < nglobal; padded query rows zeroed) in split-bf16 precision, forward and backward"""
    from spokennlp_amd import ops
    torch.manual_seed(window)
    B, heads = 2, 2
    H = heads * 64
    qkv = torch.randn(B * Lq, 3 * H, device=dev, dtype=torch.float64).requires_grad_(True)
    am = torch.ones(B, Lq, device=dev); am[-1, Lq - 70:] = 0
    mb = ((1 - am) * -30000.0)
    idx = torch.arange(Lq, device=dev)
    vis = ((idx[None, :] - idx[:, None]).abs() <= window) | (idx[None, :] < ng)
    q, k, v = [t.view(B, Lq, heads, 64).transpose(1, 2) for t in qkv.view(B, Lq, 3 * H).split(H, -1)]
    s_ = q @ k.transpose(-1, -2) / 8.0 + mb.double().view(B, 1, 1, Lq)
    s_ = s_.masked_fill(~vis[None, None], float("-inf"))
    ref = (torch.softmax(s_, -1) @ v).transpose(1, 2).reshape(B, Lq, H) * am.double()[:, :, None]
    ref = ref.reshape(B * Lq, H)
    dctx = torch.randn(B * Lq, H, device=dev)
    ref.backward(dctx.double())
    q32 = qkv.detach().float().contiguous()
    ctx, lse, qs = ops.sattn_fwd(q32, mb, B, Lq, heads, window=window, nglobal=ng)
    assert (ctx.double() - ref.detach()).abs().max().item() < 5e-5
    dqkv = ops.sattn_bwd(qs, mb, ctx, dctx, lse, B, Lq, heads, window=window, nglobal=ng)
    assert (dqkv.double() - qkv.grad).abs().max().item() < 1e-4 * max(1.0, qkv.grad.abs().max().item())


def test_fp32_attention_dropout_is_consistent_between_forward_and_backward(dev):
    """with the keep-mask fixed by the seed, ctx is LINEAR in V: <dctx, ctx(V)> == <dV, V> (adjoint identity) pins that backward applies
    the same mask as forward; the realised keep rate and the 1/(1-p) scaling are checked on a constant-V probe"""
    from spokennlp_amd import ops
    torch.manual_seed(3)
    B, Lq, heads, p = 2, 128, 2, 0.1
    H = heads * 64
    qkv = torch.randn(B * Lq, 3 * H, device=dev)
    mb = torch.zeros(B, Lq, device=dev)
    ctx, lse = ops.pattn_fwd(qkv, mb, B, Lq, heads, p=p, seed=77)
    ctx2, _ = ops.pattn_fwd(qkv, mb, B, Lq, heads, p=p, seed=77)
    ctx3, _ = ops.pattn_fwd(qkv, mb, B, Lq, heads, p=p, seed=78)
    assert torch.equal(ctx, ctx2) and not torch.equal(ctx, ctx3)
    dctx = torch.randn(B * Lq, H, device=dev)
    dqkv = ops.pattn_bwd(qkv, mb, ctx, dctx, lse, B, Lq, heads, p=p, seed=77)
    V, dV = qkv[:, 2 * H:], dqkv[:, 2 * H:]
    lhs, rhs = (dctx.double() * ctx.double()).sum().item(), (dV.double() * V.double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    ones = qkv.clone(); ones[:, 2 * H:] = 1.0                      # V = 1: ctx = sum_j keep_j p_j / (1 - p)  -> mean 1
    c1, _ = ops.pattn_fwd(ones, mb, B, Lq, heads, p=p, seed=5)
    assert abs(c1.mean().item() - 1.0) < 0.02 and c1.std().item() > 1e-3


@pytest.mark.parametrize("case", ["tiny_L64", "tiny_L128", "tiny_L100_B3"])
def test_eval_parity_vs_reference_golden(dev, case):
    from oracle import bert_ts_oracle as O
    z, sd, batch, arch = load_case(case)
    m = build_model(arch, flags_of(z, "full_eval"), sd, dev).eval()
    m.config.amdseg_precision = "parity"
    random.seed(int(z["full_eval.random_seed"]))
    with torch.no_grad():
        loss, logits, cos = m(**to_dev(batch, dev))
    ref = torch.from_numpy(z["full_eval.logits"])
    d = (logits.cpu() - ref).abs().max().item()
    print(f"{case} parity eval: max|dlogit| {d:.2e}")
    assert d < 1e-3 and abs(loss.item() - float(z["full_eval.loss"])) < 1e-3
    assert (cos.cpu() - torch.from_numpy(z["full_eval.cos"])).abs().max().item() < 1e-3
    assert O.decode_predictions(logits.cpu()[:, 0], batch["labels"][:, 0]) == O.decode_predictions(ref[:, 0], batch["labels"][:, 0])


@pytest.mark.parametrize("variant", ["train_full", "train_eop_matrix", "train_eot_list", "train_focal", "train_wce"])
def test_train_parity_vs_reference_golden(dev, variant):
    """one training step in parity precision: loss and EVERY gradient of the reference to <= 1e-3 relative"""
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, variant), sd, dev).train()
    m.config.amdseg_precision = "parity"
    random.seed(int(z[f"{variant}.random_seed"]))
    loss, logits, cos = m(**to_dev(batch, dev))
    loss.backward()
    ref_loss = float(z[f"{variant}.loss"])
    assert abs(loss.item() - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))
    assert (logits.detach().cpu() - torch.from_numpy(z[f"{variant}.logits"])).abs().max().item() < 1e-3
    params = dict(m.named_parameters())
    worst = 0.0
    for n, gv in zip(z[f"{variant}.gradnorm_names"].tolist(), z[f"{variant}.gradnorm_vals"].tolist()):
        mine = float(params[n].grad.float().norm())
        if gv < 0:
            assert mine == 0.0, n
            continue
        rel = abs(mine - gv) / max(gv, 1e-2)
        worst = max(worst, rel)
        assert rel < 1e-3, (n, mine, gv)
    full = 0
    if variant == "train_full":
        for k in z.files:
            if k.startswith("train_full.grad."):
                n = k[len("train_full.grad."):]
                ref = torch.from_numpy(z[k])
                g = params[n].grad.float().cpu()
                scale = max(float(ref.norm()), 1e-2)
                assert float((g - ref).norm()) / scale < 1e-3, (n, float((g - ref).norm()) / scale)
                full += 1
        assert full > 30
    print(variant, "parity: worst grad-norm rel err", worst, "full gradients compared", full)


def test_parity_training_loop_and_mode_switch(dev):
    """a few fused-AdamW steps in parity precision (the split weight images follow the optimiser), then switching the same model
    between precisions gives consistent logits"""
    z, sd, batch, arch = load_case("tiny_L64")
    m = build_model(arch, flags_of(z, "train_full"), sd, dev, dropout=0.1).train()
    m.config.amdseg_precision = "parity"
    b = to_dev(batch, dev)
    losses = []
    for _ in range(6):
        random.seed(0)
        loss = m(**b)[0]
        loss.backward()
        m.engine().adamw_step(2e-3)
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses
    m.eval()
    outs = {}
    for prec in ("parity", "fp32", "bf16"):
        m.config.amdseg_precision = prec
        random.seed(1)
        with torch.no_grad():
            outs[prec] = m(**b)[1].float().cpu()
    assert (outs["parity"] - outs["fp32"]).abs().max().item() < 1e-3
    assert (outs["bf16"] - outs["fp32"]).abs().max().item() < 0.1


def test_parity_precision_skips_trailing_padding_without_changing_a_bit(dev):
    """the fp32 attention kernels (pattn2) and the three split-bf16 weight-gradient launches take the same padding plan as the bf16 path
    (amdseg_bert_cfg.kend / seq_order / pad_guard / pad_runs): encoder output and layer gradients bit-identical with the plan on and off"""
    from transformers import BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=300, hidden_size=768, num_attention_heads=12, num_hidden_layers=2, intermediate_size=3072,
                     max_position_embeddings=512, num_labels=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    cfg.amdseg_precision = "parity"
    m = M(cfg).to(dev)
    eng = m.engine()
    B, L = 4, 512
    lens = [512, 300, 200, 1]
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 300, (B, L), generator=g).to(dev)
    am = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, :n] = 1
    am = am.to(dev)
    tt = torch.zeros_like(ids)
    dseq = torch.randn(B, L, 768, generator=g).to(dev) * am[:, :, None].float()
    names = [n for n in eng.fp.offsets if ".encoder.layer." in n]

    def run(skip):
        eng.skip_padded_chunks = skip
        eng.skip_padded_rows_bwd = skip
        out, ectx = eng.forward(ids, am, tt, True, seed=7, p_out=0.1)
        assert ectx.get("parity")
        out = out.clone()
        eng.backward(ectx, dseq, accumulate=False)
        torch.cuda.synchronize()
        return out, {n: eng.fp.view(eng.fp.flat_g, n).clone() for n in names}

    out_on, on = run(True)
    assert int(eng._pad_guard.item()) == 0
    out_off, off = run(False)
    assert torch.equal(out_on, out_off)
    for n in names:
        assert float(on[n].abs().max()) > 0
        assert torch.equal(on[n], off[n]), n
