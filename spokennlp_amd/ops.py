"""Thin torch-tensor wrappers over the libamdseg C ABI (tensors are storage only: data_ptr + stream go to HIP).

Every function launches on torch's current HIP stream and raises ``AmdsegError`` on a non-zero return code.
"""
import ctypes as C

import torch

from . import lib as L
from .lib import BF16, F32, EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_ADD_RES, EPI_GELU_BWD  # noqa: F401


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _chk(t, name):
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous device tensor")


def gemm_nt(A, B, epilogue=EPI_NONE, bias=None, R=None, out=None, out2=None, out_dtype=torch.bfloat16):
    """C[M,N] = A[M,K] @ B[N,K]^T with fused epilogue; A, B bf16. Returns C (and C2 for EPI_BIAS_GELU)."""
    _chk(A, "A"); _chk(B, "B")
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    if (epilogue & 0xff) == EPI_BIAS_GELU and out2 is None:
        out2 = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    rc = L.load().amdseg_gemm_nt(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), M, N, K, epilogue,
                                 _p(bias), _p(R), 0 if R is None else R.stride(0), _p(out2),
                                 0 if out2 is None else out2.stride(0), 1 if out.dtype == torch.float32 else 0, _s())
    L.check(rc, "amdseg_gemm_nt")
    return (out, out2) if (epilogue & 0xff) == EPI_BIAS_GELU else out


def gemm_tn_grouped(As, Bs, Cs, accumulate=False, colsums=None):
    """For each i: Cs[i][N_i,K_i] (+)= As[i]^T @ Bs[i]  (As[i]: [M,N_i] bf16, Bs[i]: [M,K_i] bf16, Cs[i] fp32)."""
    n = len(As)
    M = As[0].shape[0]
    vpa = (C.c_void_p * n)(*[a.data_ptr() for a in As])
    vpb = (C.c_void_p * n)(*[b.data_ptr() for b in Bs])
    vpc = (C.c_void_p * n)(*[c.data_ptr() for c in Cs])
    ia = (C.c_int * n)(*[a.stride(0) for a in As])
    ib = (C.c_int * n)(*[b.stride(0) for b in Bs])
    ic = (C.c_int * n)(*[c.stride(0) for c in Cs])
    nn = (C.c_int * n)(*[a.shape[1] for a in As])
    kk = (C.c_int * n)(*[b.shape[1] for b in Bs])
    if colsums is None:
        rc = L.load().amdseg_gemm_tn_grouped(n, vpa, ia, vpb, ib, vpc, ic, nn, kk, M, 1 if accumulate else 0, _s())
        L.check(rc, "amdseg_gemm_tn_grouped")
        return Cs
    # bias gradients: colsums[i] (fp32 [N_i] or None) (+)= column sums of As[i]
    scratch = [None if c is None else torch.empty(max((M + 127) // 128, b.shape[1] // 128) * a.shape[1], dtype=torch.float32, device=a.device)
               for a, b, c in zip(As, Bs, colsums)]
    vpo = (C.c_void_p * n)(*[None if c is None else c.data_ptr() for c in colsums])
    vps = (C.c_void_p * n)(*[None if c is None else c.data_ptr() for c in scratch])
    rc = L.load().amdseg_gemm_tn_grouped_bias(n, vpa, ia, vpb, ib, vpc, ic, nn, kk, M, 1 if accumulate else 0, vpo, vps, _s())
    L.check(rc, "amdseg_gemm_tn_grouped_bias")
    return Cs


def attn_fwd(qkv, mask_bias, B, Lseq, heads, p=0.0, seed=0, need_lse=True, scale=0.125):
    H = heads * 64
    ctx = torch.empty((B * Lseq, H), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device) if need_lse else None
    rc = L.load().amdseg_attn_fwd(_p(qkv), _p(mask_bias), _p(ctx), _p(lse), B, Lseq, heads, scale, p, seed, _s())
    L.check(rc, "amdseg_attn_fwd")
    return ctx, lse


def attn_bwd(qkv, mask_bias, ctx, dctx, lse, B, Lseq, heads, p=0.0, seed=0, scale=0.125):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_attn_bwd(_p(qkv), _p(mask_bias), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, Lseq, heads,
                                  scale, p, seed, _s())
    L.check(rc, "amdseg_attn_bwd")
    return dqkv


def attn_keepmask(B, Lseq, heads, p, seed, device, kend=None):
    """dropout keep masks of one attention layer (amdseg_attn_keepmask): uint8 buffer holding layout A then layout B"""
    lib = L.load()
    keep = torch.empty(lib.amdseg_attn_keepmask_bytes(B, Lseq, heads), dtype=torch.uint8, device=device)
    L.check(lib.amdseg_attn_keepmask(_p(keep), B, Lseq, heads, p, seed, _p(kend), _s()), "amdseg_attn_keepmask")
    return keep


def attn_fwd_keep(qkv, mask_bias, B, Lseq, heads, p, keep, need_lse=True, scale=0.125):
    H = heads * 64
    ctx = torch.empty((B * Lseq, H), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device) if need_lse else None
    L.check(L.load().amdseg_attn_fwd_keep(_p(qkv), _p(mask_bias), _p(ctx), _p(lse), B, Lseq, heads, scale, p, _p(keep), _s()), "amdseg_attn_fwd_keep")
    return ctx, lse


def attn_bwd_keep(qkv, mask_bias, ctx, dctx, lse, B, Lseq, heads, p, keep, scale=0.125):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    L.check(L.load().amdseg_attn_bwd_keep(_p(qkv), _p(mask_bias), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, Lseq, heads,
                                          scale, p, _p(keep), _s()), "amdseg_attn_bwd_keep")
    return dqkv


def attn_keepmask_band(B, Lseq, heads, p, seed, window, nglobal, device):
    """keep masks of a band (Longformer) layer: same buffer layout as attn_keepmask, only the cells the band kernels visit are written"""
    lib = L.load()
    keep = torch.zeros(lib.amdseg_attn_keepmask_bytes(B, Lseq, heads), dtype=torch.uint8, device=device)
    L.check(lib.amdseg_attn_keepmask_band(_p(keep), B, Lseq, heads, p, seed, window, nglobal, _s()), "amdseg_attn_keepmask_band")
    return keep


def attn_band_fwd_keep(qkv, mask_bias, B, Lseq, heads, window, nglobal, p, keep, scale=0.125):
    ctx = torch.empty((B * Lseq, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    L.check(L.load().amdseg_attn_band_fwd_keep(_p(qkv), _p(mask_bias), _p(ctx), _p(lse), B, Lseq, heads, scale, p, _p(keep), window, nglobal, _s()),
            "amdseg_attn_band_fwd_keep")
    return ctx, lse


def attn_band_bwd_keep(qkv, mask_bias, ctx, dctx, lse, B, Lseq, heads, window, nglobal, p, keep, scale=0.125):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    L.check(L.load().amdseg_attn_band_bwd_keep(_p(qkv), _p(mask_bias), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, Lseq, heads, scale, p,
                                               _p(keep), window, nglobal, _s()), "amdseg_attn_band_bwd_keep")
    return dqkv


def attn_list_fwd(qkv, mask_bias, B, Lseq, heads, klist, kcnt, stride, ctx=None, lse=None, scale=0.125, korder=None):
    """BigBird block-list attention (amdseg_attn_list_fwd); klist/kcnt: int32 device tensors [heads, L/64, stride] / [heads, L/64]"""
    H = heads * 64
    if ctx is None:
        ctx = torch.empty((B * Lseq, H), dtype=torch.bfloat16, device=qkv.device)
    if lse is None:
        lse = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_attn_list_fwd(_p(qkv), _p(mask_bias), _p(ctx), _p(lse), B, Lseq, heads, scale, _p(klist), _p(kcnt), stride, _p(korder), _s())
    L.check(rc, "amdseg_attn_list_fwd")
    return ctx, lse


def attn_list_f32(qkv, mask_bias, B, Lseq, heads, klist, kcnt, stride, ctx=None, scale=0.125):
    """fp32 parity-mode block-list attention (amdseg_attn_list_f32): qkv fp32 [B*L, 3*heads*64] -> ctx fp32 [B*L, heads*64]"""
    if ctx is None:
        ctx = torch.empty((B * Lseq, heads * 64), dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_attn_list_f32(_p(qkv), _p(mask_bias), _p(ctx), B, Lseq, heads, scale, _p(klist), _p(kcnt), stride, _s())
    L.check(rc, "amdseg_attn_list_f32")
    return ctx


def attn_list_bwd(qkv, mask_bias, ctx, dctx, lse, B, Lseq, heads, klist, kcnt, qlist, qcnt, stride, dqkv=None, delta=None, scale=0.125,
                  korder=None, qorder=None):
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if delta is None:
        delta = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_attn_list_bwd(_p(qkv), _p(mask_bias), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, Lseq, heads, scale,
                                       _p(klist), _p(kcnt), _p(qlist), _p(qcnt), stride, _p(korder), _p(qorder), _s())
    L.check(rc, "amdseg_attn_list_bwd")
    return dqkv


def attn_band_fwd(qkv, mask_bias, B, Lseq, heads, window, nglobal=1, p=0.0, seed=0, need_lse=True, scale=0.125):
    H = heads * 64
    ctx = torch.empty((B * Lseq, H), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device) if need_lse else None
    rc = L.load().amdseg_attn_band_fwd(_p(qkv), _p(mask_bias), _p(ctx), _p(lse), B, Lseq, heads, scale, p, seed, window, nglobal, _s())
    L.check(rc, "amdseg_attn_band_fwd")
    return ctx, lse


def attn_band_bwd(qkv, mask_bias, ctx, dctx, lse, B, Lseq, heads, window, nglobal=1, p=0.0, seed=0, scale=0.125):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B * heads * Lseq,), dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_attn_band_bwd(_p(qkv), _p(mask_bias), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, Lseq, heads,
                                       scale, p, seed, window, nglobal, _s())
    L.check(rc, "amdseg_attn_band_bwd")
    return dqkv


def attn_band_f32(qkv, mask_bias, B, Lseq, heads, window, nglobal=1, scale=0.125):
    ctx = torch.empty((B * Lseq, heads * 64), dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_attn_band_f32(_p(qkv), _p(mask_bias), _p(ctx), B, Lseq, heads, scale, window, nglobal, _s())
    L.check(rc, "amdseg_attn_band_f32")
    return ctx


# ---- Longformer global row (csrc/longformer.hip): x [B*L, H] bf16/fp32; vec [B, heads, H] fp32; planes [B, heads, L] fp32
def lf_rowvec_dot(x, vec, B, Lseq, add_tok=None, add_bh=None):
    """x: [B*L, H] (rows may be strided: a column block of a wider matrix)"""
    heads, H = vec.shape[1], vec.shape[2]
    out = torch.empty((B, heads, Lseq), dtype=torch.float32, device=x.device)
    rc = L.load().amdseg_lf_rowvec_dot_ld(_p(x), x.stride(0), _p(vec), _p(add_tok), _p(add_bh), _p(out), B, Lseq, H, heads, _dt(x), _s())
    L.check(rc, "amdseg_lf_rowvec_dot_ld")
    return out


def lf_softmax_fwd(scores, p=0.0, seed=0):
    """scores [B, heads, L] fp32 is overwritten with the probabilities; returns (p, dropped p, sum of dropped p [B, heads])."""
    B, heads, Lseq = scores.shape
    pd = torch.empty_like(scores)
    sp = torch.empty((B, heads), dtype=torch.float32, device=scores.device)
    rc = L.load().amdseg_lf_softmax_fwd(_p(scores), _p(pd), _p(sp), B * heads, Lseq, p, seed, _s())
    L.check(rc, "amdseg_lf_softmax_fwd")
    return scores, pd, sp


def lf_softmax_bwd(p_saved, dpd, p=0.0, seed=0):
    """dpd is overwritten with ds; returns (ds, recomputed dropped p)."""
    B, heads, Lseq = p_saved.shape
    pd = torch.empty_like(p_saved)
    rc = L.load().amdseg_lf_softmax_bwd(_p(p_saved), _p(dpd), _p(pd), B * heads, Lseq, p, seed, _s())
    L.check(rc, "amdseg_lf_softmax_bwd")
    return dpd, pd


def lf_wsum(x, coef, H, partials=None):
    B, heads, Lseq = coef.shape
    if partials is None:
        partials = torch.empty((B * (Lseq // 64) * heads * H,), dtype=torch.float32, device=x.device)
    y = torch.empty((B, heads, H), dtype=torch.float32, device=x.device)
    rc = L.load().amdseg_lf_wsum_ld(_p(x), x.stride(0), _p(coef), _p(partials), _p(y), B, Lseq, H, heads, _dt(x), _s())
    L.check(rc, "amdseg_lf_wsum_ld")
    return y


def lf_dx_update(dx, coefA, vecA, coefB, vecB, vt_ws=None, assign=False):
    """dx: [B*L, H] rows (may be strided); assign=True overwrites instead of accumulating"""
    B, heads, Lseq = coefA.shape
    H = vecA.shape[2]
    if vt_ws is None and dx.dtype == torch.bfloat16:
        vt_ws = torch.empty(B * H * 32, dtype=torch.bfloat16, device=dx.device)
    rc = L.load().amdseg_lf_dx_update_ld(_p(dx), dx.stride(0), 1 if assign else 0, _p(coefA), _p(vecA), _p(coefB), _p(vecB), _p(vt_ws),
                                         B, Lseq, H, heads, _dt(dx), _s())
    L.check(rc, "amdseg_lf_dx_update_ld")


# ---- PoNet token mixing (csrc/ponet.hip, csrc/ponet_global.hip)
def ponet_global_scratch(B, Lseq, H, heads, device):
    return torch.empty(int(L.load().amdseg_ponet_global_scratch_floats(B, Lseq, H, heads)), dtype=torch.float32, device=device)


def ponet_global_fwd(hq, hk, coef_mean, mask_bias, B, Lseq, H, heads, p=0.0, seed=0, scratch=None):
    """hq, hk: bf16 column blocks [B*L, H] of the projection (same row stride).  Returns (g [B,H], vecq [B,H], scores [B,heads,L], lse [B,heads])."""
    dev = hk.device
    if scratch is None:
        scratch = ponet_global_scratch(B, Lseq, H, heads, dev)
    f = dict(dtype=torch.float32, device=dev)
    vecq, g = torch.empty(B, H, **f), torch.empty(B, H, **f)
    scores, lse = torch.empty(B, heads, Lseq, **f), torch.empty(B, heads, **f)
    assert hq.stride(0) == hk.stride(0)
    rc = L.load().amdseg_ponet_global_fwd(_p(hq), _p(hk), hk.stride(0), _p(coef_mean), _p(mask_bias), B, Lseq, H, heads, p, seed, _p(scratch),
                                          _p(vecq), _p(scores), _p(lse), _p(g), _s())
    L.check(rc, "amdseg_ponet_global_fwd")
    return g, vecq, scores, lse


def ponet_global_bwd(hk, coef_mean, vecq, scores, lse, dg, dhq, dhk, B, Lseq, H, heads, p=0.0, seed=0, scratch=None, dpd_ws=None):
    """writes the dHq / dHk column blocks (bf16, same row stride) of the projection gradient"""
    dev = hk.device
    if scratch is None:
        scratch = ponet_global_scratch(B, Lseq, H, heads, dev)
    if dpd_ws is None:
        dpd_ws = torch.empty(B, heads, Lseq, dtype=torch.float32, device=dev)
    assert dhq.stride(0) == dhk.stride(0)
    rc = L.load().amdseg_ponet_global_bwd(_p(hk), hk.stride(0), _p(coef_mean), _p(vecq), _p(scores), _p(lse), _p(dg), B, Lseq, H, heads, p, seed,
                                          _p(scratch), _p(dpd_ws), _p(dhq), _p(dhk), dhk.stride(0), _s())
    L.check(rc, "amdseg_ponet_global_bwd")


def ponet_plan(mask_bias, run_start, B, Lseq):
    """work lists of the PoNet pooling kernels for one batch (csrc/ponet.hip): int32 [2 + 2*B*L] on the device, no host sync"""
    work = torch.empty(2 + 2 * B * Lseq, dtype=torch.int32, device=mask_bias.device)
    rc = L.load().amdseg_ponet_plan(_p(mask_bias), _p(run_start), _p(work), B, Lseq, _s())
    L.check(rc, "amdseg_ponet_plan")
    return work


def ponet_pool_fwd(proj, mask_bias, run_start, run_end, work, g, part, parg, ctx, B, Lseq, H):
    rc = L.load().amdseg_ponet_pool_fwd(_p(proj), proj.stride(0), _p(mask_bias), _p(run_start), _p(run_end), _p(work), _p(g), _p(part),
                                        _p(parg), _p(ctx), B, Lseq, H, _s())
    L.check(rc, "amdseg_ponet_pool_fwd")


def ponet_pool_bwd(proj, mask_bias, run_start, run_end, work, g, part, parg, dctx, dproj, psum, B, Lseq, H):
    """returns dg [B, H] fp32: the per-sequence sum of dctx * Ho over the valid tokens (gradient of the global aggregate)"""
    dg = torch.empty(B, H, dtype=torch.float32, device=proj.device)
    rc = L.load().amdseg_ponet_pool_bwd(_p(proj), proj.stride(0), _p(mask_bias), _p(run_start), _p(run_end), _p(work), _p(g), _p(part),
                                        _p(parg), _p(dctx), _p(dproj), _p(dg), _p(psum), B, Lseq, H, _s())
    L.check(rc, "amdseg_ponet_pool_bwd")
    return dg


def embed_ln_fwd(ids, type_ids, pos_ids, word, pos, typ, gamma, beta, Lseq, eps, p=0.0, seed=0, dtype=torch.bfloat16,
                 save=True):
    M = ids.numel()
    H = word.shape[1]
    dev = word.device
    out = torch.empty((M, H), dtype=dtype, device=dev)
    z = torch.empty((M, H), dtype=dtype, device=dev) if save else None
    mean = torch.empty((M,), dtype=torch.float32, device=dev) if save else None
    rstd = torch.empty((M,), dtype=torch.float32, device=dev) if save else None
    rc = L.load().amdseg_embed_ln_fwd(_p(ids), _p(type_ids), _p(pos_ids), _p(word), _p(pos), _p(typ), _p(gamma), _p(beta),
                                      _p(z), _p(out), _p(mean), _p(rstd), M, Lseq, H, word.shape[0], typ.shape[0],
                                      pos.shape[0], eps, p, seed, _dt(out), _s())
    L.check(rc, "amdseg_embed_ln_fwd")
    return out, z, mean, rstd


def embed_bwd(dz, ids, type_ids, pos_ids, dword, dpos, dtyp, Lseq, pad_id=-1, type0_holds_colsum=False):
    """type0_holds_colsum: row 0 of dtyp already holds the column sum of dz (include/amdseg.h: type_vocab < 0)"""
    M, H = dz.shape
    rc = L.load().amdseg_embed_bwd(_p(dz), _p(ids), _p(type_ids), _p(pos_ids), _p(dword), _p(dpos), _p(dtyp), M, Lseq, H,
                                   dword.shape[0], -dtyp.shape[0] if type0_holds_colsum else dtyp.shape[0], dpos.shape[0], pad_id,
                                   _dt(dz), _s())
    L.check(rc, "amdseg_embed_bwd")


def add_ln_fwd(y, resid, gamma, beta, eps, p=0.0, seed=0):
    """y is overwritten with z = resid + dropout(y); returns (out, mean, rstd)."""
    M, H = y.shape
    out = torch.empty_like(y)
    mean = torch.empty((M,), dtype=torch.float32, device=y.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=y.device)
    rc = L.load().amdseg_add_ln_fwd(_p(y), _p(resid), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), M, H, eps, p, seed,
                                    _dt(y), _s())
    L.check(rc, "amdseg_add_ln_fwd")
    return out, mean, rstd


def add_ln_fwd_keepmask(y, resid, gamma, beta, eps, p, seed, B, Lseq, heads, p_attn, seed_attn, kend=None, window=0, nglobal=0, drop_bits=None,
                        keep_z=True):
    """add_ln_fwd AND the keep masks of an attention layer as one launch (amdseg_add_ln_fwd_keepmask); returns (out, mean, rstd, keep)"""
    M, H = y.shape
    lib = L.load()
    out = torch.empty_like(y)
    mean = torch.empty((M,), dtype=torch.float32, device=y.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=y.device)
    mk = torch.zeros if window > 0 else torch.empty
    keep = mk(lib.amdseg_attn_keepmask_bytes(B, Lseq, heads), dtype=torch.uint8, device=y.device)
    rc = lib.amdseg_add_ln_fwd_keepmask(_p(y), _p(resid), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), M, H, eps, p, seed, _dt(y),
                                        _p(drop_bits), 1 if keep_z else 0, _p(keep), B, Lseq, heads, p_attn, seed_attn, _p(kend), window, nglobal, _s())
    L.check(rc, "amdseg_add_ln_fwd_keepmask")
    return out, mean, rstd, keep


LNB_ROWS = 16      # rows per workgroup of ln_bwd / rowdot_bwd (csrc/elementwise.hip)


def ln_partials_numel(M, H):
    return 3 * ((M + LNB_ROWS - 1) // LNB_ROWS) * H


def ln_bwd(dy, z, mean, rstd, gamma, p=0.0, seed=0, dgamma=None, dbeta=None, dbias=None, accumulate=False, partials=None):
    M, H = dy.shape
    dz = torch.empty_like(dy)
    dbr = torch.empty_like(dy) if p > 0 else None
    if partials is None and (dgamma is not None or dbeta is not None or dbias is not None):
        partials = torch.empty((ln_partials_numel(M, H),), dtype=torch.float32, device=dy.device)
    rc = L.load().amdseg_ln_bwd(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dbr), _p(partials), _p(dgamma),
                                _p(dbeta), _p(dbias), M, H, p, seed, 1 if accumulate else 0, _dt(dy), _s())
    L.check(rc, "amdseg_ln_bwd")
    return dz, (dbr if dbr is not None else dz)


def colsum(x, out=None, accumulate=False, ncols=None):
    M = x.shape[0]
    N = ncols or x.shape[1]
    if out is None:
        out = torch.zeros((N,), dtype=torch.float32, device=x.device)
    partials = torch.empty((((M + 127) // 128) * N,), dtype=torch.float32, device=x.device)
    rc = L.load().amdseg_colsum(_p(x), x.stride(0), _p(partials), _p(out), M, N, 1 if accumulate else 0, _dt(x), _s())
    L.check(rc, "amdseg_colsum")
    return out


def dropout(x, p, seed, out_dtype=None):
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    rc = L.load().amdseg_dropout(_p(x), _p(y), x.numel(), p, seed, _dt(x), _dt(y), _s())
    L.check(rc, "amdseg_dropout")
    return y


def cast(x, dtype):
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    rc = L.load().amdseg_cast(_p(x), _p(y), x.numel(), _dt(x), _dt(y), _s())
    L.check(rc, "amdseg_cast")
    return y


def cast_transpose(W, Wb=None, Wt=None):
    N, K = W.shape
    rc = L.load().amdseg_cast_transpose(_p(W), _p(Wb), _p(Wt), N, K, _s())
    L.check(rc, "amdseg_cast_transpose")


def rowdot_fwd(x, W, b):
    M, H = x.shape
    Cc = W.shape[0]
    out = torch.empty((M, Cc), dtype=torch.float32, device=x.device)
    rc = L.load().amdseg_rowdot_fwd(_p(x), _p(W), _p(b), _p(out), M, H, Cc, _dt(x), _s())
    L.check(rc, "amdseg_rowdot_fwd")
    return out


def rowdot_bwd(x, W, dlogits, dW=None, db=None, need_dx=True, accumulate=False):
    M, H = x.shape
    Cc = W.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    partials = None
    if dW is not None or db is not None:
        partials = torch.empty((((M + LNB_ROWS - 1) // LNB_ROWS) * (Cc * H + Cc),), dtype=torch.float32, device=x.device)
    rc = L.load().amdseg_rowdot_bwd(_p(x), _p(W), _p(dlogits), _p(dx), _p(partials), _p(dW), _p(db), M, H, Cc,
                                    1 if accumulate else 0, _dt(x), _s())
    L.check(rc, "amdseg_rowdot_bwd")
    return dx


def adamw(p, g, m, v, shadow, lr, beta1, beta2, eps, wd, step, gscale=None, zero_grad=False, chunk_flags=None):
    rc = L.load().amdseg_adamw(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), lr, beta1, beta2, eps, wd, step,
                               _p(gscale), 1 if zero_grad else 0, _p(chunk_flags), _s())
    L.check(rc, "amdseg_adamw")


def sumsq(x, out, partials, accumulate=False):
    rc = L.load().amdseg_sumsq(_p(x), x.numel(), _p(partials), _p(out), 1 if accumulate else 0, _s())
    L.check(rc, "amdseg_sumsq")


def clip_coef(sumsq_t, max_norm, extra_scale, coef, norm=None):
    rc = L.load().amdseg_clip_coef(_p(sumsq_t), max_norm, extra_scale, _p(coef), _p(norm), _s())
    L.check(rc, "amdseg_clip_coef")


def scale_(x, coef):
    rc = L.load().amdseg_scale(_p(x), x.numel(), _p(coef), _s())
    L.check(rc, "amdseg_scale")


def split3(x, out, order=0):
    """fp32 [M, K] -> bf16 [M, 3K] split image ([hi | hi | lo] for order 0, [hi | lo | hi] for order 1); csrc/parity.hip"""
    M, K = x.shape
    rc = L.load().amdseg_split3(_p(x), x.stride(0), _p(out), M, K, order, _s())
    L.check(rc, "amdseg_split3")
    return out


def split3_transpose(W, out):
    N, K = W.shape
    rc = L.load().amdseg_split3_transpose(_p(W), _p(out), N, K, _s())
    L.check(rc, "amdseg_split3_transpose")
    return out


def sattn_fwd(qkv, mask_bias, B, Lq, heads, p=0.0, keep=None, window=0, nglobal=0):
    """"parity" precision attention on split-bf16 products (amdseg_sattn_fwd): qkv fp32 [B*L, 3H] -> (ctx fp32, lse, the split image)"""
    H = heads * 64
    qs = split3(qkv, torch.empty(B * Lq, 9 * H, dtype=torch.bfloat16, device=qkv.device))
    ctx = torch.empty(B * Lq, H, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B * heads * Lq, dtype=torch.float32, device=qkv.device)
    L.check(L.load().amdseg_sattn_fwd(_p(qs), 9 * H, 6 * H, _p(mask_bias), _p(ctx), _p(lse), B, Lq, heads, 0.125, p, _p(keep), window, nglobal, _s()),
            "amdseg_sattn_fwd")
    return ctx, lse, qs


def sattn_bwd(qs, mask_bias, ctx, dctx, lse, B, Lq, heads, p=0.0, keep=None, window=0, nglobal=0):
    H = heads * 64
    dos = split3(dctx, torch.empty(B * Lq, 3 * H, dtype=torch.bfloat16, device=dctx.device))
    dqkv = torch.empty(B * Lq, 3 * H, dtype=torch.float32, device=dctx.device)
    delta = torch.empty_like(lse)
    L.check(L.load().amdseg_sattn_bwd(_p(qs), 9 * H, 6 * H, _p(mask_bias), _p(ctx), _p(dos), 3 * H, 2 * H, _p(lse), _p(delta), _p(dqkv), B, Lq,
                                      heads, 0.125, p, _p(keep), window, nglobal, _s()), "amdseg_sattn_bwd")
    return dqkv


def pattn_fwd(qkv, mask_bias, B, Lq, heads, p=0.0, seed=0):
    H = heads * 64
    ctx = torch.empty(B * Lq, H, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B * heads * Lq, dtype=torch.float32, device=qkv.device)
    rc = L.load().amdseg_pattn_fwd(_p(qkv), _p(mask_bias), _p(ctx), _p(lse), B, Lq, heads, 0.125, p, seed, _s())
    L.check(rc, "amdseg_pattn_fwd")
    return ctx, lse


def pattn_bwd(qkv, mask_bias, ctx, dctx, lse, B, Lq, heads, p=0.0, seed=0):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    rc = L.load().amdseg_pattn_bwd(_p(qkv), _p(mask_bias), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, Lq, heads, 0.125, p, seed, _s())
    L.check(rc, "amdseg_pattn_bwd")
    return dqkv
