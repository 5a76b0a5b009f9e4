"""ctypes binding of libamdseg.so (C ABI declared in include/amdseg.h).

The product path has no CPU / PyTorch fallback: if the HIP library is missing or an entry point is absent this
module raises immediately (build it with ``python -m spokennlp_amd.build`` or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMDSEG_LIB") or os.path.join(_HERE, "libamdseg.so")

BF16, F32, F32S = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_ADD_RES, EPI_GELU_BWD = 0, 1, 2, 3, 4
EPI_ACT_TANH = 0x100      # OR-ed into EPI_BIAS_GELU / EPI_GELU_BWD: gelu_new
EPI_DERIV_U8 = 0x400      # with EPI_KEEP_DERIV: the derivative as one byte per element, q = round((g' + 0.135) * 200)
EPI_KEEP_DERIV = 0x200    # OR-ed into EPI_BIAS_GELU: out2 = gelu'(pre-activation); into EPI_GELU_BWD: R is that derivative (C = (A B^T) * R)
ABI_VERSION = 14

vp, i32, f32, u64, sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t


class BertCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("heads", C.c_int32), ("I", C.c_int32),
                ("ln_eps", f32), ("p_hidden", f32), ("p_attn", f32), ("seed", u64),
                ("accumulate_grads", C.c_int32), ("dtype", C.c_int32), ("window", C.c_int32), ("nglobal", C.c_int32),
                ("nproj", C.c_int32), ("mixer", C.c_int32), ("phase", C.c_int32), ("act", C.c_int32), ("kend", vp), ("seq_order", vp), ("pad_guard", vp), ("pad_runs", vp), ("pad_counts", vp), ("ctx", vp)]


class LayerParams(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv", "wo", "w1", "w2", "wqkv_t", "wo_t", "w1_t", "w2_t",
                                  "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class LayerGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class LayerActs(C.Structure):
    _fields_ = [(n, vp) for n in ("x_in", "qkv", "ctx", "z1", "x1", "u", "h", "z2", "x_out",
                                  "lse", "mean1", "rstd1", "mean2", "rstd2", "xs", "ctx_s", "x1_s", "h_s", "keep", "qkv_s", "drop1", "drop2",
                                  "keep_next")] + [("keep_ready", C.c_int)]      # ABI 14: the next layer's masks from this layer's LayerNorm launch


class LayerWs(C.Structure):
    _fields_ = [(n, vp) for n in ("dz2", "dbr2", "du", "dx1", "dz1", "dbr1", "dctx", "dqkv", "delta", "partials",
                                  "d_out_s", "du_s", "d_ao_s", "dqkv_s", "dctx_s")]


# name -> argtypes (restype is always int unless listed in _RESTYPE); mirrors include/amdseg.h one for one
_PROTOS = {
    "amdseg_abi_version": [],
    "amdseg_error_string": [i32],
    "amdseg_gemm_nt": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp],
    "amdseg_gemm_tn_grouped": [i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(i32), C.POINTER(vp),
                               C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32, vp],
    "amdseg_gemm_tn_grouped_bias": [i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(i32), C.POINTER(vp),
                                    C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32, C.POINTER(vp), C.POINTER(vp), vp],
    "amdseg_gemm_f32_nt": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp],
    "amdseg_attn_f32": [vp, vp, vp, i32, i32, i32, f32, vp],
    "amdseg_attn_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, vp],
    "amdseg_attn_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, vp],
    "amdseg_attn_keepmask_bytes": [i32, i32, i32],
    "amdseg_attn_keepmask": [vp, i32, i32, i32, f32, u64, vp, vp],
    "amdseg_attn_keepmask_band": [vp, i32, i32, i32, f32, u64, i32, i32, vp],
    "amdseg_attn_fwd_keep": [vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp],
    "amdseg_attn_bwd_keep": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp],
    "amdseg_attn_band_fwd_keep": [vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32, vp],
    "amdseg_attn_band_bwd_keep": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32, vp],
    "amdseg_sattn_fwd": [vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32, vp],
    "amdseg_sattn_bwd": [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32, vp],
    "amdseg_attn_band_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, i32, i32, vp],
    "amdseg_attn_band_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, i32, i32, vp],
    "amdseg_attn_band_f32": [vp, vp, vp, i32, i32, i32, f32, i32, i32, vp],
    "amdseg_lf_rowvec_dot": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_lf_softmax_fwd": [vp, vp, vp, i32, i32, f32, u64, vp],
    "amdseg_lf_softmax_bwd": [vp, vp, vp, i32, i32, f32, u64, vp],
    "amdseg_lf_wsum": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_lf_dx_update": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_lf_rowvec_dot_ld": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_lf_wsum_ld": [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_lf_dx_update_ld": [vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_ponet_plan": [vp, vp, vp, i32, i32, vp],
    "amdseg_ponet_pool_fwd": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "amdseg_ponet_pool_bwd": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "amdseg_embed_ln_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, u64, i32, vp],
    "amdseg_scatter_rows_sorted": [vp, vp, vp, vp, i32, i32, i32, C.c_long, i32, vp],
    "amdseg_embed_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "amdseg_add_ln_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, u64, i32, vp],
    "amdseg_add_ln_fwd_keepmask": [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, u64, i32, vp, i32, vp, i32, i32, i32, f32, u64, vp, i32, i32, vp],
    "amdseg_ln_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, u64, i32, i32, vp],
    "amdseg_colsum": [vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_pad_plan": [vp, i32, i32, vp, vp, vp, vp, vp, f32, vp],
    "amdseg_pad_rows_guard": [vp, vp, i32, i32, i32, vp, vp],
    "amdseg_dropout": [vp, vp, sz, f32, u64, i32, i32, vp],
    "amdseg_cast": [vp, vp, sz, i32, i32, vp],
    "amdseg_cast_transpose": [vp, vp, vp, i32, i32, vp],
    "amdseg_cast_transpose_batched": [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp],
    "amdseg_cast_transpose_batched_if": [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp, vp],
    "amdseg_weights_changed": [vp, sz, vp, vp, vp],
    "amdseg_attn_list_f32": [vp, vp, vp, i32, i32, i32, f32, vp, vp, i32, vp],
    "amdseg_attn_list_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, i32, vp, vp],
    "amdseg_attn_list_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, i32, vp, vp, vp],
    "amdseg_split3": [vp, i32, vp, i32, i32, i32, vp],
    "amdseg_split3_transpose": [vp, vp, i32, i32, vp],
    "amdseg_split3_weights_batched": [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp],
    "amdseg_pattn_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, vp],
    "amdseg_pattn_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, vp],
    "amdseg_lf_global_q": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "amdseg_lf_global_out": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_lf_global_bwd_a": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_lf_global_bwd_a_ro": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_lf_global_bwd_rest": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "amdseg_ponet_global_scratch_floats": [i32, i32, i32, i32],
    "amdseg_ponet_global_fwd": [vp, vp, i32, vp, vp, i32, i32, i32, i32, f32, u64, vp, vp, vp, vp, vp, vp],
    "amdseg_ponet_global_bwd": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp, vp, vp, vp, i32, vp],
    "amdseg_lf_global_bwd_dx": [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "amdseg_lf_global_bwd_w": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_lf_dx_prep": [vp, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_lf_dx_apply": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_heads_fwd": [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, C.c_long, C.c_long, C.c_long, i32, i32, i32, f32, vp, vp,
                         C.c_long, C.c_long, i32, i32, f32, f32, f32, vp],
    "amdseg_heads_bwd_ce": [vp, i32, i32, i32, vp, vp, f32, vp, vp],
    "amdseg_heads_fwd_focal": [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, C.c_long, C.c_long, C.c_long, i32, i32, i32, f32, vp, vp,
                               C.c_long, C.c_long, i32, i32, f32, f32, f32, f32, vp],
    "amdseg_heads_bwd_ce_focal": [vp, i32, i32, i32, vp, vp, f32, f32, vp, vp],
    "amdseg_heads_bwd_rows": [vp, vp, i32, i32, vp, vp, C.c_long, C.c_long, C.c_long, i32, i32, i32, f32, vp, vp, C.c_long, C.c_long, i32, i32,
                              vp, vp, f32, f32, i32, vp, C.c_size_t, vp],
    "amdseg_ctx_create": [C.POINTER(vp)],
    "amdseg_ctx_destroy": [vp],
    "amdseg_ctx_bind": [vp],
    "amdseg_ctx_set_cu_budget": [vp, i32],
    "amdseg_ctx_cu_budget": [vp],
    "amdseg_ctx_force_small_tile": [vp, i32],
    "amdseg_ctx_prof_enable": [vp, i32],
    "amdseg_ctx_prof_reset": [vp],
    "amdseg_ctx_prof_read": [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)],
    "amdseg_allreduce_unique_id": [vp],
    "amdseg_allreduce_init": [C.POINTER(vp), vp, i32, i32],
    "amdseg_allreduce_bucket": [vp, vp, sz, i32, vp],
    "amdseg_allreduce_wait": [vp, vp],
    "amdseg_allreduce_info": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(sz)],
    "amdseg_allreduce_destroy": [vp],
    "amdseg_rowdot_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "amdseg_rowdot_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "amdseg_adamw": [vp, vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, vp, i32, vp, vp],
    "amdseg_sumsq": [vp, sz, vp, vp, i32, vp],
    "amdseg_clip_coef": [vp, f32, f32, vp, vp, vp],
    "amdseg_scale": [vp, sz, vp, vp],
    "amdseg_bert_layer_fwd": [C.POINTER(BertCfg), C.POINTER(LayerParams), C.POINTER(LayerActs), vp, i32, vp],
    "amdseg_bert_layer_bwd": [C.POINTER(BertCfg), C.POINTER(LayerParams), C.POINTER(LayerGrads), C.POINTER(LayerActs),
                              C.POINTER(LayerWs), vp, vp, vp, i32, vp],
}
_RESTYPE = {"amdseg_error_string": C.c_char_p, "amdseg_attn_keepmask_bytes": C.c_size_t, "amdseg_ponet_global_scratch_floats": C.c_size_t}

EXPORTS = tuple(_PROTOS)
_lib = None


class AmdsegError(RuntimeError):
    pass


def load(path=None):
    """Load libamdseg.so and bind every declared symbol; raises AmdsegError if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch ships its own libamdhip64.so.7; it must be in the process BEFORE libamdseg.so is dlopen'ed so that both share
    # ONE HIP runtime (and one device context / stream namespace).  Loading ours first would pull /opt/rocm's copy.
    import torch  # noqa: F401
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise AmdsegError(f"libamdseg.so not found at {p}: the HIP extension is required (no CPU fallback); "
                          f"run `python -m spokennlp_amd.build`")
    lib = C.CDLL(p)
    for name, args in _PROTOS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AmdsegError(f"libamdseg.so does not export {name}") from e
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    v = lib.amdseg_abi_version()
    if v != ABI_VERSION:
        raise AmdsegError(f"libamdseg.so ABI version mismatch: the library says {v}, this binding is written for {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().amdseg_error_string(rc)
        raise AmdsegError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")


class Ctx:
    """the explicit library context (include/amdseg.h, amdseg_ctx_*): CU budget of the tile rules, small-tile test hook, launch timer.  One per engine;
    `ptr` goes into BertCfg.ctx for the composite calls, `bound()` makes it the calling thread's context for the cfg-less entry points."""

    def __init__(self):
        h = vp()
        check(load().amdseg_ctx_create(C.byref(h)), "amdseg_ctx_create")
        self.ptr = h.value

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.amdseg_ctx_destroy(self.ptr)
        self.ptr = None

    __del__ = close

    def set_cu_budget(self, cus):
        return load().amdseg_ctx_set_cu_budget(self.ptr, int(cus))

    def cu_budget(self):
        return load().amdseg_ctx_cu_budget(self.ptr)

    def force_small_tile(self, v):
        return load().amdseg_ctx_force_small_tile(self.ptr, int(v))

    def prof_enable(self, on):
        return load().amdseg_ctx_prof_enable(self.ptr, 1 if on else 0)

    def prof_reset(self):
        return load().amdseg_ctx_prof_reset(self.ptr)

    def prof_read(self, cls):
        us, work, n = C.c_double(), C.c_double(), C.c_longlong()
        check(load().amdseg_ctx_prof_read(self.ptr, cls, C.byref(us), C.byref(work), C.byref(n)), "amdseg_ctx_prof_read")
        return us.value, work.value, n.value

    def bound(self):
        """context manager: this context is the calling thread's for the cfg-less entry points (amdseg_adamw, amdseg_gemm_nt, ...) inside the block"""
        ctx = self

        class _B:
            def __enter__(self_):
                load().amdseg_ctx_bind(ctx.ptr)
                return ctx

            def __exit__(self_, *exc):
                load().amdseg_ctx_bind(None)
                return False
        return _B()
