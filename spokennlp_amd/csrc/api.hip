// extern "C" surface of libamdseg (declared in include/amdseg.h) + the composite BertLayer forward/backward drivers.
#include "../../include/amdseg.h"
#include "amdseg_internal.h"
#include <algorithm>
#include "common.h"
#include "keepmask.h"      // km_pairs_with_rows: where layer li + 1's keep masks ride in layer li's LayerNorm launch

#define S(x) ((hipStream_t)(x))

extern "C" {

int amdseg_abi_version(void) { return AMDSEG_ABI_VERSION; }

const char* amdseg_error_string(int code) {
    switch (code) {
        case AMDSEG_OK: return "ok";
        case AMDSEG_ERR_SHAPE: return "amdseg: unsupported or misaligned shape";
        case AMDSEG_ERR_ARG: return "amdseg: bad argument (null pointer / enum / missing workspace)";
        case AMDSEG_ERR_LAUNCH: return "amdseg: kernel launch failed";
        default: {
            if (code == AMDSEG_ERR_COMM_LIB || (code > AMDSEG_ERR_COMM_BASE && code < AMDSEG_ERR_COMM_BASE + 100)) {
                const char* m = amdseg_comm_error_string_impl(code);
                return m ? m : "amdseg: RCCL call failed";
            }
            return hipGetErrorString((hipError_t)code);
        }
    }
}

int amdseg_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epilogue,
                   const float* bias, const void* R, int ldr, void* C2, int ldc2, int out_fp32, amdseg_stream_t stream) {
    return amdseg_gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, R, ldr, C2, ldc2, out_fp32, S(stream));
}
int amdseg_gemm_tn_grouped(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                           float* const* C, const int* ldc, const int* N, const int* K, int M, int accumulate,
                           amdseg_stream_t stream) {
    return amdseg_gemm_tn_grouped_impl(nprob, A, lda, B, ldb, C, ldc, N, K, M, accumulate, S(stream));
}
int amdseg_gemm_tn_grouped_bias(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* C, const int* ldc, const int* N, const int* K, int M, int accumulate,
                                float* const* colsum_out, float* const* colsum_scratch, amdseg_stream_t stream) {
    return amdseg_gemm_tn_grouped_bias_impl(nprob, A, lda, B, ldb, C, ldc, N, K, M, accumulate, colsum_out, colsum_scratch, S(stream));
}
int amdseg_gemm_f32_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int epilogue,
                       const float* bias, amdseg_stream_t stream) {
    return amdseg_gemm_f32_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, S(stream));
}
int amdseg_attn_f32(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale,
                    amdseg_stream_t stream) {
    return amdseg_attn_f32_impl(qkv, mask_bias, ctx, B, L, heads, 64, scale, 0, 0, S(stream));
}
int amdseg_attn_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                    float dropout_p, uint64_t seed, amdseg_stream_t stream) {
    return amdseg_attn_fwd_impl(qkv, mask_bias, ctx, lse, B, L, heads, scale, dropout_p, seed, 0, 0, S(stream));
}
int amdseg_attn_bwd(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                    float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, uint64_t seed,
                    amdseg_stream_t stream) {
    return amdseg_attn_bwd_impl(qkv, mask_bias, ctx, dctx, lse, delta_ws, dqkv, B, L, heads, scale, dropout_p, seed, 0, 0, S(stream));
}
size_t amdseg_attn_keepmask_bytes(int B, int L, int heads) { return amdseg_attn_keepmask_bytes_impl(B, L, heads); }
int amdseg_attn_keepmask(void* keep, int B, int L, int heads, float dropout_p, uint64_t seed, const int32_t* kend, amdseg_stream_t stream) {
    return amdseg_attn_keepmask_impl(keep, B, L, heads, dropout_p, seed, kend, S(stream));
}
int amdseg_attn_keepmask_band(void* keep, int B, int L, int heads, float dropout_p, uint64_t seed, int window, int nglobal, amdseg_stream_t stream) {
    if (window <= 0) return AMDSEG_ERR_ARG;
    return amdseg_attn_keepmask_impl(keep, B, L, heads, dropout_p, seed, nullptr, S(stream), window, nglobal);
}
int amdseg_attn_fwd_keep(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                         float dropout_p, const void* keep, amdseg_stream_t stream) {
    return amdseg_attn_fwd_impl(qkv, mask_bias, ctx, lse, B, L, heads, scale, dropout_p, 0, 0, 0, S(stream), nullptr, nullptr, keep);
}
int amdseg_attn_bwd_keep(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, const void* keep,
                         amdseg_stream_t stream) {
    return amdseg_attn_bwd_impl(qkv, mask_bias, ctx, dctx, lse, delta_ws, dqkv, B, L, heads, scale, dropout_p, 0, 0, 0, S(stream), nullptr,
                                nullptr, nullptr, keep);
}
int amdseg_sattn_fwd(const void* qs, int ldq, int lo_q, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale,
                     float dropout_p, const void* keep, int window, int nglobal, amdseg_stream_t stream) {
    return amdseg_sattn_fwd_impl(qs, ldq, lo_q, mask_bias, ctx, lse, B, L, heads, scale, dropout_p, keep, window, nglobal, S(stream));
}
int amdseg_sattn_bwd(const void* qs, int ldq, int lo_q, const float* mask_bias, const float* ctx, const void* dos, int ldo, int lo_o,
                     const float* lse, float* delta_ws, float* dqkv, int B, int L, int heads, float scale, float dropout_p, const void* keep,
                     int window, int nglobal, amdseg_stream_t stream) {
    return amdseg_sattn_bwd_impl(qs, ldq, lo_q, mask_bias, ctx, dos, ldo, lo_o, lse, delta_ws, dqkv, B, L, heads, scale, dropout_p, keep, window,
                                 nglobal, S(stream));
}
int amdseg_attn_band_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                         float dropout_p, uint64_t seed, int window, int nglobal, amdseg_stream_t stream) {
    if (window <= 0) return AMDSEG_ERR_ARG;
    return amdseg_attn_fwd_impl(qkv, mask_bias, ctx, lse, B, L, heads, scale, dropout_p, seed, window, nglobal, S(stream));
}
int amdseg_attn_band_bwd(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, uint64_t seed,
                         int window, int nglobal, amdseg_stream_t stream) {
    if (window <= 0) return AMDSEG_ERR_ARG;
    return amdseg_attn_bwd_impl(qkv, mask_bias, ctx, dctx, lse, delta_ws, dqkv, B, L, heads, scale, dropout_p, seed, window, nglobal, S(stream));
}
int amdseg_attn_band_fwd_keep(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                              float dropout_p, const void* keep, int window, int nglobal, amdseg_stream_t stream) {
    if (window <= 0) return AMDSEG_ERR_ARG;
    return amdseg_attn_fwd_impl(qkv, mask_bias, ctx, lse, B, L, heads, scale, dropout_p, 0, window, nglobal, S(stream), nullptr, nullptr, keep);
}
int amdseg_attn_band_bwd_keep(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                              float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, const void* keep,
                              int window, int nglobal, amdseg_stream_t stream) {
    if (window <= 0) return AMDSEG_ERR_ARG;
    return amdseg_attn_bwd_impl(qkv, mask_bias, ctx, dctx, lse, delta_ws, dqkv, B, L, heads, scale, dropout_p, 0, window, nglobal, S(stream),
                                nullptr, nullptr, nullptr, keep);
}
int amdseg_attn_band_f32(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale, int window,
                         int nglobal, amdseg_stream_t stream) {
    if (window <= 0) return AMDSEG_ERR_ARG;
    return amdseg_attn_f32_impl(qkv, mask_bias, ctx, B, L, heads, 64, scale, window, nglobal, S(stream));
}
int amdseg_lf_rowvec_dot(const void* x, const float* vec, const float* add_tok, const float* add_bh, float* out, int B, int L, int H,
                         int heads, int dtype, amdseg_stream_t stream) {
    return amdseg_lf_rowvec_dot_impl(x, vec, add_tok, add_bh, out, B, L, H, heads, dtype, H, S(stream));
}
int amdseg_lf_softmax_fwd(float* s_inout_p, float* pd, float* sp, int rows, int L, float dropout_p, uint64_t seed,
                          amdseg_stream_t stream) {
    return amdseg_lf_softmax_fwd_impl(s_inout_p, pd, sp, rows, L, dropout_p, seed, S(stream));
}
int amdseg_lf_softmax_bwd(const float* p_saved, float* dpd_inout_ds, float* pd, int rows, int L, float dropout_p, uint64_t seed,
                          amdseg_stream_t stream) {
    return amdseg_lf_softmax_bwd_impl(p_saved, dpd_inout_ds, pd, rows, L, dropout_p, seed, S(stream));
}
int amdseg_lf_wsum(const void* x, const float* coef, float* partials, float* y, int B, int L, int H, int heads, int dtype,
                   amdseg_stream_t stream) {
    return amdseg_lf_wsum_impl(x, coef, partials, y, B, L, H, heads, dtype, H, S(stream));
}
int amdseg_lf_dx_update(void* dx, const float* coefA, const float* vecA, const float* coefB, const float* vecB, void* vt_ws, int B,
                        int L, int H, int heads, int dtype, amdseg_stream_t stream) {
    return amdseg_lf_dx_update_impl(dx, coefA, vecA, coefB, vecB, vt_ws, B, L, H, heads, dtype, H, 0, S(stream));
}
int amdseg_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                        const float* pos, const float* type, const float* gamma, const float* beta, void* z, void* out,
                        float* mean, float* rstd, int M, int L, int H, int vocab, int type_vocab, int npos, float eps,
                        float dropout_p, uint64_t seed, int dtype, amdseg_stream_t stream) {
    return amdseg_embed_ln_fwd_impl(ids, type_ids, word, pos, type, gamma, beta, z, out, mean, rstd, M, L, H, vocab, type_vocab,
                                    npos, pos_ids, eps, dropout_p, seed, dtype, S(stream));
}
int amdseg_scatter_rows_sorted(const void* dz, const int64_t* keys, const int64_t* order, float* table, int M, int H, int nrows,
                               long skip_key, int dtype, amdseg_stream_t stream) {
    return amdseg_scatter_rows_sorted_impl(dz, keys, order, table, M, H, nrows, skip_key, dtype, S(stream));
}
int amdseg_embed_bwd(const void* dz, const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, float* dword,
                     float* dpos, float* dtype_emb, int M, int L, int H, int vocab, int type_vocab, int npos, int pad_id,
                     int dtype, amdseg_stream_t stream) {
    return amdseg_embed_bwd_impl(dz, ids, type_ids, pos_ids, dword, dpos, dtype_emb, M, L, H, vocab, type_vocab, npos, pad_id,
                                 dtype, S(stream));
}
int amdseg_add_ln_fwd(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out, float* mean,
                      float* rstd, int M, int H, float eps, float dropout_p, uint64_t seed, int dtype,
                      amdseg_stream_t stream) {
    return amdseg_add_ln_fwd_impl(y_inout_z, resid, gamma, beta, out, mean, rstd, M, H, eps, dropout_p, seed, dtype, S(stream));
}
int amdseg_add_ln_fwd_keepmask(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out, float* mean,
                               float* rstd, int M, int H, float eps, float dropout_p, uint64_t seed, int dtype,
                               void* drop_bits, int keep_z,
                               void* keep, int B, int L, int heads, float attn_dropout_p, uint64_t attn_seed, const int32_t* kend,
                               int window, int nglobal, amdseg_stream_t stream) {
    return amdseg_add_ln_fwd_km_impl(y_inout_z, resid, gamma, beta, out, mean, rstd, M, H, eps, dropout_p, seed, dtype, S(stream), drop_bits, keep_z != 0,
                                     keep, B, L, heads, attn_dropout_p, attn_seed, kend, window, nglobal);
}
int amdseg_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
                  void* dbranch, float* partials, float* dgamma, float* dbeta, float* dbias, int M, int H,
                  float dropout_p, uint64_t seed, int accumulate, int dtype, amdseg_stream_t stream) {
    return amdseg_ln_bwd_impl(dy, z, mean, rstd, gamma, dz, dbranch, partials, dgamma, dbeta, dbias, M, H, dropout_p, seed,
                              accumulate, dtype, S(stream));
}
int amdseg_colsum(const void* x, int ld, float* partials, float* out, int M, int N, int accumulate, int dtype,
                  amdseg_stream_t stream) {
    return amdseg_colsum_impl(x, ld, partials, out, M, N, accumulate, dtype, S(stream));
}
int amdseg_dropout(const void* x, void* y, size_t n, float p, uint64_t seed, int dtype_in, int dtype_out,
                   amdseg_stream_t stream) {
    return amdseg_dropout_impl(x, y, n, p, seed, dtype_in, dtype_out, S(stream));
}
int amdseg_pad_plan(const int64_t* attention_mask, int B, int L, int32_t* kend, int32_t* seq_order, int32_t* pad_runs, int32_t* pad_counts,
                    float* mask_bias, float bias, amdseg_stream_t stream) {
    return amdseg_pad_plan_impl(attention_mask, B, L, kend, seq_order, pad_runs, pad_counts, mask_bias, bias, S(stream));
}
int amdseg_pad_rows_guard(const float* x, const int32_t* kend, int B, int L, int H, int32_t* guard, amdseg_stream_t stream) {
    return amdseg_pad_rows_guard_impl(x, kend, B, L, H, guard, S(stream));
}
int amdseg_cast(const void* x, void* y, size_t n, int dtype_in, int dtype_out, amdseg_stream_t stream) {
    return amdseg_cast_impl(x, y, n, dtype_in, dtype_out, S(stream));
}
int amdseg_cast_transpose(const float* W, void* Wb, void* Wt, int N, int K, amdseg_stream_t stream) {
    return amdseg_cast_transpose_impl(W, Wb, Wt, N, K, S(stream));
}
int amdseg_lf_rowvec_dot_ld(const void* x, int ldx, const float* vec, const float* add_tok, const float* add_bh, float* out, int B, int L,
                            int H, int heads, int dtype, amdseg_stream_t stream) {
    return amdseg_lf_rowvec_dot_impl(x, vec, add_tok, add_bh, out, B, L, H, heads, dtype, ldx, S(stream));
}
int amdseg_lf_wsum_ld(const void* x, int ldx, const float* coef, float* partials, float* y, int B, int L, int H, int heads, int dtype,
                      amdseg_stream_t stream) {
    return amdseg_lf_wsum_impl(x, coef, partials, y, B, L, H, heads, dtype, ldx, S(stream));
}
int amdseg_lf_dx_update_ld(void* dx, int ldx, int assign, const float* coefA, const float* vecA, const float* coefB, const float* vecB,
                           void* vt_ws, int B, int L, int H, int heads, int dtype, amdseg_stream_t stream) {
    return amdseg_lf_dx_update_impl(dx, coefA, vecA, coefB, vecB, vt_ws, B, L, H, heads, dtype, ldx, assign, S(stream));
}
int amdseg_ponet_plan(const float* mask_bias, const int* run_start, int* work, int B, int L, amdseg_stream_t stream) {
    return amdseg_ponet_plan_impl(mask_bias, run_start, work, B, L, S(stream));
}
int amdseg_ponet_pool_fwd(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                          const float* g, void* part, void* parg, void* ctx, int B, int L, int H, amdseg_stream_t stream) {
    return amdseg_ponet_pool_fwd_impl(proj, ld, mask_bias, run_start, run_end, work, g, part, parg, ctx, B, L, H, S(stream));
}
int amdseg_ponet_pool_bwd(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                          const float* g, const void* part, const void* parg, const void* dctx, void* dproj, float* dg, float* psum, int B,
                          int L, int H, amdseg_stream_t stream) {
    return amdseg_ponet_pool_bwd_impl(proj, ld, mask_bias, run_start, run_end, work, g, part, parg, dctx, dproj, dg, psum, B, L, H, S(stream));
}
int amdseg_cast_transpose_batched(int n, const float* const* W, void* const* Wb, void* const* Wt, const int* N, const int* K,
                                  amdseg_stream_t stream) {
    return amdseg_cast_transpose_batched_impl(n, W, Wb, Wt, N, K, S(stream));
}
int amdseg_split3_weights_batched(int n, const float* const* W, void* const* out, void* const* out_t, const int* N, const int* K,
                                  amdseg_stream_t stream) {
    return amdseg_split3_weights_batched_impl(n, W, out, out_t, N, K, S(stream));
}
int amdseg_cast_transpose_batched_if(int n, const float* const* W, void* const* Wb, void* const* Wt, const int* N, const int* K,
                                     const int32_t* only_if, amdseg_stream_t stream) {
    return amdseg_cast_transpose_batched_impl(n, W, Wb, Wt, N, K, S(stream), only_if);
}
int amdseg_weights_changed(const void* x, size_t nbytes, void* state, int32_t* changed, amdseg_stream_t stream) {
    return amdseg_weights_changed_impl(x, nbytes, state, changed, S(stream));
}
int amdseg_attn_list_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                         const int* klist, const int* kcnt, int list_stride, const int* korder, amdseg_stream_t stream) {
    return amdseg_attn_list_fwd_impl(qkv, mask_bias, ctx, lse, B, L, heads, scale, klist, kcnt, list_stride, korder, S(stream));
}
int amdseg_attn_list_f32(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale,
                         const int* klist, const int* kcnt, int list_stride, amdseg_stream_t stream) {
    return amdseg_attn_list_f32_impl(qkv, mask_bias, ctx, B, L, heads, scale, klist, kcnt, list_stride, S(stream));
}
int amdseg_attn_list_bwd(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta_ws, void* dqkv, int B, int L, int heads, float scale, const int* klist, const int* kcnt,
                         const int* qlist, const int* qcnt, int list_stride, const int* korder, const int* qorder, amdseg_stream_t stream) {
    return amdseg_attn_list_bwd_impl(qkv, mask_bias, ctx, dctx, lse, delta_ws, dqkv, B, L, heads, scale, klist, kcnt, qlist, qcnt,
                                     list_stride, korder, qorder, S(stream));
}
int amdseg_rowdot_fwd(const void* x, const float* W, const float* b, float* out, int M, int H, int C, int dtype,
                      amdseg_stream_t stream) {
    return amdseg_rowdot_fwd_impl(x, W, b, out, M, H, C, dtype, S(stream));
}
int amdseg_rowdot_bwd(const void* x, const float* W, const float* dlogits, void* dx, float* partials, float* dW, float* db,
                      int M, int H, int C, int accumulate, int dtype, amdseg_stream_t stream) {
    return amdseg_rowdot_bwd_impl(x, W, dlogits, dx, partials, dW, db, M, H, C, accumulate, dtype, S(stream));
}
int amdseg_adamw(float* p, const float* g, float* m, float* v, void* bf16_shadow, size_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, const float* grad_scale, int zero_grad,
                 const unsigned char* chunk_flags, amdseg_stream_t stream) {
    return amdseg_adamw_impl(p, g, m, v, bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, chunk_flags,
                             S(stream));
}
int amdseg_sumsq(const float* x, size_t n, float* partials, float* out, int accumulate, amdseg_stream_t stream) {
    return amdseg_sumsq_impl(x, n, partials, out, accumulate, S(stream));
}
int amdseg_clip_coef(const float* sumsq, float max_norm, float extra_scale, float* coef, float* norm,
                     amdseg_stream_t stream) {
    return amdseg_clip_coef_impl(sumsq, max_norm, extra_scale, coef, norm, S(stream));
}
int amdseg_scale(float* x, size_t n, const float* coef, amdseg_stream_t stream) {
    return amdseg_scale_impl(x, n, coef, S(stream));
}

int amdseg_split3(const float* x, int ld, void* out_bf16, int M, int K, int order, amdseg_stream_t stream) {
    return amdseg_split3_impl(x, ld, out_bf16, M, K, order, S(stream));
}
int amdseg_split3_transpose(const float* W, void* out_bf16, int N, int K, amdseg_stream_t stream) {
    return amdseg_split3_transpose_impl(W, out_bf16, N, K, S(stream));
}
int amdseg_pattn_fwd(const float* qkv, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale,
                     float p_drop, uint64_t seed, amdseg_stream_t stream) {
    return amdseg_pattn_fwd_impl(qkv, mask_bias, ctx, lse, B, L, heads, scale, p_drop, seed, S(stream));
}
int amdseg_pattn_bwd(const float* qkv, const float* mask_bias, const float* ctx, const float* dctx, const float* lse, float* delta,
                     float* dqkv, int B, int L, int heads, float scale, float p_drop, uint64_t seed, amdseg_stream_t stream) {
    return amdseg_pattn_bwd_impl(qkv, mask_bias, ctx, dctx, lse, delta, dqkv, B, L, heads, scale, p_drop, seed, S(stream));
}

int amdseg_lf_global_q(const void* x, int x_dtype, const float* Wq, const float* bq, const float* Wk, float* qg, float* r, int B, int L, int H,
                       int heads, float scale, amdseg_stream_t stream) {
    return amdseg_lf_global_q_impl(x, x_dtype, Wq, bq, Wk, qg, r, B, L, H, heads, scale, S(stream));
}
int amdseg_lf_global_out(const float* Wv, const float* bv, const float* y, const float* sp, void* ctx, int ctx_dtype, int B, int L, int H,
                         int heads, amdseg_stream_t stream) {
    return amdseg_lf_global_out_impl(Wv, bv, y, sp, ctx, ctx_dtype, B, L, H, heads, S(stream));
}
int amdseg_lf_global_bwd_a(void* dctx, int dtype, const float* Wv, const float* bv, float* dout, float* dyv, float* dsp, int B, int L, int H,
                           int heads, amdseg_stream_t stream) {
    return amdseg_lf_global_bwd_a_impl(dctx, dtype, Wv, bv, dout, dyv, dsp, B, L, H, heads, S(stream));
}
int amdseg_lf_global_bwd_a_ro(const void* dctx, int dtype, const float* Wv, const float* bv, float* dout, float* dyv, float* dsp, int B, int L,
                              int H, int heads, amdseg_stream_t stream) {
    return amdseg_lf_global_bwd_a_impl((void*)dctx, dtype, Wv, bv, dout, dyv, dsp, B, L, H, heads, S(stream), 1);
}
int amdseg_lf_global_bwd_rest(const void* x, int x_dtype, void* dx, int dx_dtype, const float* Wq, const float* Wk, const float* qg,
                              const float* dout, const float* y, const float* sp, const float* dr, float* dqg, float* dWq, float* dbq,
                              float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads, float scale, amdseg_stream_t stream) {
    return amdseg_lf_global_bwd_rest_impl(x, x_dtype, dx, dx_dtype, Wq, Wk, qg, dout, y, sp, dr, dqg, dWq, dbq, dWk, dWv, dbv, B, L, H, heads,
                                          scale, S(stream));
}
size_t amdseg_ponet_global_scratch_floats(int B, int L, int H, int heads) { return amdseg_ponet_global_scratch_floats_impl(B, L, H, heads); }
int amdseg_ponet_global_fwd(const void* hq, const void* hk, int ld, const float* coef_mean, const float* mask_bias, int B, int L, int H,
                            int heads, float dropout_p, uint64_t seed, float* scratch, float* vecq, float* scores, float* lse, float* g,
                            amdseg_stream_t stream) {
    return amdseg_ponet_global_fwd_impl(hq, hk, ld, coef_mean, mask_bias, B, L, H, heads, dropout_p, seed, scratch, vecq, scores, lse, g, S(stream));
}
int amdseg_ponet_global_bwd(const void* hk, int ld, const float* coef_mean, const float* vecq, const float* scores, const float* lse,
                            const float* dg, int B, int L, int H, int heads, float dropout_p, uint64_t seed, float* scratch, float* dpd_ws,
                            void* dhq, void* dhk, int ldd, amdseg_stream_t stream) {
    return amdseg_ponet_global_bwd_impl(hk, ld, coef_mean, vecq, scores, lse, dg, B, L, H, heads, dropout_p, seed, scratch, dpd_ws, dhq, dhk, ldd,
                                        S(stream));
}
int amdseg_lf_global_bwd_dx(const float* Wq, const float* Wk, const float* dr, float* dqg, float* trow, int B, int L, int H, int heads,
                            float scale, amdseg_stream_t stream) {
    return amdseg_lf_global_bwd_dx_impl(Wq, Wk, dr, dqg, trow, B, L, H, heads, scale, S(stream));
}
int amdseg_lf_global_bwd_w(const void* x, int x_dtype, const float* qg, const float* dout, const float* y, const float* sp, const float* dr,
                           const float* dqg, float* dWq, float* dbq, float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads,
                           amdseg_stream_t stream) {
    return amdseg_lf_global_bwd_w_impl(x, x_dtype, qg, dout, y, sp, dr, dqg, dWq, dbq, dWk, dWv, dbv, B, L, H, heads, S(stream));
}
int amdseg_lf_dx_prep(const float* vecA, const float* vecB, void* vt_ws, int B, int L, int H, int heads, amdseg_stream_t stream) {
    return amdseg_lf_dx_prep_impl(vecA, vecB, vt_ws, B, L, H, heads, S(stream));
}
int amdseg_lf_dx_apply(void* dx, int ldx, const float* coefA, const float* coefB, const void* vt_ws, const float* trow, int B, int L, int H,
                       int heads, amdseg_stream_t stream) {
    return amdseg_lf_dx_apply_impl(dx, ldx, coefA, coefB, vt_ws, trow, B, L, H, heads, S(stream));
}
int amdseg_heads_fwd(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                     float* ce_unit, float* out8, float* acc, const int64_t* idx, long feat_off, long anchor_off, long lists_off,
                     int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off, long t_labels_off,
                     int nt, int Ct, float w_ts, float w_cl, float w_tssp2, amdseg_stream_t stream) {
    return amdseg_heads_fwd_impl(x, M, H, logits, labels, class_w, C, nseg, ce_unit, out8, acc, idx, feat_off, anchor_off, lists_off, n_anchor,
                                 n_list, pk, temp, Wt, bt, t_rows_off, t_labels_off, nt, Ct, w_ts, w_cl, w_tssp2, S(stream));
}
int amdseg_heads_bwd_ce(const float* gout, int M, int C, int nseg, const float* ce_unit, const float* out8, float w_ts, float* dlogits,
                        amdseg_stream_t stream) {
    return amdseg_heads_bwd_ce_impl(gout, M, C, nseg, ce_unit, out8, w_ts, dlogits, S(stream));
}
int amdseg_heads_fwd_focal(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                           float* ce_unit2, float* out12, float* acc, const int64_t* idx, long feat_off, long anchor_off, long lists_off,
                           int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off, long t_labels_off,
                           int nt, int Ct, float w_ts, float w_cl, float w_tssp2, float focal_gamma, amdseg_stream_t stream) {
    return amdseg_heads_fwd_impl(x, M, H, logits, labels, class_w, C, nseg, ce_unit2, out12, acc, idx, feat_off, anchor_off, lists_off, n_anchor,
                                 n_list, pk, temp, Wt, bt, t_rows_off, t_labels_off, nt, Ct, w_ts, w_cl, w_tssp2, S(stream), focal_gamma);
}
int amdseg_heads_bwd_ce_focal(const float* gout, int M, int C, int nseg, const float* ce_unit2, const float* out12, float w_ts, float focal_gamma,
                              float* dlogits, amdseg_stream_t stream) {
    return amdseg_heads_bwd_ce_impl(gout, M, C, nseg, ce_unit2, out12, w_ts, dlogits, S(stream), focal_gamma);
}
int amdseg_heads_bwd_rows(const float* gout, const float* x, int M, int H, float* dx, const int64_t* idx, long feat_off, long anchor_off,
                          long lists_off, int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off,
                          long t_labels_off, int nt, int Ct, float* dWt, float* dbt, float w_cl, float w_tssp2, int n_feat, void* fix,
                          size_t fix_bytes, amdseg_stream_t stream) {
    return amdseg_heads_bwd_rows_impl(gout, x, M, H, dx, idx, feat_off, anchor_off, lists_off, n_anchor, n_list, pk, temp, Wt, bt, t_rows_off,
                                      t_labels_off, nt, Ct, dWt, dbt, w_cl, w_tssp2, n_feat, fix, fix_bytes, S(stream));
}

// ---------------------------------------------------------------------------------------------------- composite layer
static inline uint64_t site_seed(uint64_t seed, int layer, int site) {
    return seed * 0x9E3779B97F4A7C15ull + (uint64_t)(layer * 8 + site + 1) * 0xD1B54A32D192ED03ull;
}
#define RET_IF(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int check_cfg(const amdseg_bert_cfg* c) {
    if (!c) return AMDSEG_ERR_ARG;
    if (c->dtype != AMDSEG_BF16 && c->dtype != AMDSEG_F32 && c->dtype != AMDSEG_F32S) return AMDSEG_ERR_ARG;
    if (c->dtype == AMDSEG_F32S && (c->mixer != 0 || (c->nproj != 0 && c->nproj != 3))) return AMDSEG_ERR_ARG;   // (band: needs acts.qkv_s, checked there)
    if (c->H != c->heads * 64 || c->B <= 0 || c->L <= 0 || c->I <= 0) return AMDSEG_ERR_SHAPE;
    const long M = (long)c->B * c->L;
    if ((M % 128) || (c->H % 128) || (c->I % 128) || (c->L % 64)) return AMDSEG_ERR_SHAPE;
    if (c->window < 0 || c->nglobal < 0 || c->phase < 0 || (c->phase > 4 && c->phase != 6)) return AMDSEG_ERR_ARG;
    if (c->nproj < 0 || c->nproj > 8 || c->mixer < 0 || c->mixer > 1 || c->act < 0 || c->act > 1) return AMDSEG_ERR_ARG;
    if (c->mixer == 1 && (c->phase == 0 || c->phase == 3 || c->phase > 2)) return AMDSEG_ERR_ARG;
    if (c->mixer == 1 && c->dtype != AMDSEG_BF16 && c->nproj != 0 && c->nproj != 3) return AMDSEG_ERR_ARG;   // fp32 parity mode: q|k|v only
    return AMDSEG_OK;
}
// phase: 0 or 3 = whole layer; 1 = first part only; 2 = second part only.  The split point is the attention context:
// a Longformer caller overwrites the global token's ctx row between forward phases 1 and 2, and consumes / zeroes its
// dctx row between backward phases 1 and 2 (see include/amdseg.h).
// Backward only: 6 = second part WITHOUT the grouped weight-gradient GEMM, 4 = that GEMM alone -- so a caller can run the
// weight gradients of layer i on a second stream under the backward of layer i-1 (they are off the critical path: nothing
// reads them before the optimiser step).
// ("parity" precision: the fused forms -- ctx image from the attention, d(ctx) image from the dgrad, GELU / GELU' + split in the FFN GEMM epilogues,
//  LayerNorm forward / backward writing the images -- are the only forms since round 6; the separate passes they replaced are in the history)
static inline int parity_unfused() { return 0; }
#define PHASE1(c) ((c)->phase == 0 || (c)->phase == 1 || (c)->phase == 3)
#define PHASE2(c) ((c)->phase == 0 || (c)->phase == 2 || (c)->phase == 3 || (c)->phase == 6)
#define PHASE_WGRAD(c) ((c)->phase == 0 || (c)->phase == 2 || (c)->phase == 3 || (c)->phase == 4)
// width of the fused input projection: 3H (q|k|v) for BERT / Longformer, nproj*H for an external token mixer (PoNet: 5H)
#define NPROJ(c) (((c)->nproj ? (c)->nproj : 3) * (c)->H)
// bf16 training: what the FFN up-projection keeps for backward in acts.u.  0: the pre-activation u (bf16; the backward GEMM's epilogue evaluates
// gelu'(u)).  1: gelu'(u) in bf16 (AMDSEG_EPI_KEEP_DERIV: the backward epilogue is one multiply -- measured neutral, its cost is reading the
// tensor, not the arithmetic).  2: gelu'(u) as ONE BYTE per element (AMDSEG_EPI_DERIV_U8): 100 MB less HBM traffic per bert-base layer in two
// epilogues that are HBM time.  Needs shapes of the 256-wide deep-pipeline tile (forms 0 / 1: and the erf GELU); forward and backward evaluate the same
// predicate on the same cfg.  AMDSEG_FFN_KEEP_DERIV = 0 / 1 / 2 picks the form.
static inline int ffn_keep_deriv(const amdseg_bert_cfg* c) {
    constexpr int mode = 2;         // (0 = u in bf16, 1 = gelu'(u) in bf16 were the other measured forms, profiles/r04_gemm_epilogue_split.md)
    const int M = c->B * c->L;
    // gelu_new (BigBird): the one-byte form only (value and derivative from ONE tanh: bigbird-base 8 x 4096 332.6 -> 336.8 seq/s; evaluated
    // separately they cost what the bytes save: 343.1 vs 344.4 on another box)
    if (!mode || c->dtype != AMDSEG_BF16 || (M % 256) || c->H < 128 || (c->H % 64) || (c->act != 0 && mode != 2)) return 0;
    const int mode_eff = mode;
    if (mode_eff == 2 && (c->I % 256) == 0) {
        // ... unless the up-projection would take the 192-wide tile for its rounds (amdseg_launch_nt_dp: M = 8192, the 4 x 2048 launch shape), which the
        // one-byte epilogue does not have: there the narrow tile is worth more than the bytes (longformer-base 4 x 2048: 368 vs 364 seq/s)
        // (counted against ALL the CUs, not the momentary budget of amdseg_set_cu_budget: forward and backward must come to the same answer)
        const int t256 = (M / 256) * (c->I / 256), t192 = (c->I % 192) == 0 ? (M / 256) * (c->I / 192) : 0, C = amdseg_num_cus();
        const bool narrow = t192 > 0 && 0.78f * (float)((t192 + C - 1) / C) < (float)((t256 + C - 1) / C);
        return narrow ? 0 : (AMDSEG_EPI_KEEP_DERIV | AMDSEG_EPI_DERIV_U8);       // (the callers OR AMDSEG_EPI_ACT_TANH in for gelu_new)
    }
    if (mode_eff == 1 && ((c->I % 256) == 0 || (c->I % 192) == 0)) return AMDSEG_EPI_KEEP_DERIV;
    return 0;
}

int amdseg_bert_layer_fwd(const amdseg_bert_cfg* c, const amdseg_bert_layer_params* p, const amdseg_bert_layer_acts* a,
                          const float* mask_bias, int li, amdseg_stream_t stream) {
    RET_IF(check_cfg(c));
    if (!p || !a || !mask_bias) return AMDSEG_ERR_ARG;
    AmdsegCtxScope ctx_scope(c->ctx);                       // tile rules and launch timer of THIS caller (NULL: the thread's bound context or the defaults)
    hipStream_t s = S(stream);
    const int M = c->B * c->L, H = c->H, I = c->I;
    if (c->dtype == AMDSEG_F32S) {
        // "parity" precision (csrc/parity.hip): fp32 activations, every product as one bf16 GEMM over K' = 3K on split images
        if (!a->xs || !a->ctx_s || !a->x1_s || !a->h_s) return AMDSEG_ERR_ARG;
        const bool ffn_fused = c->act == 0 && (M % 256) == 0 && (I % 256) == 0 && !(parity_unfused() & 4);
        if (!a->u && !ffn_fused) return AMDSEG_ERR_ARG;      // u == NULL (inference): only the fused up-projection epilogue can leave it out
        const float* fx = (const float*)a->x_in;
        if (PHASE1(c)) {
            RET_IF(amdseg_split3_impl(fx, H, a->xs, M, H, 0, s));
            const bool split_attn = a->qkv_s && (c->p_attn == 0.f || a->keep);
            if (c->window > 0 && !split_attn) return AMDSEG_ERR_ARG;      // the fp32-MFMA kernels of parity.hip have no band form
            // the projection written straight as the split image the attention reads (hi block | unused | lo block), where the 256 x 256 GEMM tiles it
            const bool fused_qkv = split_attn && (M % 256) == 0 && ((3 * H) % 256) == 0;
            if (fused_qkv)
                RET_IF(amdseg_gemm_nt_impl(a->xs, 3 * H, p->wqkv, 3 * H, a->qkv_s, 9 * H, M, 3 * H, 3 * H, AMDSEG_EPI_BIAS_SPLIT, p->bqkv, nullptr, 0,
                                           (bf16_t*)a->qkv_s + 6 * H, 9 * H, 0, s));
            else
            RET_IF(amdseg_gemm_nt_impl(a->xs, 3 * H, p->wqkv, 3 * H, a->qkv, 3 * H, M, 3 * H, 3 * H, AMDSEG_EPI_BIAS, p->bqkv, nullptr, 0, nullptr, 0, 1, s));
            if (split_attn) {
                // attention as split-bf16 products on the bf16 matrix cores (attention_split.hip); dropout from this layer's keep masks
                if (!fused_qkv) RET_IF(amdseg_split3_impl((const float*)a->qkv, 3 * H, a->qkv_s, M, 3 * H, 0, s));
                if (c->p_attn > 0.f) RET_IF(amdseg_attn_keepmask_impl(a->keep, c->B, c->L, c->heads, c->p_attn, site_seed(c->seed, li, 0), c->kend, s,
                                                                      c->window, c->nglobal));
                RET_IF(amdseg_sattn_fwd_impl(a->qkv_s, 9 * H, 6 * H, mask_bias, (float*)a->ctx, a->lse, c->B, c->L, c->heads, 0.125f, c->p_attn,
                                             c->p_attn > 0.f ? a->keep : nullptr, c->window, c->nglobal, s, c->kend, c->seq_order,
                                             (c->window == 0 && !(parity_unfused() & 1)) ? a->ctx_s : nullptr));      // (a Longformer caller still rewrites the global rows of ctx)
            } else
            RET_IF(amdseg_pattn_fwd_impl((const float*)a->qkv, mask_bias, (float*)a->ctx, a->lse, c->B, c->L, c->heads, 0.125f, c->p_attn,
                                         site_seed(c->seed, li, 0), s, c->kend, c->seq_order));
        }
        if (!PHASE2(c)) return AMDSEG_OK;
        if (!(a->qkv_s && (c->p_attn == 0.f || a->keep) && c->window == 0 && !(parity_unfused() & 1)))      // (else the split attention wrote the image itself)
            RET_IF(amdseg_split3_impl((const float*)a->ctx, H, a->ctx_s, M, H, 0, s));
        RET_IF(amdseg_gemm_nt_impl(a->ctx_s, 3 * H, p->wo, 3 * H, a->z1, H, M, H, 3 * H, AMDSEG_EPI_BIAS, p->bo, nullptr, 0, nullptr, 0, 1, s));
        const bool ln_img = !(parity_unfused() & 8);
        RET_IF(amdseg_add_ln_fwd_impl(a->z1, a->x_in, p->ln1_g, p->ln1_b, a->x1, a->mean1, a->rstd1, M, H, c->ln_eps, c->p_hidden,
                                      site_seed(c->seed, li, 1), AMDSEG_F32, s, ln_img ? a->x1_s : nullptr, a->drop1));
        if (!ln_img) RET_IF(amdseg_split3_impl((const float*)a->x1, H, a->x1_s, M, H, 0, s));
        if (ffn_fused)       // u (fp32, read by backward; NULL in inference) and the image of gelu(u) from one epilogue
            RET_IF(amdseg_gemm_nt_impl(a->x1_s, 3 * H, p->w1, 3 * H, a->u, I, M, I, 3 * H, AMDSEG_EPI_BIAS_GELU_SPLIT, p->b1, nullptr, 0, a->h_s, 3 * I, 1, s));
        else {
            RET_IF(amdseg_gemm_nt_impl(a->x1_s, 3 * H, p->w1, 3 * H, a->u, I, M, I, 3 * H, AMDSEG_EPI_BIAS, p->b1, nullptr, 0, nullptr, 0, 1, s));
            RET_IF(amdseg_gelu_fwd_split_impl((const float*)a->u, a->h_s, M, I, c->act, s));
        }
        RET_IF(amdseg_gemm_nt_impl(a->h_s, 3 * I, p->w2, 3 * I, a->z2, H, M, H, 3 * I, AMDSEG_EPI_BIAS, p->b2, nullptr, 0, nullptr, 0, 1, s));
        RET_IF(amdseg_add_ln_fwd_impl(a->z2, a->x1, p->ln2_g, p->ln2_b, a->x_out, a->mean2, a->rstd2, M, H, c->ln_eps, c->p_hidden,
                                      site_seed(c->seed, li, 2), AMDSEG_F32, s, nullptr, a->drop2));
        return AMDSEG_OK;
    }
    if (c->dtype == AMDSEG_F32) {
        // fp32 parity mode (inference): exact-fp32 MFMA GEMMs on the fp32 master weights, fp32 activations, no dropout
        if (c->p_hidden != 0.f || c->p_attn != 0.f) return AMDSEG_ERR_ARG;
        if (PHASE1(c)) {
            RET_IF(amdseg_gemm_f32_nt_impl((const float*)a->x_in, H, (const float*)p->wqkv, H, (float*)a->qkv, 3 * H, M, 3 * H, H, 1, p->bqkv, s));
            if (c->mixer == 0)
                RET_IF(amdseg_attn_f32_impl((const float*)a->qkv, mask_bias, (float*)a->ctx, c->B, c->L, c->heads, 64, 0.125f, c->window, c->nglobal, s));
        }
        if (!PHASE2(c)) return AMDSEG_OK;
        RET_IF(amdseg_gemm_f32_nt_impl((const float*)a->ctx, H, (const float*)p->wo, H, (float*)a->z1, H, M, H, H, 1, p->bo, s));
        RET_IF(amdseg_add_ln_fwd_impl(a->z1, a->x_in, p->ln1_g, p->ln1_b, a->x1, a->mean1, a->rstd1, M, H, c->ln_eps, 0.f, 0, AMDSEG_F32, s));
        RET_IF(amdseg_gemm_f32_nt_impl((const float*)a->x1, H, (const float*)p->w1, H, (float*)a->h, I, M, I, H, 2 | (c->act ? AMDSEG_EPI_ACT_TANH : 0), p->b1, s));
        RET_IF(amdseg_gemm_f32_nt_impl((const float*)a->h, I, (const float*)p->w2, I, (float*)a->z2, H, M, H, I, 1, p->b2, s));
        RET_IF(amdseg_add_ln_fwd_impl(a->z2, a->x1, p->ln2_g, p->ln2_b, a->x_out, a->mean2, a->rstd2, M, H, c->ln_eps, 0.f, 0, AMDSEG_F32, s));
        return AMDSEG_OK;
    }
    if (PHASE1(c)) {
        // q|k|v (or the mixer's) projection with bias
        const int NP = NPROJ(c);
        RET_IF(amdseg_gemm_nt_impl(a->x_in, H, p->wqkv, H, a->qkv, NP, M, NP, H, AMDSEG_EPI_BIAS, p->bqkv, nullptr, 0, nullptr, 0, 0, s));
        if (c->mixer == 0) {
            // dropout on the probabilities: decided once per layer here, read by the forward and the two backward kernels (acts.keep)
            const void* keep = (a->keep && c->p_attn > 0.f) ? a->keep : nullptr;      // full attention, or the band's cells (window > 0)
            // (keep_ready: the layer in front wrote them in the launch of its second LayerNorm -- acts.keep_next, below)
            if (keep && !(a->keep_ready && km_pairs_with_rows(c->B, c->L, c->heads, M))) RET_IF(amdseg_attn_keepmask_impl(a->keep, c->B, c->L, c->heads, c->p_attn, site_seed(c->seed, li, 0), c->kend, s,
                                                                         c->window, c->nglobal));
            RET_IF(amdseg_attn_fwd_impl(a->qkv, mask_bias, a->ctx, a->lse, c->B, c->L, c->heads, 0.125f, c->p_attn, site_seed(c->seed, li, 0),
                                        c->window, c->nglobal, s, c->kend, c->seq_order, keep,
                                        // a phase-1 call of a layer with global tokens: the caller writes their ctx rows (amdseg.h, `phase`)
                                        (c->phase == 1 && c->window > 0) ? c->nglobal : 0));
        }
    }
    if (!PHASE2(c)) return AMDSEG_OK;
    // attention output dense -> dropout -> +residual -> LN.  (Dropout and residual in the GEMM's epilogue -- 2 passes over [M, H] instead of 4 -- were
    // built and measured in round 4: the row kernel dropped 20.7 -> 13.8 us but the two GEMMs gained 14-16 us each, profiles/r04_fused_drop_res.md.)
    RET_IF(amdseg_gemm_nt_impl(a->ctx, H, p->wo, H, a->z1, H, M, H, H, AMDSEG_EPI_BIAS, p->bo, nullptr, 0, nullptr, 0, 0, s));
    RET_IF(amdseg_add_ln_fwd_impl(a->z1, a->x_in, p->ln1_g, p->ln1_b, a->x1, a->mean1, a->rstd1, M, H, c->ln_eps, c->p_hidden,
                                  site_seed(c->seed, li, 1), c->dtype, s, nullptr, a->drop1, a->u != nullptr));    // u == NULL = inference: z is not kept
    // FFN
    RET_IF(amdseg_gemm_nt_impl(a->x1, H, p->w1, H, a->h, I, M, I, H, AMDSEG_EPI_BIAS_GELU | (c->act ? AMDSEG_EPI_ACT_TANH : 0) | ffn_keep_deriv(c),
                               p->b1, nullptr, 0, a->u, I, 0, s));
    RET_IF(amdseg_gemm_nt_impl(a->h, I, p->w2, I, a->z2, H, M, H, I, AMDSEG_EPI_BIAS, p->b2, nullptr, 0, nullptr, 0, 0, s));
    // the keep masks of layer li + 1 as workgroups of this launch (VALU-bound generator under an HBM-bound row kernel: acts.keep_next, amdseg.h)
    if (a->keep_next && c->p_attn > 0.f && c->mixer == 0 && km_pairs_with_rows(c->B, c->L, c->heads, M))
        RET_IF(amdseg_add_ln_fwd_km_impl(a->z2, a->x1, p->ln2_g, p->ln2_b, a->x_out, a->mean2, a->rstd2, M, H, c->ln_eps, c->p_hidden,
                                         site_seed(c->seed, li, 2), c->dtype, s, a->drop2, a->u != nullptr,
                                         a->keep_next, c->B, c->L, c->heads, c->p_attn, site_seed(c->seed, li + 1, 0), c->kend, c->window, c->nglobal));
    else
    RET_IF(amdseg_add_ln_fwd_impl(a->z2, a->x1, p->ln2_g, p->ln2_b, a->x_out, a->mean2, a->rstd2, M, H, c->ln_eps, c->p_hidden,
                                  site_seed(c->seed, li, 2), c->dtype, s, nullptr, a->drop2, a->u != nullptr));
    return AMDSEG_OK;
}

int amdseg_bert_layer_bwd(const amdseg_bert_cfg* c, const amdseg_bert_layer_params* p, const amdseg_bert_layer_grads* g,
                          const amdseg_bert_layer_acts* a, const amdseg_bert_layer_ws* w, const float* mask_bias,
                          const void* dy, void* dx_in, int li, amdseg_stream_t stream) {
    RET_IF(check_cfg(c));
    if (c->dtype != AMDSEG_BF16 && c->dtype != AMDSEG_F32S) return AMDSEG_ERR_ARG;      // training: bf16 fast path or split-bf16 parity
    if (!p || !g || !a || !w || !mask_bias || !dy || !dx_in) return AMDSEG_ERR_ARG;
    AmdsegCtxScope ctx_scope(c->ctx);                       // the CU budget of this caller's tile rules, its launch timer
    hipStream_t s = S(stream);
    const int M = c->B * c->L, H = c->H, I = c->I, acc = c->accumulate_grads;
    const bool drop = c->p_hidden > 0.f;
    const void* d_out = drop ? w->dbr2 : w->dz2;
    const void* d_ao = drop ? w->dbr1 : w->dz1;
    // the four second-stage reductions of the layer (LN2, b1, LN1, bqkv) are queued and run as ONE kernel at the end of this
    // call; each producer therefore gets its own region of ws.partials (include/amdseg.h: amdseg_bert_layer_ws)
    const int NPd = NPROJ(c);
    // (bias-gradient regions: ceil(M/128) rows for amdseg_colsum, H/128 rows for the partials of the fused weight-gradient kernel)
    const size_t ln_part = (size_t)3 * ((M + 15) / 16) * H, cs_rows = (size_t)std::max((M + 127) / 128, (H + 127) / 128);
    float* part_ln2 = w->partials;
    float* part_b1 = part_ln2 + ln_part;
    float* part_ln1 = part_b1 + cs_rows * I;
    float* part_bqkv = part_ln1 + ln_part;
    (void)NPd;
    amdseg_reduce_defer_begin(acc);
    struct Flush { hipStream_t s; ~Flush() { amdseg_reduce_defer_flush(s); } } flush_at_exit{s};
    if (c->dtype == AMDSEG_F32S) {
        // "parity" precision: same dataflow in fp32; every GEMM operand goes through its split image (csrc/parity.hip)
        if (!w->d_out_s || !w->du_s || !w->d_ao_s || !w->dqkv_s || !a->xs || !a->ctx_s || !a->x1_s || !a->h_s) return AMDSEG_ERR_ARG;
        // full attention on the split kernels: nobody reads d(ctx) in fp32 (a Longformer caller does, between the phases: its global row)
        const bool dctx_image = a->qkv_s && w->dctx_s && (c->p_attn == 0.f || a->keep) && c->window == 0 && (M % 256) == 0 && (H % 256) == 0 &&
                                !(parity_unfused() & 2);
        if (PHASE1(c)) {
            const bool ln_img = !(parity_unfused() & 8);
            RET_IF(amdseg_ln_bwd_impl(dy, a->z2, a->mean2, a->rstd2, p->ln2_g, w->dz2, drop ? w->dbr2 : nullptr, part_ln2, g->ln2_g, g->ln2_b,
                                      g->b2, M, H, c->p_hidden, site_seed(c->seed, li, 2), acc, AMDSEG_F32, s, nullptr, nullptr, 0,
                                      ln_img ? w->d_out_s : nullptr, a->drop2));
            if (!ln_img) RET_IF(amdseg_split3_impl((const float*)d_out, H, w->d_out_s, M, H, 0, s));
            if (c->act == 0 && (M % 256) == 0 && (I % 256) == 0 && !(parity_unfused() & 4)) {
                // du = (d_out . W2) * gelu'(u) leaves the GEMM as the [hi | hi | lo] image (no fp32 du, no separate GELU' / split pass: 160 us per
                // layer at bert-base); the bias gradient is summed from the image
                RET_IF(amdseg_gemm_nt_impl(w->d_out_s, 3 * H, p->w2_t, 3 * H, w->du_s, 3 * I, M, I, 3 * H, AMDSEG_EPI_GELU_BWD_SPLIT, nullptr, a->u, I,
                                           nullptr, 0, 0, s));
                RET_IF(amdseg_colsum_split_impl(w->du_s, 3 * I, 2 * I, part_b1, g->b1, M, I, acc, s));
            } else {
                RET_IF(amdseg_gemm_nt_impl(w->d_out_s, 3 * H, p->w2_t, 3 * H, w->du, I, M, I, 3 * H, AMDSEG_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1, s));
                RET_IF(amdseg_gelu_bwd_split_impl((float*)w->du, (const float*)a->u, w->du_s, M, I, c->act, s));
                RET_IF(amdseg_colsum_impl(w->du, I, part_b1, g->b1, M, I, acc, AMDSEG_F32, s));
            }
            RET_IF(amdseg_gemm_nt_impl(w->du_s, 3 * I, p->w1_t, 3 * I, w->dx1, H, M, H, 3 * I, AMDSEG_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1, s));
            RET_IF(amdseg_add_inplace_impl((float*)w->dx1, (const float*)w->dz2, (size_t)M * H, s));
            RET_IF(amdseg_ln_bwd_impl(w->dx1, a->z1, a->mean1, a->rstd1, p->ln1_g, w->dz1, drop ? w->dbr1 : nullptr, part_ln1, g->ln1_g,
                                      g->ln1_b, g->bo, M, H, c->p_hidden, site_seed(c->seed, li, 1), acc, AMDSEG_F32, s, nullptr, nullptr, 0,
                                      ln_img ? w->d_ao_s : nullptr, a->drop1));
            if (!ln_img) RET_IF(amdseg_split3_impl((const float*)d_ao, H, w->d_ao_s, M, H, 0, s));
            if (dctx_image)      // d(ctx) straight as the hi / lo blocks the split attention backward reads (no fp32 d(ctx), no split pass)
                RET_IF(amdseg_gemm_nt_impl(w->d_ao_s, 3 * H, p->wo_t, 3 * H, w->dctx_s, 3 * H, M, H, 3 * H, AMDSEG_EPI_BIAS_SPLIT, nullptr, nullptr, 0,
                                           (bf16_t*)w->dctx_s + 2 * H, 3 * H, 0, s));
            else
            RET_IF(amdseg_gemm_nt_impl(w->d_ao_s, 3 * H, p->wo_t, 3 * H, w->dctx, H, M, H, 3 * H, AMDSEG_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1, s));
        }
        if (PHASE2(c)) {
            // the forward took the split-attention path on (qkv_s && (p_attn == 0 || keep)) and then never wrote the fp32 a->qkv the
            // fallback below reads: a caller that set qkv_s must bring dctx_s too
            if (a->qkv_s && !w->dctx_s && (c->p_attn == 0.f || a->keep)) return AMDSEG_ERR_ARG;
            if (a->qkv_s && w->dctx_s && (c->p_attn == 0.f || a->keep)) {
                if (!dctx_image) RET_IF(amdseg_split3_impl((const float*)w->dctx, H, w->dctx_s, M, H, 0, s));
                // ... whose backward writes d(q|k|v) as the [hi | hi | lo] image the next GEMMs read; the bias gradient is summed from the image
                RET_IF(amdseg_sattn_bwd_impl(a->qkv_s, 9 * H, 6 * H, mask_bias, (const float*)a->ctx, w->dctx_s, 3 * H, 2 * H, a->lse, w->delta,
                                             nullptr, c->B, c->L, c->heads, 0.125f, c->p_attn, c->p_attn > 0.f ? a->keep : nullptr, c->window,
                                             c->nglobal, s, c->kend, c->seq_order, c->pad_guard, w->dqkv_s, 9 * H));
                RET_IF(amdseg_colsum_split_impl(w->dqkv_s, 9 * H, 6 * H, part_bqkv, g->bqkv, M, 3 * H, acc, s));
            } else {
            if (c->window > 0) return AMDSEG_ERR_ARG;
            RET_IF(amdseg_pattn_bwd_impl((const float*)a->qkv, mask_bias, (const float*)a->ctx, (const float*)w->dctx, a->lse, w->delta,
                                         (float*)w->dqkv, c->B, c->L, c->heads, 0.125f, c->p_attn, site_seed(c->seed, li, 0), s, c->kend, c->seq_order,
                                         c->pad_guard));
            RET_IF(amdseg_split3_impl((const float*)w->dqkv, 3 * H, w->dqkv_s, M, 3 * H, 0, s));
            RET_IF(amdseg_colsum_impl(w->dqkv, 3 * H, part_bqkv, g->bqkv, M, 3 * H, acc, AMDSEG_F32, s));
            }
            RET_IF(amdseg_gemm_nt_impl(w->dqkv_s, 9 * H, p->wqkv_t, 9 * H, dx_in, H, M, H, 9 * H, AMDSEG_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1, s));
            RET_IF(amdseg_add_inplace_impl((float*)dx_in, (const float*)w->dz1, (size_t)M * H, s));
        }
        if (!PHASE_WGRAD(c)) return AMDSEG_OK;
        // dW = dY^T X = dYhi^T Xhi + dYhi^T Xlo + dYlo^T Xhi: three grouped launches over the hi / lo column blocks of the images
        const bf16_t* Ai[4] = {(const bf16_t*)w->d_out_s, (const bf16_t*)w->du_s, (const bf16_t*)w->d_ao_s, (const bf16_t*)w->dqkv_s};
        const bf16_t* Bi[4] = {(const bf16_t*)a->h_s, (const bf16_t*)a->x1_s, (const bf16_t*)a->ctx_s, (const bf16_t*)a->xs};
        float* C[4] = {g->w2, g->w1, g->wo, g->wqkv};
        const int N[4] = {H, I, H, 3 * H}, K[4] = {I, H, H, H};
        int lda[4], ldb[4], ldc[4];
        for (int i = 0; i < 4; ++i) { lda[i] = 3 * N[i]; ldb[i] = 3 * K[i]; ldc[i] = K[i]; }
        for (int term = 0; term < 3; ++term) {                      // (hi, hi), (hi, lo), (lo, hi)
            const void* A[4]; const void* Bm[4];
            for (int i = 0; i < 4; ++i) {
                A[i] = Ai[i] + (term == 2 ? 2 * N[i] : 0);
                Bm[i] = Bi[i] + (term == 1 ? 2 * K[i] : 0);
            }
            RET_IF(amdseg_gemm_tn_grouped_bias_impl(4, A, lda, Bm, ldb, C, ldc, N, K, M, term == 0 ? acc : 1, nullptr, nullptr, s, c->pad_runs,
                                                    c->pad_counts, c->pad_guard));
        }
        return AMDSEG_OK;
    }
    // rows of trailing padding carry exact-zero gradients all the way down (amdseg.h, amdseg_bert_cfg.pad_guard): the dgrad GEMMs skip
    // their all-padding row tiles, the weight-gradient GEMM walks only the listed token tiles
#define ZPAD (c->pad_guard ? c->kend : nullptr), c->pad_guard, c->L
    if (PHASE1(c)) {
    // LN2 backward: dz2 (residual grad), d_out = masked dz2 (grad of the FFN output dense), dln2, db2
    RET_IF(amdseg_ln_bwd_impl(dy, a->z2, a->mean2, a->rstd2, p->ln2_g, w->dz2, drop ? w->dbr2 : nullptr, part_ln2, g->ln2_g, g->ln2_b,
                              g->b2, M, H, c->p_hidden, site_seed(c->seed, li, 2), acc, c->dtype, s, ZPAD, nullptr, a->drop2));
    // du = (d_out . W2) * gelu'(u)
    RET_IF(amdseg_gemm_nt_impl(d_out, H, p->w2_t, H, w->du, I, M, I, H, AMDSEG_EPI_GELU_BWD | (c->act ? AMDSEG_EPI_ACT_TANH : 0) | ffn_keep_deriv(c),
                               nullptr, a->u, I, nullptr, 0, 0, s, ZPAD));
    // dx1 = du . W1 + dz2
    RET_IF(amdseg_gemm_nt_impl(w->du, I, p->w1_t, I, w->dx1, H, M, H, I, AMDSEG_EPI_ADD_RES, nullptr, w->dz2, H, nullptr, 0, 0, s, ZPAD));
    // (db1 = colsum(du) and dbqkv = colsum(dqkv) come out of the grouped weight-gradient GEMM below)
    // LN1 backward
    RET_IF(amdseg_ln_bwd_impl(w->dx1, a->z1, a->mean1, a->rstd1, p->ln1_g, w->dz1, drop ? w->dbr1 : nullptr, part_ln1, g->ln1_g,
                              g->ln1_b, g->bo, M, H, c->p_hidden, site_seed(c->seed, li, 1), acc, c->dtype, s, ZPAD, nullptr, a->drop1));
    // dctx = d_ao . Wo
    RET_IF(amdseg_gemm_nt_impl(d_ao, H, p->wo_t, H, w->dctx, H, M, H, H, AMDSEG_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 0, s, ZPAD));
    }
    const int NP = NPROJ(c);
    if (PHASE2(c)) {
    const void* keep_b = c->p_attn > 0.f ? a->keep : nullptr;
    if (c->mixer == 0)
        RET_IF(amdseg_attn_bwd_impl(a->qkv, mask_bias, a->ctx, w->dctx, a->lse, w->delta, w->dqkv, c->B, c->L, c->heads, 0.125f, c->p_attn,
                                    site_seed(c->seed, li, 0), c->window, c->nglobal, s, c->kend, c->seq_order, c->pad_guard,
                                    c->p_attn > 0.f ? a->keep : nullptr,
                                    // phase 6 of a band layer with global tokens: their dctx rows count as zero (amdseg.h, `phase`)
                                    (c->phase == 6 && c->window > 0) ? c->nglobal : 0));
    // dx_in = dqkv . Wqkv + dz1   (external mixer: the caller wrote ws.dqkv [M, nproj*H] between the phases)
    RET_IF(amdseg_gemm_nt_impl(w->dqkv, NP, p->wqkv_t, NP, dx_in, H, M, H, NP, AMDSEG_EPI_ADD_RES, nullptr, w->dz1, H, nullptr, 0, 0, s, ZPAD));
    }
    if (!PHASE_WGRAD(c)) return AMDSEG_OK;
    // all four weight gradients of the layer in one grouped launch: dW = dY^T X
    const void* A[4] = {d_out, w->du, d_ao, w->dqkv};
    const void* Bm[4] = {a->h, a->x1, a->ctx, a->x_in};
    float* C[4] = {g->w2, g->w1, g->wo, g->wqkv};
    const int lda[4] = {H, I, H, NP}, ldb[4] = {I, H, H, H}, ldc[4] = {I, H, H, H};
    const int N[4] = {H, I, H, NP}, K[4] = {I, H, H, H};
    float* cs_out[4] = {nullptr, g->b1, nullptr, g->bqkv};
    float* cs_scr[4] = {nullptr, part_b1, nullptr, part_bqkv};
    RET_IF(amdseg_gemm_tn_grouped_bias_impl(4, A, lda, Bm, ldb, C, ldc, N, K, M, acc, cs_out, cs_scr, s, c->pad_runs, c->pad_counts, c->pad_guard));
    return AMDSEG_OK;
}

}  // extern "C"
