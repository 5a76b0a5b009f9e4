// Internal (C++-side) launcher prototypes shared by the kernel translation units and api.cpp.
#pragma once
#include <new>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AMDSEG_MAX_GROUP 8

// dtype codes used across the C-ABI
#define AMDSEG_BF16 0
#define AMDSEG_F32 1
#define AMDSEG_F32S 2

const char* amdseg_comm_error_string_impl(int code);     /* csrc/comm.hip: AMDSEG_ERR_COMM_* */
int amdseg_gemm_nt_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        int epi, const float* bias, const void* R, int ldr, void* C2, int ldc2, int out_fp32,
                        hipStream_t stream, const int* zkend = nullptr, const int* zguard = nullptr, int zL = 0);
int amdseg_gemm_tn_grouped_impl(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* C, const int* ldc, const int* N, const int* Kp, int M, int accumulate,
                                hipStream_t stream);
int amdseg_gemm_f32_nt_impl(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                            int epi, const float* bias, hipStream_t stream);

int amdseg_embed_ln_fwd_impl(const int64_t* ids, const int64_t* type_ids, const float* word, const float* pos,
                             const float* type, const float* gamma, const float* beta, void* z, void* out, float* mean,
                             float* rstd, int M, int L, int H, int vocab, int type_vocab, int npos,
                             const int64_t* pos_ids, float eps, float p, uint64_t seed, int dtype, hipStream_t s);
int amdseg_scatter_rows_sorted_impl(const void* dz, const int64_t* keys, const int64_t* order, float* table, int M, int H, int nrows,
                                    long skip_key, int dtype, hipStream_t s);
int amdseg_embed_bwd_impl(const void* dz, const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids,
                          float* dword, float* dpos, float* dtype_emb, int M, int L, int H, int vocab, int type_vocab,
                          int npos, int pad_id, int dtype, hipStream_t s);
int amdseg_add_ln_fwd_impl(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out,
                           float* mean, float* rstd, int M, int H, float eps, float p, uint64_t seed, int dtype,
                           hipStream_t s, void* out_image = nullptr,         // out_image: `out` also as the split image [M, 3H] ("parity" precision)
                           void* keepbits = nullptr,                         // keepbits: [M * H / 8] bytes, the dropout decisions kept for ln_bwd
                           bool keep_z = true);                              // false (inference): z = resid + dropout(y) is not written back (only backward reads it)
int amdseg_add_ln_fwd_km_impl(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out, float* mean, float* rstd, int M, int H,
                              float eps, float p, uint64_t seed, int dtype, hipStream_t s, void* keepbits, bool keep_z,
                              void* keep, int B, int L, int heads, float p_attn, uint64_t seed_attn, const int* kend, int window, int nglobal);
int amdseg_ln_bwd_impl(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                       void* dz, void* dbranch, float* partials, float* dgamma, float* dbeta, float* dbias, int M,
                       int H, float p, uint64_t seed, int accumulate, int dtype, hipStream_t s,
                       const int* zkend = nullptr, const int* zguard = nullptr, int zL = 0,
                       void* dense_grad_image = nullptr,     // the dense layer's gradient (dbranch, or dz without dropout) also as the split image [M, 3H]
                       const void* keepbits = nullptr);      // the forward's dropout decisions (amdseg_add_ln_fwd_impl keepbits) instead of the hash
int amdseg_colsum_impl(const void* x, int ld, float* partials, float* out, int M, int N, int accumulate, int dtype,
                       hipStream_t s);
int amdseg_colsum_split_impl(const void* x, int ld, int lo_off, float* partials, float* out, int M, int N, int accumulate, hipStream_t s);
int amdseg_dropout_impl(const void* x, void* y, size_t n, float p, uint64_t seed, int dtype_in, int dtype_out,
                        hipStream_t s);
int amdseg_cast_transpose_impl(const float* W, void* Wb, void* Wt, int N, int K, hipStream_t s);
void amdseg_reduce_defer_begin(int accumulate);
// out[j] (+)= sum_b partials[b * stride + j], j < n (queued if a deferred batch is open)
void amdseg_reduce_rows(const float* partials, int nblocks, int stride, int n, float* out, int accumulate, hipStream_t s);
int amdseg_gemm_tn_grouped_bias_impl(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                     float* const* C, const int* ldc, const int* N, const int* Kp, int M, int accumulate,
                                     float* const* colsum_out, float* const* colsum_scratch, hipStream_t stream,
                                     const int* runs = nullptr, const int* counts = nullptr, const int* zguard = nullptr);
int amdseg_reduce_defer_flush(hipStream_t s);
int amdseg_attn_list_fwd_impl(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                              const int* klist, const int* kcnt, int list_stride, const int* korder, hipStream_t s);
int amdseg_attn_list_bwd_impl(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                              float* delta, void* dqkv, int B, int L, int heads, float scale, const int* klist, const int* kcnt,
                              const int* qlist, const int* qcnt, int list_stride, const int* korder, const int* qorder, hipStream_t s);
int amdseg_cast_transpose_batched_impl(int n, const float* const* W, void* const* Wb, void* const* Wt, const int* N, const int* K,
                                       hipStream_t s, const int* only_if = nullptr);
int amdseg_weights_changed_impl(const void* x, size_t nbytes, void* state, int* changed, hipStream_t s);
int amdseg_pad_plan_impl(const int64_t* mask, int B, int L, int* kend, int* seq_order, int* runs, int* counts, float* mask_bias, float bias,
                         hipStream_t s);
int amdseg_pad_rows_guard_impl(const float* x, const int* kend, int B, int L, int H, int* guard, hipStream_t s);
int amdseg_cast_impl(const void* x, void* y, size_t n, int dtype_in, int dtype_out, hipStream_t s);

int amdseg_attn_fwd_impl(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads,
                         float scale, float p, uint64_t seed, int window, int nglobal, hipStream_t s, const int* kend = nullptr, const int* seq_order = nullptr,
                         const void* keep = nullptr, int skip_q = 0);
int amdseg_attn_bwd_impl(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta, void* dqkv, int B, int L, int heads, float scale, float p, uint64_t seed,
                         int window, int nglobal, hipStream_t s, const int* kend = nullptr, const int* seq_order = nullptr,
                         const int* qguard = nullptr, const void* keep = nullptr, int skip_q = 0);
size_t amdseg_attn_keepmask_bytes_impl(int B, int L, int heads);
int amdseg_attn_keepmask_impl(void* keep, int B, int L, int heads, float p, uint64_t seed, const int* kend, hipStream_t s, int window = 0,
                              int nglobal = 0);
// attention_split.hip: "parity" precision attention on split-bf16 images (hi at column 0, lo at column lo_*) -- full and band
int amdseg_sattn_fwd_impl(const void* qs, int ldq, int lo_q, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale,
                          float p, const void* keep, int window, int nglobal, hipStream_t s, const int* kend = nullptr, const int* seq_order = nullptr,
                          void* ctx_image = nullptr);
int amdseg_sattn_bwd_impl(const void* qs, int ldq, int lo_q, const float* mask_bias, const float* ctx, const void* dos, int ldo, int lo_o,
                          const float* lse, float* delta, float* dqkv, int B, int L, int heads, float scale, float p, const void* keep, int window,
                          int nglobal, hipStream_t s, const int* kend = nullptr, const int* seq_order = nullptr, const int* qguard = nullptr,
                          void* dqs_image = nullptr, int ldd = 0);
int amdseg_attn_f32_impl(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, int d,
                         float scale, int window, int nglobal, hipStream_t s);

int amdseg_attn_list_f32_impl(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale,
                              const int* klist, const int* kcnt, int stride, hipStream_t s);

int amdseg_lf_rowvec_dot_impl(const void* x, const float* vec, const float* add_tok, const float* add_bh, float* out, int B, int L,
                              int H, int heads, int dtype, int ldx, hipStream_t s);
int amdseg_lf_softmax_fwd_impl(float* s_inout_p, float* pd, float* sp, int rows, int L, float p, uint64_t seed, hipStream_t s);
int amdseg_lf_softmax_bwd_impl(const float* p_saved, float* dpd_inout_ds, float* pd, int rows, int L, float p, uint64_t seed,
                               hipStream_t s);
int amdseg_lf_wsum_impl(const void* x, const float* coef, float* partials, float* y, int B, int L, int H, int heads, int dtype,
                        int ldx, hipStream_t s);
int amdseg_lf_dx_update_impl(void* dx, const float* coefA, const float* vecA, const float* coefB, const float* vecB, void* vt_ws,
                             int B, int L, int H, int heads, int dtype, int ldx, int assign, hipStream_t s);

int amdseg_lf_dx_prep_impl(const float* vecA, const float* vecB, void* vt_ws, int B, int L, int H, int heads, hipStream_t s);
int amdseg_lf_dx_apply_impl(void* dx, int ldx, const float* coefA, const float* coefB, const void* vt_ws, const float* trow, int B, int L, int H,
                            int heads, hipStream_t s);
int amdseg_lf_global_bwd_dx_impl(const float* Wq, const float* Wk, const float* dr, float* dqg, float* trow, int B, int L, int H, int heads,
                                 float scale, hipStream_t s);
int amdseg_lf_global_bwd_w_impl(const void* x, int x_dtype, const float* qg, const float* dout, const float* y, const float* sp, const float* dr,
                                const float* dqg, float* dWq, float* dbq, float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads,
                                hipStream_t s);

int amdseg_split3_weights_batched_impl(int n, const float* const* W, void* const* out, void* const* out_t, const int* N, const int* K, hipStream_t s);
size_t amdseg_ponet_global_scratch_floats_impl(int B, int L, int H, int heads);
int amdseg_ponet_global_fwd_impl(const void* hq, const void* hk, int ld, const float* coef_mean, const float* mask_bias, int B, int L, int H,
                                 int heads, float p, uint64_t seed, float* scratch, float* vecq, float* scores, float* lse, float* g,
                                 hipStream_t s);
int amdseg_ponet_global_bwd_impl(const void* hk, int ld, const float* coef_mean, const float* vecq, const float* scores, const float* lse,
                                 const float* dg, int B, int L, int H, int heads, float p, uint64_t seed, float* scratch, float* dpd_ws,
                                 void* dhq, void* dhk, int ldd, hipStream_t s);
int amdseg_ponet_plan_impl(const float* mask_bias, const int* run_start, int* work, int B, int L, hipStream_t s);
int amdseg_ponet_pool_fwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                               const float* g, void* part, void* parg, void* ctx, int B, int L, int H, hipStream_t s);
int amdseg_ponet_pool_bwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                               const float* g, const void* part, const void* parg, const void* dctx, void* dproj, float* dg, float* psum,
                               int B, int L, int H, hipStream_t s);

int amdseg_rowdot_fwd_impl(const void* x, const float* W, const float* b, float* out, int M, int H, int C, int dtype,
                           hipStream_t s);
int amdseg_rowdot_bwd_impl(const void* x, const float* W, const float* dlogits, void* dx, float* partials, float* dW,
                           float* db, int M, int H, int C, int accumulate, int dtype, hipStream_t s);

int amdseg_adamw_impl(float* p, const float* g, float* m, float* v, void* shadow, size_t n, float lr, float beta1,
                      float beta2, float eps, float wd, int step, const float* gscale, int zero_grad,
                      const unsigned char* chunk_flags, hipStream_t s);
int amdseg_sumsq_impl(const float* x, size_t n, float* partials, float* out, int accumulate, hipStream_t s);
int amdseg_clip_coef_impl(const float* sumsq, float max_norm, float extra_scale, float* coef, float* norm, hipStream_t s);
int amdseg_scale_impl(float* x, size_t n, const float* coef, hipStream_t s);

// parity.hip
int amdseg_split3_impl(const float* x, int ld, void* out, int M, int K, int order, hipStream_t s);
int amdseg_split3_transpose_impl(const float* W, void* out, int N, int K, hipStream_t s);
int amdseg_gelu_fwd_split_impl(const float* u, void* hs, int M, int I, int act, hipStream_t s);
int amdseg_gelu_bwd_split_impl(float* du, const float* u, void* dus, int M, int I, int act, hipStream_t s);
int amdseg_add_inplace_impl(float* y, const float* x, size_t n, hipStream_t s);
int amdseg_pattn_fwd_impl(const float* qkv, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale, float p,
                          uint64_t seed, hipStream_t s, const int* kend = nullptr, const int* seq_order = nullptr);
int amdseg_pattn_bwd_impl(const float* qkv, const float* mask_bias, const float* ctx, const float* dctx, const float* lse, float* delta,
                          float* dqkv, int B, int L, int heads, float scale, float p, uint64_t seed, hipStream_t s, const int* kend = nullptr,
                          const int* seq_order = nullptr, const int* qguard = nullptr);

// heads.hip
int amdseg_heads_fwd_impl(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                          float* ce_unit, float* out8, float* acc, const int64_t* idx, long feat_off, long anchor_off, long lists_off,
                          int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off,
                          long t_labels_off, int nt, int Ct, float w_ts, float w_cl, float w_tssp2, hipStream_t s, float focal_gamma = 0.f);
int amdseg_heads_bwd_ce_impl(const float* gout, int M, int C, int nseg, const float* ce_unit, const float* out8, float w_ts, float* dlogits,
                             hipStream_t s, float focal_gamma = 0.f);
int amdseg_heads_bwd_rows_impl(const float* gout, const float* x, int M, int H, float* dx, const int64_t* idx, long feat_off, long anchor_off,
                               long lists_off, int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off,
                               long t_labels_off, int nt, int Ct, float* dWt, float* dbt, float w_cl, float w_tssp2, int n_feat, void* fix,
                               size_t fix_bytes, hipStream_t s);

// lf_global.hip
int amdseg_lf_global_q_impl(const void* x, int x_dtype, const float* Wq, const float* bq, const float* Wk, float* qg, float* r, int B, int L,
                            int H, int heads, float scale, hipStream_t s);
int amdseg_lf_global_out_impl(const float* Wv, const float* bv, const float* y, const float* sp, void* ctx, int ctx_dtype, int B, int L, int H,
                              int heads, hipStream_t s);
int amdseg_lf_global_bwd_a_impl(void* dctx, int dtype, const float* Wv, const float* bv, float* dout, float* dyv, float* dsp, int B, int L,
                                int H, int heads, hipStream_t s, int keep_dctx = 0);
int amdseg_lf_global_bwd_rest_impl(const void* x, int x_dtype, void* dx, int dx_dtype, const float* Wq, const float* Wk, const float* qg,
                                   const float* dout, const float* y, const float* sp, const float* dr, float* dqg, float* dWq, float* dbq,
                                   float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads, float scale, hipStream_t s);

// CUs the launch-geometry rules may count on (gemm_dp.hip)
int amdseg_cu_budget();
int amdseg_num_cus();


// ---- the explicit library context (csrc/prof.hip; include/amdseg.h amdseg_ctx_*)
struct AmdsegProfRec { int cls; double work; hipEvent_t e0, e1; };
struct amdseg_ctx {
    int cu_budget = 0;                  // CUs the tile rules count on (0 = all): an overlapped RCCL exchange holds one CU per channel
    int force_small_tile = 0;           // test hook: exercise the 128 x 128 kernels on big shapes
    bool prof_on = false;               // launch timer
    std::vector<AmdsegProfRec> recs;    // [0, used) are live, the rest are pooled events of earlier rounds
    size_t used = 0;
    bool overflow = false;
};
amdseg_ctx* amdseg_current_ctx();       // the context of the call in progress / bound to this thread; may be NULL (defaults)
struct AmdsegCtxScope {                 // a composite entry point runs under its cfg's context (when it has one)
    amdseg_ctx* prev; bool active;
    explicit AmdsegCtxScope(amdseg_ctx* c);
    ~AmdsegCtxScope();
};
static inline int amdseg_force_small_tile() { amdseg_ctx* c = amdseg_current_ctx(); return c ? c->force_small_tile : 0; }
