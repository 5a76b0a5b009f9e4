/* libamdseg -- C ABI of the MI355X-native (gfx950) BERT token-classification encoder path.
 *
 * Drop-in boundary for the topic-segmentation hot path of alibaba-damo-academy/SpokenNLP.  The reference has no
 * native layer at all: every entry point below replaces a torch/ATen call made (through HuggingFace
 * `transformers`) from the reference's wrapper model
 *     emnlp2023-topic_segmentation/src/models/bert_for_ts.py:55-82   self.bert(...)      -> encoder layers
 *     emnlp2023-topic_segmentation/src/models/modules/loss_calculator.py:42              -> classifier
 *     transformers.Trainer (ts_sentence_seq_labeling.py:1077-1094)                       -> clip + AdamW
 * and is bound from Python with ctypes (spokennlp_amd/lib.py).
 *
 * Conventions
 *  - plain C: raw device pointers, ints, floats; no torch / C++ types cross this boundary;
 *  - every call takes a stream (hipStream_t passed as void*), is asynchronous, and returns 0 on success,
 *    AMDSEG_ERR_* (>= 1000) for argument/shape errors, or the hipError_t of a failed launch;
 *  - the caller owns every buffer (activations, saved-for-backward tensors, workspaces); the library keeps no
 *    global state;  one process per GPU, calls serialised per stream;
 *  - activations are row-major [tokens, features]; `dtype` selects bf16 (AMDSEG_BF16, the fast path) or fp32
 *    (AMDSEG_F32) for the HBM-bound kernels; MFMA GEMMs take bf16 operands and accumulate in fp32;
 *  - weights follow torch.nn.Linear: W[out, in] row-major.
 */
#ifndef AMDSEG_H
#define AMDSEG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AMDSEG_ABI_VERSION 14   /* 14: amdseg_add_ln_fwd_keepmask (the LayerNorm rows and the NEXT layer's attention-dropout keep masks as one launch), amdseg_bert_layer_acts.keep_next / .keep_ready; 13: the explicit context `amdseg_ctx` -- amdseg_ctx_create / _bind / _set_cu_budget / _prof_*, amdseg_bert_cfg.ctx -- replaces the per-thread CU budget, the small-tile hook and the process-wide launch timer of ABI 10-12; the one-kernel attention backward of ABI 12 (+ ws.dq_part) and the fused bias + dropout + residual GEMM of ABI 8 -- measured losers -- are gone; 11: amdseg_allreduce_* (the gradient exchange over RCCL behind an explicit amdseg_comm context, csrc/comm.hip); 10: AMDSEG_EPI_KEEP_DERIV (amdseg_bert_layer_acts.u holds gelu' of the FFN pre-activation in bf16 training when the shape allows); 9: amdseg_heads_bwd_rows takes n_feat / fix / fix_bytes (order-independent scatter sums), amdseg_scatter_rows_sorted; 8: amdseg_bert_layer_acts.drop1 / .drop2 (the hidden-dropout decisions of a layer kept by forward for backward), amdseg_add_ln_fwd with resid == NULL, AMDSEG_PROF_ADD_LN_FWD .. _KEEPMASK; 7: forward phase 1 of a bf16 band layer with global tokens leaves their ctx rows unwritten (amdseg_bert_cfg.phase), amdseg_lf_global_bwd_dx / _w, amdseg_lf_dx_prep / _apply; 6: amdseg_attn_keepmask / _fwd_keep / _bwd_keep, amdseg_bert_layer_acts.keep / .qkv_s, amdseg_bert_layer_ws.dctx_s, amdseg_sattn_*, amdseg_weights_changed, amdseg_cast_transpose_batched_if, AMDSEG_EPI_BIAS_SPLIT; 5: amdseg_bert_cfg.pad_guard / pad_runs / pad_counts, amdseg_pad_rows_guard; 4: amdseg_bert_cfg.kend (trailing-padding chunks of full attention are not visited); 3: amdseg_adamw chunk_flags, AMDSEG_F32S parity mode (acts / ws split images); 2: amdseg_bert_cfg.act, ws.partials regions, list attention, grouped TN with bias gradients */
#define AMDSEG_BF16 0
#define AMDSEG_F32 1
#define AMDSEG_F32S 2   /* composite layer only: fp32 activations, split-bf16 contractions ("parity" precision, forward + backward) */
#define AMDSEG_OK 0
#define AMDSEG_ERR_SHAPE 1001
#define AMDSEG_ERR_ARG 1002
#define AMDSEG_ERR_LAUNCH 1003
#define AMDSEG_ERR_COMM_LIB 1004     /* amdseg_allreduce_*: librccl could not be loaded */
#define AMDSEG_ERR_COMM_BASE 1100    /* amdseg_allreduce_*: 1100 + ncclResult_t of the failed RCCL call */
#define AMDSEG_MAX_GROUP 8

/* epilogues of amdseg_gemm_nt */
#define AMDSEG_EPI_NONE 0       /* C = A B^T                                                     */
#define AMDSEG_EPI_BIAS 1       /* C = A B^T + bias[n]                                           */
#define AMDSEG_EPI_BIAS_GELU 2  /* C2 = A B^T + bias (pre-activation; C2 may be NULL), C = gelu_erf  */
#define AMDSEG_EPI_ADD_RES 3    /* C = A B^T + R                                                  */
#define AMDSEG_EPI_GELU_BWD 4   /* C = (A B^T) * gelu_erf'(R)                                     */
#define AMDSEG_EPI_BIAS_SPLIT 5 /* C = bf16 hi of (A B^T + bias), C2 = bf16 lo = bf16(value - hi): the result as a split-bf16 image ("parity"
                                   precision, amdseg_sattn_*); M % 256 == 0, N % 256 == 0, K >= 128 (other shapes: AMDSEG_ERR_SHAPE); bias may be NULL   */
#define AMDSEG_EPI_GELU_BWD_SPLIT 6 /* C = bf16 image [M, 3N] (ldc >= 3N) = [hi | hi | lo] of (A B^T) * gelu_erf'(R), R the fp32 pre-activation
                                   (ldr in floats): the FFN input gradient of "parity" precision straight in the form the next split GEMM and the
                                   weight gradient read; same shape rule as BIAS_SPLIT; C2 unused                                        */
#define AMDSEG_EPI_BIAS_GELU_SPLIT 7 /* out_fp32 = 1: C = A B^T + bias (fp32 pre-activation), C2 = bf16 image [M, 3N] (ldc2 >= 3N) = [hi | hi | lo] of
                                   gelu_erf(C): the FFN activation of "parity" precision in the form the next split GEMM reads; same shape rule */
#define AMDSEG_EPI_KEEP_DERIV 0x200 /* OR-ed into BIAS_GELU: C2 receives gelu'(A B^T + bias) (bf16) instead of the pre-activation; OR-ed into GELU_BWD: R holds
                                       that derivative, C = (A B^T) * R -- the backward GEMM's epilogue is one multiply instead of a second GELU evaluation.
                                       erf GELU only, shapes of the deep-pipeline kernel only (M % 256 == 0, N % 192 == 0 or N % 256 == 0, K >= 128):
                                       AMDSEG_ERR_SHAPE otherwise.  What amdseg_bert_layer_fwd / _bwd use between themselves when the shape allows */
#define AMDSEG_EPI_DERIV_U8 0x400 /* with AMDSEG_EPI_KEEP_DERIV: the derivative is ONE BYTE per element, q = round((gelu' + 0.135) * 200) (absolute error <= 0.0025);
                                     C2 / R point to unsigned char [M, ld] (ld in bytes, % 8 == 0); N % 256 == 0; also with AMDSEG_EPI_ACT_TANH (gelu_new) */
#define AMDSEG_EPI_ACT_TANH 0x100 /* OR-ed into BIAS_GELU / GELU_BWD: "gelu_new" (tanh form, BigBird's hidden_act) instead of erf */

typedef void* amdseg_stream_t;  /* hipStream_t */

int amdseg_abi_version(void);
const char* amdseg_error_string(int code);

/* ---- MFMA GEMMs (csrc/gemm.hip) -------------------------------------------------------------------------------
 * gemm_nt: C[M,N] = A[M,K] . B[N,K]^T, bf16 in / fp32 accumulate, M,N multiples of 128, K multiple of 64.
 *   replaces torch.nn.functional.linear in BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput
 *   ([hf] models/bert/modeling_bert.py:175-177,282-293,325-351) and, with transposed weight shadows, its dgrad. */
int amdseg_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epilogue,
                   const float* bias, const void* R, int ldr, void* C2, int ldc2, int out_fp32, amdseg_stream_t stream);
/* gemm_tn_grouped: for each problem i: C_i[N_i,K_i] (+)= sum_m A_i[m,N_i] . B_i[m,K_i]  (fp32 out), shared M.
 *   the weight gradients dW = dY^T X of one encoder layer in ONE launch (autograd of the Linear layers above). */
int amdseg_gemm_tn_grouped(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                           float* const* C, const int* ldc, const int* N, const int* K, int M, int accumulate,
                           amdseg_stream_t stream);
/* the same plus the bias gradients of those Linear layers: colsum_out[i][N_i] (+)= sum_m A_i[m, :]  (entries / the array may be
 *   NULL); colsum_scratch[i]: >= max(ceil(M/128), K_i/128) * N_i floats.  On the 256 x 128 kernel the sums are taken from the GEMM's own A
 *   fragments (no second pass over dY). */
int amdseg_gemm_tn_grouped_bias(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* C, const int* ldc, const int* N, const int* K, int M, int accumulate,
                                float* const* colsum_out, float* const* colsum_scratch, amdseg_stream_t stream);

/* fp32 parity-mode GEMM (csrc/gemm_f32.hip): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), epilogue 0 none / 1 bias /
 * 2 bias+gelu_erf; M,N multiples of 128, K multiple of 32.  Same reference lines as amdseg_gemm_nt. */
int amdseg_gemm_f32_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int epilogue,
                       const float* bias, amdseg_stream_t stream);
/* fp32 attention (inference, no dropout), qkv [B*L, 3*heads*64] fp32 -> ctx [B*L, heads*64] fp32 */
int amdseg_attn_f32(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale,
                    amdseg_stream_t stream);

/* ---- fused attention (csrc/attention.hip), head_dim 64, L multiple of 64 -----------------------------------------
 * qkv [B*L, 3*heads*64] bf16 (q|k|v), mask_bias [B, L] fp32 additive key mask (0 or a large negative); scale is folded into the
 * register-resident operand (exact for a power of two such as 1/8; one more bf16 rounding of q / k otherwise),
 * ctx [B*L, heads*64] bf16, lse [B, heads, L] fp32 (saved for backward; may be NULL for inference).
 *   replaces eager_attention_forward ([hf] models/bert/modeling_bert.py:111-136) and its backward. */
int amdseg_attn_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                    float dropout_p, uint64_t seed, amdseg_stream_t stream);
int amdseg_attn_bwd(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                    float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, uint64_t seed,
                    amdseg_stream_t stream);

/* Dropout on the attention probabilities, decided ONCE per layer and step (ABI 6).  amdseg_attn_fwd / _bwd evaluate a stateless hash of
 * (seed, row, key) per element in each of their three kernels; these kernels are VALU-bound and that hash is ~45 % of their vector
 * instructions.  amdseg_attn_keepmask writes the keep decisions of one layer -- Bernoulli(1 - round(p * 2^16) / 2^16) bits, the same realised
 * rate as the hash -- into `keep` (amdseg_attn_keepmask_bytes(B, L, heads) = 2 * B * heads * L * L / 8 bytes) in the two lane-mask layouts the
 * kernels consume (csrc/attention.hip, "dropout keep masks": layout A for forward and dQ, its bit transpose B for dK / dV); the _keep forms of
 * forward and backward read them with scalar loads and apply them with one v_cndmask per probability.  kend: optional [B] as in
 * amdseg_bert_cfg.kend (chunks past it are not generated); keep == NULL or dropout_p == 0 falls back to the hash path.  Full attention only. */
size_t amdseg_attn_keepmask_bytes(int B, int L, int heads);
int amdseg_attn_keepmask(void* keep, int B, int L, int heads, float dropout_p, uint64_t seed, const int32_t* kend, amdseg_stream_t stream);
/* the same for band attention (amdseg_sattn_* with window > 0): only the (64-query block, 64-key chunk) cells inside the band, plus chunk 0
 * when nglobal > 0, are written (the buffer has the full-attention size) */
int amdseg_attn_keepmask_band(void* keep, int B, int L, int heads, float dropout_p, uint64_t seed, int window, int nglobal, amdseg_stream_t stream);
int amdseg_attn_fwd_keep(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                         float dropout_p, const void* keep, amdseg_stream_t stream);
int amdseg_attn_bwd_keep(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, const void* keep,
                         amdseg_stream_t stream);
/* the band kernels on keep masks generated by amdseg_attn_keepmask_band (same window / nglobal): dropout decisions of the Longformer layers */
int amdseg_attn_band_fwd_keep(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                              float dropout_p, const void* keep, int window, int nglobal, amdseg_stream_t stream);
int amdseg_attn_band_bwd_keep(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                              float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, const void* keep,
                              int window, int nglobal, amdseg_stream_t stream);

/* "Parity" precision attention on the bf16 matrix cores (csrc/attention_split.hip): every contraction of amdseg_attn_fwd / _bwd as a
 * split-bf16 product (x = hi + lo; hi.hi + hi.lo + lo.hi), fp32 softmax, accumulators and outputs.  qs = the amdseg_split3 image of the fp32
 * q|k|v projection: [B*L, ldq] bf16 with the hi parts at columns [0, 3*heads*64) and the lo parts at [lo_q, lo_q + 3*heads*64) (order 0:
 * ldq = 9H, lo_q = 6H); dos = the same for d(ctx) ([B*L, ldo], lo at lo_o; 3H / 2H); ctx, dqkv, lse, delta fp32.  dropout_p > 0 reads the
 * decisions from `keep` (amdseg_attn_keepmask; full attention only); window > 0 = the Longformer band (nglobal as amdseg_attn_band_fwd).
 * Replaces the same reference lines as amdseg_attn_fwd at the precision the reference runs in (run_finetune.sh:61-96: fp32). */
int amdseg_sattn_fwd(const void* qs, int ldq, int lo_q, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale,
                     float dropout_p, const void* keep, int window, int nglobal, amdseg_stream_t stream);
int amdseg_sattn_bwd(const void* qs, int ldq, int lo_q, const float* mask_bias, const float* ctx, const void* dos, int ldo, int lo_o,
                     const float* lse, float* delta_ws, float* dqkv, int B, int L, int heads, float scale, float dropout_p, const void* keep,
                     int window, int nglobal, amdseg_stream_t stream);

/* ---- Longformer attention (csrc/attention.hip band variants + csrc/longformer.hip global row) -------------------
 * Replaces LongformerSelfAttention.forward ([hf] models/longformer/modeling_longformer.py:482-640: sliding chunks
 * :759-868, global key columns :559-604, global rows :964-1058) as driven by the reference wrapper
 * emnlp2023-topic_segmentation/src/models/longformer_for_ts.py:55-89 (tokens 0..nglobal-1 global; the wrapper uses
 * nglobal = 1, [CLS]).  Band variants: key j visible from query i iff j < nglobal or |i-j| <= window
 * (window = attention_window/2), padded keys carry a large negative mask_bias, rows of padded queries
 * (mask_bias[q] < 0) are zeroed; only the 64-key chunks intersecting the band are streamed. */
int amdseg_attn_band_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                         float dropout_p, uint64_t seed, int window, int nglobal, amdseg_stream_t stream);
int amdseg_attn_band_bwd(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta_ws, void* dqkv, int B, int L, int heads, float scale, float dropout_p, uint64_t seed,
                         int window, int nglobal, amdseg_stream_t stream);
int amdseg_attn_band_f32(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale, int window,
                         int nglobal, amdseg_stream_t stream);
/* Global row ([hf]:964-1058) with the key_global / value_global projections folded onto the query side (see
 * csrc/longformer.hip): x [B*L, H] (dtype bf16/fp32), per-(b, head) vectors [B, heads, H] fp32, per-(b, head, token)
 * coefficient planes [B, heads, L] fp32.  heads <= 16, H <= 1024, L multiple of 64.
 *   rowvec_dot : out[b,h,j] = vec[b,h,:] . x[b,j,:] + add_tok[b,j] (may be NULL) + add_bh[b,h] (may be NULL)
 *   softmax_fwd: rows = B*heads rows of L scores; p over the scores in place, pd = dropout(p), sp[row] = sum(pd)
 *   softmax_bwd: in p (saved) and d(pd) -> ds written over d(pd); pd recomputed
 *   wsum       : y[b,h,:] = sum_j coef[b,h,j] x[b,j,:]   (partials: B * L/64 * heads * H floats)
 *   dx_update  : dx[b,j,:] += sum_h coefA[b,h,j] vecA[b,h,:] + coefB[b,h,j] vecB[b,h,:]
 *                (vt_ws: B*H*32 bf16 scratch for the MFMA path; NULL selects the scalar kernel) */
int amdseg_lf_rowvec_dot(const void* x, const float* vec, const float* add_tok, const float* add_bh, float* out, int B, int L, int H,
                         int heads, int dtype, amdseg_stream_t stream);
int amdseg_lf_softmax_fwd(float* s_inout_p, float* pd, float* sp, int rows, int L, float dropout_p, uint64_t seed,
                          amdseg_stream_t stream);
int amdseg_lf_softmax_bwd(const float* p_saved, float* dpd_inout_ds, float* pd, int rows, int L, float dropout_p, uint64_t seed,
                          amdseg_stream_t stream);
int amdseg_lf_wsum(const void* x, const float* coef, float* partials, float* y, int B, int L, int H, int heads, int dtype,
                   amdseg_stream_t stream);
int amdseg_lf_dx_update(void* dx, const float* coefA, const float* vecA, const float* coefB, const float* vecB, void* vt_ws, int B,
                        int L, int H, int heads, int dtype, amdseg_stream_t stream);

/* strided forms of the three O(L) passes above: x / dx rows are ldx elements apart (a column block of a wider matrix);
 * assign != 0 overwrites dx instead of accumulating.  Used by the PoNet global aggregation below. */
int amdseg_lf_rowvec_dot_ld(const void* x, int ldx, const float* vec, const float* add_tok, const float* add_bh, float* out, int B, int L,
                            int H, int heads, int dtype, amdseg_stream_t stream);
int amdseg_lf_wsum_ld(const void* x, int ldx, const float* coef, float* partials, float* y, int B, int L, int H, int heads, int dtype,
                      amdseg_stream_t stream);
int amdseg_lf_dx_update_ld(void* dx, int ldx, int assign, const float* coefA, const float* vecA, const float* coefB, const float* vecB,
                           void* vt_ws, int B, int L, int H, int heads, int dtype, amdseg_stream_t stream);

/* ---- PoNet token mixing (csrc/ponet.hip) ---------------------------------------------------------------------------
 * Replaces the pooling branches of modelscope's PoNetSelfAttention as called through
 * alimeeting4mug/src/models/modeling_ponet.py:68-79 (source NOT in the reference tree: semantics = oracle/ponet_oracle.py,
 * parity unpinned).  proj [B*L, ld >= 5H] bf16 = (Hq | Hk | Ho | Hl | Hs); run_start / run_end [B*L] int32 = first / last
 * position (within the sequence) of the token's segment run (segment_ids non-decreasing, padding constant per run);
 * g [B, H] fp32 global aggregate.  amdseg_ponet_plan builds, once per batch, the list of run starts in `work` (int32 [2 + 2*B*L]:
 * work[1] = number of runs, work[2 + B*L ..) = token index of every run's first token) from run_start.  part = [B*L, H] 32-bit words:
 * forward leaves, in the row of a run's first token, key = order-preserving image of the bf16 run maximum << 16 | (0xFFFF - argmax
 * position) per column (folded with atomicMax: deterministic), read again by backward; parg is unused (kept for call compatibility, may
 * be NULL); forward writes ctx [B*L, H]; backward writes the Ho, Hl, Hs column blocks of dproj [B*L, ld] and dg [B, H] = per-sequence
 * sum of dctx * Ho (the gradient of g; zeroed here), using psum = [B*L, H] fp32 scratch (run sums, fp32 atomicAdd: run-to-run differences
 * in the last bit).  run_end is not read.  L <= 65535, L % 8 == 0, H % 8 == 0. */
int amdseg_ponet_plan(const float* mask_bias, const int* run_start, int* work, int B, int L, amdseg_stream_t stream);
int amdseg_ponet_pool_fwd(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                          const float* g, void* part, void* parg, void* ctx, int B, int L, int H, amdseg_stream_t stream);
int amdseg_ponet_pool_bwd(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                          const float* g, const void* part, const void* parg, const void* dctx, void* dproj, float* dg, float* psum, int B,
                          int L, int H, amdseg_stream_t stream);
/* The global aggregation branch of the same mixer (csrc/ponet_global.hip), bf16 column blocks Hq / Hk of the projection (row stride ld):
 *   qbar_b = sum_j coef_mean[b, j] Hq[b, j, :];  s[b, h, j] = qbar_b[head h] . Hk[b, j, head h] / 8 + mask_bias[b, j];  p = softmax_j s;
 *   g[b, c] = sum_j dropout(p)[b, head(c), j] Hk[b, j, c]     (heads of 64 columns, H = heads * 64 <= 1024, L % 64 == 0)
 * forward writes vecq = qbar / 8 [B, H], the raw scores [B, heads, L], lse [B, heads] (all three read again by backward) and g [B, H];
 * backward takes dg [B, H] (amdseg_ponet_pool_bwd) and WRITES the dHq and dHk column blocks (row stride ldd).  scratch:
 * amdseg_ponet_global_scratch_floats(B, L, H, heads) floats = B (L/64 (H + 2 heads) + H); dpd_ws: [B, heads, L] floats.  Dropout decisions are those of amdseg_lf_softmax_fwd
 * with the same seed.  Replaces the amdseg_lf_* formulation of rounds 1-2 (10 / 12 launches and 12 x the arithmetic per layer). */
size_t amdseg_ponet_global_scratch_floats(int B, int L, int H, int heads);
int amdseg_ponet_global_fwd(const void* hq, const void* hk, int ld, const float* coef_mean, const float* mask_bias, int B, int L, int H,
                            int heads, float dropout_p, uint64_t seed, float* scratch, float* vecq, float* scores, float* lse, float* g,
                            amdseg_stream_t stream);
int amdseg_ponet_global_bwd(const void* hk, int ld, const float* coef_mean, const float* vecq, const float* scores, const float* lse,
                            const float* dg, int B, int L, int H, int heads, float dropout_p, uint64_t seed, float* scratch, float* dpd_ws,
                            void* dhq, void* dhk, int ldd, amdseg_stream_t stream);

/* ---- HBM-bound row kernels (csrc/elementwise.hip) ---------------------------------------------------------------
 * embeddings + LayerNorm + dropout ([hf] models/bert/modeling_bert.py:53-108); tables are the fp32 masters.
 * pos_ids may be NULL (position = token index % L); z (pre-LN sum), mean, rstd are saved for backward (may be NULL) */
int amdseg_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                        const float* pos, const float* type, const float* gamma, const float* beta, void* z, void* out,
                        float* mean, float* rstd, int M, int L, int H, int vocab, int type_vocab, int npos, float eps,
                        float dropout_p, uint64_t seed, int dtype, amdseg_stream_t stream);
/* scatter of dz [M, H] into the three tables (fp32 atomics; rows with ids == pad_id add nothing to dword).  type_vocab < 0: the table has
   -type_vocab rows and row 0 of dtype_emb ALREADY holds the column sum of dz over all rows (amdseg_ln_bwd's dbias output of the embedding
   LayerNorm): rows of type t != 0 move their gradient from row 0 to row t, rows of type 0 add nothing (no hot-row atomics) */
int amdseg_embed_bwd(const void* dz, const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, float* dword,
                     float* dpos, float* dtype_emb, int M, int L, int H, int vocab, int type_vocab, int npos, int pad_id,
                     int dtype, amdseg_stream_t stream);
/* the same scatter for ONE table without atomics (ABI 9; the engine's deterministic mode): table[keys[r]] += dz[r] for every row r with
   0 <= keys[r] < nrows and keys[r] != skip_key (-1: none), where `order` [M] is a STABLE argsort of keys (torch.sort(keys, stable=True)).
   The rows of one key are summed front to back by one workgroup per 256 columns and added with a plain store: bit-reproducible.
   amdseg_embed_bwd with vocab == 0 / npos == 0 leaves the word / position table to this call. */
int amdseg_scatter_rows_sorted(const void* dz, const int64_t* keys, const int64_t* order, float* table, int M, int H, int nrows,
                               long skip_key, int dtype, amdseg_stream_t stream);
/* z = resid + dropout(y) (written over y), out = LayerNorm(z)   ([hf] modeling_bert.py:282-293, 340-351).
 * resid == NULL (ABI 8): y_inout_z already holds z -- LayerNorm only, y_inout_z is not written, dropout_p ignored */
int amdseg_add_ln_fwd(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out, float* mean,
                      float* rstd, int M, int H, float eps, float dropout_p, uint64_t seed, int dtype,
                      amdseg_stream_t stream);
/* ABI 14: amdseg_add_ln_fwd AND amdseg_attn_keepmask (window == 0) / amdseg_attn_keepmask_band (window > 0) of `keep` as ONE launch: the row
 * kernel is HBM-bound, the mask generator VALU-bound, and as interleaved workgroups of one grid the generator runs under the rows' memory latency
 * (bert-base 32 x 512: 20.7 + 18.5 us as two launches).  Every output is bit-identical to the two calls.  amdseg_bert_layer_fwd uses it for the
 * second LayerNorm of layer li and the masks of layer li + 1 (amdseg_bert_layer_acts.keep_next). */
int amdseg_add_ln_fwd_keepmask(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out, float* mean,
                               float* rstd, int M, int H, float eps, float dropout_p, uint64_t seed, int dtype,
                               void* drop_bits, int keep_z,
                               void* keep, int B, int L, int heads, float attn_dropout_p, uint64_t attn_seed, const int32_t* kend,
                               int window, int nglobal, amdseg_stream_t stream);
/* LayerNorm backward: dz (residual-stream grad), dbranch = dropout-masked dz (NULL when p == 0), and the column
 * reductions dgamma, dbeta, dbias (= colsum(dbranch)); partials = workspace of 3*ceil(M/16)*H floats */
int amdseg_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
                  void* dbranch, float* partials, float* dgamma, float* dbeta, float* dbias, int M, int H,
                  float dropout_p, uint64_t seed, int accumulate, int dtype, amdseg_stream_t stream);
/* out[N] (+)= column sums of x[M, ld];  partials = workspace of ceil(M/128)*N floats  (bias gradients) */
int amdseg_colsum(const void* x, int ld, float* partials, float* out, int M, int N, int accumulate, int dtype,
                  amdseg_stream_t stream);
/* The padding plan of a batch, computed on the device from the int64 attention mask [B, L] (L % 64 == 0, B <= 8192): kend[B], seq_order[B],
   pad_runs[B][2], pad_counts[2] as amdseg_bert_cfg documents them, and (mask_bias != NULL) the additive key mask
   mask_bias[b, p] = (1 - mask[b, p]) * bias ([hf] modeling_utils.py get_extended_attention_mask; the reference has no plan: it multiplies
   the padded positions like any other) */
int amdseg_pad_plan(const int64_t* attention_mask, int B, int L, int32_t* kend, int32_t* seq_order, int32_t* pad_runs, int32_t* pad_counts,
                    float* mask_bias, float bias, amdseg_stream_t stream);
/* *guard = 1 if any element of x (fp32 [B*L, H], H % 4 == 0) in a row at a position >= kend[b] is not an exact zero (NaN counts), else 0:
   the check behind amdseg_bert_cfg.pad_guard.  Reads only those rows.  (No reference counterpart: the reference computes the padded rows.) */
int amdseg_pad_rows_guard(const float* x, const int32_t* kend, int B, int L, int H, int32_t* guard, amdseg_stream_t stream);
int amdseg_dropout(const void* x, void* y, size_t n, float p, uint64_t seed, int dtype_in, int dtype_out,
                   amdseg_stream_t stream);
int amdseg_cast(const void* x, void* y, size_t n, int dtype_in, int dtype_out, amdseg_stream_t stream);
/* fp32 master W[N,K] -> bf16 Wb[N,K] and bf16 transpose Wt[K,N] (either may be NULL); N,K multiples of 64 */
int amdseg_cast_transpose(const float* W, void* Wb, void* Wt, int N, int K, amdseg_stream_t stream);
/* the same for n matrices in one launch (host pointer tables; Wb or Wt may be NULL as a whole): the per-step refresh of
 * the bf16 compute shadows after the optimiser step ([hf] trainer.py optimizer.step -> next forward).  W == NULL as a whole: the
 * bf16 copies Wb are the INPUT (amdseg_adamw wrote them through its `shadow` argument) and only the transposes Wt are written */
int amdseg_cast_transpose_batched(int n, const float* const* W, void* const* Wb, void* const* Wt, const int* N, const int* K,
                                  amdseg_stream_t stream);
/* "were these weights written behind the library's back?", answered on the device.  amdseg_weights_changed: 64-bit content checksum of
 * x[0 .. nbytes) (nbytes % 16 == 0) against the one taken by the previous call; *changed = 1 if it differs (always on the first call);
 * state = 16 bytes of zero-initialised device memory owned by the caller, one per buffer.  amdseg_cast_transpose_batched_if: the batched
 * refresh as a no-op when *only_if == 0.  Together: an inference forward re-derives the bf16 compute copies after ANY write to the fp32
 * masters (`p.data.copy_`, raw pointers), at ~80 us per 340 MB instead of a host read or a manual `mark_weights_dirty()`. */
int amdseg_weights_changed(const void* x, size_t nbytes, void* state, int32_t* changed, amdseg_stream_t stream);
int amdseg_cast_transpose_batched_if(int n, const float* const* W, void* const* Wb, void* const* Wt, const int* N, const int* K,
                                     const int32_t* only_if, amdseg_stream_t stream);
/* Block-list attention = BigBird block-sparse attention ([hf] models/big_bird/modeling_big_bird.py
 * BigBirdBlockSparseAttention.bigbird_block_sparse_attention, reached from the reference through
 * emnlp2023-topic_segmentation/src/models/bigbird_for_ts.py:27 BigBirdModel).  qkv/ctx/lse/mask_bias as amdseg_attn_fwd (bf16,
 * head dim 64, L % 64 == 0, mask_bias = -10000 * (1 - attention_mask) as the reference adds it).  Query block i of head h visits
 * the key blocks klist[(h * L/64 + i) * list_stride + 0 .. kcnt[h * L/64 + i]) in order, duplicates included (one softmax over the
 * concatenation, as the reference's torch.cat of key blocks); qlist/qcnt are the transposed lists per (head, key block) with the
 * same multiplicities (device int32 arrays, built by the host mirror spokennlp_amd/bigbird_plan.py).  Rows of padded queries
 * (mask_bias < 0 at the query's own position) come out zero and carry no gradient (reference: context_layer * from_mask).  korder / qorder (optional, [heads][L/64]): the block index the r-th
 * workgroup of a head works on -- sorted by decreasing list length so that the few very long rows (first / last block) start first. */
int amdseg_attn_list_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                         const int* klist, const int* kcnt, int list_stride, const int* korder, amdseg_stream_t stream);
int amdseg_attn_list_bwd(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta_ws, void* dqkv, int B, int L, int heads, float scale, const int* klist, const int* kcnt,
                         const int* qlist, const int* qcnt, int list_stride, const int* korder, const int* qorder, amdseg_stream_t stream);
/* fp32 parity-mode list attention (inference): qkv / ctx fp32, same lists; L <= 4096 (at most 64 listed key blocks per query block) */
int amdseg_attn_list_f32(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale,
                         const int* klist, const int* kcnt, int list_stride, amdseg_stream_t stream);
/* small-C linear heads: logits[M,C] = x[M,H] W[C,H]^T + b, C <= 4
 * (modules/loss_calculator.py:17,42 classifier; modules/tssp.py:14,31) and their backward */
int amdseg_rowdot_fwd(const void* x, const float* W, const float* b, float* out, int M, int H, int C, int dtype,
                      amdseg_stream_t stream);
int amdseg_rowdot_bwd(const void* x, const float* W, const float* dlogits, void* dx, float* partials, float* dW, float* db,
                      int M, int H, int C, int accumulate, int dtype, amdseg_stream_t stream);

/* ---- optimiser (csrc/optim.hip): torch.optim.AdamW + clip_grad_norm_ semantics over flat fp32 buffers ----------
 * chunk_flags (nullable = decay everywhere): one byte per 64 consecutive elements, bit 0 = weight decay applies, bit 1 = frozen
 * (no update) -- the two parameter groups of [hf] trainer.py:1168-1215 (biases / LayerNorm undecayed) on ONE flat buffer. */
int amdseg_adamw(float* p, const float* g, float* m, float* v, void* bf16_shadow, size_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, const float* grad_scale, int zero_grad,
                 const unsigned char* chunk_flags, amdseg_stream_t stream);
int amdseg_sumsq(const float* x, size_t n, float* partials, float* out, int accumulate, amdseg_stream_t stream);
int amdseg_clip_coef(const float* sumsq, float max_norm, float extra_scale, float* coef, float* norm,
                     amdseg_stream_t stream);
int amdseg_scale(float* x, size_t n, const float* coef, amdseg_stream_t stream);

/* ---- "parity" precision (csrc/parity.hip): fp32 activations, fp32-grade contractions on the bf16 MFMA pipes.  x = hi + lo with
 * hi = bf16(x), lo = bf16(x - hi); a product A . B^T runs as ONE bf16 GEMM over K' = 3K on the images A' = [Ahi | Ahi | Alo],
 * B' = [Bhi | Blo | Bhi] (amdseg_gemm_nt with out_fp32 = 1).  The reference computes in fp32 throughout
 * (run_finetune.sh:61-96: no --fp16 / --bf16); this mode is what the 1e-3 logit tolerance of the north star is pinned with, forward AND
 * backward.  order: 0 = activation / gradient image [hi | hi | lo], 1 = weight image [hi | lo | hi]. */
int amdseg_split3(const float* x, int ld, void* out_bf16, int M, int K, int order, amdseg_stream_t stream);
/* W [N,K] fp32 -> [K, 3N] = [Wt_hi | Wt_lo | Wt_hi]: the weight image of the dgrad dX = dY . W written as an NT product */
int amdseg_split3_transpose(const float* W, void* out_bf16, int N, int K, amdseg_stream_t stream);
/* both weight images of n matrices in one launch (host arrays of device pointers, as amdseg_cast_transpose_batched): out[i] [N_i, 3K_i] =
 * amdseg_split3(order 1), out_t[i] [K_i, 3N_i] = amdseg_split3_transpose; N_i, K_i multiples of 64; out or out_t may be NULL */
int amdseg_split3_weights_batched(int n, const float* const* W, void* const* out, void* const* out_t, const int* N, const int* K,
                                  amdseg_stream_t stream);
/* fp32 attention ([hf] models/bert/modeling_bert.py:111-136) with saved log-sum-exp and hash dropout; qkv [M,3H] fp32, head_dim 64 */
int amdseg_pattn_fwd(const float* qkv, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale,
                     float p_drop, uint64_t seed, amdseg_stream_t stream);
int amdseg_pattn_bwd(const float* qkv, const float* mask_bias, const float* ctx, const float* dctx, const float* lse, float* delta,
                     float* dqkv, int B, int L, int heads, float scale, float p_drop, uint64_t seed, amdseg_stream_t stream);

/* ---- Longformer global [CLS] row, the O(heads * H^2) algebra around the O(L) passes above (csrc/lf_global.hip; [hf]
 * models/longformer/modeling_longformer.py:964-1058 with the key / value projections folded onto the query side):
 *   q:    qg [B,heads,64] = (Wq x0 + bq) * scale,  r [B,heads,H] = Wk_h^T qg_h          (x0 = row b*L of x, dtype x_dtype)
 *   out:  ctx[b*L, h*64 + e] = Wv_h y_h + bv_h * sp                                      (y, sp from amdseg_lf_wsum / lf_softmax_fwd)
 *   bwd_a: dout = dctx[b*L] (then zeroed), dyv = Wv_h^T dout_h, dsp = bv_h . dout_h
 *   bwd_rest: dqg = Wk_h dr_h * scale;  dWv += dout (x) y, dbv += dout * sp, dWk += qg (x) dr;  dWq += dqg (x) x0, dbq += dqg;
 *             dx[b*L] += Wq^T dqg      (weight gradients ACCUMULATE into the caller's flat gradient views) */
int amdseg_lf_global_q(const void* x, int x_dtype, const float* Wq, const float* bq, const float* Wk, float* qg, float* r, int B, int L, int H,
                       int heads, float scale, amdseg_stream_t stream);
int amdseg_lf_global_out(const float* Wv, const float* bv, const float* y, const float* sp, void* ctx, int ctx_dtype, int B, int L, int H,
                         int heads, amdseg_stream_t stream);
int amdseg_lf_global_bwd_a(void* dctx, int dtype, const float* Wv, const float* bv, float* dout, float* dyv, float* dsp, int B, int L, int H,
                           int heads, amdseg_stream_t stream);
/* the same without zeroing dctx[:, 0] (read-only): for a caller that runs backward phase 6 (AMDSEG_BF16, window > 0: the attention backward
 * treats the dctx rows of the first nglobal tokens as zero whatever they hold) and this kernel on another stream at the same time */
int amdseg_lf_global_bwd_a_ro(const void* dctx, int dtype, const float* Wv, const float* bv, float* dout, float* dyv, float* dsp, int B, int L,
                              int H, int heads, amdseg_stream_t stream);
int amdseg_lf_global_bwd_rest(const void* x, int x_dtype, void* dx, int dx_dtype, const float* Wq, const float* Wk, const float* qg,
                              const float* dout, const float* y, const float* sp, const float* dr, float* dqg, float* dWq, float* dbq,
                              float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads, float scale, amdseg_stream_t stream);
/* amdseg_lf_global_bwd_rest in two calls, for a caller whose critical path waits for dx only: _dx computes dqg [B, H] and WRITES
 * trow[b, :] = Wq^T dqg[b] [B, H] fp32 (the global row's contribution to dx[b, 0, :]; added by amdseg_lf_dx_apply) -- it needs neither dx nor x;
 * _w adds every weight / bias gradient of the three global projections (reads the dqg of _dx) and may run any time before the optimiser. */
int amdseg_lf_global_bwd_dx(const float* Wq, const float* Wk, const float* dr, float* dqg, float* trow, int B, int L, int H, int heads,
                            float scale, amdseg_stream_t stream);
int amdseg_lf_global_bwd_w(const void* x, int x_dtype, const float* qg, const float* dout, const float* y, const float* sp, const float* dr,
                           const float* dqg, float* dWq, float* dbq, float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads,
                           amdseg_stream_t stream);
/* amdseg_lf_dx_update (bf16 dx) in two calls: _prep packs vecA / vecB [B, heads, H] into the MFMA operand image vt_ws [B, H, 32] bf16 (does not
 * touch dx: may run early, on another stream); _apply is the read-modify-write pass  dx[b, j, :] += sum_h coefA[b, h, j] vecA[b, h, :] +
 * coefB[b, h, j] vecB[b, h, :]  and, with trow != NULL, dx[b, 0, :] += trow[b, :]. */
int amdseg_lf_dx_prep(const float* vecA, const float* vecB, void* vt_ws, int B, int L, int H, int heads, amdseg_stream_t stream);
int amdseg_lf_dx_apply(void* dx, int ldx, const float* coefA, const float* coefB, const void* vt_ws, const float* trow, int B, int L, int H,
                       int heads, amdseg_stream_t stream);

/* ---- fused loss heads of the training step (csrc/heads.hip): token cross-entropy over the classifier logits (loss_calculator.py:25-57,
 * utils.py:141-182 weighted CE, ignore_index -100; `nseg` equal row segments = anchor half | augmented half, one mean each), CSSL InfoNCE in
 * list form (cssl.py:82-116: anchors x (pk positive + negative) lists of row indices) and TSSP Linear(H, Ct) + CE (tssp.py:16-36).
 *   out8:  [0..1] CE per segment, [2] CSSL, [3] TSSP, [4] total = w_ts * (CE_0 + CE_1) + w_cl * CSSL + w_tssp2 * TSSP, [5..6] 1 / sum of CE
 *          weights per segment (read by backward);  ce_unit [M,C] and acc [32 * ceil(M / 256) + n_anchor + nt] are caller-owned scratch
 *          (per-wave CE partials and per-row loss terms, summed in a fixed order: the loss is bit-reproducible).
 *   idx:   ONE int64 device buffer holding every index list of the step (built on the host with the reference's `random` call order);
 *          feat_off -> seq row of feature f; anchor_off -> feature index of anchor i (-1: anchor i = feature i); lists_off ->
 *          [n_list][n_anchor] feature indices (the first pk lists are the positives); t_rows_off / t_labels_off -> TSSP rows / classes.
 * Backward: amdseg_heads_bwd_ce writes dlogits [M,C] = d total / d logits * gout[0]; the caller runs amdseg_rowdot_bwd on it (which WRITES
 * dx [M,H]); amdseg_heads_bwd_rows then ADDS the CSSL / TSSP row gradients into dx and ACCUMULATES dWt [Ct,H], dbt [Ct].  These are scatter
 * sums (a feature row sits in the lists of many anchors); they are taken as 64-bit fixed-point integers (2^-40 units) in `fix` -- caller-owned
 * scratch of >= 8 * ((n_feat + nt) * H + Ct * H + Ct) bytes, n_feat = number of CSSL feature rows (entries at feat_off); zeroed by the call --
 * and added to their destinations by one writer each, so the gradients do not depend on the order the atomics land in. */
int amdseg_heads_fwd(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                     float* ce_unit, float* out8, float* acc, const int64_t* idx, long feat_off, long anchor_off, long lists_off,
                     int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off, long t_labels_off,
                     int nt, int Ct, float w_ts, float w_cl, float w_tssp2, amdseg_stream_t stream);
int amdseg_heads_bwd_ce(const float* gout, int M, int C, int nseg, const float* ce_unit, const float* out8, float w_ts, float* dlogits,
                        amdseg_stream_t stream);
/* the same with the reference's focal loss (focal_loss_gamma != 0, modules/utils.py:141-168): FocalLoss overwrites its `reduction` with 'mean'
 * before the base class's forward runs, so its value is  mean_i (1 - p_i,t_i)^gamma  x  the scalar (weighted) mean CE, t_i = the label (0 on ignored
 * rows), the first mean over ALL rows of the segment -- reproduced as is.  ce_unit2: [M, 2C] scratch (CE unit | focal-factor unit);
 * out12: 12 floats ([8..9] mean focal factor, [10..11] mean CE per segment; [0..7] as amdseg_heads_fwd with [0..1] = their product). */
int amdseg_heads_fwd_focal(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                           float* ce_unit2, float* out12, float* acc, const int64_t* idx, long feat_off, long anchor_off, long lists_off,
                           int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off, long t_labels_off,
                           int nt, int Ct, float w_ts, float w_cl, float w_tssp2, float focal_gamma, amdseg_stream_t stream);
int amdseg_heads_bwd_ce_focal(const float* gout, int M, int C, int nseg, const float* ce_unit2, const float* out12, float w_ts, float focal_gamma,
                              float* dlogits, amdseg_stream_t stream);
int amdseg_heads_bwd_rows(const float* gout, const float* x, int M, int H, float* dx, const int64_t* idx, long feat_off, long anchor_off,
                          long lists_off, int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off,
                          long t_labels_off, int nt, int Ct, float* dWt, float* dbt, float w_cl, float w_tssp2, int n_feat, void* fix,
                          size_t fix_bytes, amdseg_stream_t stream);

/* NOTE on additive key masks (`mask_bias` of every attention entry point): 0 for visible keys, a MODERATE negative number for masked ones
 * (the host mirror uses -30000; the kernels fold mask and log-sum-exp into the fp32 MFMA accumulator start, so a magnitude like 1e30
 * would cost all fp32 digits of (mask - lse) in rows whose visible keys are all masked).  exp() of a masked score is an exact 0 next to
 * any real score either way, which is what the reference's finfo.min mask yields. */

/* ---- the explicit library context (ABI 13; csrc/prof.hip).  EVERYTHING libamdseg remembers between two calls lives in an amdseg_ctx the
 * caller creates: (1) the CU budget of the launch-geometry rules -- a process whose backward overlaps an RCCL all-reduce sets it to (CUs - RCCL
 * channels): the GEMM tile width is chosen by rounds of workgroups against this number (a 256-tile grid on 240 free CUs is two rounds);
 * spokennlp_amd/engine.py does so around backward when world > 1 --, (2) a test hook that routes GEMMs to the 128 x 128 kernels, (3) the launch
 * timer (start / stop events of every launch of the dominant kernel classes INSIDE real training steps: bench.py's `roofline`).
 * Which context a call uses: amdseg_bert_cfg.ctx for the composite layer calls; entry points without a cfg use the context the CALLING THREAD
 * bound with amdseg_ctx_bind() (NULL unbinds); with neither, the defaults apply: every CU, the tile rules' own choice, no timing.  One context
 * per engine / GPU; calls on one context are serialised by its owner (the 1-process-per-GPU model).  The RCCL communicator `amdseg_comm` below is the second explicit
 * context: the RCCL communicator.  Besides the two, the library holds only immutable per-process facts (CU count, kernel attributes, dlopen handle). */
typedef struct amdseg_ctx amdseg_ctx;
int amdseg_ctx_create(amdseg_ctx** out);
int amdseg_ctx_destroy(amdseg_ctx* ctx);
int amdseg_ctx_bind(amdseg_ctx* ctx);                                 /* the calling thread's context for cfg-less entry points; NULL unbinds */
int amdseg_ctx_set_cu_budget(amdseg_ctx* ctx, int cus);               /* 0 = all; returns the previous value */
int amdseg_ctx_cu_budget(const amdseg_ctx* ctx);
int amdseg_ctx_force_small_tile(amdseg_ctx* ctx, int v);              /* test hook; returns the previous value */
/* launch timer: enable(1) arms it (returns the previous state, < 0 on failure); read() syncs the device and returns, for one launch class, the
 * summed kernel spans (us), the summed algorithmic work (FLOPs or bytes) and the number of launches since the last reset */
int amdseg_ctx_prof_enable(amdseg_ctx* ctx, int on);
int amdseg_ctx_prof_reset(amdseg_ctx* ctx);
int amdseg_ctx_prof_read(amdseg_ctx* ctx, int cls, double* total_us, double* total_work, long long* launches);
#define AMDSEG_PROF_GEMM_NT 0      /* gemm_nt_dp_kernel: forward + dgrad projection GEMMs */
#define AMDSEG_PROF_GEMM_TN 1      /* gemm_tn_dp_kernel: grouped weight-gradient GEMM */
#define AMDSEG_PROF_ATTN_FWD 2
#define AMDSEG_PROF_ATTN_BWD_DQ 3
#define AMDSEG_PROF_ATTN_BWD_DKV 4
/* HBM-bound row / stream kernels (ABI 8): `work` of these classes is the launch's ALGORITHMIC BYTES, not FLOPs */
#define AMDSEG_PROF_ADD_LN_FWD 5   /* add_ln_fwd_kernel: read y, resid; write z, out                      = 4 * M * H * sizeof(act) */
#define AMDSEG_PROF_LN_BWD 6       /* ln_bwd_kernel: read dy, z; write dz (+ dbranch with dropout)        = (3 or 4) * M * H * sizeof(act) */
#define AMDSEG_PROF_ADAMW 7        /* adamw_kernel: p, g, m, v in; p, m, v out (+ bf16 shadow, + zeroed g) = 28 (+2) (+4) B per parameter */
#define AMDSEG_PROF_KEEPMASK 8     /* attn_keepmask_kernel: the mask bytes written */

/* ---- composite: one BertLayer forward / backward ([hf] models/bert/modeling_bert.py:374-416) -------------------- */
typedef struct amdseg_bert_cfg {
    int32_t B, L, H, heads, I;      /* batch, sequence, hidden, heads (H = heads*64), intermediate */
    float ln_eps, p_hidden, p_attn; /* dropout probabilities (0 in eval) */
    uint64_t seed;                  /* dropout seed of this step; per-layer/site streams are derived from it */
    int32_t accumulate_grads;       /* weight grads: 0 overwrite, 1 add into existing */
    int32_t dtype;                  /* AMDSEG_BF16 (train + inference), AMDSEG_F32 (inference, exact fp32 MFMA: the layer params then
                                       point at the fp32 master weights, activations are fp32) or AMDSEG_F32S (train + inference, fp32
                                       activations: params w* = weight images [N,3K] of amdseg_split3(order 1), w*_t = images [K,3N] of
                                       amdseg_split3_transpose; BERT attention only: window = 0, mixer = 0) */
    int32_t window, nglobal;        /* Longformer layers: band attention (see amdseg_attn_band_fwd); 0, 0 = BERT */
    int32_t nproj, mixer;           /* mixer 0: softmax attention over the q|k|v projection (nproj 0 or 3).  mixer 1: external
                                       token mixer (PoNet): the projection is nproj*H wide (acts.qkv, ws.dqkv, wqkv, wqkv_t, bqkv
                                       sized accordingly), forward phase 1 stops after it and the caller fills acts.ctx;
                                       the caller fills ws.dqkv from ws.dctx before backward phase 2; phase must be 1 or 2 */
    int32_t phase;                  /* 0 or 3 = whole layer.  1 / 2 = the part before / after the attention context:
                                       forward 1 = QKV projection + attention (writes acts.ctx), 2 = the rest;
                                       backward 1 = from dy down to ws.dctx, 2 = attention backward, dx_in, all weight
                                       gradients.  A Longformer caller writes the global token's ctx row between the
                                       forward phases (AMDSEG_BF16, window > 0, nglobal > 0: forward phase 1 does NOT store
                                       the ctx rows of the first nglobal tokens of a sequence, so the caller may write them
                                       from another stream while phase 1 runs; the fp32 dtypes store a band row there that
                                       the caller overwrites AFTER phase 1) and consumes + zeroes its dctx row between
                                       backward phases (AMDSEG_BF16, window > 0, backward phase 6: the attention backward takes the
                                       dctx rows of the first nglobal tokens as zero itself -- amdseg_lf_global_bwd_a_ro).
                                       Backward only: 6 = phase 2 without the grouped weight-gradient GEMM, 4 = that GEMM
                                       alone (e.g. on a second stream, under the next layer's backward; the caller orders
                                       the streams and must not reuse ws before it has run). */
    int32_t act;                    /* FFN activation: 0 = exact (erf) GELU "gelu", 1 = "gelu_new" (BigBird) */
    const int32_t* kend;            /* optional device [B] (full attention, bf16 path): every key at a position >= kend[b] carries a
                                       mask <= -5000 (trailing padding).  The attention kernels then do not visit the 64-key chunks past
                                       it -- they contribute exact zeros, results are bit-identical -- and write dK = dV = 0 there.
                                       kend[b] == 0 (no unmasked key) and NULL keep every chunk. */
    const int32_t* seq_order;       /* optional device [B] (with kend): a permutation of the sequences, longest visible length first; the
                                       attention launches then dispatch their workgroups in that order so that the short sequences fill
                                       the tail (results do not depend on it) */
    const int32_t* pad_guard;       /* backward, bf16 path, softmax-attention layers (mixer 0); optional, needs kend.  Device int written by
                                       amdseg_pad_rows_guard() on the gradient the backward starts from: 0 = every row at a position >=
                                       kend[b] of that gradient is an exact zero.  Such rows stay exact zeros through every layer (a
                                       masked key gets p = 0, so dK = dV = 0; its own query row has dctx = 0, so dQ = 0; LayerNorm, GELU
                                       and dropout backward map 0 to 0), so with *pad_guard == 0 the input-gradient GEMMs skip the K loop
                                       of 256-row tiles made of them (L % 256 == 0) and the weight-gradient GEMM walks only pad_runs:
                                       bit-identical results, less work.  *pad_guard != 0 or NULL: dense. */
    const int32_t* pad_runs;        /* with pad_guard: device [B][2] = {first, end} runs of 64-token tiles t (rows 64t .. 64t+63) that hold a
                                       position < kend -- per sequence b with kend[b] > 0: {b*L/64, b*L/64 + ceil(kend[b]/64)}; L % 64 == 0 */
    const int32_t* pad_counts;      /* with pad_guard: device int[2] = {tiles in those runs, number of runs} */
    amdseg_ctx* ctx;                /* (ABI 13) the caller's context: CU budget of the tile rules, launch timer.  NULL: the context bound to the calling
                                       thread, else the defaults */
} amdseg_bert_cfg;

typedef struct amdseg_bert_layer_params {   /* bf16 compute shadows (+ transposes for dgrad), fp32 vectors */
    const void *wqkv, *wo, *w1, *w2;        /* [3H,H] [H,H] [I,H] [H,I] */
    const void *wqkv_t, *wo_t, *w1_t, *w2_t;/* [H,3H] [H,H] [H,I] [I,H] (backward only) */
    const float *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} amdseg_bert_layer_params;

typedef struct amdseg_bert_layer_grads {    /* fp32, views into the flat gradient buffer */
    float *wqkv, *wo, *w1, *w2, *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} amdseg_bert_layer_grads;

typedef struct amdseg_bert_layer_acts {     /* caller-owned activations; all but x_in are written by forward */
    const void* x_in;                       /* [M,H] layer input */
    void *qkv, *ctx, *z1, *x1, *u, *h, *z2, *x_out;   /* [M,3H] [M,H] [M,H] [M,H] [M,I] [M,I] [M,H] [M,H]; u may be NULL in
                                                         inference (only backward reads it); the bf16 forward then also leaves z1 / z2 holding the
                                                         dense outputs instead of the pre-LayerNorm sums (backward's other input).  In bf16 training
                                                         u holds what amdseg_bert_layer_bwd of the SAME cfg expects: the pre-activation, or gelu' of
                                                         it as one byte per element (AMDSEG_EPI_DERIV_U8) where the shape allows */
    float *lse, *mean1, *rstd1, *mean2, *rstd2;       /* [B*heads*L] [M] [M] [M] [M] */
    /* AMDSEG_F32S only: bf16 split images [hi | hi | lo] of x_in, ctx, x1 and gelu(u): [M,3H] [M,3H] [M,3H] [M,3I] (written by
     * forward, read by the GEMMs of forward and by the weight gradients of backward; `h` is unused in that mode) */
    void *xs, *ctx_s, *x1_s, *h_s;
    /* optional (bf16 path, full attention, p_attn > 0): amdseg_attn_keepmask_bytes(B, L, heads) bytes per layer.  Forward fills it with this
     * step's dropout keep masks (amdseg_attn_keepmask) and all three attention kernels read them instead of hashing; backward must get the
     * buffer its forward wrote.  NULL: the stateless hash path. */
    void* keep;
    /* optional, AMDSEG_F32S: bf16 split image [M, 9H] = [hi | hi | lo] of the fp32 q|k|v projection (amdseg_split3 layout).  When given (and
     * ws.dctx_s for backward) the attention runs on the bf16 matrix cores as three split products (csrc/attention_split.hip) instead of the
     * fp32-MFMA kernels of csrc/parity.hip; forward writes it, backward reads it (acts.qkv itself is then only forward scratch). */
    void* qkv_s;
    /* optional (training, p_hidden > 0), ABI 8: M * H / 8 bytes each.  The forward's dropout + residual + LayerNorm kernels store the keep
     * decisions of the two hidden-state dropouts ([hf] modeling_bert.py:292 BertSelfOutput.dropout, :350 BertOutput.dropout) as one byte per
     * 8 consecutive elements (bit e = element e kept; the same stateless hash as before decides them), and the LayerNorm backward reads
     * them instead of re-evaluating the hash per element.  NULL: the hash is evaluated in both directions (same decisions). */
    void *drop1, *drop2;
    /* optional, ABI 14 (bf16 path, p_attn > 0): keep_next = the `keep` buffer of layer li + 1.  Phase 2 of this layer's forward then writes that
     * layer's keep masks (seed of layer li + 1, this cfg's B / L / heads / kend / window / nglobal) in the launch of its second LayerNorm
     * (amdseg_add_ln_fwd_keepmask); the acts of layer li + 1 must carry keep_ready != 0, which tells ITS forward that `keep` already holds this
     * step's masks and must not be generated again.  Same bits as without the pair.  NULL / 0: every layer generates its own masks.
     * The library pairs only where the shape makes it pay (generator workgroups at most half the row workgroups: csrc/keepmask.h
     * km_pairs_with_rows); both layers evaluate that rule on the same cfg, so a pair set up here never goes out of step. */
    void* keep_next;
    int keep_ready;
} amdseg_bert_layer_acts;

typedef struct amdseg_bert_layer_ws {       /* backward scratch, reusable across layers */
    void *dz2, *dbr2, *du, *dx1, *dz1, *dbr1, *dctx, *dqkv;  /* [M,H] [M,H] [M,I] [M,H] [M,H] [M,H] [M,H] [M,3H] */
    float *delta, *partials;                /* [B*heads*L]; partials: 6*ceil(M/16)*H + max(ceil(M/128), ceil(H/128))*(I + nproj*H) floats (one
                                               region per deferred reduction: LN2, b1, LN1, bqkv) */
    /* AMDSEG_F32S only: split images of the four gradient operands d(FFN out), du, d(attention out), dqkv: [M,3H] [M,3I] [M,3H] [M,9H] */
    void *d_out_s, *du_s, *d_ao_s, *dqkv_s;
    void* dctx_s;                           /* optional, AMDSEG_F32S with acts.qkv_s: split image [M, 3H] of d(ctx) (scratch) */
} amdseg_bert_layer_ws;

int amdseg_bert_layer_fwd(const amdseg_bert_cfg* cfg, const amdseg_bert_layer_params* p, const amdseg_bert_layer_acts* a,
                          const float* mask_bias, int layer_idx, amdseg_stream_t stream);
/* dy: gradient w.r.t. x_out [M,H]; dx_in: gradient w.r.t. x_in [M,H] (output) */
int amdseg_bert_layer_bwd(const amdseg_bert_cfg* cfg, const amdseg_bert_layer_params* p, const amdseg_bert_layer_grads* g,
                          const amdseg_bert_layer_acts* a, const amdseg_bert_layer_ws* ws, const float* mask_bias,
                          const void* dy, void* dx_in, int layer_idx, amdseg_stream_t stream);

/* ---- gradient exchange of pure data parallelism (csrc/comm.hip) -- ABI 11 ----------------------------------------------------
 * replaces torch DDP's bucketed all-reduce, which the reference gets from `python -m torch.distributed.launch --nproc_per_node N`
 * (emnlp2023-topic_segmentation/run_finetune.sh:61) + transformers.Trainer's model wrapping.  One process per GPU; `amdseg_comm` is the
 * explicit context SURVEY 8(b) asks for (the RCCL communicator of this rank on the CURRENT device, a side stream, two events) -- no
 * global state.  The Python host of this repo does the same exchange through torch.distributed (spokennlp_amd/dp.py: the process group,
 * launcher contract and Trainer integration live there; backend "nccl" IS RCCL); these entry points give a host without PyTorch the
 * same thing.  RCCL is bound at run time (dlopen; AMDSEG_RCCL_LIB overrides the search), never at link time.
 *   unique_id: rank 0 obtains the 128-byte rendezvous id (ncclGetUniqueId) and ships it to the other ranks by its own means;
 *   init:      collective over the `world` ranks (ncclCommInitRank); rank r's communicator lives on the device current at the call;
 *   bucket:    `buf[0..n)` (fp32 or bf16, in place) becomes the element-wise SUM over the ranks.  Issued relative to `compute_stream`:
 *              the reduction starts when everything queued on that stream so far has run (the slice is final) and runs on the context's
 *              side stream, so the kernels queued on compute_stream afterwards overlap it (issue buckets in the order backward finishes
 *              them: encoder layers last to first, then the embedding tables, then the heads; the mean = 1/world goes into
 *              amdseg_adamw_step's grad_scale);
 *   wait:      compute_stream waits for every bucket issued so far (call it once, in front of the gradient norm / AdamW);
 *   info:      rank / world / elements issued since the last wait;   destroy: drains the side stream, frees the communicator.
 * Errors: AMDSEG_ERR_COMM_LIB (no librccl), AMDSEG_ERR_COMM_BASE + ncclResult_t, or a hipError_t; amdseg_error_string knows them all. */
#define AMDSEG_COMM_ID_BYTES 128
typedef struct amdseg_comm amdseg_comm;
int amdseg_allreduce_unique_id(void* id128);
int amdseg_allreduce_init(amdseg_comm** out, const void* id128, int rank, int world);
int amdseg_allreduce_bucket(amdseg_comm* comm, void* buf, size_t n, int dtype, amdseg_stream_t compute_stream);
int amdseg_allreduce_wait(amdseg_comm* comm, amdseg_stream_t compute_stream);
int amdseg_allreduce_info(const amdseg_comm* comm, int* rank, int* world, size_t* pending_elements);
int amdseg_allreduce_destroy(amdseg_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* AMDSEG_H */
