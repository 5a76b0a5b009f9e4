// Multi-head self-attention in "parity" precision on the bf16 matrix cores: every contraction of the flash kernels of attention.hip
// evaluated as a split-bf16 product (x = hi + lo, A.B = Ahi.Bhi + Ahi.Blo + Alo.Bhi, the dropped lo.lo term is 2^-16 relative), fp32
// softmax / accumulators / outputs.  Same reference lines as attention.hip ([hf] models/bert/modeling_bert.py:111-136 forward + autograd
// backward) at the precision the reference runs in (run_finetune.sh:61-96: fp32, no --fp16 / --bf16).
//
// Why: the fp32 attention of parity.hip (pattn2_*: flash-style on v_mfma_f32_16x16x4_f32, 157 TFLOP/s) was 37 % of the "parity" training
// step (0.36 / 0.52 / 0.77 ms per layer forward / dQ / dK+dV at bert-base, profiles/r02_parity_v1_kernel_stats.md).  Three
// v_mfma_f32_16x16x32_bf16 products cost 3/16 of that matrix time; the softmax side is the bf16 kernel's plus the hi / lo split of the
// probabilities (and of dS in backward), 5 more vector instructions per element pair.
//
// Layout: the operands arrive as the split images amdseg_split3 writes -- qs [B*L][ldq] bf16 with the hi parts of q|k|v at columns
// [0, 3H) and the lo parts at [lo_q, lo_q + 3H); dos [B*L][ldo] likewise for dO -- so K / V (Q / dO) tiles are staged hi and lo by the same
// LDS-DMA as in attention.hip (4 tiles of [64][64] bf16 per stage instead of 2, two stages = 64 KiB); ctx, dqkv, lse, delta are fp32.
// Structure, orientation (S^T = K Q^T ...), chunk skipping (kend / seq_order / qguard), band visibility and the keep-mask layouts are
// those of attention.hip; dropout is read from the layer's keep masks only (no hash path here; band: the cells its kernels visit).
#include "attention_common.h"

struct SAttnArgs {
    const bf16_t* qs; int ldq, lo_q;
    const float* mask_bias; float* ctx; float* lse;
    const bf16_t* dos; int ldo, lo_o;
    float* delta; float* dqkv;
    bf16_t* ctx_img;                        // optional (forward): ctx ALSO as the split image [B*L][3H] = [hi | hi | lo] the output projection reads
    bf16_t* dqs; int ldd;                   // optional: write d(q|k|v) as the split image [B*L][ldd] = [hi | hi | lo] (blocks 3H wide: the A operand
                                            // of the next split GEMMs) instead of fp32 dqkv
    int B, L, heads;
    float scale, inv_keep; uint32_t thresh16;
    int window, nglobal;
    const int* kend; const int* seq_order; const int* qguard;
    const uint64_t* keepA; const uint64_t* keepB;
};

#define SA_STG 32768                                   // bytes per LDS stage: 4 tiles
#define SA_LDS (2 * SA_STG + 1024)

// 8 fp32 values (two accumulator fragments) -> bf16 hi and bf16 lo = bf16(x - hi)
__device__ __forceinline__ void sa_split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    union { uint32_t u[4]; bf16x8 v; } H, Lo;
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        H.u[i] = pack2bf(x[2 * i], x[2 * i + 1]);
        Lo.u[i] = pack2bf(x[2 * i] - __uint_as_float(H.u[i] << 16), x[2 * i + 1] - __uint_as_float(H.u[i] & 0xffff0000u));
    }
    hi = H.v; lo = Lo.v;
}
// four consecutive gradient values: fp32 to `f`, or as hi | hi | lo to the image position `img` (blocks W apart)
__device__ __forceinline__ void sa_store4(float* f, bf16_t* img, int W, float x0, float x1, float x2, float x3) {
    if (!img) { *reinterpret_cast<float4*>(f) = make_float4(x0, x1, x2, x3); return; }
    uint2 hi, lo;
    hi.x = pack2bf(x0, x1); hi.y = pack2bf(x2, x3);
    lo.x = pack2bf(x0 - __uint_as_float(hi.x << 16), x1 - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack2bf(x2 - __uint_as_float(hi.y << 16), x3 - __uint_as_float(hi.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(img) = hi; *reinterpret_cast<uint2*>(img + W) = hi; *reinterpret_cast<uint2*>(img + 2 * W) = lo;
}
// acc += Ahi.Bhi + Ahi.Blo + Alo.Bhi   (A = the LDS-side fragment, B = the register-side one)
#define SA_MFMA3(acc, ah, al, bh, bl) do { \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0); } while (0)

__device__ __forceinline__ int sa_visible_chunks(const SAttnArgs& a, int b, int nch) {
    if (!a.kend) return nch;
    const int ke = a.kend[b];
    return ke > 0 ? min(nch, (ke + CH - 1) / CH) : nch;
}

// ------------------------------------------------------------------------------------------------ forward
template <bool BAND, bool KM>
__global__ __launch_bounds__(256, 2) void sattn_fwd_kernel(SAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 4;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, i16 = l & 15;
    int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    attn_xcd_remap(qb, h, b, a.heads);
    if (!BAND && a.seq_order) b = a.seq_order[b];
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const int q = qb * (NW * 16) + w * 16 + i16;
    const uint64_t prow = ((uint64_t)(b * a.heads + h)) * a.L + q;
#define sK(i) (smem + (i) * SA_STG)
#define sV(i) (smem + (i) * SA_STG + 8192)
#define sKl(i) (smem + (i) * SA_STG + 16384)
#define sVl(i) (smem + (i) * SA_STG + 24576)
#define sM(i) (smem + 2 * SA_STG + (i) * 256)
    bf16x8 fqh[2], fql[2];
    {
        const bf16_t* qp = a.qs + (tok0 + q) * a.ldq + h * HD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                    // the softmax scale folded into both halves of q (exact for a power of two)
            fqh[kk] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + kk * 32 + g * 8), a.scale);
            fql[kk] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + a.lo_q + kk * 32 + g * 8), a.scale);
        }
    }
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;
    const bf16_t* kbase = a.qs + tok0 * a.ldq + H + h * HD;
    const bf16_t* vbase = a.qs + tok0 * a.ldq + 2 * H + h * HD;
    int c0 = 0, c1 = a.L / CH - 1, extra = 0;
    bool pad_block = false;
    if (BAND) {
        const int q_lo = qb * (NW * 16), q_hi = q_lo + NW * 16 - 1;
        c0 = q_lo > a.window ? (q_lo - a.window) / CH : 0;
        c1 = min(c1, (q_hi + a.window) / CH);
        extra = (a.nglobal > 0 && c0 > 0) ? 1 : 0;
        if (a.kend && a.kend[b] > 0) {
            const int ke = a.kend[b];
            pad_block = q_lo >= ke;
            c1 = min(c1, (ke - 1) / CH);
        }
    }
    float* op = a.ctx + (tok0 + q) * H + h * HD;
    if (BAND && pad_block) {                                // every row of the block is a padded query: zero rows, LSE = +inf
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<float4*>(op + d * 16 + g * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.lse && g == 0) a.lse[prow] = INFINITY;
        return;
    }
    int nch = c1 - c0 + 1 + extra;
    if (!BAND) nch = sa_visible_chunks(a, b, nch);
#define CHUNK_OF(t) ((BAND && extra && (t) == 0) ? 0 : c0 + (t) - extra)
#define SA_STAGE_KV(t, i) do { const size_t ro_ = (size_t)CHUNK_OF(t) * CH * a.ldq; \
        at_stage<NW>(kbase + ro_, a.ldq, sK(i), w, l); at_stage<NW>(vbase + ro_, a.ldq, sV(i), w, l); \
        at_stage<NW>(kbase + a.lo_q + ro_, a.ldq, sKl(i), w, l); at_stage<NW>(vbase + a.lo_q + ro_, a.ldq, sVl(i), w, l); \
        if (w == 0) at_stage_f32x64(a.mask_bias + tok0 + CHUNK_OF(t) * CH, sM(i), l); } while (0)
    KeepWords kw;
    const size_t kcell0 = (((size_t)(b * a.heads + h)) * (a.L / 16) + (size_t)qb * NW + __builtin_amdgcn_readfirstlane(w)) * (a.L / CH);
    if (KM && nch > 0) km_load(kw, a.keepA, kcell0 + CHUNK_OF(0));
    if (nch > 0) SA_STAGE_KV(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the Q fragments (see attention.hip)
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = ch & 1;
        if (ch + 1 < nch) SA_STAGE_KV(ch + 1, cur ^ 1);
        float4 mbc[4];
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) mbc[fc] = *reinterpret_cast<const float4*>(sM(cur) + (fc * 16 + g * 4) * 4);
        const char* tK = sK(cur); const char* tV = sV(cur); const char* tKl = sKl(cur); const char* tVl = sVl(cur);
        const int key0 = CHUNK_OF(ch) * CH;
        f32x4 s[4];
        {
            bf16x8 fkh[4][2], fkl[4][2];
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) { fkh[fc][kk] = at_frag(tK, fc * 16 + i16, kk * 4 + g); fkl[fc][kk] = at_frag(tKl, fc * 16 + i16, kk * 4 + g); }
#pragma unroll
            for (int fc = 0; fc < 4; ++fc) s[fc] = (f32x4){mbc[fc].x, mbc[fc].y, mbc[fc].z, mbc[fc].w};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {                // consecutive MFMAs on different accumulators
#pragma unroll
                for (int fc = 0; fc < 4; ++fc) s[fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkh[fc][kk], fqh[kk], s[fc], 0, 0, 0);
#pragma unroll
                for (int fc = 0; fc < 4; ++fc) s[fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkh[fc][kk], fql[kk], s[fc], 0, 0, 0);
#pragma unroll
                for (int fc = 0; fc < 4; ++fc) s[fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkl[fc][kk], fqh[kk], s[fc], 0, 0, 0);
            }
        }
        if (BAND) {
            const int wq_lo = qb * (NW * 16) + w * 16;
            if (!(key0 >= wq_lo + 15 - a.window && key0 + CH - 1 <= wq_lo + a.window)) {
#pragma unroll
                for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (band_masked(q, key0 + fc * 16 + g * 4 + r, a.window, a.nglobal)) s[fc][r] = -INFINITY;
            }
        }
        float cmax = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
        cmax = fmaxf(fmaxf(cmax, s[0][3]), s[1][0]);
        cmax = fmaxf(fmaxf(cmax, s[1][1]), s[1][2]);
        cmax = fmaxf(fmaxf(cmax, s[1][3]), s[2][0]);
        cmax = fmaxf(fmaxf(cmax, s[2][1]), s[2][2]);
        cmax = fmaxf(fmaxf(cmax, s[2][3]), s[3][0]);
        cmax = fmaxf(fmaxf(cmax, s[3][1]), s[3][2]);
        cmax = fmaxf(cmax, s[3][3]) * LOG2E;
        if (__any(cmax > m_run)) {
            cmax = xor_reduce_max_g(cmax);
            const float m_new = fmaxf(m_run, cmax);
            const float alpha = (BAND && m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_part *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
        }
        const float m_use = (BAND && m_run == -INFINITY) ? 0.f : m_run;
        {
            const f32x2 sc2v = {LOG2E, LOG2E}, negm = {-m_use, -m_use};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const f32x2 t = (f32x2){s[fc][rp * 2], s[fc][rp * 2 + 1]} * sc2v + negm;
                    const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    ps2 += e;
                    s[fc][rp * 2] = e.x; s[fc][rp * 2 + 1] = e.y;
                }
            l_part += ps2.x + ps2.y;
        }
        if (KM) {
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[fc][r] = km_sel(s[fc][r], kw.m[fc * 4 + r]);
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q], P = Ph + Pl, V = Vh + Vl
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            bf16x8 fph, fpl;
            sa_split8(s[2 * kp], s[2 * kp + 1], fph, fpl);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 fvh = at_frag_tr(tV, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
                const bf16x8 fvl = at_frag_tr(tVl, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
                SA_MFMA3(o[d], fvh, fvl, fph, fpl);
            }
        }
        if (KM && ch + 1 < nch) {
            asm volatile("" ::: "memory");
            km_load(kw, a.keepA, kcell0 + CHUNK_OF(ch + 1));
        }
    }
    const float lsum = xor_reduce_sum_g(l_part);
    float inv = (a.thresh16 ? a.inv_keep : 1.0f) / lsum;
    float lse_q = (m_run + __builtin_amdgcn_logf(lsum)) * LN2;
    const bool padq = BAND && a.mask_bias[tok0 + q] < 0.f;
    if (padq) { inv = 0.f; lse_q = INFINITY; }
#pragma unroll
    for (int d = 0; d < 4; ++d)
        *reinterpret_cast<float4*>(op + d * 16 + g * 4) = padq ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                               : make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
    if (a.ctx_img) {
        const int Hh = a.heads * HD;
        bf16_t* ip = a.ctx_img + (size_t)(tok0 + q) * 3 * Hh + h * HD;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            sa_store4(nullptr, ip + d * 16 + g * 4, Hh, padq ? 0.f : o[d][0] * inv, padq ? 0.f : o[d][1] * inv, padq ? 0.f : o[d][2] * inv,
                      padq ? 0.f : o[d][3] * inv);
    }
    if (a.lse && g == 0) a.lse[prow] = lse_q;
#undef CHUNK_OF
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <bool BAND, bool KM>
__global__ __launch_bounds__(256, 2) void sattn_bwd_dq_kernel(SAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 4;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, i16 = l & 15;
    int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    attn_xcd_remap(qb, h, b, a.heads);
    if (!BAND && a.seq_order) b = a.seq_order[b];
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const int q = qb * (NW * 16) + w * 16 + i16;
    const uint64_t prow = ((uint64_t)(b * a.heads + h)) * a.L + q;
    float* dqp = a.dqkv ? a.dqkv + (tok0 + q) * (size_t)(3 * H) + h * HD : nullptr;
    bf16_t* dqi = a.dqs ? a.dqs + (tok0 + q) * (size_t)a.ldd + h * HD : nullptr;
    if (!BAND && a.qguard && a.kend) {                      // a query block of trailing padding whose dO rows are exact zeros: dQ = 0, delta = 0
        const int ke = a.kend[b];
        if (ke > 0 && qb * (NW * 16) >= ke && *a.qguard == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d) sa_store4(dqp + d * 16 + g * 4, dqi ? dqi + d * 16 + g * 4 : nullptr, 3 * H, 0.f, 0.f, 0.f, 0.f);
            if (g == 0) a.delta[prow] = 0.f;
            return;
        }
    }
    bf16x8 fqh[2], fql[2], fdh[2], fdl[2];
    float delta_q;
    {
        const bf16_t* qp = a.qs + (tok0 + q) * a.ldq + h * HD;
        const bf16_t* dp = a.dos + (tok0 + q) * a.ldo + h * HD;
        const float* cp = a.ctx + (tok0 + q) * H + h * HD;
        float acc = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            fqh[kk] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + kk * 32 + g * 8), a.scale);
            fql[kk] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + a.lo_q + kk * 32 + g * 8), a.scale);
            fdh[kk] = *reinterpret_cast<const bf16x8*>(dp + kk * 32 + g * 8);
            fdl[kk] = *reinterpret_cast<const bf16x8*>(dp + a.lo_o + kk * 32 + g * 8);
            float ov[8];
            ld8<float>(cp + kk * 32 + g * 8, ov);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc += ov[e] * (__uint_as_float((uint32_t)(uint16_t)fdh[kk][e] << 16) + __uint_as_float((uint32_t)(uint16_t)fdl[kk][e] << 16));
        }
        delta_q = xor_reduce_sum_g(acc);
        if (g == 0) a.delta[prow] = delta_q;
    }
    const float nlse = -a.lse[prow];
    const float ikeep = a.thresh16 ? a.inv_keep : 1.0f;
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* kbase = a.qs + tok0 * a.ldq + H + h * HD;
    const bf16_t* vbase = a.qs + tok0 * a.ldq + 2 * H + h * HD;
    int c0 = 0, c1 = a.L / CH - 1, extra = 0;
    bool pad_block = false;
    if (BAND) {
        const int q_lo = qb * (NW * 16), q_hi = q_lo + NW * 16 - 1;
        c0 = q_lo > a.window ? (q_lo - a.window) / CH : 0;
        c1 = min(c1, (q_hi + a.window) / CH);
        extra = (a.nglobal > 0 && c0 > 0) ? 1 : 0;
        if (a.kend && a.kend[b] > 0) {
            const int ke = a.kend[b];
            pad_block = q_lo >= ke;
            c1 = min(c1, (ke - 1) / CH);
        }
    }
    if (BAND && pad_block) {
#pragma unroll
        for (int d = 0; d < 4; ++d) sa_store4(dqp + d * 16 + g * 4, dqi ? dqi + d * 16 + g * 4 : nullptr, 3 * H, 0.f, 0.f, 0.f, 0.f);
        if (g == 0) a.delta[prow] = 0.f;
        return;
    }
    int nch = c1 - c0 + 1 + extra;
    if (!BAND) nch = sa_visible_chunks(a, b, nch);
#define CHUNK_OF(t) ((BAND && extra && (t) == 0) ? 0 : c0 + (t) - extra)
    KeepWords kw;
    const size_t kcell0 = (((size_t)(b * a.heads + h)) * (a.L / 16) + (size_t)qb * NW + __builtin_amdgcn_readfirstlane(w)) * (a.L / CH);
    if (KM && nch > 0) km_load(kw, a.keepA, kcell0 + CHUNK_OF(0));
    if (nch > 0) SA_STAGE_KV(0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = ch & 1;
        if (ch + 1 < nch) SA_STAGE_KV(ch + 1, cur ^ 1);
        float4 mbc[4];
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) mbc[fc] = *reinterpret_cast<const float4*>(sM(cur) + (fc * 16 + g * 4) * 4);
        const char* tK = sK(cur); const char* tV = sV(cur); const char* tKl = sKl(cur); const char* tVl = sVl(cur);
        const int key0 = CHUNK_OF(ch) * CH;
        bool edge = false;
        if (BAND) {
            const int wq_lo = qb * (NW * 16) + w * 16;
            edge = !(key0 >= wq_lo + 15 - a.window && key0 + CH - 1 <= wq_lo + a.window);
        }
        f32x4 ds[4], sacc4[4], pacc4[4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 fkh[2][2], fkl[2][2], fvh[2][2], fvl[2][2];
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int r = (hf * 2 + f2) * 16 + i16, c = kk * 4 + g;
                    fkh[f2][kk] = at_frag(tK, r, c); fkl[f2][kk] = at_frag(tKl, r, c);
                    fvh[f2][kk] = at_frag(tV, r, c); fvl[f2][kk] = at_frag(tVl, r, c);
                }
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2) {
                const float4 m4 = mbc[hf * 2 + f2];
                sacc4[hf * 2 + f2] = (f32x4){m4.x + nlse, m4.y + nlse, m4.z + nlse, m4.w + nlse};
                pacc4[hf * 2 + f2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int f2 = 0; f2 < 2; ++f2) {
                        const int fi = hf * 2 + f2;
                        sacc4[fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pr == 2 ? fkl[f2][kk] : fkh[f2][kk], pr == 1 ? fql[kk] : fqh[kk], sacc4[fi], 0, 0, 0);
                        pacc4[fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pr == 2 ? fvl[f2][kk] : fvh[f2][kk], pr == 1 ? fdl[kk] : fdh[kk], pacc4[fi], 0, 0, 0);
                    }
        }
        if (KM) {
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int r = 0; r < 4; ++r) pacc4[fc][r] = km_sel(pacc4[fc][r], kw.m[fc * 4 + r]);
        }
        const f32x2 sc2v = {LOG2E, LOG2E}, ikv = {ikeep, ikeep}, ndl = {-delta_q, -delta_q};
#pragma unroll
        for (int fc = 0; fc < 4; ++fc)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const f32x2 t = (f32x2){sacc4[fc][rp * 2], sacc4[fc][rp * 2 + 1]} * sc2v;
                f32x2 pe = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                if (BAND && edge) {
                    if (band_masked(q, key0 + fc * 16 + g * 4 + rp * 2, a.window, a.nglobal)) pe.x = 0.f;
                    if (band_masked(q, key0 + fc * 16 + g * 4 + rp * 2 + 1, a.window, a.nglobal)) pe.y = 0.f;
                }
                const f32x2 d2 = pe * ((f32x2){pacc4[fc][rp * 2], pacc4[fc][rp * 2 + 1]} * ikv + ndl);
                ds[fc][rp * 2] = d2.x; ds[fc][rp * 2 + 1] = d2.y;
            }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            bf16x8 fsh, fsl;
            sa_split8(ds[2 * kp], ds[2 * kp + 1], fsh, fsl);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 fth = at_frag_tr(tK, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
                const bf16x8 ftl = at_frag_tr(tKl, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
                SA_MFMA3(dq[d], fth, ftl, fsh, fsl);
            }
        }
        if (KM && ch + 1 < nch) {
            asm volatile("" ::: "memory");
            km_load(kw, a.keepA, kcell0 + CHUNK_OF(ch + 1));
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
        sa_store4(dqp + d * 16 + g * 4, dqi ? dqi + d * 16 + g * 4 : nullptr, 3 * H, dq[d][0] * a.scale, dq[d][1] * a.scale, dq[d][2] * a.scale, dq[d][3] * a.scale);
#undef CHUNK_OF
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool BAND, bool KM>
__global__ __launch_bounds__(256, 2) void sattn_bwd_dkv_kernel(SAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 4;
#define sQ(i) (smem + (i) * SA_STG)
#define sO(i) (smem + (i) * SA_STG + 8192)
#define sQl(i) (smem + (i) * SA_STG + 16384)
#define sOl(i) (smem + (i) * SA_STG + 24576)
#define sL(i) (smem + 2 * SA_STG + (i) * 512)
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, i16 = l & 15;
    int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (!BAND) { attn_xcd_remap(kb, h, b, a.heads); if (a.seq_order) b = a.seq_order[b]; }
    else {
        int bh;
        attn_1d_order(blockIdx.x, a.L / (NW * 16), a.heads * a.B, 1, kb, bh);
        h = bh % a.heads; b = bh / a.heads;
    }
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const int key = kb * (NW * 16) + w * 16 + i16;
    const uint64_t bh64 = (uint64_t)(b * a.heads + h);
    float* dkp = a.dqkv ? a.dqkv + (tok0 + key) * (size_t)(3 * H) + H + h * HD : nullptr;
    float* dvp = a.dqkv ? a.dqkv + (tok0 + key) * (size_t)(3 * H) + 2 * H + h * HD : nullptr;
    bf16_t* dki = a.dqs ? a.dqs + (tok0 + key) * (size_t)a.ldd + H + h * HD : nullptr;
    bf16_t* dvi = a.dqs ? a.dqs + (tok0 + key) * (size_t)a.ldd + 2 * H + h * HD : nullptr;
    if (!BAND && a.kend) {
        const int ke = a.kend[b];
        if (ke > 0 && kb * (NW * 16) >= ke) {               // a key block wholly in the trailing padding: dK = dV = 0 exactly
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                sa_store4(dkp + d * 16 + g * 4, dki ? dki + d * 16 + g * 4 : nullptr, 3 * H, 0.f, 0.f, 0.f, 0.f);
                sa_store4(dvp + d * 16 + g * 4, dvi ? dvi + d * 16 + g * 4 : nullptr, 3 * H, 0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
    }
    bf16x8 fkh[2], fkl[2], fvh[2], fvl[2];
    {
        const bf16_t* kp = a.qs + (tok0 + key) * a.ldq + H + h * HD;
        const bf16_t* vp = a.qs + (tok0 + key) * a.ldq + 2 * H + h * HD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            fkh[kk] = frag_scale(*reinterpret_cast<const bf16x8*>(kp + kk * 32 + g * 8), a.scale);
            fkl[kk] = frag_scale(*reinterpret_cast<const bf16x8*>(kp + a.lo_q + kk * 32 + g * 8), a.scale);
            fvh[kk] = *reinterpret_cast<const bf16x8*>(vp + kk * 32 + g * 8);
            fvl[kk] = *reinterpret_cast<const bf16x8*>(vp + a.lo_q + kk * 32 + g * 8);
        }
    }
    const float mbs = a.mask_bias[tok0 + key];
    const float ikeep = a.thresh16 ? a.inv_keep : 1.0f;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const bf16_t* qbase = a.qs + tok0 * a.ldq + h * HD;
    const bf16_t* obase = a.dos + tok0 * a.ldo + h * HD;
    int c0 = 0, c1 = a.L / CH - 1;
    if (BAND && !(a.nglobal > 0 && kb * (NW * 16) < a.nglobal)) {
        const int k_lo = kb * (NW * 16), k_hi = k_lo + NW * 16 - 1;
        c0 = k_lo > a.window ? (k_lo - a.window) / CH : 0;
        c1 = min(c1, (k_hi + a.window) / CH);
    }
    if (BAND && a.kend && a.kend[b] > 0) {
        const int ke = a.kend[b];
        c1 = min(c1, (ke - 1) / CH);
        if (kb * (NW * 16) >= ke) c1 = c0 - 1;
    }
    if (!BAND && a.qguard && a.kend) {
        const int ke = a.kend[b];
        if (ke > 0 && *a.qguard == 0) c1 = min(c1, (ke - 1) / CH);
    }
    const int nch = c1 - c0 + 1;
#define SA_STAGE_QO(t, i) do { const int qc_ = c0 + (t); \
        at_stage<NW>(qbase + (size_t)qc_ * CH * a.ldq, a.ldq, sQ(i), w, l); at_stage<NW>(obase + (size_t)qc_ * CH * a.ldo, a.ldo, sO(i), w, l); \
        at_stage<NW>(qbase + a.lo_q + (size_t)qc_ * CH * a.ldq, a.ldq, sQl(i), w, l); at_stage<NW>(obase + a.lo_o + (size_t)qc_ * CH * a.ldo, a.ldo, sOl(i), w, l); \
        if (w == 0) at_stage_f32x64(a.lse + bh64 * a.L + (size_t)qc_ * CH, sL(i), l); \
        if (w == 1) at_stage_f32x64(a.delta + bh64 * a.L + (size_t)qc_ * CH, sL(i) + 256, l); } while (0)
    KeepWords kw;
    const size_t kcell0 = ((size_t)bh64 * (a.L / 16) + (size_t)kb * NW + __builtin_amdgcn_readfirstlane(w)) * (a.L / CH);
    if (KM && nch > 0) km_load(kw, a.keepB, kcell0 + c0);
    if (nch > 0) SA_STAGE_QO(0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = ch & 1;
        if (ch + 1 < nch) SA_STAGE_QO(ch + 1, cur ^ 1);
        float4 lsc[4], dlc[4];
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) {
            lsc[qf] = *reinterpret_cast<const float4*>(sL(cur) + (qf * 16 + g * 4) * 4);
            dlc[qf] = *reinterpret_cast<const float4*>(sL(cur) + 256 + (qf * 16 + g * 4) * 4);
        }
        const char* tQ = sQ(cur); const char* tO = sO(cur); const char* tQl = sQl(cur); const char* tOl = sOl(cur);
        const int q0 = (c0 + ch) * CH;
        bool edge = false;
        if (BAND) {
            const int wk_lo = kb * (NW * 16) + w * 16;
            edge = !(q0 >= wk_lo + 15 - a.window && q0 + CH - 1 <= wk_lo + a.window);
        }
        f32x4 pd[4], ds[4], sacc4[4], pacc4[4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 qh[2][2], ql[2][2], oh[2][2], ol[2][2];
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int r = (hf * 2 + f2) * 16 + i16, c = kk * 4 + g;
                    qh[f2][kk] = at_frag(tQ, r, c); ql[f2][kk] = at_frag(tQl, r, c);
                    oh[f2][kk] = at_frag(tO, r, c); ol[f2][kk] = at_frag(tOl, r, c);
                }
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2) {
                const float4 l4 = lsc[hf * 2 + f2];
                sacc4[hf * 2 + f2] = (f32x4){mbs - l4.x, mbs - l4.y, mbs - l4.z, mbs - l4.w};
                pacc4[hf * 2 + f2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int f2 = 0; f2 < 2; ++f2) {
                        const int fi = hf * 2 + f2;
                        sacc4[fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pr == 2 ? ql[f2][kk] : qh[f2][kk], pr == 1 ? fkl[kk] : fkh[kk], sacc4[fi], 0, 0, 0);
                        pacc4[fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pr == 2 ? ol[f2][kk] : oh[f2][kk], pr == 1 ? fvl[kk] : fvh[kk], pacc4[fi], 0, 0, 0);
                    }
        }
        const f32x2 sc2v = {LOG2E, LOG2E}, ikv = {ikeep, ikeep};
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) {
            const float dl[4] = {dlc[qf].x, dlc[qf].y, dlc[qf].z, dlc[qf].w};
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const f32x2 t = (f32x2){sacc4[qf][rp * 2], sacc4[qf][rp * 2 + 1]} * sc2v;
                f32x2 pe = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                if (BAND && edge) {
                    if (band_masked(q0 + qf * 16 + g * 4 + rp * 2, key, a.window, a.nglobal)) pe.x = 0.f;
                    if (band_masked(q0 + qf * 16 + g * 4 + rp * 2 + 1, key, a.window, a.nglobal)) pe.y = 0.f;
                }
                f32x2 pk = pe;
                float p0 = pacc4[qf][rp * 2], p1 = pacc4[qf][rp * 2 + 1];
                if (KM) {
                    const uint64_t m0 = kw.m[qf * 4 + rp * 2], m1 = kw.m[qf * 4 + rp * 2 + 1];
                    pk.x = km_sel(pe.x, m0); pk.y = km_sel(pe.y, m1);
                    p0 = km_sel(p0, m0); p1 = km_sel(p1, m1);
                }
                const f32x2 d2 = pe * ((f32x2){p0, p1} * ikv - (f32x2){dl[rp * 2], dl[rp * 2 + 1]});
                pd[qf][rp * 2] = pk.x; pd[qf][rp * 2 + 1] = pk.y;
                ds[qf][rp * 2] = d2.x; ds[qf][rp * 2 + 1] = d2.y;
            }
        }
        // dV^T[d][key] += dO^T[d][q] P_drop[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            bf16x8 fph, fpl, fsh, fsl;
            sa_split8(pd[2 * qp], pd[2 * qp + 1], fph, fpl);
            sa_split8(ds[2 * qp], ds[2 * qp + 1], fsh, fsl);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 foh = at_frag_tr(tO, (2 * qp) * 16 + g * 4, (2 * qp + 1) * 16 + g * 4, d * 16, l);
                const bf16x8 fol = at_frag_tr(tOl, (2 * qp) * 16 + g * 4, (2 * qp + 1) * 16 + g * 4, d * 16, l);
                SA_MFMA3(dv[d], foh, fol, fph, fpl);
                const bf16x8 fqth = at_frag_tr(tQ, (2 * qp) * 16 + g * 4, (2 * qp + 1) * 16 + g * 4, d * 16, l);
                const bf16x8 fqtl = at_frag_tr(tQl, (2 * qp) * 16 + g * 4, (2 * qp + 1) * 16 + g * 4, d * 16, l);
                SA_MFMA3(dk[d], fqth, fqtl, fsh, fsl);
            }
        }
        if (KM && ch + 1 < nch) {
            asm volatile("" ::: "memory");
            km_load(kw, a.keepB, kcell0 + c0 + ch + 1);
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        sa_store4(dkp + d * 16 + g * 4, dki ? dki + d * 16 + g * 4 : nullptr, 3 * H, dk[d][0] * a.scale, dk[d][1] * a.scale, dk[d][2] * a.scale, dk[d][3] * a.scale);
        sa_store4(dvp + d * 16 + g * 4, dvi ? dvi + d * 16 + g * 4 : nullptr, 3 * H, dv[d][0] * ikeep, dv[d][1] * ikeep, dv[d][2] * ikeep, dv[d][3] * ikeep);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static int sattn_fill(SAttnArgs& a, int B, int L, int heads, float scale, float p, int window, int nglobal, const void* keep) {
    if (B <= 0 || L <= 0 || heads <= 0 || (L % CH)) return AMDSEG_ERR_SHAPE;
    if (p < 0.f || p >= 1.f || window < 0 || nglobal < 0 || nglobal > CH) return AMDSEG_ERR_ARG;
    a.B = B; a.L = L; a.heads = heads; a.scale = scale; a.window = window; a.nglobal = window > 0 ? nglobal : 0;
    uint32_t th = (uint32_t)(p * 65536.0f + 0.5f);
    if (p > 0.f && th == 0) th = 1;
    if (th && !keep) return AMDSEG_ERR_ARG;                   // dropout here is read from the layer's keep masks (amdseg_attn_keepmask)
    a.thresh16 = th;
    a.inv_keep = th ? 65536.0f / (float)(65536u - th) : 1.0f;
    a.keepA = (const uint64_t*)keep; a.keepB = a.keepA ? a.keepA + (size_t)B * heads * L * (size_t)L / 64 : nullptr;
    return AMDSEG_OK;
}
template <typename K>
static int sattn_lds(K kernel) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SA_LDS);
    return e == hipSuccess ? AMDSEG_OK : (int)e;
}
#define SA_LAUNCH(kern, grid) do { static bool set_ = false; if (!set_) { int rc_ = sattn_lds(kern); if (rc_) return rc_; set_ = true; } \
        hipLaunchKernelGGL(kern, grid, dim3(256), SA_LDS, s, a); } while (0)

int amdseg_sattn_fwd_impl(const void* qs, int ldq, int lo_q, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale,
                          float p, const void* keep, int window, int nglobal, hipStream_t s, const int* kend, const int* seq_order, void* ctx_image) {
    if (!qs || !mask_bias || !ctx) return AMDSEG_ERR_ARG;
    SAttnArgs a = {};
    int rc = sattn_fill(a, B, L, heads, scale, p, window, nglobal, keep);
    if (rc) return rc;
    a.qs = (const bf16_t*)qs; a.ldq = ldq; a.lo_q = lo_q; a.mask_bias = mask_bias; a.ctx = ctx; a.lse = lse; a.ctx_img = (bf16_t*)ctx_image;
    a.kend = kend; a.seq_order = (window > 0 || !kend) ? nullptr : seq_order;
    const dim3 grid(L / 64, heads, B);
    if (window > 0 && a.thresh16) SA_LAUNCH((sattn_fwd_kernel<true, true>), grid);
    else if (window > 0) SA_LAUNCH((sattn_fwd_kernel<true, false>), grid);
    else if (a.thresh16) SA_LAUNCH((sattn_fwd_kernel<false, true>), grid);
    else SA_LAUNCH((sattn_fwd_kernel<false, false>), grid);
    return amdseg_launch_status();
}

int amdseg_sattn_bwd_impl(const void* qs, int ldq, int lo_q, const float* mask_bias, const float* ctx, const void* dos, int ldo, int lo_o,
                          const float* lse, float* delta, float* dqkv, int B, int L, int heads, float scale, float p, const void* keep, int window,
                          int nglobal, hipStream_t s, const int* kend, const int* seq_order, const int* qguard, void* dqs_image, int ldd) {
    if (!qs || !mask_bias || !ctx || !dos || !lse || !delta || (!dqkv && !dqs_image)) return AMDSEG_ERR_ARG;
    if (dqs_image && ldd < 9 * heads * HD) return AMDSEG_ERR_SHAPE;
    SAttnArgs a = {};
    int rc = sattn_fill(a, B, L, heads, scale, p, window, nglobal, keep);
    if (rc) return rc;
    a.qs = (const bf16_t*)qs; a.ldq = ldq; a.lo_q = lo_q; a.mask_bias = mask_bias; a.ctx = (float*)ctx; a.lse = (float*)lse;
    a.dos = (const bf16_t*)dos; a.ldo = ldo; a.lo_o = lo_o; a.delta = delta; a.dqkv = dqs_image ? nullptr : dqkv;
    a.dqs = (bf16_t*)dqs_image; a.ldd = ldd;
    a.kend = kend; a.seq_order = (window > 0 || !kend) ? nullptr : seq_order; a.qguard = (window > 0 || !kend) ? nullptr : qguard;
    const dim3 grid(L / 64, heads, B);
    if (window > 0 && a.thresh16) {
        SA_LAUNCH((sattn_bwd_dq_kernel<true, true>), grid);
        SA_LAUNCH((sattn_bwd_dkv_kernel<true, true>), dim3((L / 64) * heads * B));
    } else if (window > 0) {
        SA_LAUNCH((sattn_bwd_dq_kernel<true, false>), grid);
        SA_LAUNCH((sattn_bwd_dkv_kernel<true, false>), dim3((L / 64) * heads * B));
    } else if (a.thresh16) {
        SA_LAUNCH((sattn_bwd_dq_kernel<false, true>), grid);
        SA_LAUNCH((sattn_bwd_dkv_kernel<false, true>), grid);
    } else {
        SA_LAUNCH((sattn_bwd_dq_kernel<false, false>), grid);
        SA_LAUNCH((sattn_bwd_dkv_kernel<false, false>), grid);
    }
    return amdseg_launch_status();
}
