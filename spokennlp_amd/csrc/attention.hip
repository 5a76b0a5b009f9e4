// Fused multi-head self-attention (head_dim 64) forward + backward for gfx950, bf16 MFMA / fp32 softmax.
//
// Replaces eager_attention_forward of [hf] models/bert/modeling_bert.py:111-136 (scores = QK^T*d^-1/2 + additive
// key mask -> softmax -> dropout -> .V) and its autograd backward; in-tree 4.x copy of the same arithmetic:
// mmvts/src/models/cross_encoder/bert_model.py:308-352.
//
// Layout: qkv is the fused projection output [B*L, 3H] bf16 (q | k | v, head h at columns h*64); ctx is [B*L, H].
// One 256-thread workgroup = 4 waves x 16 rows (64 query rows, or 64 key rows in the dK/dV kernel) of one (b, h);
// the other operand streams through LDS in 64-row chunks by global_load_lds DMA, double buffered.
// All matmuls are v_mfma_f32_16x16x32_bf16 computed in the *transposed* orientation that keeps the softmax row
// index in the lane id (S^T = K Q^T, O^T = V^T P^T ...), so row max / sum / LSE / delta are lane-local scalars and
// P never leaves registers; k-strided operands are gathered with ds_read_b64_tr_b16, so no transposed copy of
// K, V, Q or dO is ever written to HBM.  The score matrix is never materialised; backward recomputes it from the
// saved log-sum-exp.  Dropout uses the stateless (seed, element) hash of common.h, re-evaluated in backward.
#include "attention_common.h"

// ------------------------------------------------------------------------------------------------ dropout keep masks
// The stateless hash above is evaluated per element in all three kernels: ~85 of the forward kernel's ~186 vector instructions per 64-key
// chunk (8 hash words, 16 compares, 16 selects), the same again in dQ and more in dK/dV -- and these kernels are VALU-bound.  The KM
// instantiations read the keep decisions instead, computed ONCE per layer and step by attn_keepmask_kernel and stored exactly as the
// consumers' v_cndmask wants them: one 64-bit LANE MASK per accumulator register.  The S^T accumulators of a wave are 16 registers
// (fragment fc, element r) x 64 lanes (i16 = row within the wave's 16, g = l >> 4), so the masks of one (16 rows, 64-key chunk) cell are 16
// words = 128 B, fetched with two scalar loads (no vector instruction, no VGPR) and applied with ONE v_cndmask per element.
//   layout A (forward, dQ: lane = query row):  word [bh][q / 16][key / 64][fc * 4 + r], bit g * 16 + i16  <->  q = 16 (q/16) + i16,
//                                              key = 64 (key/64) + fc * 16 + g * 4 + r
//   layout B (dK/dV: lane = key row):          word [bh][key / 16][q / 64][qf * 4 + r], bit g * 16 + i16  <->  key = 16 (key/16) + i16,
//                                              q = 64 (q/64) + qf * 16 + g * 4 + r
// B is the bit transpose of A in two index digits.  The generator makes the bits of a 64 x 64 (query, key) block bit-sliced -- each lane
// holds 64 Bernoulli(keep) bits, built from 16-bit-precision comparisons evaluated 64 at a time with AND / OR on xorshift words, the same
// realised rate (65536 - thresh16) / 65536 as the hash path -- writes them as the block's 64 A words, then exchanges two lane-index bits with
// two word-index bits (two DPP butterfly steps) and swaps two 2-bit digits inside the word (two delta swaps) to get the 64 B words.
// ~250 vector instructions per 4096 elements, against ~250 per 1024 elements and kernel for the hash.
__global__ __launch_bounds__(256) void attn_keepmask_kernel(KeepMaskArgs a) { km_block_body(a, (int)blockIdx.x); }

// consumer side.  The words sit in the constant address space so that the (wave-uniform) loads are scalar loads; inverse_ballot turns a
// wave-uniform 64-bit word into the lane predicate of a select, i.e. v_cndmask with that SGPR pair as its mask operand.
// ------------------------------------------------------------------------------------------------ forward
#define ATTN_FWD_BOUNDS __launch_bounds__(NW * 64, 2)
template <int NW, bool BAND, bool LIST = false, bool KM = false>
__global__ ATTN_FWD_BOUNDS void attn_fwd_kernel(AttnArgs a) {
    // two K / V chunk buffers.  (A third one -- the chunk after next in flight, counted vmcnt -- measured SLOWER in the step, 50.3 vs 48.8 us per
    // launch, round 4: the forward kernel does not wait for its K / V tiles, one chunk of lead covers them; removed in round 6.)
    constexpr int NB = 2;
    __shared__ __attribute__((aligned(16))) char smem[NB * 16384 + NB * 256];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, i16 = l & 15;
    int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (LIST) {                                             // 1-D launch: one (b, h) at a time per XCD, its longest lists first
        int r, bh;
        attn_1d_order(blockIdx.x, a.L / (NW * 16), a.heads * a.B, 2, r, bh);
        h = bh % a.heads; b = bh / a.heads;
        qb = a.korder ? a.korder[h * (a.L / CH) + r] : r;
    } else attn_xcd_remap(qb, h, b, a.heads);
    // sequences in decreasing visible length (list scheduling: with ~3 workgroups per slot the short ones must come LAST to fill the tail;
    // in batch order skipping 16 % of the chunks shortened the launch by 1-6 %)
    if (!BAND && !LIST && a.seq_order) b = a.seq_order[b];
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const int q = qb * (NW * 16) + w * 16 + i16;                       // this lane's query row (shared by the 4 g-groups)
#define bufM(i) (smem + NB * 16384 + (i) * 256)
    const uint64_t prow = ((uint64_t)(b * a.heads + h)) * a.L + q;
#define bufK(i) (smem + (i) * 16384)
#define bufV(i) (smem + 8192 + (i) * 16384)

    // Q as the B operand of S^T = K Q^T : lane holds Q[q][kk*32 + g*8 .. +8]
    // The softmax scale is folded into the Q fragments: exact for a power of two (head_dim 64: 1/8; scores bit-identical to scaling them
    // afterwards), one more bf16 rounding of q otherwise.  The additive key mask then starts the MFMA accumulators as it comes out of LDS --
    // no mask / scale multiplies per chunk (8 packed multiplies of this VALU-bound loop)
    bf16x8 fq[2];
    {
        const bf16_t* qp = a.qkv + (tok0 + q) * a.H3 + h * HD;
        fq[0] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + g * 8), a.scale);
        fq[1] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + 32 + g * 8), a.scale);
    }
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;
    const float sc2 = LOG2E;
    const uint32_t salt = pdrop_salt(pdrop_seedmix(a.seed), prow);
    const uint32_t gsalt = (uint32_t)(g * 2) * 0x9e3779b9u;            // this lane group's keys g*4 .. g*4+3 = key pairs g*2, g*2+1
    const uint32_t thr_hi = a.thresh16 << 16;

    const bf16_t* kbase = a.qkv + tok0 * a.H3 + H + h * HD;
    const bf16_t* vbase = a.qkv + tok0 * a.H3 + 2 * H + h * HD;
    int c0 = 0, c1 = a.L / CH - 1, extra = 0;
    bool pad_block = false;
    if (BAND) {
        const int q_lo = qb * (NW * 16), q_hi = q_lo + NW * 16 - 1;
        c0 = q_lo > a.window ? (q_lo - a.window) / CH : 0;
        c1 = min(c1, (q_hi + a.window) / CH);
        extra = (a.nglobal > 0 && c0 > 0) ? 1 : 0;
        if (a.kend && a.kend[b] > 0) {                      // trailing padding (kend, see attn_visible_chunks): chunks past it add exact zeros;
            const int ke = a.kend[b];                       // a query block wholly inside it is a block of zero rows ([hf] longformer :579)
            pad_block = q_lo >= ke;
            c1 = min(c1, (ke - 1) / CH);
        }
    }
    if (BAND && pad_block) {                                // (workgroup-uniform) every row of the block is a padded query: zero rows, LSE = +inf
        bf16_t* op0 = a.ctx + (tok0 + q) * H + h * HD;
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<uint2*>(op0 + d * 16 + g * 4) = make_uint2(0u, 0u);
        if (a.lse && g == 0) a.lse[prow] = INFINITY;
        return;
    }
    int nch = c1 - c0 + 1 + extra;
    if (!BAND && !LIST) nch = attn_visible_chunks(a, b, nch);
    const int* lst = nullptr;
    ListWalk lw;
    int c_cur = 0, c_nxt = 0, ch_ = 0;                      // LIST: the chunks of iteration ch_ and ch_ + 1
    if (LIST) {
        const int row = h * (a.L / CH) + qb; nch = a.kcnt[row]; lst = a.klist + (size_t)row * a.list_stride;
        lw.init(lst, nch, l); c_cur = lw.next(l); c_nxt = lw.next(l);
    }
    (void)lst;
#define CHUNK_OF(t) (LIST ? ((t) == ch_ ? c_cur : c_nxt) : (BAND && extra && (t) == 0) ? 0 : c0 + (t) - extra)
    KeepWords kw;
    const size_t kcell0 = (((size_t)(b * a.heads + h)) * (a.L / 16) + (size_t)qb * NW + __builtin_amdgcn_readfirstlane(w)) * (a.L / CH);
    if (KM && nch > 0) km_load(kw, a.keepA, kcell0 + CHUNK_OF(0));
    at_stage<NW>(kbase + (size_t)CHUNK_OF(0) * CH * a.H3, a.H3, bufK(0), w, l);
    at_stage<NW>(vbase + (size_t)CHUNK_OF(0) * CH * a.H3, a.H3, bufV(0), w, l);
    // the additive key mask of a chunk travels with its K/V tiles (a global load issued where it is consumed costs a full
    // L2 round trip per key fragment: 4 exposed latencies per chunk)
    if (w == 0) at_stage_f32x64(a.mask_bias + tok0 + CHUNK_OF(0) * CH, bufM(0), l);
    // a wait the compiler can see: otherwise it places the vmcnt wait for the Q fragments (plain global loads) at their first use
    // INSIDE the loop, where it drains the chunk prefetch that was just issued, every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    int cur = 0;
    for (int ch = 0; ch < nch; ++ch) {
        ch_ = ch;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = ch & 1;
        const int nxt = cur ^ 1;
        if (ch + (NB - 1) < nch) {
            at_stage<NW>(kbase + (size_t)CHUNK_OF(ch + NB - 1) * CH * a.H3, a.H3, bufK(nxt), w, l);
            at_stage<NW>(vbase + (size_t)CHUNK_OF(ch + NB - 1) * CH * a.H3, a.H3, bufV(nxt), w, l);
            if (w == 0) at_stage_f32x64(a.mask_bias + tok0 + CHUNK_OF(ch + NB - 1) * CH, bufM(nxt), l);
        }
        float4 mbc[4];
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) mbc[fc] = *reinterpret_cast<const float4*>(bufM(cur) + (fc * 16 + g * 4) * 4);
        const char* tK = bufK(cur);
        const char* tV = bufV(cur);
        const int key0 = CHUNK_OF(ch) * CH;
        // S^T[key][q]: 4 key frags of 16; all K fragments first, then the MFMAs with the k-step outermost so that
        // consecutive MFMAs are independent.  The accumulators START from the additive key mask (the scale sits in Q), so the masked
        // score comes out of the MFMA and no separate scale-and-mask pass over the 16 scores is needed
        // (this kernel is VALU-bound: ~250 vector instructions per 64-key chunk against 16 MFMAs, profiles/r01_gemm_experiments.md)
        f32x4 s[4];
        {
            bf16x8 fk[4][2];
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fk[fc][kk] = at_frag(tK, fc * 16 + i16, kk * 4 + g);
#pragma unroll
            for (int fc = 0; fc < 4; ++fc) s[fc] = (f32x4){mbc[fc].x, mbc[fc].y, mbc[fc].z, mbc[fc].w};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int fc = 0; fc < 4; ++fc) s[fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[fc][kk], fq[kk], s[fc], 0, 0, 0);
        }
        if (BAND) {
            const int wq_lo = qb * (NW * 16) + w * 16;                  // wave-uniform: chunk wholly inside every row's band?
            if (!(key0 >= wq_lo + 15 - a.window && key0 + CH - 1 <= wq_lo + a.window)) {
#pragma unroll
                for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (band_masked(q, key0 + fc * 16 + g * 4 + r, a.window, a.nglobal)) s[fc][r] = -INFINITY;
            }
        }
        // running maximum in the log2 domain (m = max raw score * scale * log2 e; scale > 0): v_max3 chain over the raw scores
        float cmax = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
        cmax = fmaxf(fmaxf(cmax, s[0][3]), s[1][0]);
        cmax = fmaxf(fmaxf(cmax, s[1][1]), s[1][2]);
        cmax = fmaxf(fmaxf(cmax, s[1][3]), s[2][0]);
        cmax = fmaxf(fmaxf(cmax, s[2][1]), s[2][2]);
        cmax = fmaxf(fmaxf(cmax, s[2][3]), s[3][0]);
        cmax = fmaxf(fmaxf(cmax, s[3][1]), s[3][2]);
        cmax = fmaxf(cmax, s[3][3]);
        cmax *= sc2;
        // wave-uniform: rescale only when some row's running max grew -- i.e. when some LANE's own 16 scores exceed its row's running maximum;
        // the exchange between the four lanes of a row (two LDS-crossbar round trips on the loop's critical path) happens only then
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
        const float m_use = (BAND && m_run == -INFINITY) ? 0.f : m_run;   // a row whose visited keys were all masked so far
        // p = 2^(s * scale * log2e - m): packed fp32 fma (two scores per instruction), native 2^x, packed partial sums
        {
            const f32x2 sc2v = {sc2, sc2}, negm = {-m_use, -m_use};
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
            // keep decisions of this (16 rows, chunk) cell as 16 lane masks in SGPRs (attn_keepmask_kernel): one select per probability
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[fc][r] = km_sel(s[fc][r], kw.m[fc * 4 + r]);
        } else if (a.thresh16) {
            // keep-mask: one hash word per key pair, 16 bits per probability.  The high field is compared as the whole word against
            // thresh << 16, the low field after one shift; dropped entries become 0 and the 1 / keep-rate factor is applied ONCE to
            // the finished output row (no per-element and / multiply)
            const uint32_t salt_c = salt + (uint32_t)(key0 >> 1) * 0x9e3779b9u;
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const uint32_t u = pdrop_mix(salt_c + (uint32_t)((fc * 16 + rp * 2) >> 1) * 0x9e3779b9u + gsalt);
                    s[fc][rp * 2] = (u << 16) >= thr_hi ? s[fc][rp * 2] : 0.f;
                    s[fc][rp * 2 + 1] = u >= thr_hi ? s[fc][rp * 2 + 1] : 0.f;
                }
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const bf16x8 fp = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x8 fv = at_frag_tr(tV, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp, o[d], 0, 0, 0);
            }
        }
        if (LIST) { c_cur = c_nxt; c_nxt = lw.next(l); }
        if (KM && ch + 1 < nch) {
            // the next chunk's words, issued behind the last LDS wait of this iteration: scalar loads share lgkmcnt with the LDS and return out
            // of order, so an LDS wait with one of them in flight has to be lgkmcnt(0) and would sit out the load's latency
            asm volatile("" ::: "memory");
            km_load(kw, a.keepA, kcell0 + CHUNK_OF(ch + 1));
        }
    }
    const float lsum = xor_reduce_sum_g(l_part);
    float inv = (a.thresh16 ? a.inv_keep : 1.0f) / lsum;
    float lse_q = (m_run + __builtin_amdgcn_logf(lsum)) * LN2;          // natural-log LSE, as backward expects
    const bool padq = (BAND || LIST) && a.mask_bias[tok0 + q] < 0.f;   // padded query: zero row (Longformer :579, BigBird context_layer * from_mask), p == 0 in backward
    if (padq) { inv = 0.f; lse_q = INFINITY; }
    bf16_t* op = a.ctx + (tok0 + q) * H + h * HD;
    if (!(BAND && q < a.skip_q)) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint2 pk;
            pk.x = padq ? 0u : pack2bf(o[d][0] * inv, o[d][1] * inv);
            pk.y = padq ? 0u : pack2bf(o[d][2] * inv, o[d][3] * inv);
            *reinterpret_cast<uint2*>(op + d * 16 + g * 4) = pk;
        }
    }
    if (a.lse && g == 0) a.lse[prow] = lse_q;
#undef CHUNK_OF
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
__global__ void attn_delta_kernel(const bf16_t* ctx, const bf16_t* dctx, float* delta, int B, int L, int heads) {
    // one thread per (token, head, 8-col chunk) -> 8 lanes per (token, head)
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * L * heads * 8;
    if (gid >= total) return;
    const int c = gid & 7;
    const size_t th = gid >> 3;
    const int h = th % heads;
    const size_t tok = th / heads;
    const int H = heads * HD;
    float x[8], y[8];
    ld8<bf16_t>(ctx + tok * H + h * HD + c * 8, x);
    ld8<bf16_t>(dctx + tok * H + h * HD + c * 8, y);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e] * y[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (c == 0) {
        const int b = tok / L, qq = tok % L;
        delta[((size_t)(b * heads + h)) * L + qq] = s;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
template <int NW, bool BAND, bool LIST = false, bool KM = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int NB = 2;                                   // (the K / V buffer count the bufM macro of the forward kernel refers to)
    __shared__ __attribute__((aligned(16))) char smem[32768 + 512];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, i16 = l & 15;
    int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (LIST) {                                             // 1-D launch: one (b, h) at a time per XCD, its longest lists first
        int r, bh;
        attn_1d_order(blockIdx.x, a.L / (NW * 16), a.heads * a.B, 2, r, bh);
        h = bh % a.heads; b = bh / a.heads;
        qb = a.korder ? a.korder[h * (a.L / CH) + r] : r;
    } else attn_xcd_remap(qb, h, b, a.heads);
    // sequences in decreasing visible length (list scheduling: with ~3 workgroups per slot the short ones must come LAST to fill the tail;
    // in batch order skipping 16 % of the chunks shortened the launch by 1-6 %)
    if (!BAND && !LIST && a.seq_order) b = a.seq_order[b];
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const int q = qb * (NW * 16) + w * 16 + i16;
    const uint64_t prow = ((uint64_t)(b * a.heads + h)) * a.L + q;
#define bufK(i) (smem + (i) * 16384)
#define bufV(i) (smem + 8192 + (i) * 16384)

    if (!BAND && !LIST && a.qguard && a.kend) {             // (workgroup-uniform) a query block of trailing padding whose dO rows are exact zeros:
        const int ke = a.kend[b];                           // delta = rowsum(dO * O) = 0, dS = P * (dP - delta) = 0, dQ = 0
        if (ke > 0 && qb * (NW * 16) >= ke && *a.qguard == 0) {
            bf16_t* op0 = a.dqkv + (tok0 + q) * a.H3 + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) *reinterpret_cast<uint2*>(op0 + d * 16 + g * 4) = make_uint2(0u, 0u);
            if (g == 0) a.delta[prow] = 0.f;
            return;
        }
    }
    bf16x8 fq[2], fdo[2];
    {
        const bf16_t* qp = a.qkv + (tok0 + q) * a.H3 + h * HD;
        fq[0] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + g * 8), a.scale);           // the scale folded into Q, as in the forward kernel
        fq[1] = frag_scale(*reinterpret_cast<const bf16x8*>(qp + 32 + g * 8), a.scale);
        const bf16_t* dp = a.dctx + (tok0 + q) * H + h * HD;
        fdo[0] = *reinterpret_cast<const bf16x8*>(dp + g * 8);
        fdo[1] = *reinterpret_cast<const bf16x8*>(dp + 32 + g * 8);
        if (BAND && q < a.skip_q) {                         // a global token's row: its dO counts as zero (delta = 0, dP = 0, dQ row = 0)
#pragma unroll
            for (int e = 0; e < 8; ++e) { fdo[0][e] = 0; fdo[1][e] = 0; }
        }
    }
    // delta_q = rowsum(dO * O) of this lane's query row: computed here from the fragments already in registers (was a separate
    // launch); written out for the dK/dV kernel, which runs after this one on the same stream
    float delta_q;
    {
        const bf16_t* op = a.ctx + (tok0 + q) * H + h * HD;
        const bf16x8 fo0 = *reinterpret_cast<const bf16x8*>(op + g * 8), fo1 = *reinterpret_cast<const bf16x8*>(op + 32 + g * 8);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc += __uint_as_float((uint32_t)(uint16_t)fo0[e] << 16) * __uint_as_float((uint32_t)(uint16_t)fdo[0][e] << 16);
            acc += __uint_as_float((uint32_t)(uint16_t)fo1[e] << 16) * __uint_as_float((uint32_t)(uint16_t)fdo[1][e] << 16);
        }
        delta_q = xor_reduce_sum_g(acc);
        if (g == 0) a.delta[prow] = delta_q;
    }
    const float sc2 = LOG2E;
    const float nlse_s = -a.lse[prow];                                  // accumulator start: mask - lse, see below
    const uint32_t salt = pdrop_salt(pdrop_seedmix(a.seed), prow);
    const uint32_t gsalt = (uint32_t)(g * 2) * 0x9e3779b9u;
    const uint32_t thr_hi = a.thresh16 << 16;
    const float ikeep = a.thresh16 ? a.inv_keep : 1.0f;
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bf16_t* kbase = a.qkv + tok0 * a.H3 + H + h * HD;
    const bf16_t* vbase = a.qkv + tok0 * a.H3 + 2 * H + h * HD;
    int c0 = 0, c1 = a.L / CH - 1, extra = 0;
    bool pad_block = false;
    if (BAND) {
        const int q_lo = qb * (NW * 16), q_hi = q_lo + NW * 16 - 1;
        c0 = q_lo > a.window ? (q_lo - a.window) / CH : 0;
        c1 = min(c1, (q_hi + a.window) / CH);
        extra = (a.nglobal > 0 && c0 > 0) ? 1 : 0;
        if (a.kend && a.kend[b] > 0) {                      // trailing padding (kend, see attn_visible_chunks): chunks past it add exact zeros;
            const int ke = a.kend[b];                       // a query block wholly inside it is a block of zero rows ([hf] longformer :579)
            pad_block = q_lo >= ke;
            c1 = min(c1, (ke - 1) / CH);
        }
    }
    if (BAND && pad_block) {                                // padded query rows: p == 0 in backward (LSE = +inf) -> dQ = 0; delta finite for dK/dV
        bf16_t* op0 = a.dqkv + (tok0 + q) * a.H3 + h * HD;
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<uint2*>(op0 + d * 16 + g * 4) = make_uint2(0u, 0u);
        if (g == 0) a.delta[prow] = 0.f;
        return;
    }
    int nch = c1 - c0 + 1 + extra;
    if (!BAND && !LIST) nch = attn_visible_chunks(a, b, nch);
    const int* lst = nullptr;
    ListWalk lw;
    int c_cur = 0, c_nxt = 0, ch_ = 0;                      // LIST: the chunks of iteration ch_ and ch_ + 1
    if (LIST) {
        const int row = h * (a.L / CH) + qb; nch = a.kcnt[row]; lst = a.klist + (size_t)row * a.list_stride;
        lw.init(lst, nch, l); c_cur = lw.next(l); c_nxt = lw.next(l);
    }
    (void)lst;
#define CHUNK_OF(t) (LIST ? ((t) == ch_ ? c_cur : c_nxt) : (BAND && extra && (t) == 0) ? 0 : c0 + (t) - extra)
    KeepWords kw;
    const size_t kcell0 = (((size_t)(b * a.heads + h)) * (a.L / 16) + (size_t)qb * NW + __builtin_amdgcn_readfirstlane(w)) * (a.L / CH);
    if (KM && nch > 0) km_load(kw, a.keepA, kcell0 + CHUNK_OF(0));
    at_stage<NW>(kbase + (size_t)CHUNK_OF(0) * CH * a.H3, a.H3, bufK(0), w, l);
    at_stage<NW>(vbase + (size_t)CHUNK_OF(0) * CH * a.H3, a.H3, bufV(0), w, l);
    if (w == 0) at_stage_f32x64(a.mask_bias + tok0 + CHUNK_OF(0) * CH, bufM(0), l);      // key mask rides with the tiles
    for (int ch = 0; ch < nch; ++ch) {
        ch_ = ch;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = ch & 1;
        if (ch + 1 < nch) {
            at_stage<NW>(kbase + (size_t)CHUNK_OF(ch + 1) * CH * a.H3, a.H3, bufK(cur ^ 1), w, l);
            at_stage<NW>(vbase + (size_t)CHUNK_OF(ch + 1) * CH * a.H3, a.H3, bufV(cur ^ 1), w, l);
            if (w == 0) at_stage_f32x64(a.mask_bias + tok0 + CHUNK_OF(ch + 1) * CH, bufM(cur ^ 1), l);
        }
        float4 mbc[4];
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) mbc[fc] = *reinterpret_cast<const float4*>(bufM(cur) + (fc * 16 + g * 4) * 4);
        const char* tK = bufK(cur);
        const char* tV = bufV(cur);
        const int key0 = CHUNK_OF(ch) * CH;
        bool edge = false;
        if (BAND) {
            const int wq_lo = qb * (NW * 16) + w * 16;
            edge = !(key0 >= wq_lo + 15 - a.window && key0 + CH - 1 <= wq_lo + a.window);
        }
        f32x4 ds[4];
        f32x4 sacc4[4], pacc4[4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                               // two fragment pairs at a time: 4 independent MFMA chains
            bf16x8 fk[2][2], fv[2][2];
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    fk[f2][kk] = at_frag(tK, (hf * 2 + f2) * 16 + i16, kk * 4 + g);
                    fv[f2][kk] = at_frag(tV, (hf * 2 + f2) * 16 + i16, kk * 4 + g);
                }
            // the score accumulators start from mask - lse (the scale sits in Q): p = 2^(acc * log2e) needs ONE multiply per element
            // after the MFMA instead of scale, mask and lse terms (VALU-bound kernel)
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2) {
                const float4 m4 = mbc[hf * 2 + f2];
                const f32x2 n2 = {nlse_s, nlse_s};
                const f32x2 lo2 = (f32x2){m4.x, m4.y} + n2, hi2 = (f32x2){m4.z, m4.w} + n2;                  // packed add
                sacc4[hf * 2 + f2] = (f32x4){lo2.x, lo2.y, hi2.x, hi2.y};
                pacc4[hf * 2 + f2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2) {
                    sacc4[hf * 2 + f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[f2][kk], fq[kk], sacc4[hf * 2 + f2], 0, 0, 0);
                    pacc4[hf * 2 + f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[f2][kk], fdo[kk], pacc4[hf * 2 + f2], 0, 0, 0);
                }
        }
        const f32x2 sc2v = {sc2, sc2}, ikv = {ikeep, ikeep}, ndl = {-delta_q, -delta_q};
        if (KM) {                                                       // dropped entries: dP = 0, from the stored lane masks (layout A)
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int r = 0; r < 4; ++r) pacc4[fc][r] = km_sel(pacc4[fc][r], kw.m[fc * 4 + r]);
        } else if (a.thresh16) {                                        // dropped entries: dP = 0 (the 1 / keep-rate factor is ikv below)
            const uint32_t salt_c = salt + (uint32_t)(key0 >> 1) * 0x9e3779b9u + gsalt;
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const uint32_t u = pdrop_mix(salt_c + (uint32_t)(fc * 8 + rp) * 0x9e3779b9u);
                    pacc4[fc][rp * 2] = (u << 16) >= thr_hi ? pacc4[fc][rp * 2] : 0.f;
                    pacc4[fc][rp * 2 + 1] = u >= thr_hi ? pacc4[fc][rp * 2 + 1] : 0.f;
                }
        }
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) {
            const f32x4 sacc = sacc4[fc];
            const f32x4 pacc = pacc4[fc];
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const f32x2 t = (f32x2){sacc[rp * 2], sacc[rp * 2 + 1]} * sc2v;
                f32x2 pe = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                if (BAND && edge) {
                    if (band_masked(q, key0 + fc * 16 + g * 4 + rp * 2, a.window, a.nglobal)) pe.x = 0.f;
                    if (band_masked(q, key0 + fc * 16 + g * 4 + rp * 2 + 1, a.window, a.nglobal)) pe.y = 0.f;
                }
                const f32x2 d2 = pe * ((f32x2){pacc[rp * 2], pacc[rp * 2 + 1]} * ikv + ndl);
                ds[fc][rp * 2] = d2.x; ds[fc][rp * 2 + 1] = d2.y;
            }
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const bf16x8 fds = pack8(ds[2 * kp], ds[2 * kp + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x8 fk = at_frag_tr(tK, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
                dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fds, dq[d], 0, 0, 0);
            }
        }
        if (LIST) { c_cur = c_nxt; c_nxt = lw.next(l); }
        if (KM && ch + 1 < nch) {                                       // behind the last LDS wait of the iteration (see the forward kernel)
            asm volatile("" ::: "memory");
            km_load(kw, a.keepA, kcell0 + CHUNK_OF(ch + 1));
        }
    }
    bf16_t* op = a.dqkv + (tok0 + q) * a.H3 + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint2 pk;
        pk.x = pack2bf(dq[d][0] * a.scale, dq[d][1] * a.scale);
        pk.y = pack2bf(dq[d][2] * a.scale, dq[d][3] * a.scale);
        *reinterpret_cast<uint2*>(op + d * 16 + g * 4) = pk;
    }
#undef CHUNK_OF
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <int NW, bool BAND, bool LIST = false, bool KM = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int NB = 2;
    __shared__ __attribute__((aligned(16))) char smem[32768 + 1024];
#define bufL(i) (smem + 32768 + (i) * 512)
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, i16 = l & 15;
    int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (!BAND && !LIST) { attn_xcd_remap(kb, h, b, a.heads); if (a.seq_order) b = a.seq_order[b]; }
    if (BAND) {
        // 1-D launch.  The key block holding the global keys sees EVERY query chunk (L/64 instead of ~2W/64 + 1): those
        // B*heads long workgroups come first in dispatch order and land round-robin on all XCDs, so they overlap the
        // short ones instead of forming a tail on one XCD (block ids = 0 mod 64 all map to XCD 0 in a 3-D launch).
        // Round 2: one (b, h) at a time per XCD (attn_1d_order), its global key block first; the long workgroup then runs under the
        // next pairs' short ones.
        int bh;
        attn_1d_order(blockIdx.x, a.L / (NW * 16), a.heads * a.B, 1, kb, bh);
        h = bh % a.heads; b = bh / a.heads;
    }
    if (LIST) {                                             // 1-D launch: one (b, h) at a time per XCD, its longest lists first
        int r, bh2;
        attn_1d_order(blockIdx.x, a.L / (NW * 16), a.heads * a.B, 2, r, bh2);
        h = bh2 % a.heads; b = bh2 / a.heads;
        kb = a.qorder ? a.qorder[h * (a.L / CH) + r] : r;
    }
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const int key = kb * (NW * 16) + w * 16 + i16;                    // this lane's key row
    const uint64_t bh = (uint64_t)(b * a.heads + h);
    if (!BAND && !LIST && a.kend) {                         // a key block wholly in the trailing padding: dK = dV = 0 exactly
        const int ke = a.kend[b];
        if (ke > 0 && kb * (NW * 16) >= ke) {
            const int Hq = a.heads * HD;
            bf16_t* okp0 = a.dqkv + (tok0 + key) * a.H3 + Hq + h * HD;
            bf16_t* ovp0 = a.dqkv + (tok0 + key) * a.H3 + 2 * Hq + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                *reinterpret_cast<uint2*>(okp0 + d * 16 + g * 4) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(ovp0 + d * 16 + g * 4) = make_uint2(0u, 0u);
            }
            return;
        }
    }
#define bufQ(i) (smem + (i) * 16384)
#define bufO(i) (smem + 8192 + (i) * 16384)

    // K, V rows of this lane's key as B operands (B[k=d][j=key])
    bf16x8 fk[2], fv[2];
    {
        const bf16_t* kp = a.qkv + (tok0 + key) * a.H3 + H + h * HD;
        fk[0] = frag_scale(*reinterpret_cast<const bf16x8*>(kp + g * 8), a.scale);           // the softmax scale, folded into this side's
        fk[1] = frag_scale(*reinterpret_cast<const bf16x8*>(kp + 32 + g * 8), a.scale);      // register-resident operand (K here, Q in fwd / dQ)
        const bf16_t* vp = a.qkv + (tok0 + key) * a.H3 + 2 * H + h * HD;
        fv[0] = *reinterpret_cast<const bf16x8*>(vp + g * 8);
        fv[1] = *reinterpret_cast<const bf16x8*>(vp + 32 + g * 8);
    }
    const float sc2 = LOG2E;
    const float mbs = a.mask_bias[tok0 + key];                          // accumulator start: mask - lse_row, see below
    // keep-mask word of (row, key): pdrop_bits(pdrop_salt(seedmix, row), key & ~1) = mix(seedmix + row * C1 + (key >> 1) * C2);
    // the key part is a per-lane constant, the row part one multiply per chunk + compile-time offsets
    const uint32_t kc = pdrop_seedmix(a.seed) + (uint32_t)(key >> 1) * 0x9e3779b9u;
    const uint32_t ksh = (key & 1) ? 0u : 16u;                          // this key's 16-bit field, moved to the top of the word
    const uint32_t thr_hi = a.thresh16 << 16;
    const float ikeep = a.thresh16 ? a.inv_keep : 1.0f;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const bf16_t* qbase = a.qkv + tok0 * a.H3 + h * HD;
    const bf16_t* obase = a.dctx + tok0 * H + h * HD;
    // query chunks that can see this key block: the band around it -- or every chunk if the block holds a global key
    int c0 = 0, c1 = a.L / CH - 1;
    if (BAND && !(a.nglobal > 0 && kb * (NW * 16) < a.nglobal)) {
        const int k_lo = kb * (NW * 16), k_hi = k_lo + NW * 16 - 1;
        c0 = k_lo > a.window ? (k_lo - a.window) / CH : 0;
        c1 = min(c1, (k_hi + a.window) / CH);
    }
    if (BAND && a.kend && a.kend[b] > 0) {                  // padded query rows have p == 0 (LSE = +inf): their chunks add exact zeros; a key block
        const int ke = a.kend[b];                           // past the last unmasked key gets dK = dV = 0 (the loop below is empty for it)
        c1 = min(c1, (ke - 1) / CH);
        if (kb * (NW * 16) >= ke) c1 = c0 - 1;
    }
    if (!BAND && !LIST && a.qguard && a.kend) {             // query chunks of trailing padding with exact-zero dO rows (AttnArgs.qguard): dP = 0 and
        const int ke = a.kend[b];                           // delta = 0 there, so they add exact zeros to dK and dV
        if (ke > 0 && *a.qguard == 0) c1 = min(c1, (ke - 1) / CH);
    }
    int nch = c1 - c0 + 1;
    const int* lst = nullptr;
    ListWalk lw;
    int c_cur = 0, c_nxt = 0, ch_ = 0;
    if (LIST) {
        const int row = h * (a.L / CH) + kb; nch = a.qcnt[row]; lst = a.qlist + (size_t)row * a.list_stride;
        lw.init(lst, nch, l); c_cur = lw.next(l); c_nxt = lw.next(l);
    }
    (void)lst;
#define QCHUNK_OF(t) (LIST ? ((t) == ch_ ? c_cur : c_nxt) : c0 + (t))
    KeepWords kw;
    const size_t kcell0 = ((size_t)bh * (a.L / 16) + (size_t)kb * NW + __builtin_amdgcn_readfirstlane(w)) * (a.L / CH);
    if (KM && nch > 0) km_load(kw, a.keepB, kcell0 + QCHUNK_OF(0));
    at_stage<NW>(qbase + (size_t)QCHUNK_OF(0) * CH * a.H3, a.H3, bufQ(0), w, l);
    at_stage<NW>(obase + (size_t)QCHUNK_OF(0) * CH * H, H, bufO(0), w, l);
    // LSE and delta of the chunk's 64 query rows ride with the Q / dO tiles (were 8 exposed global loads per chunk)
    if (w == 0) at_stage_f32x64(a.lse + bh * a.L + (size_t)QCHUNK_OF(0) * CH, bufL(0), l);
    if (w == 1) at_stage_f32x64(a.delta + bh * a.L + (size_t)QCHUNK_OF(0) * CH, bufL(0) + 256, l);
    for (int ch = 0; ch < nch; ++ch) {
        ch_ = ch;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = ch & 1;
        if (ch + 1 < nch) {
            const int qn = QCHUNK_OF(ch + 1);
            at_stage<NW>(qbase + (size_t)qn * CH * a.H3, a.H3, bufQ(cur ^ 1), w, l);
            at_stage<NW>(obase + (size_t)qn * CH * H, H, bufO(cur ^ 1), w, l);
            if (w == 0) at_stage_f32x64(a.lse + bh * a.L + (size_t)qn * CH, bufL(cur ^ 1), l);
            if (w == 1) at_stage_f32x64(a.delta + bh * a.L + (size_t)qn * CH, bufL(cur ^ 1) + 256, l);
        }
        float4 lsc[4], dlc[4];
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) {
            lsc[qf] = *reinterpret_cast<const float4*>(bufL(cur) + (qf * 16 + g * 4) * 4);
            dlc[qf] = *reinterpret_cast<const float4*>(bufL(cur) + 256 + (qf * 16 + g * 4) * 4);
        }
        const char* tQ = bufQ(cur);
        const char* tO = bufO(cur);
        const int q0 = QCHUNK_OF(ch) * CH;
        bool edge = false;
        if (BAND) {
            const int wk_lo = kb * (NW * 16) + w * 16;
            edge = !(q0 >= wk_lo + 15 - a.window && q0 + CH - 1 <= wk_lo + a.window) || q0 < a.skip_q;
        }
        f32x4 pd[4], ds[4];     // P_drop[q][key], dS[q][key] : lane key = i16, q = qf*16 + g*4 + r
        f32x4 sacc4[4], pacc4[4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                               // two fragment pairs at a time: 4 independent MFMA chains
            bf16x8 fqa[2][2], foa[2][2];
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    fqa[f2][kk] = at_frag(tQ, (hf * 2 + f2) * 16 + i16, kk * 4 + g);
                    foa[f2][kk] = at_frag(tO, (hf * 2 + f2) * 16 + i16, kk * 4 + g);
                }
            // score accumulators start from mask_key - lse_row (see the dQ kernel)
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2) {
                const float4 l4 = lsc[hf * 2 + f2];
                const f32x2 m2 = {mbs, mbs};
                const f32x2 lo2 = m2 - (f32x2){l4.x, l4.y}, hi2 = m2 - (f32x2){l4.z, l4.w};                  // packed add
                sacc4[hf * 2 + f2] = (f32x4){lo2.x, lo2.y, hi2.x, hi2.y};
                pacc4[hf * 2 + f2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2) {
                    sacc4[hf * 2 + f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqa[f2][kk], fk[kk], sacc4[hf * 2 + f2], 0, 0, 0);
                    pacc4[hf * 2 + f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(foa[f2][kk], fv[kk], pacc4[hf * 2 + f2], 0, 0, 0);
                }
        }
        const f32x2 sc2v = {sc2, sc2}, ikv = {ikeep, ikeep};
        // (tried and measured slower on this kernel: sharing the hash word of a key pair between lanes 2t / 2t + 1 through DPP,
        // 86 -> 113 us; a separate copy of the element loop per dropout setting, 86 -> 102 us at 186 VGPRs)
        const uint32_t rowc = kc + (uint32_t)(bh * a.L + q0 + g * 4) * 0x85ebca6bu;      // + (qf * 16 + r) * C1 per element
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) {
            const f32x4 sacc = sacc4[qf];
            f32x4 pacc = pacc4[qf];
            const float dl[4] = {dlc[qf].x, dlc[qf].y, dlc[qf].z, dlc[qf].w};
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const f32x2 t = (f32x2){sacc[rp * 2], sacc[rp * 2 + 1]} * sc2v;
                f32x2 pe = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                if (BAND && edge) {
                    // (a query in front of skip_q: its dO counts as zero -- p = 0 removes its terms from dV and, through dS = p (dP - delta), from dK)
                    const int qa = q0 + qf * 16 + g * 4 + rp * 2;
                    if (band_masked(qa, key, a.window, a.nglobal) || qa < a.skip_q) pe.x = 0.f;
                    if (band_masked(qa + 1, key, a.window, a.nglobal) || qa + 1 < a.skip_q) pe.y = 0.f;
                }
                f32x2 pk = pe;                                           // P_drop without the 1 / keep-rate factor (applied to dV at the end)
                if (KM) {                                                // stored lane masks (layout B: lane = key row)
                    const uint64_t m0 = kw.m[qf * 4 + rp * 2], m1 = kw.m[qf * 4 + rp * 2 + 1];
                    pk.x = km_sel(pe.x, m0); pk.y = km_sel(pe.y, m1);
                    pacc[rp * 2] = km_sel(pacc[rp * 2], m0); pacc[rp * 2 + 1] = km_sel(pacc[rp * 2 + 1], m1);
                } else if (a.thresh16) {
                    const bool k0 = (pdrop_mix(rowc + (uint32_t)(qf * 16 + rp * 2) * 0x85ebca6bu) << ksh) >= thr_hi;
                    const bool k1 = (pdrop_mix(rowc + (uint32_t)(qf * 16 + rp * 2 + 1) * 0x85ebca6bu) << ksh) >= thr_hi;
                    pk.x = k0 ? pe.x : 0.f; pk.y = k1 ? pe.y : 0.f;
                    pacc[rp * 2] = k0 ? pacc[rp * 2] : 0.f; pacc[rp * 2 + 1] = k1 ? pacc[rp * 2 + 1] : 0.f;
                }
                const f32x2 d2 = pe * ((f32x2){pacc[rp * 2], pacc[rp * 2 + 1]} * ikv - (f32x2){dl[rp * 2], dl[rp * 2 + 1]});
                pd[qf][rp * 2] = pk.x; pd[qf][rp * 2 + 1] = pk.y;
                ds[qf][rp * 2] = d2.x; ds[qf][rp * 2 + 1] = d2.y;
            }
        }
        // dV^T[d][key] += dO^T[d][q] P_drop[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            const bf16x8 fp = pack8(pd[2 * qp], pd[2 * qp + 1]);
            const bf16x8 fds = pack8(ds[2 * qp], ds[2 * qp + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x8 fo = at_frag_tr(tO, (2 * qp) * 16 + g * 4, (2 * qp + 1) * 16 + g * 4, d * 16, l);
                dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo, fp, dv[d], 0, 0, 0);
                bf16x8 fqt = at_frag_tr(tQ, (2 * qp) * 16 + g * 4, (2 * qp + 1) * 16 + g * 4, d * 16, l);
                dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, fds, dk[d], 0, 0, 0);
            }
        }
        if (LIST) { c_cur = c_nxt; c_nxt = lw.next(l); }
        if (KM && ch + 1 < nch) {                                       // behind the last LDS wait of the iteration (see the forward kernel)
            asm volatile("" ::: "memory");
            km_load(kw, a.keepB, kcell0 + QCHUNK_OF(ch + 1));
        }
    }
    bf16_t* okp = a.dqkv + (tok0 + key) * a.H3 + H + h * HD;
    bf16_t* ovp = a.dqkv + (tok0 + key) * a.H3 + 2 * H + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint2 pk;
        pk.x = pack2bf(dk[d][0] * a.scale, dk[d][1] * a.scale);
        pk.y = pack2bf(dk[d][2] * a.scale, dk[d][3] * a.scale);
        *reinterpret_cast<uint2*>(okp + d * 16 + g * 4) = pk;
        uint2 pv;
        pv.x = pack2bf(dv[d][0] * ikeep, dv[d][1] * ikeep);
        pv.y = pack2bf(dv[d][2] * ikeep, dv[d][3] * ikeep);
        *reinterpret_cast<uint2*>(ovp + d * 16 + g * 4) = pv;
    }
}

static int attn_fill(AttnArgs& a, int B, int L, int heads, float scale, float p, uint64_t seed, int window, int nglobal) {
    if (B <= 0 || L <= 0 || heads <= 0 || (L % CH)) return AMDSEG_ERR_SHAPE;
    if (p < 0.f || p >= 1.f || window < 0 || nglobal < 0 || nglobal > CH) return AMDSEG_ERR_ARG;
    a.window = window; a.nglobal = window > 0 ? nglobal : 0;
    a.B = B; a.L = L; a.heads = heads; a.H3 = 3 * heads * HD; a.scale = scale; a.seed = seed;
    uint32_t th = (uint32_t)(p * 65536.0f + 0.5f);
    if (p > 0.f && th == 0) th = 1;
    a.thresh16 = th;
    a.inv_keep = th ? 65536.0f / (float)(65536u - th) : 1.0f;    // unbiased for the realised drop rate th/65536
    return AMDSEG_OK;
}

// size in bytes of the keep-mask buffer of one layer (layout A, then layout B)
size_t amdseg_attn_keepmask_bytes_impl(int B, int L, int heads) { return (size_t)B * heads * L * (size_t)L / 8 * 2; }

// the keep masks of one attention layer and step (both layouts); the hash seed is the one the hash path would take
int amdseg_attn_keepmask_impl(void* keep, int B, int L, int heads, float p, uint64_t seed, const int* kend, hipStream_t s, int window, int nglobal) {
    KeepMaskArgs k;
    const int rc = km_fill(k, keep, B, L, heads, p, seed, kend, window, nglobal);
    if (rc) return rc;
    AMDSEG_LAUNCH_PROF(AMDSEG_PROF_KEEPMASK, 2.0 * B * heads * (double)L * L / 8.0, attn_keepmask_kernel, dim3((unsigned)km_blocks(B, L, heads)), dim3(256), 0, s, k);
    return amdseg_launch_status();
}

int amdseg_attn_fwd_impl(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads,
                         float scale, float p, uint64_t seed, int window, int nglobal, hipStream_t s, const int* kend, const int* seq_order,
                         const void* keep, int skip_q) {
    if (!qkv || !mask_bias || !ctx) return AMDSEG_ERR_ARG;
    AttnArgs a = {};
    int rc = attn_fill(a, B, L, heads, scale, p, seed, window, nglobal);
    if (rc) return rc;
    a.skip_q = window > 0 ? skip_q : 0;
    a.kend = kend; a.seq_order = (window > 0 || !kend) ? nullptr : seq_order; a.qguard = nullptr;
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = lse;
    a.keepA = (const uint64_t*)keep; a.keepB = a.keepA ? a.keepA + (size_t)B * heads * L * (size_t)L / 64 : nullptr;
    if (keep && a.thresh16 && window == 0) {                // dropout decisions read from the layer's keep masks (attn_keepmask_kernel)
        const double work_km = 4.0 * B * heads * (double)L * (double)L * HD;
        if (L % 128 == 0) AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_FWD, work_km, (attn_fwd_kernel<8, false, false, true>), dim3(L / 128, heads, B), dim3(512), 0, s, a);
        else AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_FWD, work_km, (attn_fwd_kernel<4, false, false, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        return amdseg_launch_status();
    }
    // algorithmic FLOPs: QK^T + PV over the visible keys (full: L, band: 2W + 1 + G)
    const double span = window > 0 ? (double)(2 * window + 1 + a.nglobal) : (double)L;
    const double work = 4.0 * B * heads * (double)L * span * HD;
    if (window > 0) {
        // band: 64-query workgroups visit (64 + 2W) / 64 = 9 chunks at W = 256, 128-query ones 10 -- measured 152 vs 172 us per layer at
        // longformer-base (full attention is the other way round: 8 waves share every K / V tile)
        if (keep && a.thresh16) AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_FWD, work, (attn_fwd_kernel<4, true, false, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        else AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_FWD, work, (attn_fwd_kernel<4, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
    } else {
        if (L % 128 == 0) AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_FWD, work, (attn_fwd_kernel<8, false>), dim3(L / 128, heads, B), dim3(512), 0, s, a);
        else AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_FWD, work, (attn_fwd_kernel<4, false>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
    }
    return amdseg_launch_status();
}

int amdseg_attn_bwd_impl(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                         float* delta, void* dqkv, int B, int L, int heads, float scale, float p, uint64_t seed,
                         int window, int nglobal, hipStream_t s, const int* kend, const int* seq_order, const int* qguard, const void* keep, int skip_q) {
    if (!qkv || !mask_bias || !ctx || !dctx || !lse || !delta || !dqkv) return AMDSEG_ERR_ARG;
    AttnArgs a = {};
    int rc = attn_fill(a, B, L, heads, scale, p, seed, window, nglobal);
    if (rc) return rc;
    a.skip_q = window > 0 ? skip_q : 0;
    a.kend = kend; a.seq_order = (window > 0 || !kend) ? nullptr : seq_order; a.qguard = (window > 0 || !kend) ? nullptr : qguard;
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = (float*)lse;
    a.dctx = (const bf16_t*)dctx; a.delta = delta; a.dqkv = (bf16_t*)dqkv;
    a.keepA = (const uint64_t*)keep; a.keepB = a.keepA ? a.keepA + (size_t)B * heads * L * (size_t)L / 64 : nullptr;
    if (keep && a.thresh16 && window == 0) {                // the forward's keep masks again (layout A for dQ, layout B for dK/dV)
        const double unit_km = 2.0 * B * heads * (double)L * (double)L * HD;
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DQ, 2.0 * unit_km, (attn_bwd_dq_kernel<4, false, false, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, 3.0 * unit_km, (attn_bwd_dkv_kernel<4, false, false, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        return amdseg_launch_status();
    }
    const size_t total = (size_t)B * L * heads * 8;
    (void)total;           // delta = rowsum(dO * O) is produced by the dQ kernel (attn_delta_kernel is kept for reference / tests)
    // backward kernels need 144-168 VGPRs: 4-wave workgroups keep 3 waves per SIMD resident (8-wave ones would spill or halve occupancy)
    // algorithmic FLOPs of backward = 2.5 x forward (dV, dP, dQ, dK + the S recomputation counted once): dQ kernel 3 products, dK/dV 4
    const double span = window > 0 ? (double)(2 * window + 1 + a.nglobal) : (double)L;
    const double unit = 2.0 * B * heads * (double)L * span * HD;
    if (window > 0 && keep && a.thresh16) {                 // band, dropout decisions from the keep masks generated for the band's cells
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DQ, 2.0 * unit, (attn_bwd_dq_kernel<4, true, false, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, 3.0 * unit, (attn_bwd_dkv_kernel<4, true, false, true>), dim3((L / 64) * heads * B), dim3(256), 0, s, a);
    } else if (window > 0) {
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DQ, 2.0 * unit, (attn_bwd_dq_kernel<4, true>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, 3.0 * unit, (attn_bwd_dkv_kernel<4, true>), dim3((L / 64) * heads * B), dim3(256), 0, s, a);
    } else {
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DQ, 2.0 * unit, (attn_bwd_dq_kernel<4, false>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, 3.0 * unit, (attn_bwd_dkv_kernel<4, false>), dim3(L / 64, heads, B), dim3(256), 0, s, a);
    }
    return amdseg_launch_status();
}


// ------------------------------------------------------------------------------------------------ block-list attention
// BigBird block-sparse attention ([hf] models/big_bird/modeling_big_bird.py: bigbird_block_sparse_attention): query block i (64
// rows) attends to an explicit list of 64-key blocks (global first/last blocks, the 3-block window, the per-head random blocks);
// the reference concatenates those key blocks and takes ONE softmax over the concatenation, which is what streaming the listed
// blocks through the online softmax computes -- including blocks that are listed more than once.  klist/kcnt: per (head, query
// block); qlist/qcnt: the transposed lists per (head, key block) for dK/dV, with the same multiplicities.  No dropout on the
// probabilities (the reference's block-sparse path has none).
static int attn_list_check(int L, const int* klist, const int* kcnt, int stride) {
    if (!klist || !kcnt || stride <= 0) return AMDSEG_ERR_ARG;
    if (L % CH) return AMDSEG_ERR_SHAPE;
    return AMDSEG_OK;
}

int amdseg_attn_list_fwd_impl(const void* qkv, const float* mask_bias, void* ctx, float* lse, int B, int L, int heads, float scale,
                              const int* klist, const int* kcnt, int list_stride, const int* korder, hipStream_t s) {
    if (!qkv || !mask_bias || !ctx) return AMDSEG_ERR_ARG;
    AttnArgs a = {};
    int rc = attn_fill(a, B, L, heads, scale, 0.f, 0, 0, 0);
    if (rc) return rc;
    if ((rc = attn_list_check(L, klist, kcnt, list_stride))) return rc;
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = lse;
    a.klist = klist; a.kcnt = kcnt; a.list_stride = list_stride; a.korder = korder;
    hipLaunchKernelGGL((attn_fwd_kernel<4, false, true>), dim3((L / 64) * heads * B), dim3(256), 0, s, a);
    return amdseg_launch_status();
}

int amdseg_attn_list_bwd_impl(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse,
                              float* delta, void* dqkv, int B, int L, int heads, float scale, const int* klist, const int* kcnt,
                              const int* qlist, const int* qcnt, int list_stride, const int* korder, const int* qorder, hipStream_t s) {
    if (!qkv || !mask_bias || !ctx || !dctx || !lse || !delta || !dqkv) return AMDSEG_ERR_ARG;
    AttnArgs a = {};
    int rc = attn_fill(a, B, L, heads, scale, 0.f, 0, 0, 0);
    if (rc) return rc;
    if ((rc = attn_list_check(L, klist, kcnt, list_stride)) || (rc = attn_list_check(L, qlist, qcnt, list_stride))) return rc;
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = (float*)lse;
    a.dctx = (const bf16_t*)dctx; a.delta = delta; a.dqkv = (bf16_t*)dqkv;
    a.klist = klist; a.kcnt = kcnt; a.qlist = qlist; a.qcnt = qcnt; a.list_stride = list_stride; a.korder = korder; a.qorder = qorder;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<4, false, true>), dim3((L / 64) * heads * B), dim3(256), 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<4, false, true>), dim3((L / 64) * heads * B), dim3(256), 0, s, a);
    return amdseg_launch_status();
}
