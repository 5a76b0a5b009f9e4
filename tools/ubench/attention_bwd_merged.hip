// Attention backward as ONE kernel (full attention, head_dim 64, L % 256 == 0): dQ, dK and dV of a (batch, head) pair from one evaluation of
// P = exp(S - lse) and dS = P o (keep o dP / (1 - p) - delta).
// (same reference lines as attention.hip: the autograd backward of BertSelfAttention, [hf] models/bert/modeling_bert.py:300-340.)
//
// The two-kernel form (attention.hip: attn_bwd_dq_kernel query-stationary, attn_bwd_dkv_kernel key-stationary) evaluates S, P, dP and dS twice --
// 7 matmul units for 5 algorithmic -- and both kernels are VALU-issue bound on exactly that elementwise work.  Here:
//   * one 16-wave workgroup per (batch, head); wave w owns 16 keys of the current 256-key block exactly as a dkv wave does (K_w, V_w as register
//     B operands, dK^T / dV^T [64 d x 16 keys] in 32 accumulator registers, S / dP in the [q][key] orientation whose accumulator layout IS the B
//     operand of the dV / dK products) and the 16 waves share every Q / dO chunk (one LDS-DMA piece per wave and chunk instead of four);
//   * dQ needs the reduction over KEYS, i.e. across waves: every wave writes its dS^T slab (bf16, [key][q], 8 B per lane and row fragment)
//     into a [256 keys][64 q] LDS image, and after ONE barrier wave w computes the 16 x 16 tile (d-tile w & 3, q-tile w >> 2) of the chunk's
//     dQ^T = K^T dS^T over the 256 keys: both operands are transposed gathers (ds_read_b64_tr_b16) of row-major LDS images -- the K block
//     staged once per key block, and the dS^T image;
//   * the second key block of a 512-token sequence adds to the first one's dQ: the partial sits in `dq_part` as raw per-lane float4 dumps
//     (written and read back by the SAME lane of the SAME wave: no cross-thread visibility question, deterministic order of addition), the
//     last visible key block converts and stores bf16;
//   * delta = rowsum(dO o O) of a chunk is computed one chunk ahead from two 8-byte global loads per lane issued two chunks ahead (the dQ
//     kernel used to produce it): no separate launch, no extra barrier;
//   * trailing padding (kend / qguard, as in the two-kernel form): key blocks without an unmasked key get dK = dV = 0, query chunks of
//     exact-zero dO rows get dQ = 0; neither is visited.
//   * the dQ tiles of chunk c are computed during step c + 1 (two dS^T images): one barrier per step, and the latency-bound gather chain of the dQ
//     product runs beside the other waves' elementwise work instead of in a phase of its own.
// LDS: K block 32 KiB + two dS^T images 64 KiB + two (Q, dO) chunk buffers 32 KiB + lse / delta 1 KiB = 129 KiB.
#include "attention_common.h"

#define MG_KB 256
// timing probes (wrong results; tools/dbg/mg_timeline.py, profiles/r05_attn_bwd_merged.md): compiled in only under -DAMDSEG_PROBES, which also makes
// amdseg_abi_version() negative so that lib.load() refuses the library as a product
#ifdef AMDSEG_PROBES
#define MG_DBG(bit) (a.skip_q & (bit))
#else
#define MG_DBG(bit) false
#endif
#define MG_OFF_K 0
#define MG_OFF_EX 32768
#define MG_OFF_QO 98304
#define MG_OFF_L 131072
#define MG_LDS (131072 + 1024)

// 8 of the 16 lane-mask words of a (16 rows, chunk) cell: the words of row fragments 2 * hf and 2 * hf + 1 (scalar loads; see km_load)
struct KeepHalf { uint64_t m[8]; };
__device__ __forceinline__ void km_load_half(KeepHalf& k, const uint64_t* base, size_t cell, int hf) {
    km_cptr p = (km_cptr)(uintptr_t)(base + cell * 16 + hf * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) k.m[i] = p[i];
}

// KT = 16-key tiles per wave: 1 -> 16 waves (1024 threads, 4 waves per SIMD, 128 VGPRs), 2 -> 8 waves (512 threads, 2 per SIMD, 256 VGPRs: every Q / dO
// fragment read from LDS feeds two key tiles, every K fragment of the dQ step two q-tiles -- a quarter fewer issue slots per unit of work).
// One workgroup per (256-key block, batch, head): blockIdx = kb * (B * heads) + pair, so every block-0 workgroup is dispatched before any block-1 one.
// `sync` (the tail of the caller's scratch, zeroed once by the caller): [0] generation of the launch, [1] workgroups finished, [2 + pair * nkb + kb]
// = generation + 1 once block kb's dQ partial of that pair is complete.  The partials cross workgroups as agent-scope (L2-coherent) relaxed atomics,
// published by a flag store behind s_waitcnt vmcnt(0); the block-(kb + 1) workgroup of the pair spins on that flag before its first partial read.
template <bool KM, int KT>
__global__ __launch_bounds__(1024 / KT, 1) void attn_bwd_merged_kernel(AttnArgs a, float* dq_part, uint32_t* sync) {
    constexpr int NW = 16 / KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63, g = l >> 4, i16 = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (MG_DBG(32)) return;
    uint64_t* tl = MG_DBG(256) ? reinterpret_cast<uint64_t*>(sync + 2 + 4096) + (size_t)blockIdx.x * 4 : nullptr;      // timeline probe: start / after the partner wait / end, XCC
    if (tl && threadIdx.x == 0) { tl[0] = wall_clock64(); uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); tl[3] = xcc; }
    const int nbh = a.B * a.heads;
    const int kb = blockIdx.x / nbh;
    const int pair = blockIdx.x - kb * nbh;
    int b = pair / a.heads;
    const int h = pair - b * a.heads;
    if (a.seq_order) b = a.seq_order[b];                     // longest sequences first
    const int H = a.heads * HD;
    const size_t tok0 = (size_t)b * a.L;
    const uint64_t bh = (uint64_t)(b * a.heads + h);
    const int nq_all = a.L / CH, nkb = a.L / MG_KB;
    const uint32_t token = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    uint32_t* flags = sync + 2 + (size_t)bh * nkb;
    int ke = a.L;
    bool padded = false;
    if (a.kend) { const int k_ = a.kend[b]; if (k_ > 0) { ke = k_; padded = true; } }
    const int nkbv = min(nkb, (ke + MG_KB - 1) / MG_KB);    // key blocks that hold an unmasked key
    int nqv = nq_all;                                       // query chunks whose dO rows are not known to be exact zeros
    if (padded && a.qguard && *a.qguard == 0) nqv = min(nq_all, (ke + CH - 1) / CH);
    const int dt = w & 3, qt0 = (w >> 2) * KT;              // this wave's 16 x 16 tiles of a chunk's dQ^T: d-tile dt, q-tiles qt0 .. qt0 + KT - 1
    const bool last = kb == nkbv - 1;                       // this block converts and stores dQ; earlier ones leave fp32 partials

#define mgK(t) (smem + MG_OFF_K + (t) * 8192)
#define mgEX(i, t) (smem + MG_OFF_EX + (i) * 32768 + (t) * 8192)
#define mgQ(i) (smem + MG_OFF_QO + (i) * 16384)
#define mgO(i) (smem + MG_OFF_QO + 8192 + (i) * 16384)
#define mgLse(i) (smem + MG_OFF_L + (i) * 512)
#define mgDl(i) (smem + MG_OFF_L + 256 + (i) * 512)

    if (kb >= nkbv || nqv == 0) {
        // a key block without an unmasked key (or a sequence without any gradient): dK = dV = 0 for its rows
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = kb * MG_KB + (w * KT + kt) * 16 + i16;
            bf16_t* okp0 = a.dqkv + (tok0 + key) * a.H3 + H + h * HD;
            bf16_t* ovp0 = a.dqkv + (tok0 + key) * a.H3 + 2 * H + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                *reinterpret_cast<uint2*>(okp0 + d * 16 + g * 4) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(ovp0 + d * 16 + g * 4) = make_uint2(0u, 0u);
            }
        }
    }
    if (kb == 0)                                            // query chunks that are never visited: dQ = 0
        for (int c = nqv; c < nq_all; ++c)
#pragma unroll
            for (int j = 0; j < KT; ++j)
                *reinterpret_cast<uint2*>(a.dqkv + (tok0 + c * CH + (qt0 + j) * 16 + i16) * a.H3 + h * HD + dt * 16 + g * 4) = make_uint2(0u, 0u);

    if (kb < nkbv && nqv > 0) {
    const bf16_t* qbase = a.qkv + tok0 * a.H3 + h * HD;
    const bf16_t* obase = a.dctx + tok0 * H + h * HD;
    const bf16_t* cbase = a.ctx + tok0 * H + h * HD;
    const float sc2 = LOG2E;
    const float ikeep = (KM && a.thresh16) ? a.inv_keep : 1.0f;

    // lane terms of the per-step addresses, formed once: a step then adds wave-uniform (scalar) bases
    uint32_t st_off[KT];                                                // chunk pieces (LDS-DMA): byte offset of this lane's 16 B inside the chunk's rows
    uint32_t dl_off[KT];                                                // delta rows: byte offset of this lane's 8 B of O / dO
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int p = w + q * NW;                                       // pieces 0-7: Q rows 8p .. 8p + 8; pieces 8-15: dO rows
        const int r = (p & 7) * 8 + (l >> 3), cc = (l & 7) ^ swz(r);
        st_off[q] = (uint32_t)((r * (p < 8 ? a.H3 : H) + cc * 8) * 2);
        dl_off[q] = (uint32_t)((((w * KT + q) * 4 + g) * H + i16 * 4) * 2);
    }
    // one chunk = Q tile (8 1-KiB pieces) + dO tile (8 pieces) + 64 LSE values (wave 0): 16 / NW pieces per wave
    auto stage_chunk = [&](int c, int buf) {
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            const int p = w + q * NW, R0 = (p & 7) * 8;
            if (p < 8) amdseg_glds16_saddr(qbase + (size_t)c * CH * a.H3, st_off[q], mgQ(buf) + R0 * 128);
            else amdseg_glds16_saddr(obase + (size_t)c * CH * H, st_off[q], mgO(buf) + R0 * 128);
        }
        if (w == 0) at_stage_f32x64(a.lse + bh * a.L + (size_t)c * CH, mgLse(buf), l);
    };
    // delta = rowsum(dO o O) of a chunk: wave w covers rows (w * KT + j) * 4 + g, lane i16 the columns i16*4 .. +4 (8 B of O and of dO)
    uint2 do2[KT], o2[KT];
    auto delta_load = [&](int c) {
        const char* cb = reinterpret_cast<const char*>(cbase + (size_t)c * CH * H);
        const char* ob = reinterpret_cast<const char*>(obase + (size_t)c * CH * H);
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            o2[j] = *reinterpret_cast<const uint2*>(cb + dl_off[j]);
            do2[j] = *reinterpret_cast<const uint2*>(ob + dl_off[j]);
        }
    };
    auto delta_put = [&](int buf) {
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            float s = __uint_as_float(o2[j].x << 16) * __uint_as_float(do2[j].x << 16) + __uint_as_float(o2[j].x & 0xffff0000u) * __uint_as_float(do2[j].x & 0xffff0000u) +
                      __uint_as_float(o2[j].y << 16) * __uint_as_float(do2[j].y << 16) + __uint_as_float(o2[j].y & 0xffff0000u) * __uint_as_float(do2[j].y & 0xffff0000u);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            if (i16 == 0) reinterpret_cast<float*>(mgDl(buf))[(w * KT + j) * 4 + g] = s;
        }
    };
    // dQ^T tiles (dt, qt0 + j) of chunk c over the block's 256 keys: A[m = d][k = key] from the K block, B[k = key][n = q] from the dS^T image of
    // that chunk -- both transposed gathers of row-major images
    float* part0 = dq_part + (((size_t)bh * nq_all) * 16 + (dt + 4 * qt0)) * 256 + l * 4;      // + c * 4096 per chunk, + j * 1024 per q-tile
    auto dq_tiles = [&](int c) {
        const int ib = c & 1;
        f32x4 dq[KT];
        f32x4 p4[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (kb > 0 && !MG_DBG(2)) {
                const float* pp = part0 + (size_t)c * 4096 + j * 1024;
#pragma unroll
                for (int e = 0; e < 4; ++e) p4[j][e] = __hip_atomic_load(pp + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // k slots of this product: (g, j < 4) <-> key ks*32 + g*4 + j, (g, j >= 4) <-> key ks*32 + 16 + g*4 + (j - 4) -- the row pattern of the dV / dK
            // gathers: a 32-lane group touches 8 CONSECUTIVE rows per gather (conflict free, tile64.h; rows g*8 .. had 2-way conflicts on every gather)
            const int r0 = (ks & 1) * 32 + g * 4;
            const bf16x8 fa = at_frag_tr(mgK(ks >> 1), r0, r0 + 16, dt * 16, l);
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const bf16x8 fb = at_frag_tr(mgEX(ib, ks >> 1), r0, r0 + 16, (qt0 + j) * 16, l);
                dq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, dq[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if (kb > 0 && !MG_DBG(2)) dq[j] += p4[j];
            if (MG_DBG(2)) { if (dq[j][0] == 12345.f) a.dqkv[0] = 1; }
            else if (!last) {
                float* pp = part0 + (size_t)c * 4096 + j * 1024;
#pragma unroll
                for (int e = 0; e < 4; ++e) __hip_atomic_store(pp + e, dq[j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                uint2 pq;
                pq.x = pack2bf(dq[j][0] * a.scale, dq[j][1] * a.scale); pq.y = pack2bf(dq[j][2] * a.scale, dq[j][3] * a.scale);
                *reinterpret_cast<uint2*>(a.dqkv + (tok0 + c * CH + (qt0 + j) * 16 + i16) * a.H3 + h * HD + dt * 16 + g * 4) = pq;
            }
        }
    };

    const int key0 = kb * MG_KB + w * KT * 16 + i16;                    // this lane's key row of key tile 0 (+ 16 per further tile)
    // K block -> LDS (row-major [256 keys][64 d] as four 64 x 64 tiles, 32 / NW 1-KiB pieces per wave); the first chunk; delta of chunk 0
    {
        const bf16_t* kblk = a.qkv + (tok0 + (size_t)kb * MG_KB) * a.H3 + H + h * HD;
#pragma unroll
        for (int q = 0; q < 2 * KT; ++q) {
            const int p = w * 2 * KT + q, t = p >> 3, R0 = (p & 7) * 8, r = R0 + (l >> 3), cc = (l & 7) ^ swz(r);
            at_glds16(kblk + (size_t)(t * 64 + r) * a.H3 + cc * 8, mgK(t) + R0 * 128);
        }
    }
    stage_chunk(0, 0);
    delta_load(0);
    // K, V rows of this lane's keys as B operands (B[k = d][n = key]); the softmax scale folded into K (exact: a power of two)
    bf16x8 fk[KT][2], fv[KT][2];
    float mbs[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const bf16_t* kp = a.qkv + (tok0 + key0 + kt * 16) * a.H3 + H + h * HD;
        fk[kt][0] = frag_scale(*reinterpret_cast<const bf16x8*>(kp + g * 8), a.scale);
        fk[kt][1] = frag_scale(*reinterpret_cast<const bf16x8*>(kp + 32 + g * 8), a.scale);
        const bf16_t* vp = kp + H;
        fv[kt][0] = *reinterpret_cast<const bf16x8*>(vp + g * 8);
        fv[kt][1] = *reinterpret_cast<const bf16x8*>(vp + 32 + g * 8);
        mbs[kt] = a.mask_bias[tok0 + key0 + kt * 16];
    }
    f32x4 dk[KT][4], dv[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int d = 0; d < 4; ++d) { dk[kt][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[kt][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const size_t kcell0 = ((size_t)bh * (a.L / 16) + (size_t)kb * 16 + (size_t)w * KT) * (a.L / CH);     // + kt * (L / CH) per key tile, + chunk
    delta_put(0);                                                       // (the loads above are waited for by the compiler)
    if (nqv > 1) delta_load(1);
    if (kb > 0 && !MG_DBG(128)) {                                  // the previous block's partial of this pair must be complete before the first read
        if (tid == 0) while (__hip_atomic_load(flags + kb - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token) __builtin_amdgcn_s_sleep(8);
        // (the barrier at the top of step 0 orders every wave behind it)
    }
    if (tl && tid == 0) tl[1] = wall_clock64();
    // lane term of the dS^T slab address: row (ktile & 3) * 16 + i16 of image tile ktile >> 2; 16-B chunk (qf * 2 + (g >> 1)) ^ swz(row), 8-B half g & 1
    uint32_t ex_off[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int ktile = w * KT + kt, row = (ktile & 3) * 16 + i16;
        ex_off[kt] = (uint32_t)((ktile >> 2) * 8192 + row * 128 + (((g >> 1) ^ swz(row)) << 4) + (g & 1) * 8);       // qf enters as ^ (qf << 5): swz is even
    }

    for (int c = 0; c < (MG_DBG(64) ? 0 : nqv); ++c) {
        const int cur = c & 1;
        // everything this step reads has landed -- except that the youngest operations of step c - 1 are the stores of chunk c - 2's dQ tiles
        // (gfx950 retires vector memory operations in issue order, tools/ubench/vmcnt_order.cpp): those stay in flight
        if (c < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (last) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * KT) : "memory");
        __syncthreads();                                                // ONE barrier per step: chunk c (at c == 0: the K block) landed, step c - 1 -- the dS^T image of chunk c - 1 included -- is over everywhere
        if (c + 1 < nqv) {
            if (!MG_DBG(8)) stage_chunk(c + 1, cur ^ 1);
            delta_put(cur ^ 1);                                         // delta of chunk c + 1 from the registers loaded one step ago ...
            if (c + 2 < nqv && !MG_DBG(4)) delta_load(c + 2);      // ... and the rows of chunk c + 2
        }
        const char* tQ = mgQ(cur);
        const char* tO = mgO(cur);
        const f32x2 sc2v = {sc2, sc2}, ikv = {ikeep, ikeep};
#ifdef MG_HF_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int hf = 0; hf < 2; ++hf) {                                // two row fragments (32 queries) at a time
            KeepHalf kw[KT];
            if (KM) {                                                   // this half's lane masks: scalar loads issued beside the fragment reads below
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) km_load_half(kw[kt], a.keepB, kcell0 + (size_t)kt * (a.L / CH) + c, hf);
            }
            f32x4 sacc[KT][2], pacc[KT][2];
            float4 dlc[2];
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2) {
                const float4 l4 = *reinterpret_cast<const float4*>(mgLse(cur) + ((hf * 2 + f2) * 16 + g * 4) * 4);
                dlc[f2] = *reinterpret_cast<const float4*>(mgDl(cur) + ((hf * 2 + f2) * 16 + g * 4) * 4);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const f32x2 m2 = {mbs[kt], mbs[kt]};
                    const f32x2 lo2 = m2 - (f32x2){l4.x, l4.y}, hi2 = m2 - (f32x2){l4.z, l4.w};          // accumulator start: key mask - row LSE (packed)
                    sacc[kt][f2] = (f32x4){lo2.x, lo2.y, hi2.x, hi2.y};
                    pacc[kt][f2] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2) {
                    const bf16x8 fqa = at_frag(tQ, (hf * 2 + f2) * 16 + i16, kk * 4 + g);
                    const bf16x8 foa = at_frag(tO, (hf * 2 + f2) * 16 + i16, kk * 4 + g);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        sacc[kt][f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqa, fk[kt][kk], sacc[kt][f2], 0, 0, 0);
                        pacc[kt][f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(foa, fv[kt][kk], pacc[kt][f2], 0, 0, 0);
                    }
                }
            bf16x8 fp[KT], fds[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 pd[2], ds[2];                                     // P_drop[q][key], dS[q][key]: lane key = i16, q = qf*16 + g*4 + r
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2) {
                    const float dl[4] = {dlc[f2].x, dlc[f2].y, dlc[f2].z, dlc[f2].w};
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        if (MG_DBG(16)) { pd[f2][rp * 2] = sacc[kt][f2][rp * 2]; pd[f2][rp * 2 + 1] = sacc[kt][f2][rp * 2 + 1]; ds[f2][rp * 2] = pacc[kt][f2][rp * 2]; ds[f2][rp * 2 + 1] = pacc[kt][f2][rp * 2 + 1]; continue; }
                        const f32x2 t = (f32x2){sacc[kt][f2][rp * 2], sacc[kt][f2][rp * 2 + 1]} * sc2v;
                        const f32x2 pe = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                        f32x2 pk = pe;
                        f32x2 dp = {pacc[kt][f2][rp * 2], pacc[kt][f2][rp * 2 + 1]};
                        if (KM) {
                            const uint64_t m0 = kw[kt].m[f2 * 4 + rp * 2], m1 = kw[kt].m[f2 * 4 + rp * 2 + 1];
                            pk.x = km_sel(pe.x, m0); pk.y = km_sel(pe.y, m1);
                            dp.x = km_sel(dp.x, m0); dp.y = km_sel(dp.y, m1);
                        }
                        const f32x2 d2 = pe * (dp * ikv - (f32x2){dl[rp * 2], dl[rp * 2 + 1]});
                        pd[f2][rp * 2] = pk.x; pd[f2][rp * 2 + 1] = pk.y;
                        ds[f2][rp * 2] = d2.x; ds[f2][rp * 2 + 1] = d2.y;
                    }
                }
                fp[kt] = pack8(pd[0], pd[1]);
                fds[kt] = pack8(ds[0], ds[1]);
                // dS^T slabs of the two row fragments: 4 consecutive q (8 B) at row = key, column qf*16 + g*4 -- the words just packed
                {
                    union { bf16x8 v; uint2 u[2]; } x;
                    x.v = fds[kt];
                    char* ex = smem + MG_OFF_EX + cur * 32768;
                    *reinterpret_cast<uint2*>(ex + (ex_off[kt] ^ (uint32_t)((hf * 2) << 5))) = x.u[0];
                    *reinterpret_cast<uint2*>(ex + (ex_off[kt] ^ (uint32_t)((hf * 2 + 1) << 5))) = x.u[1];
                }
            }
            // dV^T[d][key] += dO^T[d][q] P_drop[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]   (k = the 32 queries of this half)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 fo = at_frag_tr(tO, (2 * hf) * 16 + g * 4, (2 * hf + 1) * 16 + g * 4, d * 16, l);
                const bf16x8 fqt = at_frag_tr(tQ, (2 * hf) * 16 + g * 4, (2 * hf + 1) * 16 + g * 4, d * 16, l);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    dv[kt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo, fp[kt], dv[kt][d], 0, 0, 0);
                    dk[kt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, fds[kt], dk[kt][d], 0, 0, 0);
                }
            }
#ifndef MG_NO_SCHED
            __builtin_amdgcn_sched_barrier(0);                          // (keeps the two halves' transients from being live together)
#endif
        }
        // dQ of the PREVIOUS chunk: its image was completed by the barrier at the top of this step; independent of everything above, so the
        // latency-bound gather chain of one wave runs beside the other waves' elementwise work instead of in a phase of its own
        if (c > 0 && !MG_DBG(1)) dq_tiles(c - 1);
    }
    __syncthreads();
    if (!MG_DBG(1)) dq_tiles(nqv - 1);
    // dK, dV rows of this lane's keys
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        bf16_t* okp = a.dqkv + (tok0 + key0 + kt * 16) * a.H3 + H + h * HD;
        bf16_t* ovp = okp + H;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint2 pk, pv;
            pk.x = pack2bf(dk[kt][d][0] * a.scale, dk[kt][d][1] * a.scale); pk.y = pack2bf(dk[kt][d][2] * a.scale, dk[kt][d][3] * a.scale);
            *reinterpret_cast<uint2*>(okp + d * 16 + g * 4) = pk;
            pv.x = pack2bf(dv[kt][d][0] * ikeep, dv[kt][d][1] * ikeep); pv.y = pack2bf(dv[kt][d][2] * ikeep, dv[kt][d][3] * ikeep);
            *reinterpret_cast<uint2*>(ovp + d * 16 + g * 4) = pv;
        }
    }
    if (!last) {                                                        // publish this block's partial: every wave's stores are complete, then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + kb, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
    if (tl && tid == 0) tl[2] = wall_clock64();
    // ---- generation bookkeeping: the last workgroup of the launch resets the counter and moves the generation on (no host-side state, no memset per launch)
    if (tid == 0) {
        const uint32_t done = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// fp32 partials [B * heads * L * 64] + the sync words (generation, finished count, one flag per (pair, key block)), rounded to 256 B
static size_t mg_part_floats(int B, int L, int heads) { return (size_t)B * heads * (size_t)L * HD; }
size_t amdseg_attn_bwd_merged_scratch_bytes_impl(int B, int L, int heads) {
    const size_t sync_words = 2 + (size_t)B * heads * (size_t)(L / MG_KB > 0 ? L / MG_KB : 1);
#ifdef AMDSEG_PROBES          // + the timeline probe's records behind 4096 flag words
    return mg_part_floats(B, L, heads) * sizeof(float) + (2 + 4096) * 4 + (sync_words - 2) * 32 + 256;
#else
    return mg_part_floats(B, L, heads) * sizeof(float) + ((sync_words * 4 + 255) / 256) * 256;
#endif
}

bool amdseg_attn_bwd_merged_ok(int L, float p, const void* keep) { return L % MG_KB == 0 && L >= MG_KB && (p == 0.f || keep != nullptr); }

int amdseg_attn_bwd_merged_impl(const void* qkv, const float* mask_bias, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                                void* dq_part, int B, int L, int heads, float scale, float p, hipStream_t s, const int* kend, const int* seq_order,
                                const int* qguard, const void* keep) {
    if (!qkv || !mask_bias || !ctx || !dctx || !lse || !dqkv || !dq_part) return AMDSEG_ERR_ARG;
    if (L % MG_KB || L < MG_KB) return AMDSEG_ERR_SHAPE;
    if (p > 0.f && !keep) return AMDSEG_ERR_ARG;             // dropout decisions come from the layer's keep masks (attn_keepmask_kernel)
    AttnArgs a = {};
    a.B = B; a.L = L; a.heads = heads; a.H3 = 3 * heads * HD; a.scale = scale; a.seed = 0;
    uint32_t th = (uint32_t)(p * 65536.0f + 0.5f);
    if (p > 0.f && th == 0) th = 1;
    a.thresh16 = th;
    a.inv_keep = th ? 65536.0f / (float)(65536u - th) : 1.0f;
    a.kend = kend; a.seq_order = kend ? seq_order : nullptr; a.qguard = kend ? qguard : nullptr;
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = (float*)lse;
    a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
    a.keepA = (const uint64_t*)keep; a.keepB = a.keepA ? a.keepA + (size_t)B * heads * L * (size_t)L / 64 : nullptr;
    static bool attr_done = false;
    if (!attr_done) {
        const void* ks[4] = {reinterpret_cast<const void*>(&attn_bwd_merged_kernel<true, 1>), reinterpret_cast<const void*>(&attn_bwd_merged_kernel<false, 1>),
                             reinterpret_cast<const void*>(&attn_bwd_merged_kernel<true, 2>), reinterpret_cast<const void*>(&attn_bwd_merged_kernel<false, 2>)};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, MG_LDS);
            if (e != hipSuccess) return (int)e;
        }
        attr_done = true;
    }
#ifdef AMDSEG_PROBES
    { const char* e = getenv("AMDSEG_MG_DEBUG"); a.skip_q = e ? atoi(e) : 0; }      // 1 no dQ product, 2 no partial / dQ traffic, 4 no delta loads, 8 no chunk DMA, 16 no elementwise, 32 empty kernel, 64 no loop, 128 no partner wait, 256 timeline
#endif
    int kt_sel = 2;                                          // AMDSEG_ATTN_MERGED_KT=1: the 16-wave form (default: 8 waves x 32 keys); read per call
    { const char* e = getenv("AMDSEG_ATTN_MERGED_KT"); if (e && atoi(e) == 1) kt_sel = 1; }
    const double work = 10.0 * B * heads * (double)L * (double)L * HD;      // 5 products (S, dP, dV, dK, dQ): the algorithmic backward
    float* part = (float*)dq_part;
    uint32_t* sync = reinterpret_cast<uint32_t*>(part + mg_part_floats(B, L, heads));
    const dim3 grid((unsigned)(B * heads * (L / MG_KB)));
    if (kt_sel == 1) {
        if (th) AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, work, (attn_bwd_merged_kernel<true, 1>), grid, dim3(1024), MG_LDS, s, a, part, sync);
        else AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, work, (attn_bwd_merged_kernel<false, 1>), grid, dim3(1024), MG_LDS, s, a, part, sync);
    } else {
        if (th) AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, work, (attn_bwd_merged_kernel<true, 2>), grid, dim3(512), MG_LDS, s, a, part, sync);
        else AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ATTN_BWD_DKV, work, (attn_bwd_merged_kernel<false, 2>), grid, dim3(512), MG_LDS, s, a, part, sync);
    }
    return amdseg_launch_status();
}
