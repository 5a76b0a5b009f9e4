// Generator of the attention-dropout keep masks (layouts and method: the comment block "dropout keep masks" at the top of attention.hip).
// Its own header because two launches run it: attn_keepmask_kernel (attention.hip) alone, and add_ln_fwd_km_kernel (elementwise.hip) next to the
// LayerNorm rows of the layer in front -- the generator is VALU-bound, the row kernel HBM-bound, so one grid of interleaved workgroups hides most of it.
#pragma once
#include "common.h"
#include "tile64.h"

__device__ __forceinline__ uint32_t km_xs32(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
__device__ __forceinline__ uint64_t km_delta_swap(uint64_t v, uint64_t m, int d) {   // swaps the bits selected by m with the bits d above them
    const uint64_t t = ((v >> d) ^ v) & m;
    return v ^ t ^ (t << d);
}
// lane <-> word exchange of index bit j (j = 1, 2) of a 32-bit half: new[X][p] = old[X with bit j := p_j][p with bit j := X_j]
template <int J>
__device__ __forceinline__ uint32_t km_exchange(uint32_t x, bool upper) {
    constexpr uint32_t M = J == 1 ? 0x55555555u : 0x33333333u;                      // positions whose bit j is clear
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, J == 1 ? 0xB1 : 0x4E, 0xf, 0xf, true);   // lane ^ j (quad_perm)
    return upper ? ((x & ~M) | ((y >> J) & M)) : ((x & M) | ((y << J) & ~M));
}

struct KeepMaskArgs {
    uint64_t* A; uint64_t* Bm;
    int B, L, heads; uint32_t thresh16; uint64_t seed;
    const int* kend;                        // optional [B]: chunks past the last unmasked key are never read by the consumers (attn_visible_chunks)
    int window, nglobal;                    // band attention (window > 0): only the (query block, key chunk) cells its kernels visit are generated
    uint32_t tn[16];                        // per threshold bit i: 0 if bit i of thresh16 is set, ~0 otherwise (host-filled, see the kernel)
};

// one wave per (bh, 64-query block, group of KM_CG key chunks)
#define KM_CG 4
// (the body as a device function of the BLOCK index: attn_keepmask_kernel runs it alone; add_ln_fwd_km_kernel of elementwise.hip runs it in the same
//  launch as the LayerNorm of the layer in front -- round 6, "horizontal fusion" of a VALU-bound generator with an HBM-bound row kernel)
__device__ __forceinline__ void km_block_body(const KeepMaskArgs& a, int block) {
    const int l = threadIdx.x & 63;
    const int nblk = a.L / CH, ngrp = (nblk + KM_CG - 1) / KM_CG;
    const int wid = block * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the index arithmetic below is scalar
    const int total = a.B * a.heads * nblk * ngrp;
    if (wid >= total) return;
    const int grp = wid % ngrp, qblk = (wid / ngrp) % nblk, bh = wid / (ngrp * nblk);
    int nvis = nblk;
    if (a.kend) { const int ke = a.kend[bh / a.heads]; if (ke > 0) nvis = min(nblk, (ke + CH - 1) / CH); }
    // two xorshift32 streams per lane (the low and the high half of the lane's 64 bits), seeded by a strong hash of (seed, wave, lane)
    uint32_t xl = mix32((uint32_t)a.seed ^ mix32((uint32_t)(a.seed >> 32) + 0x9e3779b9u + (uint32_t)wid * 128u + (uint32_t)l));
    uint32_t xh = mix32(xl ^ (0x85ebca6bu + (uint32_t)l));
    xl |= xl == 0; xh |= xh == 0;
    const int qa = l >> 4, ka = (l >> 2) & 3, c2 = l & 3;             // A: lane = (qa, ka, kc); B after the exchange: lane = (qa, ka, qc)
    const size_t rows16 = (size_t)a.L / 16;
    for (int cc = 0; cc < KM_CG; ++cc) {
        const int chunk = grp * KM_CG + cc;
        if (chunk >= nvis) break;
        if (a.window > 0) {                                 // the band around the query block, + chunk 0 when it holds global keys
            const int q_lo = qblk * CH, q_hi = q_lo + CH - 1;
            const bool in_band = chunk * CH + CH - 1 >= q_lo - a.window && chunk * CH <= q_hi + a.window;
            if (!in_band && !(a.nglobal > 0 && chunk == 0)) continue;
        }
        // keep <=> u >= thresh16 for a uniform 16-bit u, evaluated bit-serially from the lowest set bit of the threshold upwards on 64 lanes x
        // 64 independent u's at once: ge = t_i ? (u_i & ge) : (u_i | ge)
        // = majority(u_i, ge, tn_i) with tn_i = t_i ? 0 : ~0 -- ONE v_bitop3 per half and round.  tn comes from the kernel arguments (SGPRs):
        // written as a select on the threshold bit the optimiser turns it back into and + or + v_cndmask (3 instructions).  Rounds below the
        // lowest set bit of the threshold leave ge at all ones, so all 16 rounds run unconditionally
        uint32_t lo = 0xffffffffu, hi = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            xl = km_xs32(xl); xh = km_xs32(xh);
            lo = (xl & lo) | (a.tn[i] & (xl | lo));
            hi = (xh & hi) | (a.tn[i] & (xh | hi));
        }
        // A: lane (qa, ka, kc) holds word (w = qa, fc = ka, r = kc) with bits (g = kb, i16 = (qb, qc))
        a.A[(((size_t)bh * rows16 + (size_t)qblk * 4 + qa) * nblk + chunk) * 16 + ka * 4 + c2] = ((uint64_t)hi << 32) | lo;
        // exchange the lane digit kc with the word digit qc ...
        lo = km_exchange<1>(lo, l & 1); hi = km_exchange<1>(hi, l & 1);
        lo = km_exchange<2>(lo, l & 2); hi = km_exchange<2>(hi, l & 2);
        // ... and swap the two upper 2-bit digits of the bit index, (kb, qb, kc) -> (qb, kb, kc): a 4 x 4 transpose of the word's nibbles
        uint64_t v = ((uint64_t)hi << 32) | lo;
        v = km_delta_swap(v, 0x0000F0F00000F0F0ull, 12);
        v = km_delta_swap(v, 0x00000000FF00FF00ull, 24);
        // B: lane (qa, ka, qc) holds word (w' = ka, qf = qa, r' = qc) with bits (g' = qb, i16' = (kb, kc))
        a.Bm[(((size_t)bh * rows16 + (size_t)chunk * 4 + ka) * nblk + qblk) * 16 + qa * 4 + c2] = v;
    }
}


// blocks of 256 threads the generator needs for one layer
static inline long km_blocks(int B, int L, int heads) {
    const int nblk = L / CH, ngrp = (nblk + KM_CG - 1) / KM_CG;
    return ((long)B * heads * nblk * ngrp + 3) / 4;
}


// amdseg_bert_layer_fwd pairs the generator with the second LayerNorm of the layer in front only where that was measured to pay: the generator's
// blocks at most half the row blocks (bert-base L = 512: 0.375 -> 39.0 us as two launches, 32.3 as one; 12 heads at L = 1024: 0.75 -> 27.5 either way;
// 16 heads x 1024 x 1024 against H = 1024 rows: 1.0 -> 60.7 apart, 70.6 together -- profiles/r06_ln_keepmask_pairing.md).  Both layers of a pair
// evaluate this on the same cfg, so "the layer in front wrote my masks" and "I write the next layer's masks" always agree.
static inline bool km_pairs_with_rows(int B, int L, int heads, int M) { return km_blocks(B, L, heads) * 2 <= ((long)M + 3) / 4; }

// host side: thresh16 as attn_fill (attention.hip) rounds it, the per-bit words of the comparison, layout B behind layout A
static inline int km_fill(KeepMaskArgs& k, void* keep, int B, int L, int heads, float p, uint64_t seed, const int* kend, int window, int nglobal) {
    if (!keep) return AMDSEG_ERR_ARG;
    if (B <= 0 || L <= 0 || heads <= 0 || (L % CH)) return AMDSEG_ERR_SHAPE;
    if (p < 0.f || p >= 1.f || window < 0 || nglobal < 0 || nglobal > CH) return AMDSEG_ERR_ARG;
    uint32_t th = (uint32_t)(p * 65536.0f + 0.5f);
    if (p > 0.f && th == 0) th = 1;
    k = KeepMaskArgs{};
    k.A = (uint64_t*)keep; k.Bm = k.A + (size_t)B * heads * L * (size_t)L / 64;
    k.B = B; k.L = L; k.heads = heads; k.thresh16 = th; k.seed = seed; k.kend = window > 0 ? nullptr : kend;
    k.window = window; k.nglobal = window > 0 ? nglobal : 0;
    for (int i = 0; i < 16; ++i) k.tn[i] = ((th >> i) & 1) ? 0u : 0xffffffffu;
    return AMDSEG_OK;
}
