// PoNet token mixing for gfx950 (alimeeting4mug/src/models/modeling_ponet.py:68-79 -> modelscope PoNetModel, NOT in the
// reference tree: semantics restated in oracle/ponet_oracle.py, parity unpinned).  HBM-bound row kernels over the fused
// projection output  proj [M, 5H] bf16 = (Hq | Hk | Ho | Hl | Hs):
//
//   segment max-pooling  S_n = max of Hs over the valid tokens of n's segment (a contiguous run of equal segment_ids),
//   local max-pooling    L_n = max of Hl over the valid tokens n-1, n, n+1,
//   fusion               ctx_n = (g + S_n) * Ho_n + L_n        (g = global-attention aggregate, [B, H] fp32, see host mirror)
// and their backward.  One wave per token row (or per 8 consecutive rows), 16 B per lane (the LayerNorm recipe).
// Round 2 layout of the work (round 1: five-launch trees whose inactive waves -- 7 of 8 -- still held 130+ VGPRs each while they loaded
// three words and exited; the forward ran at 0.23 of HBM):
//   plan      (once per batch)  two compact work lists built with an atomic counter each: LEVEL-A leaders = valid tokens with
//                               (pos - run_start) % 8 == 0, and RUN leaders = valid tokens with pos == run_start;
//   forward   pn_a_max  : persistent waves over the level-A list, 8 Hs rows in flight each -> partial max / argmax rows (plane A)
//             pn_r_max  : persistent waves over the run list, fold the run's plane-A rows (8 in flight) -> S / argmax row of the run
//             pn_combine: every token: ctx = (g + S) * Ho + max(Hl[n-1..n+1])
//   backward  pn_bwd_tok: one wave per 8 CONSECUTIVE tokens, Hl and dctx rows kept in a sliding register window (3.75 row loads per
//                         token instead of 10); writes dHo, dHl, zeroes padded rows, and accumulates E = dctx * Ho per run SEGMENT
//                         (flushed at run boundaries and at the end of the group) -- E itself is never materialised;
//             pn_r_sum  : run leaders fold their segment sums -> run total G (and add it into dg, the gradient of the global aggregate)
//             pn_route  : dHs_j = [argmax of j's run == j] * G.
// Algorithmic bytes per token: forward read Hs, Ho, Hl + write ctx = 4 * H * 2 B (25.2 MB per 4096-token sequence and layer);
// backward read dctx, Ho, Hl + write dHo, dHl, dHs = 6 * H * 2 B.
#include <algorithm>
#include "common.h"
#include "amdseg_internal.h"

#define PN_MAXCH 2                // 8-column chunks per lane: H <= 1024
#define PN_SUB 64

struct PnArgs {
    const bf16_t* proj; int ld;                // fused projections, row stride (5H)
    const float* mask_bias;                    // [M], < 0 = padded token
    const int* run_start; const int* run_end;  // [M] index (within the sequence) of the first / last token of the token's run
    bf16_t* part; unsigned short* parg;        // [M, H] sub-leader rows: partial max of Hs and its argmax token index
    bf16_t* part2; unsigned short* parg2;      // [M, H] run-leader rows: the folded run maximum / argmax
    float* psum2;                              // [M, H] run-leader rows: the folded run sum of E
    bf16_t* partA; unsigned short* pargA; float* psumA;    // [M, H] level-A rows (fan-in 8) of the two trees
    const float* g;                            // [B, H]
    bf16_t* ctx;                               // [M, H]
    const bf16_t* dctx;                        // [M, H]
    bf16_t* dproj;                             // [M, 5H] gradient of the projections (Ho, Hl, Hs columns written here)
    bf16_t* E;                                 // unused (round 1: [M, H] dctx * Ho)
    float* psum;                               // [M, H] segment rows: partial sums of E = dctx * Ho
    float* dg;                                 // [B, H] gradient of the global aggregate g: sum of E over the valid tokens of a sequence
    const int* work;                           // plan: [0] #level-A leaders, [1] #run leaders, [2 .. 2+M) level-A list, [2+M .. 2+2M) run list
    int M, L, H;
};

template <int NCH>
__device__ __forceinline__ void pn_load(const bf16_t* row, int nch, int l, float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) ld8<bf16_t>(row + c * 8, v[i]);
    }
}
template <int NCH>
__device__ __forceinline__ void pn_store(bf16_t* row, int nch, int l, const float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) st8<bf16_t>(row + c * 8, v[i]);
    }
}
template <int NCH>
__device__ __forceinline__ void pn_fill(float (&v)[NCH][8], float x) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = x;
}

// ---------------------------------------------------------------------------------------------------- plan
__global__ __launch_bounds__(256) void pn_plan_kernel(const float* __restrict__ mask_bias, const int* __restrict__ run_start, int* work, int M, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= M || mask_bias[n] < 0.f) return;
    const int pos = n % L, rs = run_start[n];
    if (((pos - rs) & 7) == 0) work[2 + atomicAdd(work, 1)] = n;
    if (pos == rs) work[2 + M + atomicAdd(work + 1, 1)] = n;
}

// ---------------------------------------------------------------------------------------------------- level A: max of 8 Hs rows
__global__ __launch_bounds__(256) void pn_a_max_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4, count = a.work[0], nch = a.H >> 3;
    for (int item = blockIdx.x * 4 + w; item < count; item += nwaves) {
        const int n = a.work[2 + item];
        const int b = n / a.L, pos = n - b * a.L, re = a.run_end[n];
        float v[8][PN_MAXCH][8]; bool ok[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t row = (size_t)b * a.L + min(pos + k, re);
            ok[k] = pos + k <= re && a.mask_bias[row] >= 0.f;
            pn_load<PN_MAXCH>(a.proj + row * a.ld + 4 * a.H, nch, l, v[k]);
        }
        float mx[PN_MAXCH][8]; unsigned short arg[PN_MAXCH][8];
        pn_fill<PN_MAXCH>(mx, -INFINITY);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) arg[i][e] = (unsigned short)pos;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!ok[k]) continue;
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[k][i][e] > mx[i][e]) { mx[i][e] = v[k][i][e]; arg[i][e] = (unsigned short)(pos + k); }     // first maximum wins
        }
        pn_store<PN_MAXCH>(a.partA + (size_t)n * a.H, nch, l, mx);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                uint4 pk;
                pk.x = arg[i][0] | ((uint32_t)arg[i][1] << 16); pk.y = arg[i][2] | ((uint32_t)arg[i][3] << 16);
                pk.z = arg[i][4] | ((uint32_t)arg[i][5] << 16); pk.w = arg[i][6] | ((uint32_t)arg[i][7] << 16);
                *reinterpret_cast<uint4*>(a.pargA + (size_t)n * a.H + c * 8) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- run level: fold the plane-A rows
// run leader n folds rows run_start, run_start + 8, ... <= run_end of plane A (8 rows in flight) into part2 / parg2 at its own row
__global__ __launch_bounds__(256) void pn_r_max_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4, count = a.work[1], nch = a.H >> 3;
    for (int item = blockIdx.x * 4 + w; item < count; item += nwaves) {
        const int n = a.work[2 + a.M + item];
        const int b = n / a.L, rs = n - b * a.L, re = a.run_end[n];
        float S[PN_MAXCH][8]; unsigned short arg[PN_MAXCH][8];
        pn_fill<PN_MAXCH>(S, -INFINITY);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) arg[i][e] = (unsigned short)rs;
        for (int k0 = rs; k0 <= re; k0 += 64) {
            float v[8][PN_MAXCH][8]; uint4 pa[8][PN_MAXCH]; bool ok[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = k0 + 8 * k;
                const size_t row = (size_t)b * a.L + min(t, re);
                ok[k] = t <= re && a.mask_bias[row] >= 0.f;              // (a padded level-A position was never written)
                pn_load<PN_MAXCH>(a.partA + row * a.H, nch, l, v[k]);
#pragma unroll
                for (int i = 0; i < PN_MAXCH; ++i)
                    if (l + 64 * i < nch) pa[k][i] = *reinterpret_cast<const uint4*>(a.pargA + row * a.H + (l + 64 * i) * 8);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!ok[k]) continue;
#pragma unroll
                for (int i = 0; i < PN_MAXCH; ++i) {
                    const uint32_t pw[4] = {pa[k][i].x, pa[k][i].y, pa[k][i].z, pa[k][i].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (v[k][i][e] > S[i][e]) { S[i][e] = v[k][i][e]; arg[i][e] = (unsigned short)((pw[e >> 1] >> ((e & 1) * 16)) & 0xffffu); }
                }
            }
        }
        pn_store<PN_MAXCH>(a.part2 + (size_t)n * a.H, nch, l, S);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                uint4 pk;
                pk.x = arg[i][0] | ((uint32_t)arg[i][1] << 16); pk.y = arg[i][2] | ((uint32_t)arg[i][3] << 16);
                pk.z = arg[i][4] | ((uint32_t)arg[i][5] << 16); pk.w = arg[i][6] | ((uint32_t)arg[i][7] << 16);
                *reinterpret_cast<uint4*>(a.parg2 + (size_t)n * a.H + c * 8) = pk;
            }
        }
    }
}

// single-row read of the folded run maximum (row of the run leader)
template <bool ARG>
__device__ __forceinline__ void pn_run_max(const PnArgs& a, int b, int rs, int re, int nch, int l, float (&S)[PN_MAXCH][8],
                                           unsigned short (&arg)[PN_MAXCH][8]) {
    const size_t row = (size_t)b * a.L + rs;
    pn_load<PN_MAXCH>(a.part2 + row * a.H, nch, l, S);
    if (ARG) {
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const uint4 pk = *reinterpret_cast<const uint4*>(a.parg2 + row * a.H + c * 8);
                const uint32_t pw[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) arg[i][e] = (unsigned short)((pw[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward fusion
__global__ __launch_bounds__(256) void pn_combine_fwd_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    if (n >= a.M) return;
    const int nch = a.H >> 3;
    float out[PN_MAXCH][8];
    if (a.mask_bias[n] < 0.f) {
        pn_fill<PN_MAXCH>(out, 0.f);
        pn_store<PN_MAXCH>(a.ctx + (size_t)n * a.H, nch, l, out);
        return;
    }
    const int b = n / a.L, pos = n - b * a.L;
    float S[PN_MAXCH][8]; unsigned short dummy[PN_MAXCH][8];
    pn_run_max<false>(a, b, a.run_start[n], a.run_end[n], nch, l, S, dummy);
    const bf16_t* prow = a.proj + (size_t)n * a.ld;
    float ho[PN_MAXCH][8], lm[PN_MAXCH][8], t[PN_MAXCH][8];
    pn_load<PN_MAXCH>(prow + 2 * a.H, nch, l, ho);
    pn_load<PN_MAXCH>(prow + 3 * a.H, nch, l, lm);
    if (pos > 0 && a.mask_bias[n - 1] >= 0.f) {
        pn_load<PN_MAXCH>(prow - a.ld + 3 * a.H, nch, l, t);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) lm[i][e] = fmaxf(lm[i][e], t[i][e]);
    }
    if (pos + 1 < a.L && a.mask_bias[n + 1] >= 0.f) {
        pn_load<PN_MAXCH>(prow + a.ld + 3 * a.H, nch, l, t);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) lm[i][e] = fmaxf(lm[i][e], t[i][e]);
    }
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            float gv[8];
            ld8<float>(a.g + (size_t)b * a.H + c * 8, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) out[i][e] = (gv[e] + S[i][e]) * ho[i][e] + lm[i][e];
        }
    }
    pn_store<PN_MAXCH>(a.ctx + (size_t)n * a.H, nch, l, out);
}

// ---------------------------------------------------------------------------------------------------- backward, 8 consecutive tokens per wave
// dHo = dctx * (g + S);  dHl_j = sum over the valid neighbours n of j (incl. j) of dctx_n * [first maximum of {Hl_{n-1}, Hl_n, Hl_{n+1}} == j];
// E = dctx * Ho accumulated per run segment into psum rows (segment = maximal piece of a run inside this group of 8; its row = its first token)
__global__ __launch_bounds__(256) void pn_bwd_tok_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int t0 = (blockIdx.x * 4 + w) * 8;                  // first token of the group (L % 8 == 0: a group never straddles sequences)
    if (t0 >= a.M) return;
    const int nch = a.H >> 3, b = t0 / a.L, p0 = t0 - b * a.L;
    const size_t seq0 = (size_t)b * a.L;
    auto valid_at = [&](int p) { return p >= 0 && p < a.L && a.mask_bias[seq0 + p] >= 0.f; };
    // sliding windows: hl[k] = Hl row p + k - 2 (k = 0..4), dcw[k] = dctx row p + k - 1 (k = 0..2); -inf / 0 outside the valid range
    float hl[5][PN_MAXCH][8], dcw[3][PN_MAXCH][8];
    bool hv[5], dv[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                              // rows p0-2 .. p0+1 (row p0+2 is loaded by the first iteration)
        hv[k + 1] = valid_at(p0 + k - 2);
        if (hv[k + 1]) pn_load<PN_MAXCH>(a.proj + (seq0 + p0 + k - 2) * a.ld + 3 * a.H, nch, l, hl[k + 1]); else pn_fill<PN_MAXCH>(hl[k + 1], -INFINITY);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                              // dctx rows p0-1, p0
        dv[k + 1] = valid_at(p0 + k - 1);
        if (dv[k + 1]) pn_load<PN_MAXCH>(a.dctx + (seq0 + p0 + k - 1) * a.H, nch, l, dcw[k + 1]); else pn_fill<PN_MAXCH>(dcw[k + 1], 0.f);
    }
    float gv[PN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
        if (l + 64 * i < nch) ld8<float>(a.g + (size_t)b * a.H + (l + 64 * i) * 8, gv[i]);
    float seg[PN_MAXCH][8], S[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(seg, 0.f); pn_fill<PN_MAXCH>(S, 0.f);
    int seg_row = -1, cur_rs = -1;                             // open segment's psum row; run whose S is loaded
    // (requesting the next token's rows one iteration ahead was tried: 126 -> 200 us, the three extra row buffers push the wave past 256 VGPRs)
    for (int t = 0; t < 8; ++t) {
        const int p = p0 + t;
        const size_t n = seq0 + p;
        // shift the windows and bring in Hl row p + 2, dctx row p + 1
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hv[k] = hv[k + 1];
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) hl[k][i][e] = hl[k + 1][i][e];
        }
        hv[4] = valid_at(p + 2);
        if (hv[4]) pn_load<PN_MAXCH>(a.proj + (n + 2) * a.ld + 3 * a.H, nch, l, hl[4]); else pn_fill<PN_MAXCH>(hl[4], -INFINITY);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            dv[k] = dv[k + 1];
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) dcw[k][i][e] = dcw[k + 1][i][e];
        }
        dv[2] = valid_at(p + 1);
        if (dv[2]) pn_load<PN_MAXCH>(a.dctx + (n + 1) * a.H, nch, l, dcw[2]); else pn_fill<PN_MAXCH>(dcw[2], 0.f);
        float ho[PN_MAXCH][8];
        pn_load<PN_MAXCH>(a.proj + n * a.ld + 2 * a.H, nch, l, ho);
        bf16_t* drow = a.dproj + n * a.ld;
        if (!hv[2]) {                                          // padded token: zero gradients, close the open segment
            float z[PN_MAXCH][8];
            pn_fill<PN_MAXCH>(z, 0.f);
            pn_store<PN_MAXCH>(drow + 2 * a.H, nch, l, z); pn_store<PN_MAXCH>(drow + 3 * a.H, nch, l, z); pn_store<PN_MAXCH>(drow + 4 * a.H, nch, l, z);
            if (seg_row >= 0) {
#pragma unroll
                for (int i = 0; i < PN_MAXCH; ++i)
                    if (l + 64 * i < nch) st8<float>(a.psum + (seq0 + seg_row) * a.H + (l + 64 * i) * 8, seg[i]);
                seg_row = -1;
            }
            continue;
        }
        const int rs = a.run_start[n];
        if (rs != cur_rs) {                                    // new run: flush the open segment, load the run's S row
            if (seg_row >= 0) {
#pragma unroll
                for (int i = 0; i < PN_MAXCH; ++i)
                    if (l + 64 * i < nch) st8<float>(a.psum + (seq0 + seg_row) * a.H + (l + 64 * i) * 8, seg[i]);
            }
            pn_fill<PN_MAXCH>(seg, 0.f);
            seg_row = p; cur_rs = rs;
            pn_load<PN_MAXCH>(a.part2 + (seq0 + rs) * a.H, nch, l, S);
        } else if (seg_row < 0) { pn_fill<PN_MAXCH>(seg, 0.f); seg_row = p; }
        float o1[PN_MAXCH][8], dl[PN_MAXCH][8];
        pn_fill<PN_MAXCH>(dl, 0.f);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = dcw[1][i][e];
                o1[i][e] = d * (gv[i][e] + S[i][e]);
                seg[i][e] += d * ho[i][e];
            }
        // local max-pool backward: neighbour m = p + k - 1 (k = 0..2) has the window hl[k], hl[k+1], hl[k+2]; this token is hl[2]
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!hv[k + 1]) continue;                          // the neighbour itself must be a valid token
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x0 = hl[k][i][e], x1 = hl[k + 1][i][e], x2 = hl[k + 2][i][e];
                    const int am = (x0 >= x1 && x0 >= x2) ? k : ((x1 >= x2) ? k + 1 : k + 2);      // first maximum of the window
                    if (am == 2) dl[i][e] += dcw[k][i][e];
                }
        }
        pn_store<PN_MAXCH>(drow + 2 * a.H, nch, l, o1);
        pn_store<PN_MAXCH>(drow + 3 * a.H, nch, l, dl);
    }
    if (seg_row >= 0) {
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
            if (l + 64 * i < nch) st8<float>(a.psum + (seq0 + seg_row) * a.H + (l + 64 * i) * 8, seg[i]);
    }
}

// run leader: G = sum of the run's segment rows (run_start and every later multiple of 8 up to run_end) -> psum2[leader]; dg[b] += G
__global__ __launch_bounds__(256) void pn_r_sum_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4, count = a.work[1], nch = a.H >> 3;
    for (int item = blockIdx.x * 4 + w; item < count; item += nwaves) {
        const int n = a.work[2 + a.M + item];
        const int b = n / a.L, rs = n - b * a.L, re = a.run_end[n];
        float G[PN_MAXCH][8];
        pn_fill<PN_MAXCH>(G, 0.f);
        for (int k = rs; k <= re; k = (k & ~7) + 8) {
            const size_t row = (size_t)b * a.L + k;
            if (a.mask_bias[row] < 0.f) continue;              // (runs of valid tokens never contain padding; defensive)
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i) {
                const int c = l + 64 * i;
                if (c < nch) {
                    float v[8];
                    ld8<float>(a.psum + row * a.H + c * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) G[i][e] += v[e];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                st8<float>(a.psum2 + (size_t)n * a.H + c * 8, G[i]);
                if (a.dg)
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(a.dg + (size_t)b * a.H + c * 8 + e, G[i][e]);
            }
        }
    }
}

// dHs_j = [argmax of j's run == j] * (sum of E over the run)
__global__ __launch_bounds__(256) void pn_bwd_route_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= a.M || a.mask_bias[j] < 0.f) return;                       // padded rows were zeroed by pn_bwd_token_kernel
    const int b = j / a.L, pos = j - b * a.L, rs = a.run_start[j], re = a.run_end[j], nch = a.H >> 3;
    float S[PN_MAXCH][8]; unsigned short arg[PN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) arg[i][e] = 0xffffu;
    pn_run_max<true>(a, b, rs, re, nch, l, S, arg);
    float G[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(G, 0.f);
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) ld8<float>(a.psum2 + ((size_t)b * a.L + rs) * a.H + c * 8, G[i]);
    }
    float out[PN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) out[i][e] = (arg[i][e] == (unsigned short)pos) ? G[i][e] : 0.f;
    pn_store<PN_MAXCH>(a.dproj + (size_t)j * a.ld + 4 * a.H, nch, l, out);
}

// ---------------------------------------------------------------------------------------------------- launchers
static int pn_check(int B, int L, int H, int ld) {
    if (B <= 0 || L <= 0 || L > 65535 || (L % 8) || H <= 0 || (H % 8) || H > 8 * 64 * PN_MAXCH || ld < 5 * H || (ld % 8)) return AMDSEG_ERR_SHAPE;
    return AMDSEG_OK;
}
#define PN_PERSIST_BLOCKS 512           // 2048 waves = what the chip holds at ~180 VGPRs per wave (2 per SIMD): no second round of launches

int amdseg_ponet_plan_impl(const float* mask_bias, const int* run_start, int* work, int B, int L, hipStream_t s) {
    if (!mask_bias || !run_start || !work) return AMDSEG_ERR_ARG;
    if (B <= 0 || L <= 0) return AMDSEG_ERR_SHAPE;
    hipError_t e = hipMemsetAsync(work, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    const int M = B * L;
    hipLaunchKernelGGL(pn_plan_kernel, dim3((M + 255) / 256), dim3(256), 0, s, mask_bias, run_start, work, M, L);
    return amdseg_launch_status();
}

int amdseg_ponet_pool_fwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                               const float* g, void* part, void* parg, void* ctx, int B, int L, int H, hipStream_t s) {
    // part / parg hold TWO [M, H] planes each: folded run-leader rows (read by every token of the run, and by backward), level-A rows
    if (!proj || !mask_bias || !run_start || !run_end || !work || !g || !part || !parg || !ctx) return AMDSEG_ERR_ARG;
    int rc = pn_check(B, L, H, ld);
    if (rc) return rc;
    PnArgs a = {};
    a.proj = (const bf16_t*)proj; a.ld = ld; a.mask_bias = mask_bias; a.run_start = run_start; a.run_end = run_end; a.g = g; a.work = work;
    a.ctx = (bf16_t*)ctx; a.M = B * L; a.L = L; a.H = H;
    a.part2 = (bf16_t*)part; a.parg2 = (unsigned short*)parg;
    a.partA = a.part2 + (size_t)a.M * H; a.pargA = a.parg2 + (size_t)a.M * H;
    const int pb = std::min(PN_PERSIST_BLOCKS, (a.M + 3) / 4);
    hipLaunchKernelGGL(pn_a_max_kernel, dim3(pb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_r_max_kernel, dim3(pb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_combine_fwd_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    return amdseg_launch_status();
}

int amdseg_ponet_pool_bwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                               const float* g, const void* part, const void* parg, const void* dctx, void* dproj, float* dg, float* psum,
                               int B, int L, int H, hipStream_t s) {
    // psum: two [M, H] fp32 planes (segment rows, folded run-leader rows); dg [B, H] is ZEROED here and receives the per-sequence sum of dctx * Ho
    if (!proj || !mask_bias || !run_start || !run_end || !work || !g || !part || !parg || !dctx || !dproj || !dg || !psum) return AMDSEG_ERR_ARG;
    int rc = pn_check(B, L, H, ld);
    if (rc) return rc;
    PnArgs a = {};
    a.proj = (const bf16_t*)proj; a.ld = ld; a.mask_bias = mask_bias; a.run_start = run_start; a.run_end = run_end; a.g = g; a.work = work;
    a.dctx = (const bf16_t*)dctx; a.dproj = (bf16_t*)dproj; a.psum = psum; a.dg = dg; a.M = B * L; a.L = L; a.H = H;
    a.part2 = (bf16_t*)part; a.parg2 = (unsigned short*)parg; a.psum2 = psum + (size_t)a.M * H;
    hipError_t e = hipMemsetAsync(dg, 0, (size_t)B * H * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int pb = std::min(PN_PERSIST_BLOCKS, (a.M + 3) / 4);
    hipLaunchKernelGGL(pn_bwd_tok_kernel, dim3((a.M / 8 + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_r_sum_kernel, dim3(pb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_bwd_route_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    return amdseg_launch_status();
}
