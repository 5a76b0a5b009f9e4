// PoNet token mixing for gfx950 (alimeeting4mug/src/models/modeling_ponet.py:68-79 -> modelscope PoNetModel, NOT in the
// reference tree: semantics restated in oracle/ponet_oracle.py, parity unpinned).  HBM-bound row kernels over the fused
// projection output  proj [M, 5H] bf16 = (Hq | Hk | Ho | Hl | Hs):
//
//   segment max-pooling  S_n = max of Hs over the valid tokens of n's segment (a contiguous run of equal segment_ids),
//   local max-pooling    L_n = max of Hl over the valid tokens n-1, n, n+1,
//   fusion               ctx_n = (g + S_n) * Ho_n + L_n        (g = global-attention aggregate, [B, H] fp32, see host mirror)
// and their backward.
//
// Round-2 (second) layout.  The first two versions reduced runs with trees of per-run waves (level-A rows, then one wave per run folding
// them): the run-level kernels were latency chains -- one wave walks the rows of a 500-token run one dependent load after the other -- and
// cost 54 + 42 us per layer although they move a few MB; the row kernels used 16 B per lane on 96 chunks per row (H = 768), i.e. half of
// every second wave instruction idle, at 170-250 VGPRs.  Now:
//   * every kernel maps a wave to (16 consecutive tokens) x (256 columns): lanes 0-31 walk tokens 0-7, lanes 32-63 tokens 8-15, each lane
//     owning 8 columns (16 B) -> all lanes busy for H = 768, every row access of a half-wave is 512 contiguous bytes, all row loads of
//     a stream are issued before the first use (18-30 rows in flight per wave);
//   * the run maximum is ONE 32-bit word per (run, column):  key = order-preserving image of the bf16 value << 16 | (0xFFFF - position),
//     folded with atomicMax into the row of the run's first token: value and "first maximum wins" argmax in one commutative, hence
//     deterministic, reduction -- no tree, no second kernel.  A stream folds its tokens in registers and issues one atomic per run piece
//     and column (transposed through LDS so that an atomic instruction covers whole 128-B lines).  Atomics on ONE address retire at
//     ~0.15-0.3 us each (memory-side), so the pieces that can continue into a neighbouring stream -- the first and the last of every
//     stream -- are first merged across the 8 streams of a workgroup (64 .. 256 consecutive tokens) in LDS: a 4096-token run costs
//     16-64 atomics per column instead of 512 (measured before the merge: backward 80 -> 213 us on one run per sequence);
//   * backward accumulates E = dctx * Ho per run piece in registers and adds it (fp32 atomicAdd, same workgroup merge) into the run's row
//     of G, and the workgroup's total into dg; the routing kernel then writes dHs_j = [argmax == j] * G.
//   forward   pn_zero(keys of the run-start rows) -> pn_segmax -> pn_combine
//   backward  pn_zero(G rows)                     -> pn_bwd_tok -> pn_bwd_route
// Algorithmic bytes per token: forward read Hs, Ho, Hl + write ctx = 4 * H * 2 B (25.2 MB per 4096-token sequence and layer);
// backward read dctx, Ho, Hl + write dHo, dHl, dHs = 6 * H * 2 B.
#include <algorithm>
#include "common.h"
#include "amdseg_internal.h"

struct PnArgs {
    const bf16_t* proj; int ld;                // fused projections, row stride (>= 5H)
    const float* mask_bias;                    // [M], < 0 = padded token
    const int* run_start;                      // [M] position (within the sequence) of the first token of the token's run
    uint32_t* keys;                            // [M, H] run maximum keys, row of the run's first token
    float* G;                                  // [M, H] run sums of E = dctx * Ho, row of the run's first token
    const float* g;                            // [B, H]
    bf16_t* ctx;                               // [M, H]
    const bf16_t* dctx;                        // [M, H]
    bf16_t* dproj;                             // [M, ld] gradient of the projections (Ho, Hl, Hs column blocks written here)
    float* dg;                                 // [B, H] gradient of g: sum of E over the valid tokens of a sequence
    const int* work;                           // plan: [1] = #runs, [2 + M ..) = token index of every run's first token
    int M, L, H;
};

// ---- bf16 <-> order-preserving 16-bit image (unsigned compare == float compare; -0 is folded onto +0)
__device__ __forceinline__ uint32_t pn_ord(uint32_t bits) {
    bits = bits == 0x8000u ? 0u : bits;
    return bits ^ ((bits & 0x8000u) ? 0xffffu : 0x8000u);
}
__device__ __forceinline__ float pn_unord(uint32_t o) {
    const uint32_t bits = o ^ ((o & 0x8000u) ? 0x8000u : 0xffffu);
    return __uint_as_float(bits << 16);
}
__device__ __forceinline__ void pn_unpack(const uint4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pn_pack(const float (&v)[8]) {
    uint4 r; r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]); r.z = pack2bf(v[4], v[5]); r.w = pack2bf(v[6], v[7]);
    return r;
}

// wave -> (16-token block, 256-column group); lane -> (8-token stream, 8 columns)
struct PnLane {
    int n0, b, p0, col, l31, half; bool act, tok; size_t seq0;      // tok: the stream exists (half-wave uniform); act: ... and so do the lane's columns
    __device__ __forceinline__ bool init(const PnArgs& a) {
        const int l = threadIdx.x & 63;
        const int ncg = (a.H + 255) >> 8, tbb = blockIdx.x / ncg, cg = blockIdx.x - tbb * ncg, tb = tbb * 4 + (threadIdx.x >> 6);
        half = l >> 5; l31 = l & 31;
        col = cg * 256 + l31 * 8;
        n0 = tb * 16 + half * 8;
        tok = n0 < a.M; act = tok && col < a.H;
        b = min(n0, a.M - 1) / a.L; p0 = n0 - b * a.L; seq0 = (size_t)b * a.L;
        return tb * 16 < a.M;                  // wave-uniform: anything to do (kernels with a workgroup barrier must not exit on it)
    }
};

// ---------------------------------------------------------------------------------------------------- plan: the list of run starts
__global__ __launch_bounds__(256) void pn_plan_kernel(const int* __restrict__ run_start, int* work, int M, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= M) return;
    if (n % L == run_start[n]) work[2 + M + atomicAdd(work + 1, 1)] = n;
}

// zero the rows of the run starts of a [M, H] 32-bit plane (one wave per run; H % 8 == 0)
__global__ __launch_bounds__(256) void pn_zero_kernel(uint32_t* plane, const int* work, int M, int H) {
    const int l = threadIdx.x & 63, count = work[1];
    for (int item = blockIdx.x * 4 + (threadIdx.x >> 6); item < count; item += gridDim.x * 4) {
        uint4* row = reinterpret_cast<uint4*>(plane + (size_t)work[2 + M + item] * H);
        for (int c = l; c < (H >> 2); c += 64) row[c] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// one run piece of a stream -> the run's row, through a half-wave transpose in LDS: lane i holds columns i*8 .. +8 of the half's 256;
// instruction j then covers columns j*32 + lane = one 128-B line per half-wave (the direct form touches 8 lines with 4 of 32 bytes each)
template <typename T, typename F>
__device__ __forceinline__ void pn_flush(T* lds_half, const T (&v)[8], int l31, T* row, int col0, int H, F atomic_op) {
    *reinterpret_cast<uint4*>(lds_half + l31 * 8) = make_uint4(__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]),
                                                                __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3]));
    *reinterpret_cast<uint4*>(lds_half + l31 * 8 + 4) = make_uint4(__builtin_bit_cast(uint32_t, v[4]), __builtin_bit_cast(uint32_t, v[5]),
                                                                    __builtin_bit_cast(uint32_t, v[6]), __builtin_bit_cast(uint32_t, v[7]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave: LDS operations complete in order; this also pins the compiler's order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const T x = lds_half[j * 32 + l31];
        if (col0 + j * 32 + l31 < H) atomic_op(row + col0 + j * 32 + l31, x);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the image is rewritten by the next piece
}

// ---- token meta data of a stream: lane i (of its half-wave) holds mask / run_start of position p_first + i; PN_AT broadcasts entry i to the
// half.  Loaded with ONE instruction each and -- like every row load below -- independent of any other load's result: the first version
// asked `mask_bias[n] >= 0 ? row : 0` per token, which the compiler has to serialise (8 dependent round trips per stream: 34 us for 50 MB).
// Rows of padded tokens are loaded like any other (valid memory) and ignored by the validity flags.
__device__ __forceinline__ void pn_meta(const PnArgs& a, size_t seq0, int p_first, int cnt, int l31, bool tok, float& mv, int& rs) {
    const int p = p_first + l31;
    const bool in = tok && l31 < cnt && p >= 0 && p < a.L;
    mv = in ? a.mask_bias[seq0 + p] : -1.f;
    rs = in ? a.run_start[seq0 + p] : -1;
}
#define PN_AT(v, i) __shfl(v, hbase | (i), 64)
// row n of a [M, ld] matrix, clamped into the buffer (rows outside the token's own sequence are loaded and ignored: see pn_meta)
__device__ __forceinline__ const uint4* pn_row(const bf16_t* base, int n, int M, int ld, int col) {
    return reinterpret_cast<const uint4*>(base + (size_t)(uint32_t)min(max(n, 0), M - 1) * (uint32_t)ld + col);
}

// ---- workgroup merge of the stream-boundary pieces.  Stream si (0..7, in token order) deposits its FIRST piece in entry 2 si and -- when it
// has more than one -- its LAST piece in entry 2 si + 1 (row id = global row of the run's first token, -1 = no piece); thread c then walks
// the 16 entries of column c in order, folds neighbours of the same run and issues one atomic per run (whole 128-B lines per instruction).
template <typename T>
struct PnMerge { T val[16][256]; int row[16]; };
template <typename T>
__device__ __forceinline__ void pn_deposit(PnMerge<T>& m, int entry, int row, const T (&v)[8], int l31) {
    if (row >= 0) {
        *reinterpret_cast<uint4*>(&m.val[entry][l31 * 8]) = make_uint4(__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]),
                                                                      __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3]));
        *reinterpret_cast<uint4*>(&m.val[entry][l31 * 8 + 4]) = make_uint4(__builtin_bit_cast(uint32_t, v[4]), __builtin_bit_cast(uint32_t, v[5]),
                                                                          __builtin_bit_cast(uint32_t, v[6]), __builtin_bit_cast(uint32_t, v[7]));
    }
    if (l31 == 0) m.row[entry] = row;
}
template <typename T, typename OP, typename AT>
__device__ __forceinline__ void pn_merge_flush(const PnMerge<T>& m, T* plane, int col0, int H, OP op, AT atomic_op) {
    const int c = threadIdx.x;                               // 256 threads = the 256 columns of the group
    if (col0 + c >= H) return;
    int cur = -1; T acc = T(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = m.row[i];
        if (r < 0) continue;
        const T v = m.val[i][c];
        if (r == cur) acc = op(acc, v);
        else { if (cur >= 0) atomic_op(plane + (size_t)cur * H + col0 + c, acc); cur = r; acc = v; }
    }
    if (cur >= 0) atomic_op(plane + (size_t)cur * H + col0 + c, acc);
}

// ---------------------------------------------------------------------------------------------------- forward 1: run maxima
// NB batches of 8 tokens per stream (a wave covers 16 * NB consecutive tokens, a workgroup 64 * NB; the next batch's rows are requested
// before the current batch is folded).  Measured on 8 x 4096 tokens, 70-token runs: NB = 1 18.8 us, 2 20.9, 4 24.9 -- the kernel is
// instruction-bound (~170 instructions per token and lane), more and shorter waves hide it better than fewer atomics help: NB = 1 is launched.
template <int NB>
__global__ __launch_bounds__(256) void pn_segmax_kernel(PnArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tr[4][2][256];
    __shared__ __attribute__((aligned(16))) PnMerge<uint32_t> mg;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ncg = (a.H + 255) >> 8, tbb = blockIdx.x / ncg, cg = blockIdx.x - tbb * ncg, tb = tbb * 4 + w;
    const int half = l >> 5, hbase = half << 5, l31 = l & 31, col = cg * 256 + l31 * 8, col0 = cg * 256, si = w * 2 + half;
    const int n0 = tb * 16 * NB + half * 8 * NB;                         // (8 * NB) | L: a stream never straddles sequences
    const bool tok = n0 < a.M, act = tok && col < a.H;
    const int b = min(n0, a.M - 1) / a.L, p0 = n0 - b * a.L;
    const size_t seq0 = (size_t)b * a.L;
    uint32_t* lds_half = tr[w][half];
    float mv; int rsv;
    pn_meta(a, seq0, p0, 8 * NB, l31, tok, mv, rsv);
    uint4 raw[2][8];
    auto request = [&](int bt, int buf) {
#pragma unroll
        for (int k = 0; k < 8; ++k) raw[buf][k] = act ? *pn_row(a.proj + 4 * a.H, n0 + bt * 8 + k, a.M, a.ld, col) : make_uint4(0u, 0u, 0u, 0u);
    };
    uint32_t key[8];
    int cur = -1; bool first_open = true;                   // first_open: the open piece is the stream's first one
    auto close_piece = [&]() {                              // the open piece ends inside the stream
        if (first_open) { pn_deposit<uint32_t>(mg, 2 * si, (int)seq0 + cur, key, l31); first_open = false; }
        else pn_flush<uint32_t>(lds_half, key, l31, a.keys + (seq0 + cur) * a.H, col0, a.H, [](uint32_t* p, uint32_t x) { atomicMax(p, x); });
    };
    request(0, 0);
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        const int buf = bt & 1;
        if (bt + 1 < NB) request(bt + 1, buf ^ 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!(PN_AT(mv, bt * 8 + k) >= 0.f)) continue;               // half-wave uniform: lanes without columns still take part
            const int rs = PN_AT(rsv, bt * 8 + k);
            if (rs != cur) {
                if (cur >= 0) close_piece();
                cur = rs;
#pragma unroll
                for (int e = 0; e < 8; ++e) key[e] = 0u;
            }
            const uint32_t wd[4] = {raw[buf][k].x, raw[buf][k].y, raw[buf][k].z, raw[buf][k].w};
            const uint32_t low = 0xffffu - (uint32_t)(p0 + bt * 8 + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t bits = (e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffffu);
                key[e] = max(key[e], (pn_ord(bits) << 16) | low);
            }
        }
    }
    // stream end: the open piece is the first (entry 2 si, no last) or the last one (entry 2 si + 1)
    if (cur < 0) { pn_deposit<uint32_t>(mg, 2 * si, -1, key, l31); pn_deposit<uint32_t>(mg, 2 * si + 1, -1, key, l31); }
    else if (first_open) { pn_deposit<uint32_t>(mg, 2 * si, (int)seq0 + cur, key, l31); pn_deposit<uint32_t>(mg, 2 * si + 1, -1, key, l31); }
    else pn_deposit<uint32_t>(mg, 2 * si + 1, (int)seq0 + cur, key, l31);
    __syncthreads();
    pn_merge_flush<uint32_t>(mg, a.keys, col0, a.H, [](uint32_t x, uint32_t y) { return max(x, y); }, [](uint32_t* p, uint32_t x) { atomicMax(p, x); });
}

// the run's S and argmax position for this lane's 8 columns
__device__ __forceinline__ void pn_load_S(const PnArgs& a, size_t row, int col, float (&S)[8], int (&arg)[8]) {
    const uint4 k0 = *reinterpret_cast<const uint4*>(a.keys + row * a.H + col), k1 = *reinterpret_cast<const uint4*>(a.keys + row * a.H + col + 4);
    const uint32_t w[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = pn_unord(w[e] >> 16); arg[e] = 0xffff - (int)(w[e] & 0xffffu); }
}

// ---------------------------------------------------------------------------------------------------- forward 2: fusion
__global__ __launch_bounds__(256) void pn_combine_fwd_kernel(PnArgs a) {
    PnLane q;
    if (!q.init(a)) return;
    const int hbase = q.half << 5;
    float mv; int rsv;
    pn_meta(a, q.seq0, q.p0 - 1, 10, q.l31, q.tok, mv, rsv);            // entry i <-> position p0 - 1 + i
    // (no early exit for lanes without columns: PN_AT reads lanes 0-9 of the half-wave, which must stay active)
    uint4 ho[8], hl[10];
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 10; ++k) hl[k] = q.act ? *pn_row(a.proj + 3 * a.H, q.n0 + k - 1, a.M, a.ld, q.col) : z4;
#pragma unroll
    for (int k = 0; k < 8; ++k) ho[k] = q.act ? *pn_row(a.proj + 2 * a.H, q.n0 + k, a.M, a.ld, q.col) : z4;
    float gv[8], S[8]; int arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gv[e] = 0.f; S[e] = 0.f; }
    if (q.act) ld8<float>(a.g + (size_t)q.b * a.H + q.col, gv);
    int cur = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float out[8];
        if (PN_AT(mv, k + 1) >= 0.f) {
            const int rs = PN_AT(rsv, k + 1);
            if (rs != cur) { cur = rs; if (q.act) pn_load_S(a, q.seq0 + cur, q.col, S, arg); }
            float h[8], x[8], y[8];
            pn_unpack(ho[k], h); pn_unpack(hl[k + 1], x);
            if (PN_AT(mv, k) >= 0.f) { pn_unpack(hl[k], y);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], y[e]); }
            if (PN_AT(mv, k + 2) >= 0.f) { pn_unpack(hl[k + 2], y);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], y[e]); }
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (gv[e] + S[e]) * h[e] + x[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = 0.f;
        }
        if (q.act) *reinterpret_cast<uint4*>(a.ctx + ((size_t)q.n0 + k) * a.H + q.col) = pn_pack(out);
    }
}

// ---------------------------------------------------------------------------------------------------- backward 1: token rows
// dHo = dctx * (g + S);  dHl_j = sum over the valid neighbours m of j (incl. j) of dctx_m * [first maximum of {Hl_{m-1}, Hl_m, Hl_{m+1}} == j];
// E = dctx * Ho summed per run piece and added into the run's G row.  Padded rows get zero gradients.
__global__ __launch_bounds__(256) void pn_bwd_tok_kernel(PnArgs a) {
    __shared__ __attribute__((aligned(16))) float tr[4][2][256];
    __shared__ __attribute__((aligned(16))) PnMerge<float> mg;
    __shared__ __attribute__((aligned(16))) float totl[8][256];
    __shared__ int totb[8];
    PnLane q;
    q.init(a);                                                           // (no early exit: workgroup barrier below)
    const int hbase = q.half << 5, si = (threadIdx.x >> 6) * 2 + q.half;
    float mv; int rsv;
    pn_meta(a, q.seq0, q.p0 - 2, 12, q.l31, q.tok, mv, rsv);            // entry i <-> position p0 - 2 + i
    uint4 ho[8], hl[12], dc[10];
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 12; ++k) hl[k] = q.act ? *pn_row(a.proj + 3 * a.H, q.n0 + k - 2, a.M, a.ld, q.col) : z4;     // Hl rows p0-2 .. p0+9
#pragma unroll
    for (int k = 0; k < 10; ++k) dc[k] = q.act ? *pn_row(a.dctx, q.n0 + k - 1, a.M, a.H, q.col) : z4;                // dctx rows p0-1 .. p0+8
#pragma unroll
    for (int k = 0; k < 8; ++k) ho[k] = q.act ? *pn_row(a.proj + 2 * a.H, q.n0 + k, a.M, a.ld, q.col) : z4;
    float gv[8], S[8], seg[8], tot[8]; int arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gv[e] = 0.f; S[e] = 0.f; seg[e] = 0.f; tot[e] = 0.f; }
    if (q.act) ld8<float>(a.g + (size_t)q.b * a.H + q.col, gv);
    float* lds_half = tr[threadIdx.x >> 6][q.half];
    const int col0 = q.col - q.l31 * 8;
    bool first_open = true;
    auto close_piece = [&](int cur_) {                      // the open piece ends inside the stream
#pragma unroll
        for (int e = 0; e < 8; ++e) tot[e] += seg[e];
        if (first_open) { pn_deposit<float>(mg, 2 * si, (int)q.seq0 + cur_, seg, q.l31); first_open = false; }
        else pn_flush<float>(lds_half, seg, q.l31, a.G + (q.seq0 + cur_) * a.H, col0, a.H, [](float* p, float x) { atomicAdd(p, x); });
    };
    bool hv[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) hv[k] = PN_AT(mv, k) >= 0.f;
    int cur = -1;
    // hlf[j] = Hl row p + j - 2 as floats (-inf where invalid), slid one row per token
    float hlf[5][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pn_unpack(hl[j], hlf[j + 1]);
        if (!hv[j]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hlf[j + 1][e] = -INFINITY;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) hlf[j][e] = hlf[j + 1][e];
        pn_unpack(hl[k + 4], hlf[4]);
        if (!hv[k + 4]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hlf[4][e] = -INFINITY;
        }
        if (!q.tok) continue;                              // (half-wave uniform)
        bf16_t* drow = a.dproj + ((size_t)q.n0 + k) * a.ld + q.col;
        if (!hv[k + 2]) {                                  // padded token: zero gradients
            if (q.act) {
                *reinterpret_cast<uint4*>(drow + 2 * a.H) = z4;
                *reinterpret_cast<uint4*>(drow + 3 * a.H) = z4;
            }
            continue;
        }
        const int rs = PN_AT(rsv, k + 2);
        if (rs != cur) {
            if (cur >= 0) close_piece(cur);
            cur = rs;
            if (q.act) pn_load_S(a, q.seq0 + cur, q.col, S, arg);
#pragma unroll
            for (int e = 0; e < 8; ++e) seg[e] = 0.f;
        }
        float d[3][8], h[8], o1[8], dl[8];
        pn_unpack(dc[k], d[0]); pn_unpack(dc[k + 1], d[1]); pn_unpack(dc[k + 2], d[2]);        // dctx of the neighbours p-1, p, p+1
        pn_unpack(ho[k], h);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o1[e] = d[1][e] * (gv[e] + S[e]);
            seg[e] += d[1][e] * h[e];
            dl[e] = 0.f;
        }
        // neighbour m = p + j - 1 (j = 0..2) has the window hlf[j], hlf[j+1], hlf[j+2]; this token is hlf[2]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!hv[k + 1 + j]) continue;                  // the neighbour itself must be a valid token
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x0 = hlf[j][e], x1 = hlf[j + 1][e], x2 = hlf[j + 2][e];
                const int am = (x0 >= x1 && x0 >= x2) ? j : ((x1 >= x2) ? j + 1 : j + 2);      // first maximum of the window
                if (am == 2) dl[e] += d[j][e];
            }
        }
        if (q.act) {
            *reinterpret_cast<uint4*>(drow + 2 * a.H) = pn_pack(o1);
            *reinterpret_cast<uint4*>(drow + 3 * a.H) = pn_pack(dl);
        }
    }
    // stream end: the open piece is the first (entry 2 si, no last) or the last one (entry 2 si + 1); the stream's total goes to dg
    if (cur < 0) { pn_deposit<float>(mg, 2 * si, -1, seg, q.l31); pn_deposit<float>(mg, 2 * si + 1, -1, seg, q.l31); }
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) tot[e] += seg[e];
        if (first_open) { pn_deposit<float>(mg, 2 * si, (int)q.seq0 + cur, seg, q.l31); pn_deposit<float>(mg, 2 * si + 1, -1, seg, q.l31); }
        else pn_deposit<float>(mg, 2 * si + 1, (int)q.seq0 + cur, seg, q.l31);
    }
    *reinterpret_cast<float4*>(&totl[si][q.l31 * 8]) = make_float4(tot[0], tot[1], tot[2], tot[3]);
    *reinterpret_cast<float4*>(&totl[si][q.l31 * 8 + 4]) = make_float4(tot[4], tot[5], tot[6], tot[7]);
    if (q.l31 == 0) totb[si] = (q.tok && cur >= 0) ? q.b : -1;
    __syncthreads();
    pn_merge_flush<float>(mg, a.G, col0, a.H, [](float x, float y) { return x + y; }, [](float* p, float x) { atomicAdd(p, x); });
    if (a.dg && col0 + (int)threadIdx.x < a.H) {            // the workgroup's sum of E per sequence (a workgroup of 64 tokens may straddle sequences)
        const int c = threadIdx.x;
        int cb = -1; float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bi = totb[i];
            if (bi < 0) continue;
            if (bi != cb) { if (cb >= 0) atomicAdd(a.dg + (size_t)cb * a.H + col0 + c, acc); cb = bi; acc = 0.f; }
            acc += totl[i][c];
        }
        if (cb >= 0) atomicAdd(a.dg + (size_t)cb * a.H + col0 + c, acc);
    }
}

// ---------------------------------------------------------------------------------------------------- backward 2: routing
// dHs_j = [argmax of j's run == j] * G(run)
__global__ __launch_bounds__(256) void pn_bwd_route_kernel(PnArgs a) {
    PnLane q;
    if (!q.init(a)) return;
    const int hbase = q.half << 5;
    float mv; int rsv;
    pn_meta(a, q.seq0, q.p0, 8, q.l31, q.tok, mv, rsv);
    float S[8], G[8]; int arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { G[e] = 0.f; arg[e] = -1; }
    int cur = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!q.tok) continue;
        const size_t n = (size_t)q.n0 + k;
        const int rs = PN_AT(rsv, k), pos = q.p0 + k;
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = 0.f;
        if (rs != cur) {
            cur = rs;
            if (q.act) { pn_load_S(a, q.seq0 + cur, q.col, S, arg); ld8<float>(a.G + (q.seq0 + cur) * a.H + q.col, G); }
        }
        if (PN_AT(mv, k) >= 0.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = arg[e] == pos ? G[e] : 0.f;
        }
        if (q.act) *reinterpret_cast<uint4*>(a.dproj + n * a.ld + 4 * a.H + q.col) = pn_pack(out);
    }
}

// ---------------------------------------------------------------------------------------------------- launchers
static int pn_check(int B, int L, int H, int ld) {
    if (B <= 0 || L <= 0 || L > 65535 || (L % 8) || H <= 0 || (H % 8) || ld < 5 * H || (ld % 8)) return AMDSEG_ERR_SHAPE;
    return AMDSEG_OK;
}
static int pn_grid(int M, int H, int tok_per_block = 64) { return (M + tok_per_block - 1) / tok_per_block * ((H + 255) / 256); }

int amdseg_ponet_plan_impl(const float* mask_bias, const int* run_start, int* work, int B, int L, hipStream_t s) {
    if (!mask_bias || !run_start || !work) return AMDSEG_ERR_ARG;
    if (B <= 0 || L <= 0) return AMDSEG_ERR_SHAPE;
    hipError_t e = hipMemsetAsync(work, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    const int M = B * L;
    hipLaunchKernelGGL(pn_plan_kernel, dim3((M + 255) / 256), dim3(256), 0, s, run_start, work, M, L);
    return amdseg_launch_status();
}

int amdseg_ponet_pool_fwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                               const float* g, void* part, void* parg, void* ctx, int B, int L, int H, hipStream_t s) {
    // part: [M, H] 32-bit keys (run maximum | argmax) in the rows of the run starts, read again by backward; parg: unused since round 2
    if (!proj || !mask_bias || !run_start || !work || !g || !part || !ctx) return AMDSEG_ERR_ARG;
    (void)run_end; (void)parg;
    int rc = pn_check(B, L, H, ld);
    if (rc) return rc;
    PnArgs a = {};
    a.proj = (const bf16_t*)proj; a.ld = ld; a.mask_bias = mask_bias; a.run_start = run_start; a.g = g; a.work = work;
    a.ctx = (bf16_t*)ctx; a.M = B * L; a.L = L; a.H = H; a.keys = (uint32_t*)part;
    hipLaunchKernelGGL(pn_zero_kernel, dim3(std::min(512, (a.M + 3) / 4)), dim3(256), 0, s, a.keys, work, a.M, H);
    hipLaunchKernelGGL(pn_segmax_kernel<1>, dim3(pn_grid(a.M, H)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_combine_fwd_kernel, dim3(pn_grid(a.M, H)), dim3(256), 0, s, a);
    return amdseg_launch_status();
}

int amdseg_ponet_pool_bwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end, const int* work,
                               const float* g, const void* part, const void* parg, const void* dctx, void* dproj, float* dg, float* psum,
                               int B, int L, int H, hipStream_t s) {
    // psum: [M, H] fp32, the run sums G in the rows of the run starts; dg [B, H] is ZEROED here and receives the per-sequence sum of dctx * Ho
    if (!proj || !mask_bias || !run_start || !work || !g || !part || !dctx || !dproj || !dg || !psum) return AMDSEG_ERR_ARG;
    (void)run_end; (void)parg;
    int rc = pn_check(B, L, H, ld);
    if (rc) return rc;
    PnArgs a = {};
    a.proj = (const bf16_t*)proj; a.ld = ld; a.mask_bias = mask_bias; a.run_start = run_start; a.g = g; a.work = work;
    a.dctx = (const bf16_t*)dctx; a.dproj = (bf16_t*)dproj; a.G = psum; a.dg = dg; a.M = B * L; a.L = L; a.H = H;
    a.keys = (uint32_t*)part;
    hipError_t e = hipMemsetAsync(dg, 0, (size_t)B * H * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pn_zero_kernel, dim3(std::min(512, (a.M + 3) / 4)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(a.G), work, a.M, H);
    hipLaunchKernelGGL(pn_bwd_tok_kernel, dim3(pn_grid(a.M, H)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_bwd_route_kernel, dim3(pn_grid(a.M, H)), dim3(256), 0, s, a);
    return amdseg_launch_status();
}
