// Shared device helpers of the attention kernels (attention.hip: bf16; attention_split.hip: split-bf16 "parity" precision).
#pragma once
#include <type_traits>
#include "common.h"
#include "amdseg_internal.h"
#include "prof.h"
#include "tile64.h"
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// dropout on attention probabilities: one 32-bit hash per aligned key pair, 16 bits per element.
// element (row = bh*L + q, key); keep iff 16-bit field >= thresh16
// (one mix32 round over (pair index ^ per-row salt): enough decorrelation for dropout, 1/3 of the integer ops)
__device__ __forceinline__ uint32_t pdrop_seedmix(uint64_t seed) {           // wave-uniform, once per kernel
    return mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u));
}
// per-row salt: ONE multiply -- the dK/dV kernel needs it for 16 different rows per lane and chunk, the avalanche is done
// by the mix32 round of pdrop_bits over (salt + pair index * golden ratio)
__device__ __forceinline__ uint32_t pdrop_salt(uint32_t seedmix, uint64_t row) {
    return seedmix + (uint32_t)row * 0x85ebca6bu;
}
__device__ __forceinline__ uint32_t pdrop_bits(uint32_t salt, int key_even) {
    // one multiply round: the argument is already a sum of odd-constant multiples of (row, pair), and v_mul_lo_u32 runs at
    // quarter rate -- the two-round mix32 made dropout a third of the forward kernel's time
    uint32_t x = salt + (uint32_t)(key_even >> 1) * 0x9e3779b9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    return x;
}
// the same word as pdrop_bits(salt, key_even) when the caller has already formed  salt + (key_even >> 1) * 0x9e3779b9
__device__ __forceinline__ uint32_t pdrop_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ float xor_reduce_max_g(float v) {   // across the 4 lane groups sharing l&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float xor_reduce_sum_g(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// 64 consecutive floats (a chunk's key mask / LSE / delta) global -> LDS by DMA, 4 B per lane; issued by ONE wave and
// covered by the same vmcnt(0) + barrier hand-off as the tiles
__device__ __forceinline__ void at_stage_f32x64(const float* g, char* lds, int l) {
    amdseg_glds4(g + l, lds);
}

struct AttnArgs {
    const bf16_t* qkv; const float* mask_bias; bf16_t* ctx; float* lse;
    const bf16_t* dctx; float* delta; bf16_t* dqkv;
    int B, L, heads, H3;    // H3 = 3*H row stride of qkv
    float scale, inv_keep; uint32_t thresh16; uint64_t seed;
    int window, nglobal;    // band attention (Longformer): one-sided window W (0 = full attention), leading global tokens G
    // block-list attention (BigBird block-sparse): row (h, block) of klist / qlist holds kcnt / qcnt block indices to visit, in order
    // and WITH multiplicity (a key block listed twice counts twice in the softmax, as in the reference's concatenated key matrices)
    const int* klist; const int* kcnt; const int* qlist; const int* qcnt; int list_stride;
    const int* korder; const int* qorder;   // optional [heads][L/64]: block index handled by the r-th workgroup of a head (longest lists first)
    const int* kend;                        // optional [B] (full attention): keys at positions >= kend[b] are all masked (<= -5000); 0 = none unmasked
    const int* seq_order;                   // optional [B] with kend: the sequence the b-th group of workgroups works on (longest first)
    const int* qguard;                      // optional (backward, with kend): *qguard == 0 <=> the dctx rows at positions >= kend[b] are exact zeros
                                            // (amdseg_bert_cfg.pad_guard): those query rows get dQ = 0 and add nothing to dK / dV, so they are not visited
    int skip_q;                             // forward, band: the ctx rows of the first skip_q queries of every sequence are NOT stored (their LSE is): the
                                            // caller of a phase-1 layer call writes the global tokens' rows itself, from another stream if it likes.
                                            // backward, band: the dctx rows of those queries count as ZERO whatever they hold (their gradient belongs to
                                            // the caller's global-row backward, which may read them while these kernels run)
    const uint64_t* keepA;                  // dropout keep bits written by attn_keepmask_kernel (KM instantiations), lane-mask layouts A (forward,
    const uint64_t* keepB;                  // dQ) and B (dK/dV), see "dropout keep masks" below
};

// Band ("sliding window + global") visibility, [hf] models/longformer/modeling_longformer.py:524-604 restated as a mask:
// key j is visible from query i  <=>  j < G (global key, re-added as extra column :559-568)  or  |i - j| <= W (:744-822);
// padded keys carry -inf in mask_bias as in the full-attention path.  Rows of padded queries are zeroed (:579).
// A workgroup only streams the 64-key chunks that intersect its band (+ chunk 0 for the global keys).
__device__ __forceinline__ bool band_masked(int q, int key, int W, int G) {
    const int d = key - q;
    return key >= G && (d > W || d < -W);
}

// Walks a block list in order: 64 entries live in one VGPR (lane i holds entry base + i), v_readlane picks the next one; a scalar
// load per chunk would put a global-memory round trip in front of every chunk's DMA address.
struct ListWalk {
    const int* lst; int n, lv, t;
    __device__ __forceinline__ void init(const int* p, int count, int l) { lst = p; n = count; t = 0; lv = l < n ? p[l] : 0; }
    __device__ __forceinline__ int next(int l) {            // entry t, then t + 1; entries past the end read as 0
        if (t > 0 && (t & 63) == 0) lv = (t + l) < n ? lst[t + l] : 0;
        const int v = __builtin_amdgcn_readlane(lv, t & 63);
        ++t;
        return v;
    }
};

// XCD-aware placement for the 3-D launches (x = row block, y = head, z = batch).  The hardware deals consecutive workgroup ids round-robin
// over the 8 XCDs, so the row blocks of ONE (batch, head) -- which all stream the same K / V (or Q / dO) rows -- landed on 8 different L2s
// and every XCD fetched those rows from the fabric on its own: PMC FETCH_SIZE of the dQ kernel was 468 MB per launch against ~110 MB of
// distinct data (6.2 TB/s over its 75 us: the kernel was fabric-bound, profiles/r02_pmc_instep.md).  Remapped so that workgroup ids
// xcd, xcd + 8, xcd + 16, ... (= one XCD, in dispatch order) walk the row blocks of the same (batch, head) before moving to the next one.
__device__ __forceinline__ void attn_xcd_remap(int& rb, int& h, int& b, int heads) {
    const int nrb = gridDim.x, nbh = gridDim.y * gridDim.z;
    if (nbh & 7) return;                                    // needs a multiple of 8 (batch, head) pairs; tiny shapes keep the plain order
    const int id = blockIdx.x + nrb * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, t = id >> 3;
    const int bh = (t / nrb) * 8 + xcd;
    rb = t % nrb; h = bh % heads; b = bh / heads;
}

// The 1-D launches (block lists, band dK/dV): workgroup id -> (rank r of the row block inside its pair, (batch, head) pair) such that one XCD
// (ids = xcd mod 8, in dispatch order) works through ONE pair at a time, its longest blocks first.  The previous rank-major order (all pairs
// at rank r before rank r + 1) kept 12 pairs' K / V (or Q / dO) in flight per XCD -- 12 MB against a 4-MB L2 -- and the band dK/dV order
// put neighbouring key blocks, which stream the same Q / dO chunks, on 8 different XCDs: PMC FETCH_SIZE 1.07 GB per launch against 0.2 GB
// of distinct data at L = 4096 (profiles/r02_pmc_longformer.txt).
// The `nfirst` longest blocks of EVERY pair (the global key block of the band, the two global blocks of a block list: 6-8 x the work of the
// others) still go out first, rank-major, so that they run under the short ones instead of forming a tail (pair-at-a-time for all ranks:
// 251 -> 237 seq/s at longformer-base although the traffic fell); pair bh sits on XCD bh % 8 in both parts.
__device__ __forceinline__ void attn_1d_order(int id, int nblk, int nbh, int nfirst, int& r, int& bh) {
    if ((nbh & 7) || id < nfirst * nbh) { r = id / nbh; bh = id - r * nbh; return; }
    id -= nfirst * nbh;
    const int xcd = id & 7, t = id >> 3, rest = nblk - nfirst;
    r = nfirst + t % rest; bh = (t / rest) * 8 + xcd;
}

// Trailing padding (AttnArgs.kend, handed down by the composite layer call from amdseg_bert_cfg.kend).  A key whose additive mask is
// <= -5000 has exp(score + mask - max) == 0 EXACTLY in fp32 whenever its row sees at least one unmasked key, so the 64-key chunks past the
// last unmasked key of a sequence add exact zeros to every sum (forward, dQ) and the key blocks there get dK = dV = 0: not visiting them
// changes no bit.  kend[b] == 0 (no unmasked key: the softmax is uniform over the masked keys, as in the reference) keeps every chunk.
// (A per-workgroup scan of the mask row instead of the precomputed kend cost as much as the skipped chunks saved.)
__device__ __forceinline__ int attn_visible_chunks(const AttnArgs& a, int b, int nch) {
    if (!a.kend) return nch;
    const int ke = a.kend[b];
    return ke > 0 ? min(nch, (ke + CH - 1) / CH) : nch;
}

#include "keepmask.h"      // the generator of the dropout keep masks (its own header: elementwise.hip runs it inside a LayerNorm launch)

// ---- consumer side of the dropout keep masks (written by attn_keepmask_kernel, attention.hip: layouts documented there)
typedef const __attribute__((address_space(4))) uint64_t* km_cptr;
struct KeepWords { uint64_t m[16]; };
__device__ __forceinline__ void km_load(KeepWords& k, const uint64_t* base, size_t cell) {
    km_cptr p = (km_cptr)(uintptr_t)(base + cell * 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) k.m[i] = p[i];
}
__device__ __forceinline__ float km_sel(float x, uint64_t lanes) {     // lane's bit set ? x : 0  ->  v_cndmask_b32 v, 0, v, s[n:n+1]
    // (the builtin, not inline asm: the compiler does not give an asm statement the wait states a read of a fresh MFMA result needs)
    return __builtin_amdgcn_inverse_ballot_w64(lanes) ? x : 0.f;
}

// eight bf16 values times s, rounded back to bf16 (exact for a power of two)
__device__ __forceinline__ bf16x8 frag_scale(bf16x8 f, float s) {
    union { bf16x8 v; uint32_t u[4]; } x, y;
    x.v = f;
#pragma unroll
    for (int i = 0; i < 4; ++i) y.u[i] = pack2bf(__uint_as_float(x.u[i] << 16) * s, __uint_as_float(x.u[i] & 0xffff0000u) * s);
    return y.v;
}

