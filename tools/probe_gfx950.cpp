// Hardware-semantics probe for gfx950 (MI355X). Test infrastructure only: it pins the
// lane<->element maps that the kernels in spokennlp_amd/csrc rely on
//   (1) MFMA 16x16x32 / 32x32x16 bf16 operand + accumulator layout,
//   (2) ds_read_b64_tr_b16 (LDS transpose read) gather pattern,
//   (3) global_load_lds (direct global->LDS DMA) destination order,
// and prints device properties. Build: hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.cpp -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- (1) MFMA with the ASSUMED layouts -------------------------------------------------
// 16x16x32: A[i][k]: lane l holds i = l&15, k = (l>>4)*8 + j (j=0..7); B[k][j]: lane holds col l&15, k same.
//           D: lane holds col = l&15, rows (l>>4)*4 + r, r = 0..3
__global__ void mfma16(const unsigned short* A, const unsigned short* B, float* D) {   // A 16x32 row-major, B 32x16 row-major
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 15) * 32 + (l >> 4) * 8 + j];
        b[j] = (short)B[((l >> 4) * 8 + j) * 16 + (l & 15)];
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
// 32x32x16: A[i][k]: i = l&31, k = (l>>5)*8 + j; B[k][j]: col l&31; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
__global__ void mfma32(const unsigned short* A, const unsigned short* B, float* D) {   // A 32x16, B 16x32
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 31) * 16 + (l >> 5) * 8 + j];
        b[j] = (short)B[((l >> 5) * 8 + j) * 32 + (l & 31)];
    }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// raw dump: one-hot A slot (lane la, elem ja), B all ones -> which (lane, reg) of D are nonzero
__global__ void mfma16_raw(int la, int ja, int mode, float* Draw) {
    int l = threadIdx.x;
    bf16x8 a, b;
    short one = (short)0x3f80;
    for (int j = 0; j < 8; ++j) {
        if (mode == 0) { a[j] = (l == la && j == ja) ? one : 0; b[j] = one; }
        else           { b[j] = (l == la && j == ja) ? one : 0; a[j] = one; }
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Draw[l * 4 + r] = acc[r];
}

// ---- (2) ds_read_b64_tr_b16 ------------------------------------------------------------
// LDS holds lds[i] = i (u16). mode 0: lane l address = element (l*4)           (dense, lane-linear 8 B each)
//                              mode 1: lane l address = 16-lane group g=l>>4, i=l&15: element g*256 + (i>>2)*64 + (i&3)*4
//                                      (a [4 rows][16 cols] block with row pitch 64 elements)
//                              mode 2: all lanes of a group same address (element g*64)
__global__ void trprobe(int mode, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, i = l & 15;
    int e = mode == 0 ? l * 4 : mode == 1 ? g * 256 + (i >> 2) * 64 + (i & 3) * 4 : g * 64;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + e));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

// ---- (3) global_load_lds 16 B ----------------------------------------------------------
// each lane reads 16 B from src + perm(lane)*16 ; LDS destination base is wave-uniform.
__global__ void gldsprobe(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0xdeadbeef;
    __syncthreads();
    int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    int p = (l * 7) & 63;   // a permutation of lanes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (w * 64 + p) * 4),
                                     (__attribute__((address_space(3))) void*)(lds + w * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s arch=%s CUs=%d clock=%d kHz memclock=%d kHz L2=%d smemPerBlock=%zu regsPerBlock=%d totalGlobal=%.1f GB warp=%d\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize,
           p.sharedMemPerBlock, p.regsPerBlock, p.totalGlobalMem / 1e9, p.warpSize);
    int rc = 0;
    // (1a) 16x16x32
    {
        std::vector<unsigned short> A(16 * 32), B(32 * 16);
        std::vector<float> Af(16 * 32), Bf(32 * 16), D(256), R(256, 0.f);
        srand(1);
        for (int i = 0; i < 512; ++i) { Af[i] = (float)((rand() % 17) - 8); A[i] = f2bf(Af[i]); Bf[i] = (float)((rand() % 13) - 6); B[i] = f2bf(Bf[i]); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += Af[i * 32 + k] * Bf[k * 16 + j]; R[i * 16 + j] = s; }
        unsigned short *dA, *dB; float* dD;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        mfma16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        float md = 0; for (int i = 0; i < 256; ++i) md = fmaxf(md, fabsf(D[i] - R[i]));
        printf("[mfma16x16x32 assumed layout] maxdiff=%g %s\n", md, md == 0 ? "PASS" : "FAIL");
        if (md != 0) {
            rc = 1;
            float* dR; CK(hipMalloc(&dR, 1024)); std::vector<float> raw(256);
            for (int mode = 0; mode < 2; ++mode) for (int la = 0; la < 64; la += 1) for (int ja = 0; ja < 8; ja += 7) {
                mfma16_raw<<<1, 64>>>(la, ja, mode, dR); CK(hipDeviceSynchronize());
                CK(hipMemcpy(raw.data(), dR, 1024, hipMemcpyDeviceToHost));
                printf("raw mode=%d lane=%d j=%d nz:", mode, la, ja);
                int c = 0; for (int i = 0; i < 256 && c < 20; ++i) if (raw[i] != 0) { printf(" (%d,%d)", i / 4, i % 4); ++c; }
                printf("\n");
            }
        }
    }
    // (1b) 32x32x16
    {
        std::vector<unsigned short> A(512), B(512);
        std::vector<float> Af(512), Bf(512), D(1024), R(1024, 0.f);
        srand(2);
        for (int i = 0; i < 512; ++i) { Af[i] = (float)((rand() % 17) - 8); A[i] = f2bf(Af[i]); Bf[i] = (float)((rand() % 13) - 6); B[i] = f2bf(Bf[i]); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += Af[i * 16 + k] * Bf[k * 32 + j]; R[i * 32 + j] = s; }
        unsigned short *dA, *dB; float* dD;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        mfma32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        float md = 0; for (int i = 0; i < 1024; ++i) md = fmaxf(md, fabsf(D[i] - R[i]));
        printf("[mfma32x32x16 assumed layout] maxdiff=%g %s\n", md, md == 0 ? "PASS" : "FAIL");
        if (md != 0) rc = 1;
    }
    // (2) tr read
    {
        unsigned short* dO; CK(hipMalloc(&dO, 512)); std::vector<unsigned short> o(256);
        for (int mode = 0; mode < 3; ++mode) {
            trprobe<<<1, 64>>>(mode, dO); CK(hipDeviceSynchronize());
            CK(hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost));
            printf("[tr_b16 mode %d]\n", mode);
            for (int l = 0; l < 64; ++l) { printf(" l%02d:%4d %4d %4d %4d", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]); if ((l & 3) == 3) printf("\n"); }
        }
        // check the hypothesis: within a 16-lane group, result[l][j] = data supplied by lane (j*4 + (i>>2)) element (i&3)
        trprobe<<<1, 64>>>(1, dO); CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            int g = l >> 4, i = l & 15;
            int expect = g * 256 + j * 64 + i;    // row j, column i of the [4][16] block with pitch 64
            if (o[l * 4 + j] != expect) ++bad;
        }
        printf("[tr_b16 hypothesis: lane i elem j = block[row j][col i]] %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) rc = 1;
    }
    // (3) global_load_lds
    {
        unsigned *dS, *dO; CK(hipMalloc(&dS, 4096)); CK(hipMalloc(&dO, 4096));
        std::vector<unsigned> s(1024), o(1024);
        for (int i = 0; i < 1024; ++i) s[i] = i;
        CK(hipMemcpy(dS, s.data(), 4096, hipMemcpyHostToDevice));
        gldsprobe<<<1, 256>>>(dS, dO); CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dO, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
            int p = (l * 7) & 63;
            unsigned expect = (w * 64 + p) * 4 + q;     // LDS dest = base + lane*16 B holds that lane's source
            if (o[w * 256 + l * 4 + q] != expect) ++bad;
        }
        printf("[global_load_lds 16B: dest = wave base + lane*16] %s (bad=%d) first words: %u %u %u %u %u %u %u %u\n",
               bad ? "FAIL" : "PASS", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
        if (bad) rc = 1;
    }
    printf("probe rc=%d\n", rc);
    return rc;
}
