// microbenchmark: global_load_lds (LDS-DMA) fill bandwidth per CU vs in-flight depth / footprint / row width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// each wave issues DEPTH 1-KiB DMAs, waits for all, repeats ITERS times. footprint per block = fp bytes (wraps)
template <int DEPTH, int MODE>
__global__ __launch_bounds__(256) void k(const char* src, size_t fp_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * fp_bytes;
    size_t off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            size_t o = (off + (size_t)(d * 4 + w) * 1024) & (fp_bytes - 1);
            const char* g;
            if (MODE == 0) g = base + o + l * 16;                                // 1 KiB contiguous (8 x 128 B lines)
            else if (MODE == 1) g = base + (o & ~(size_t)1023) * 1 + (l >> 3) * 128 + (l & 7) * 16 ;  // same as 0
            else g = base + (o & (fp_bytes / 2 - 1)) * 2 + (l >> 2) * 128 + (l & 3) * 16;   // 64-B half rows (16 lines per instr)
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(smem + (d * 4 + w) * 1024), 16, 0, 0);
        }
        off += (size_t)DEPTH * 4096;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = ((float*)smem)[0];
}

template <int DEPTH, int MODE>
void run(const char* src, size_t fp, int blocks_per_cu, const char* tag) {
    int iters = 2000 / DEPTH;
    size_t lds = (size_t)DEPTH * 4096;
    CK(hipFuncSetAttribute((const void*)k<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<DEPTH, MODE><<<blocks, 256, lds>>>(src, fp, 10, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<DEPTH, MODE><<<blocks, 256, lds>>>(src, fp, iters, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = (double)blocks * iters * DEPTH * 4096.0;
    double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s depth=%2d blocks/CU=%d fp/block=%7zu KB  %7.2f TB/s  %6.1f B/clk/CU(@2.1GHz)  inflight/CU=%4zu KB\n", tag, DEPTH, blocks_per_cu, fp / 1024, tbs,
           tbs * 1e12 / 256 / 2.1e9, (size_t)DEPTH * 4 * blocks_per_cu);
}

int main() {
    size_t total = (size_t)2 << 30;
    char* src; CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
    // L2-resident: 64 KB per block (2 blocks/CU * 32 CUs * 64 KB = 4 MB per XCD .. borderline) -> use 32 KB
    run<4, 0>(src, 32768, 2, "L2-res 1KiB rows");
    run<8, 0>(src, 32768, 2, "L2-res 1KiB rows");
    run<16, 0>(src, 32768, 2, "L2-res 1KiB rows");
    run<8, 0>(src, 32768, 1, "L2-res 1KiB rows");
    run<16, 0>(src, 32768, 1, "L2-res 1KiB rows");
    run<8, 0>(src, 32768, 4, "L2-res 1KiB rows");
    run<8, 2>(src, 32768, 2, "L2-res 64B half rows");
    run<16, 2>(src, 32768, 2, "L2-res 64B half rows");
    // MALL/HBM: 2 MB per block (1 GB total footprint > MALL)
    run<8, 0>(src, (size_t)2 << 20, 2, "HBM-stream 1KiB rows");
    run<16, 0>(src, (size_t)2 << 20, 2, "HBM-stream 1KiB rows");
    run<16, 0>(src, (size_t)2 << 20, 4, "HBM-stream 1KiB rows");
    // MALL-resident: 256 KB per block * 512 blocks = 128 MB
    run<8, 0>(src, (size_t)256 << 10, 2, "MALL-res 1KiB rows");
    run<16, 0>(src, (size_t)256 << 10, 2, "MALL-res 1KiB rows");
    return 0;
}
