// microbenchmark: LDS-DMA fill bandwidth when the data is L2-resident but NOT L1-resident
// (every workgroup of an XCD walks the same 2 MiB region, 64 KiB apart from its neighbours -> L1 never hits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int DEPTH>
__global__ __launch_bounds__(256) void k(const char* src, size_t region, int iters, int shared_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const char* base = shared_per_xcd ? src + (size_t)xcd * region : src + (size_t)blockIdx.x * region;
    size_t off = shared_per_xcd ? ((size_t)idx * 65536) & (region - 1) : 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            size_t o = (off + (size_t)(d * 4 + w) * 1024) & (region - 1);
            __builtin_amdgcn_global_load_lds(GLB_PTR(base + o + l * 16), LDS_PTR(smem + (d * 4 + w) * 1024), 16, 0, 0);
        }
        off += (size_t)DEPTH * 4096;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int DEPTH>
void run(const char* src, size_t region, int bpc, int shared, const char* tag) {
    int iters = 4000 / DEPTH;
    CK(hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    int blocks = 256 * bpc;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<DEPTH><<<blocks, 256, DEPTH * 4096>>>(src, region, 20, shared);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<DEPTH><<<blocks, 256, DEPTH * 4096>>>(src, region, iters, shared);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double tbs = (double)blocks * iters * DEPTH * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-28s depth=%2d blocks/CU=%d region=%5zu KB inflight/CU=%3d KB: %6.2f TB/s\n", tag, DEPTH, bpc, region / 1024, DEPTH * 4 * bpc, tbs);
}

int main() {
    size_t total = (size_t)4 << 30;
    char* src; CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
    run<4>(src, 2 << 20, 2, 1, "L2-shared(2MB/XCD)");
    run<8>(src, 2 << 20, 2, 1, "L2-shared(2MB/XCD)");
    run<16>(src, 2 << 20, 2, 1, "L2-shared(2MB/XCD)");
    run<16>(src, 2 << 20, 1, 1, "L2-shared(2MB/XCD)");
    run<28>(src, 2 << 20, 1, 1, "L2-shared(2MB/XCD)");
    run<8>(src, 2 << 20, 4, 1, "L2-shared(2MB/XCD)");
    run<8>(src, 1 << 20, 2, 1, "L2-shared(1MB/XCD)");
    run<8>(src, 8 << 20, 2, 1, "L2-spill(8MB/XCD)");
    run<16>(src, 8 << 20, 2, 1, "L2-spill(8MB/XCD)");
    run<8>(src, 64 << 10, 2, 0, "private 64KB/block");
    run<8>(src, 128 << 10, 2, 0, "private 128KB/block");
    return 0;
}
