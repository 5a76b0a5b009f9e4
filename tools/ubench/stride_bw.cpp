// microbenchmark: LDS-DMA fill bandwidth for GEMM-like strided tile reads: each wave instruction fetches 8 rows x 128 B,
// rows `ld` bytes apart; a workgroup walks a 256-row panel along k.  Does the row stride (leading dimension) matter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 512 threads; per K-step each wave issues 4 DMAs = 32 rows x 128 B -> block covers 256 rows x 128 B = 32 KiB per step
__global__ __launch_bounds__(512) void k(const char* src, size_t ld, int panels_per_xcd, int ksteps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int panel = xcd * panels_per_xcd + idx % panels_per_xcd;     // blocks of an XCD share panels (L2 reuse)
    const char* base = src + (size_t)panel * 256 * ld;
    for (int it = 0; it < iters; ++it)
        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = w * 32 + q * 8 + (l >> 3);
                __builtin_amdgcn_global_load_lds(GLB_PTR(base + (size_t)r * ld + (size_t)ks * 128 + (l & 7) * 16),
                                                 LDS_PTR(smem + ((ks & 1) * 32 + w * 4 + q) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void run(const char* src, size_t ld, int ksteps, int panels_per_xcd) {
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    int iters = 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<<<256, 512, 65536>>>(src, ld, panels_per_xcd, ksteps, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<<<256, 512, 65536>>>(src, ld, panels_per_xcd, ksteps, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double tbs = 256.0 * iters * ksteps * 32768.0 / (ms * 1e-3) / 1e12;
    printf("ld=%6zu B  ksteps=%3d panels/XCD=%2d (sharing %2dx): %6.2f TB/s\n", ld, ksteps, panels_per_xcd, 32 / panels_per_xcd, tbs);
}

int main() {
    size_t total = (size_t)4 << 30;
    char* src; CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
    size_t lds_[] = {1536, 1536 + 128, 4608, 4608 + 128, 6144, 6144 + 128, 6144 + 256, 8192, 8192 + 128};
    for (size_t ld : lds_) {
        int ks = (int)((ld / 128) < 48 ? ld / 128 : 48);
        run(src, ld, ks, 8);      // 8 panels per XCD, each read by 4 blocks
        run(src, ld, ks, 32);     // no sharing: every block its own panel
    }
    return 0;
}
