// microbenchmark: register-staged global->LDS (global_load_dwordx4 -> ds_write_b128) fill bandwidth per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void k(const char* src, size_t fp_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * fp_bytes;
    size_t off = 0;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        uint4 r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            size_t o = (off + (size_t)(d * 4 + w) * 1024) & (fp_bytes - 1);
            r[d] = *reinterpret_cast<const uint4*>(base + o + l * 16);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<uint4*>(smem + (d * 4 + w) * 1024 + l * 16) = r[d];
        off += (size_t)DEPTH * 4096;
    }
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = ((float*)smem)[0] + acc.x;
}

template <int DEPTH>
void run(const char* src, size_t fp, int blocks_per_cu, const char* tag) {
    int iters = 2000 / DEPTH;
    size_t lds = (size_t)DEPTH * 4096;
    CK(hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<DEPTH><<<blocks, 256, lds>>>(src, fp, 10, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<DEPTH><<<blocks, 256, lds>>>(src, fp, iters, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = (double)blocks * iters * DEPTH * 4096.0;
    double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-30s depth=%2d blocks/CU=%d fp/block=%7zu KB  %7.2f TB/s  %6.1f B/clk/CU(@2.1GHz)\n", tag, DEPTH, blocks_per_cu, fp / 1024, tbs, tbs * 1e12 / 256 / 2.1e9);
}

int main() {
    size_t total = (size_t)2 << 30;
    char* src; CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
    run<4>(src, 32768, 1, "L2-res regstage");
    run<8>(src, 32768, 1, "L2-res regstage");
    run<4>(src, 32768, 2, "L2-res regstage");
    run<8>(src, 32768, 2, "L2-res regstage");
    run<16>(src, 32768, 2, "L2-res regstage");
    run<8>(src, 32768, 4, "L2-res regstage");
    run<8>(src, (size_t)2 << 20, 2, "HBM regstage");
    return 0;
}
