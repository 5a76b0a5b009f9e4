// ablation harness for the ping-pong GEMM: which of {DMA, LDS fragment reads, MFMA} sets the K-step time?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
unsigned long long* g_amdseg_dbg = nullptr;
#include "../../spokennlp_amd/csrc/gemm.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
int main() {
    const int M = 16384;
    int shapes[2][2] = {{768, 3072}, {2304, 768}};
    for (auto& sh : shapes) {
        int N = sh[0], K = sh[1];
        bf16_t *A, *B, *C;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(B, 0x3c, (size_t)N * K * 2));
#ifdef ABL_RANDOM
        { std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K); srand(1);
          for (auto& x : hA) x = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
          for (auto& x : hB) x = (unsigned short)(0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
          CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice)); }
#endif
        if (!g_amdseg_dbg) { CK(hipMalloc(&g_amdseg_dbg, 8192)); CK(hipMemset(g_amdseg_dbg, 0, 8192)); }
        for (int i = 0; i < 3; ++i) amdseg_gemm_nt_impl(A, K, B, K, C, N, M, N, K, 0, nullptr, nullptr, 0, nullptr, 0, 0, 0);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) amdseg_gemm_nt_impl(A, K, B, K, C, N, M, N, K, 0, nullptr, nullptr, 0, nullptr, 0, 0, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%s N=%d K=%d: %.1f us (%.0f TF-equivalent)", ABL_NAME, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
#ifdef AMDSEG_CLOCK_PROBE
        {
            unsigned long long h[512];
            CK(hipMemcpy(h, g_amdseg_dbg, sizeof(h), hipMemcpyDeviceToHost));
            double c = 0, r = 0; for (int i = 0; i < 256; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
            printf("  | shader clock during kernel: %.0f MHz (cycles %.0f / realtime ticks %.0f @100MHz)", c / r * 100.0, c / 256, r / 256);
        }
#endif
        printf("\n");
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
