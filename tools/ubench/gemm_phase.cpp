// phase-timer harness for gemm_nt_kernel: where do the cycles of a K-step go?  (build with -DAMDSEG_PHASE_TIMERS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
unsigned long long* g_amdseg_dbg = nullptr;
#include "../../spokennlp_amd/csrc/gemm.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
int main() {
    const int M = 16384;
    int shapes[3][2] = {{2304, 768}, {768, 768}, {768, 3072}};
    for (auto& sh : shapes) {
        int N = sh[0], K = sh[1];
        bf16_t *A, *B, *C; float* bias;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4));
        CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(B, 0x3c, (size_t)N * K * 2)); CK(hipMemset(bias, 0, N * 4));
        int nb = (M / 128) * (N / 128);
        CK(hipMalloc(&g_amdseg_dbg, (size_t)nb * 64 * 8 + 4096)); CK(hipMemset(g_amdseg_dbg, 0, (size_t)nb * 64 * 8));
        for (int i = 0; i < 3; ++i) amdseg_gemm_nt_impl(A, K, B, K, C, N, M, N, K, 1, bias, nullptr, 0, nullptr, 0, 0, 0);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        amdseg_gemm_nt_impl(A, K, B, K, C, N, M, N, K, 1, bias, nullptr, 0, nullptr, 0, 0, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int nk = K / 64;
        if (M % 256 == 0 && N % 192 == 0) {
            int nbp = (M / 256) * (N / 192);
            std::vector<unsigned long long> h((size_t)nbp * 64);
            CK(hipMemcpy(h.data(), g_amdseg_dbg, h.size() * 8, hipMemcpyDeviceToHost));
            for (int g = 0; g < 2; ++g) {
                double s[5] = {0, 0, 0, 0, 0};
                for (int b = 0; b < nbp; ++b) for (int w = g * 4; w < g * 4 + 4; ++w) for (int j = 0; j < 5; ++j) s[j] += h[((size_t)b * 8 + w) * 8 + j];
                for (int j = 0; j < 5; ++j) s[j] /= (nbp * 4.0);
                printf("N=%d K=%d pp G%d: %.1f us (%.0f TF) | per wave: mem %.0f mfma %.0f barrier %.0f loop %.0f epilogue %.0f | per K-step: mem %.0f mfma %.0f bar %.0f (2 phases each)\n",
                       N, K, g, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, s[0], s[1], s[2], s[3], s[4], s[0] / nk, s[1] / nk, s[2] / nk);
            }
        } else {
        std::vector<unsigned long long> h((size_t)nb * 16);
        CK(hipMemcpy(h.data(), g_amdseg_dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < (size_t)nb * 4; ++i) for (int j = 0; j < 4; ++j) s[j] += h[i * 4 + j];
        for (int j = 0; j < 4; ++j) s[j] /= (nb * 4.0);
        printf("N=%d K=%d: %.1f us (%.0f TF) | per wave avg cycles: wait+barrier %.0f  compute %.0f  loop total %.0f  epilogue %.0f | per K-step: wait %.0f comp %.0f\n",
               N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, s[0], s[1], s[2], s[3], s[0] / nk, s[1] / nk);
        }
        hipFree(A); hipFree(B); hipFree(C); hipFree(bias); hipFree(g_amdseg_dbg);
    }
    return 0;
}
