// A binder of include/amdseg.h WITHOUT Python or PyTorch: plain HIP runtime + the C ABI of libamdseg.so.
// What the drop-in boundary promises: extern "C" entry points over raw device pointers, a stream per call, int return codes,
// caller-owned buffers.  This program allocates with hipMalloc, runs one projection GEMM with its bias epilogue (the torch.nn.Linear of
// [hf] models/bert/modeling_bert.py:175-177), checks sampled outputs against a double-precision CPU product, exercises the error path, and runs the
// RCCL-backed gradient exchange context with a world of one rank.  Built by __graft_entry__.build(); run by tests/test_gpu_kernels.py.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/cabi_demo.cpp -Iinclude -Lspokennlp_amd -lamdseg -Wl,-rpath,'$ORIGIN/../spokennlp_amd' -o tools/cabi_demo.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "amdseg.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    printf("amdseg ABI %d\n", amdseg_abi_version());
    if (amdseg_abi_version() != AMDSEG_ABI_VERSION) { printf("FAIL: header / library ABI mismatch\n"); return 1; }
    const int M = 1024, N = 768, K = 768;
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    std::vector<float> hbias(N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd() * 0.1f);
    for (auto& v : hbias) v = rnd();
    void *dA, *dB, *dC; float* dbias;
    HIPCHK(hipMalloc(&dA, hA.size() * 2)); HIPCHK(hipMalloc(&dB, hB.size() * 2)); HIPCHK(hipMalloc(&dC, hC.size() * 2));
    HIPCHK(hipMalloc((void**)&dbias, N * 4));
    HIPCHK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    hipStream_t st; HIPCHK(hipStreamCreate(&st));
    int rc = amdseg_gemm_nt(dA, K, dB, K, dC, N, M, N, K, AMDSEG_EPI_BIAS, dbias, nullptr, 0, nullptr, 0, 0, st);
    if (rc != AMDSEG_OK) { printf("FAIL amdseg_gemm_nt: %d %s\n", rc, amdseg_error_string(rc)); return 1; }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 2000; ++t) {
        const int m = (t * 7919) % M, n = (t * 104729) % N;
        double acc = hbias[n];
        for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hB[(size_t)n * K + k]);
        const double d = fabs(acc - (double)bf2f(hC[(size_t)m * N + n])) / fmax(1.0, fabs(acc));
        if (d > worst) worst = d;
    }
    printf("gemm_nt + bias: worst relative difference to a double-precision product over 2000 samples %.3g (bf16 output rounding 3.9e-3)\n", worst);
    if (!(worst < 8e-3)) { printf("FAIL: GEMM result\n"); return 1; }
    // argument errors are codes with a message
    rc = amdseg_gemm_nt(dA, K, dB, K, dC, N, M + 1, N, K, AMDSEG_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 0, st);
    printf("misaligned shape -> %d (%s)\n", rc, amdseg_error_string(rc));
    if (rc != AMDSEG_ERR_SHAPE) { printf("FAIL: expected AMDSEG_ERR_SHAPE\n"); return 1; }
    // the gradient exchange behind its explicit context, world of one rank
    char uid[AMDSEG_COMM_ID_BYTES];
    rc = amdseg_allreduce_unique_id(uid);
    if (rc != AMDSEG_OK) { printf("FAIL unique_id: %d %s\n", rc, amdseg_error_string(rc)); return 1; }
    amdseg_comm* comm = nullptr;
    rc = amdseg_allreduce_init(&comm, uid, 0, 1);
    if (rc != AMDSEG_OK) { printf("FAIL init: %d %s\n", rc, amdseg_error_string(rc)); return 1; }
    float* g; const size_t ng = 1 << 20;
    HIPCHK(hipMalloc((void**)&g, ng * 4));
    std::vector<float> hg(ng); for (auto& v : hg) v = rnd();
    HIPCHK(hipMemcpyAsync(g, hg.data(), ng * 4, hipMemcpyHostToDevice, st));
    rc = amdseg_allreduce_bucket(comm, g, ng, AMDSEG_F32, st);
    if (rc == AMDSEG_OK) rc = amdseg_allreduce_wait(comm, st);
    if (rc != AMDSEG_OK) { printf("FAIL allreduce: %d %s\n", rc, amdseg_error_string(rc)); return 1; }
    std::vector<float> back(ng);
    HIPCHK(hipMemcpyAsync(back.data(), g, ng * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (memcmp(back.data(), hg.data(), ng * 4) != 0) { printf("FAIL: a one-rank sum must return its input\n"); return 1; }
    rc = amdseg_allreduce_destroy(comm);
    printf("allreduce (world 1): identity, destroy -> %d\n", rc);
    // the explicit library context (ABI 13): created, set, bound for a cfg-less call, destroyed -- all state the library keeps is in it
    amdseg_ctx* cx = nullptr;
    rc = amdseg_ctx_create(&cx);
    if (rc != AMDSEG_OK || !cx) { printf("FAIL ctx_create: %d\n", rc); return 1; }
    if (amdseg_ctx_set_cu_budget(cx, 240) != 0 || amdseg_ctx_cu_budget(cx) != 240) { printf("FAIL ctx budget\n"); return 1; }
    amdseg_ctx_prof_enable(cx, 1); amdseg_ctx_prof_reset(cx);
    amdseg_ctx_bind(cx);
    rc = amdseg_gemm_nt(dA, K, dB, K, dC, N, M, N, K, AMDSEG_EPI_BIAS, dbias, nullptr, 0, nullptr, 0, 0, st);     // timed by THIS context's launch timer
    amdseg_ctx_bind(nullptr);
    double us = 0, work = 0; long long launches = 0;
    int rc2 = amdseg_ctx_prof_read(cx, AMDSEG_PROF_GEMM_NT, &us, &work, &launches);
    printf("ctx: budget 240, bound gemm rc %d, launch timer read rc %d: %lld launch(es), %.1f us\n", rc, rc2, launches, us);
    if (rc != AMDSEG_OK || rc2 != AMDSEG_OK) return 1;
    amdseg_ctx_destroy(cx);
    printf("CABI_DEMO_OK\n");
    return 0;
}
