// Does vmcnt retire loads and stores in ISSUE ORDER on gfx950?  (the counted waits of a GEMM main loop that keeps LDS-DMA in flight
// across epilogue stores depend on it.)  Per lane: a slow load (cold line, HBM), then a store to a hot line, then s_waitcnt vmcnt(1):
// in-order retirement means the LOAD is done when the wait returns (only the store may be pending).  The loaded register is preset to a
// sentinel and copied out straight after the wait, all inside one asm block so the compiler adds no wait of its own.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/vmcnt_order.cpp -o /tmp/vmcnt_order && /tmp/vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* cold, unsigned* hot, unsigned* out, size_t stride_words, int variant) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned* src = cold + gid * stride_words;       // one cold 128-B line per lane: slow, many misses in flight
    unsigned* dst = hot + (gid & 1023);                    // L2-hot destination
    unsigned v = 0xDEADBEEFu, got;
    if (variant == 0) {          // load, store, vmcnt(1)
        asm volatile("global_load_dword %0, %2, off\n\t"
                     "global_store_dword %3, %4, off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %1, %0\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "+v"(v), "=v"(got) : "v"(src), "v"(dst), "v"((unsigned)gid) : "memory");
    } else {                     // store first (slow: cold line), then a fast load (hot), vmcnt(1): the STORE must be done -- cannot be observed
        asm volatile("global_load_dword %0, %2, off\n\t"   // from the wave; kept as the mirror case: 4 stores then the load, vmcnt(4)
                     "global_store_dword %3, %4, off\n\t"
                     "global_store_dword %3, %4, off offset:4\n\t"
                     "global_store_dword %3, %4, off offset:8\n\t"
                     "global_store_dword %3, %4, off offset:12\n\t"
                     "s_waitcnt vmcnt(4)\n\t"
                     "v_mov_b32 %1, %0\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "+v"(v), "=v"(got) : "v"(src), "v"(dst), "v"((unsigned)gid) : "memory");
    }
    out[gid] = got;
}
int main() {
    const int blocks = 4096, threads = 256;
    const size_t n = (size_t)blocks * threads, stride = 64;           // 256 B apart: every lane its own line, 268 MB of cold data
    unsigned *cold, *hot, *out;
    hipMalloc(&cold, n * stride * 4); hipMalloc(&hot, 1 << 20); hipMalloc(&out, n * 4);
    std::vector<unsigned> h(n * stride);
    for (size_t i = 0; i < n; ++i) h[i * stride] = (unsigned)i * 2654435761u + 1u;
    hipMemcpy(cold, h.data(), n * stride * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> o(n);
    for (int variant = 0; variant < 2; ++variant) {
        size_t stale = 0, wrong = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(out, 0, n * 4);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, cold, hot, out, stride, variant);
            hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < n; ++i) {
                if (o[i] == 0xDEADBEEFu) ++stale;
                else if (o[i] != (unsigned)i * 2654435761u + 1u) ++wrong;
            }
        }
        printf("variant %d (%s): %zu lanes x 20 launches, stale (load not retired at the counted wait) = %zu, wrong = %zu\n", variant,
               variant == 0 ? "load, store, vmcnt(1)" : "load, 4 stores, vmcnt(4)", n, stale, wrong);
    }
    return 0;
}
