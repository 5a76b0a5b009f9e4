// probe: occupy H compute units for a bounded time (one 512-thread workgroup with 128 KiB of LDS per CU, spinning on the 100-MHz real-time counter),
// the way an RCCL ring's channels do while an all-reduce overlaps backward -- run beside bench.py to see what the step loses when a GEMM grid
// sized for 256 CUs finds fewer (tools/dbg/cu_hog.sh).  usage: cu_hog <workgroups> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void hog(unsigned long long ticks, unsigned* sink) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = (char)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); ++n; }
    if (n == 0xffffffffu) sink[0] = smem[(threadIdx.x + 1) & 511];
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 16;
    const double secs = argc > 2 ? atof(argv[2]) : 10.0;
    if (secs > 60.0) return 2;
    unsigned* sink; if (hipMalloc(&sink, 4) != hipSuccess) return 3;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hog), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return 4;
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    if (rate <= 0) rate = 100000;
    hipLaunchKernelGGL(hog, dim3(wgs), dim3(512), 131072, 0, (unsigned long long)(secs * rate * 1000.0), sink);
    printf("hog: %d workgroups for %.1f s (wall clock %d kHz)\n", wgs, secs, rate); fflush(stdout);
    hipError_t e = hipDeviceSynchronize();
    printf("hog done: %s\n", hipGetErrorString(e));
    return 0;
}
