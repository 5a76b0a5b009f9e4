// Launch timing of the dominant kernels INSIDE real training steps (measurement plumbing for bench.py's `roofline`;
// include/amdseg.h: amdseg_prof_*).  A profiled launch goes through hipExtLaunchKernelGGL with a start and a stop event: the two
// events are filled from the dispatch packet's OWN completion-signal timestamps (what rocprofv3 --kernel-trace reads), so the span
// is the kernel's, without the marker-to-marker gap that a hipEventRecord pair around a launch adds (~5 us on this stack, 5-10 % of
// a 60 us GEMM).  Profiling off (the default): the plain launch, nothing else.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

// host side (prof.hip): events of the next launch of class `cls` doing `work` algorithmic FLOPs; false when profiling is off
bool amdseg_prof_events(int cls, double work, hipEvent_t* start, hipEvent_t* stop);

#define AMDSEG_LAUNCH_PROF(cls, work, kernel, grid, block, lds, stream, ...)                                    \
    do {                                                                                                          \
        hipEvent_t pe0_, pe1_;                                                                                    \
        if (amdseg_prof_events((cls), (work), &pe0_, &pe1_))                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pe0_, pe1_, 0, __VA_ARGS__);                  \
        else                                                                                                      \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                    \
    } while (0)
