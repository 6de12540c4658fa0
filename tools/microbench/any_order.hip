// Does hipExtAnyOrderLaunch let a kernel start beside the one in front of it on the SAME stream? K1: one workgroup busy for ~T us; K2: a
// full-chip kernel busy for ~T us. Sequential: ~2T + gap; overlapped: ~T.   hipcc --offload-arch=gfx950 -O2 any_order.hip -o any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned x = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) *sink = x;
}
int main()
{
    unsigned* sink; hipMalloc(&sink, 4);
    hipStream_t s; hipStreamCreate(&s);
    const unsigned long long T = 2000ull * 30; // ~30 us of s_memtime ticks at 100 MHz? (readcyclecounter = shader clock ~2 GHz: 60k cycles)
    for (int mode = 0; mode < 3; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            hipStreamSynchronize(s);
            auto a = std::chrono::steady_clock::now();
            for (int k = 0; k < 50; ++k) {
                hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s, T, sink);
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(255), dim3(1024), 0, s, T, sink);
                else hipExtLaunchKernelGGL(spin, dim3(255), dim3(1024), 0, s, nullptr, nullptr, mode == 1 ? 1 /*hipExtAnyOrderLaunch*/ : 0, T, sink);
            }
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / 50;
            if (us < best) best = us;
        }
        printf("%s: %.2f us per pair\n", mode == 0 ? "hipLaunchKernelGGL, hipLaunchKernelGGL" : (mode == 1 ? "second launch hipExtAnyOrderLaunch      " : "second launch hipExtLaunchKernel flags 0"), best);
    }
    return 0;
}
