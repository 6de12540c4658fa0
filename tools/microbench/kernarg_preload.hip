// Does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N) shorten a launch on this stack?
// One-workgroup kernels launched back to back on one stream: a dependent load chain that starts from a kernel argument,
// i.e. the shape of this path's small launches (cull of a few commands, HZB tail). Build twice:
//   hipcc -O3 --offload-arch=gfx950 -o kp0 kernarg_preload.hip
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -o kp16 kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void chain(const float* in, float* out, const float* in2, int n, int m, float s, float t, int pad0, int pad1)
{
    const int i = threadIdx.x;
    if (i < n) {
        const float a = in[i];
        const float b = in2[(int)a + (i & m)];
        out[i] = a * s + b * t + (float)(pad0 + pad1);
    }
}

__global__ __launch_bounds__(256) void empty(float* out, int n) { if (n < 0) out[0] = 1.0f; }

int main()
{
    float *in, *in2, *out;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&in2, 4096)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(in, 0, 4096)); CK(hipMemset(in2, 0, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int which = 0; which < 2; ++which) {
        auto launch = [&]() {
            if (which == 0) hipLaunchKernelGGL(empty, dim3(1), dim3(256), 0, st, out, 1);
            else hipLaunchKernelGGL(chain, dim3(1), dim3(256), 0, st, in, out, in2, 256, 15, 1.0f, 2.0f, 0, 0);
        };
        for (int k = 0; k < 2000; ++k) launch();
        CK(hipStreamSynchronize(st));
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, st));
            for (int k = 0; k < 4000; ++k) launch();
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            t.push_back(ms * 1e3f / 4000);
        }
        std::sort(t.begin(), t.end());
        printf("%-28s %6.2f us per launch (median of 7 x 4000 back-to-back launches; min %.2f)\n", which == 0 ? "empty kernel" : "two dependent loads", t[3], t[0]);
    }
    return 0;
}
