// Dependent-chain issue rate on gfx950: N independent v_fma_f32 chains per wave (1, 2, 4), at 8/4/2/1 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float s)               \
    {                                                                                         \
        extern __shared__ float pad[];                                                        \
        float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;        \
        float t = s * 1.5f + threadIdx.x;                                                     \
        for (int it = 0; it < iters; ++it) asm volatile(REP16(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t)); \
        float r = a0 + a1 + a2 + a3;                                                          \
        if (r == 12345.678f) out[0] = r + pad[0];                                             \
    }
KERNEL(chain1, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n")
KERNEL(chain2, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n")
KERNEL(chain4, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(chain1_cvt, "v_cvt_f32_u32 %0, %0\n v_fma_f32 %0, %0, %4, %0\n v_cvt_u32_f32 %0, %0\n v_fma_f32 %0, %0, %4, %0\n")
KERNEL(chain1_mix, "v_fma_mix_f32 %0, %0, %4, %0 op_sel_hi:[1,0,0]\n v_fma_f32 %0, %0, %4, %0\n v_fma_mix_f32 %0, %0, %4, %0 op_sel_hi:[1,0,0]\n v_fma_f32 %0, %0, %4, %0\n")
KERNEL(chain1_rcp, "v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n")
typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t k)
{
    float* d; (void)hipMalloc(&d, 4);
    printf("%-12s", name);
    const int ws[] = {8, 4, 2, 1};
    for (int w : ws) {
        const int iters = 1024;
        const size_t lds = w == 8 ? 0 : (size_t)(160 * 1024 / w - 1024);
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int blocks = 256 * w * 2;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<<<blocks, 256, lds>>>(d, 8, 1.0001f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<<<blocks, 256, lds>>>(d, iters, 1.0001f);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("  w%d %6.2f", w, ms * 1e6 / ((double)iters * 64 * w * 2) * 2.4);
        fflush(stdout);
    }
    printf("\n");
    (void)hipFree(d);
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per wave-instruction per SIMD (nominal 2.4 GHz)\n");
    run("chain1", chain1); run("chain2", chain2); run("chain4", chain4); run("chain1_cvt", chain1_cvt); run("chain1_mix", chain1_mix); run("chain1_rcp", chain1_rcp);
    return 0;
}
