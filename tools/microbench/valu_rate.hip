// VALU issue-rate microbenchmark for gfx950: plain v_fma_f32 vs v_pk_fma_f32 vs transcendental, at 1..8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
    float a[16];
    float2v p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = float2v{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(float2v{s, s}));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(float2v{s, s}));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %0, %1, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(s));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const char* name, int instr_per_iter, int blocks_per_cu)
{
    float* d; hipMalloc(&d, 4);
    const int iters = 4096, cus = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<cus * blocks_per_cu, 256>>>(d, 16, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<cus * blocks_per_cu, 256>>>(d, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: blocks_per_cu waves (256 threads = 4 waves = 1 per SIMD)
    const double wave_instrs = (double)iters * instr_per_iter * blocks_per_cu;
    const double ns_per_instr = ms * 1e6 / wave_instrs;
    printf("%-16s waves/SIMD=%d  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, blocks_per_cu, ns_per_instr, ns_per_instr * 2.4);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 16, w);
        run<1>("v_pk_fma_f32", 8, w);
        run<4>("v_pk_mul_f32", 8, w);
        run<2>("v_rcp_f32", 16, w);
        run<3>("v_cndmask_b32", 16, w);
        run<5>("v_cvt_f32_f16", 16, w);
        run<6>("v_fma_mix_f32", 16, w);
    }
    return 0;
}
