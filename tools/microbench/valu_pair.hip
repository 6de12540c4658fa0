// What shares a VALU issue slot on gfx950? Streams of two independent v_fma_f32 chains per wave (which pair: ~2.4 cycles per
// instruction per SIMD, tools/microbench/valu_rate.hip, table 5) with other instructions BETWEEN the two halves of a pair: a scalar ALU op,
// an LDS read, a side-pipe VALU op; and dependent / independent neighbours in both orders. Every scalar register the asm touches
// is an operand and SCC is declared clobbered (table 3.s k_mix_fma_salu clobbered s4/s5 and SCC behind the compiler's back: the
// loop's own compare lives in SCC, and the kernel never ended).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, NV, ASM)                                                                               \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float s)                             \
    {                                                                                                       \
        __shared__ float lds[512];                                                                          \
        lds[threadIdx.x] = s; lds[threadIdx.x + 256] = s;                                                   \
        __syncthreads();                                                                                    \
        float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, d0 = 0, d1 = 0;      \
        float t = s * 1.5f + threadIdx.x;                                                                   \
        unsigned s0 = (unsigned)iters, s1 = 7u;                                                             \
        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + threadIdx.x * 4u; \
        for (int it = 0; it < iters; ++it)                                                                  \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+v"(d0), "+v"(d1) : "v"(t), "v"(la) : "scc"); \
        float r = a0 + a1 + a2 + a3 + d0 + d1 + (float)(s0 + s1);                                           \
        if (r == 12345.678f) out[0] = r;                                                                    \
    }                                                                                                       \
    static const int NAME##_nv = NV;
// operands: %0-%3 fma chains, %4 %5 sgprs, %6 %7 scratch vgprs, %8 multiplier, %9 lds address
KERNEL(pair_mm,        2, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n")
KERNEL(dep_mm,         2, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %0, %0, %8, %0\n")
KERNEL(pair_m_salu_m,  2, "v_fma_f32 %0, %0, %8, %0\n s_add_u32 %4, %4, 1\n v_fma_f32 %1, %1, %8, %1\n")
KERNEL(pair_mm_salu,   2, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n s_add_u32 %4, %4, 1\n")
KERNEL(pair_m_2salu_m, 2, "v_fma_f32 %0, %0, %8, %0\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 3\n v_fma_f32 %1, %1, %8, %1\n")
KERNEL(pair_m_lds_m,   2, "v_fma_f32 %0, %0, %8, %0\n ds_read_b32 %6, %9\n v_fma_f32 %1, %1, %8, %1\n s_waitcnt lgkmcnt(0)\n")
KERNEL(pair_mm_lds,    2, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n ds_read_b32 %6, %9\n s_waitcnt lgkmcnt(0)\n")
KERNEL(pair_ms,        2, "v_fma_f32 %0, %0, %8, %0\n v_floor_f32 %1, %1\n")
KERNEL(pair_sm,        2, "v_floor_f32 %1, %1\n v_fma_f32 %0, %0, %8, %0\n")
KERNEL(pair_ss,        2, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n")
KERNEL(dep_ms,         2, "v_fma_f32 %0, %0, %8, %0\n v_floor_f32 %0, %0\n")
KERNEL(quad_mmss,      4, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n")
KERNEL(quad_msms,      4, "v_fma_f32 %0, %0, %8, %0\n v_floor_f32 %2, %2\n v_fma_f32 %1, %1, %8, %1\n v_floor_f32 %3, %3\n")
KERNEL(quad_dep_aabb,  4, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %1, %1, %8, %1\n")
KERNEL(quad_dep_abab,  4, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n")
KERNEL(tri_dep_aab,    3, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n")
KERNEL(pair_m_mix,     2, "v_fma_f32 %0, %0, %8, %0\n v_fma_mix_f32 %1, %1, %8, %1 op_sel_hi:[1,0,0]\n")
KERNEL(pair_m_sgprsrc, 2, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %4, %1\n")
KERNEL(pair_sg_sg,     2, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %5, %1\n")
typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t k, int nv)
{
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 4096, waves_per_simd = 4;
    const int blocks = 256 * waves_per_simd; // 4 waves per block, 1024 SIMDs: waves_per_simd waves on every SIMD
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) k<<<blocks, 256>>>(d, iters, 1.0001f); // a few ms of work first: the chip's clock ramps
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<<<blocks, 256>>>(d, iters, 1.0001f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double group_ns = ms * 1e6 / ((double)iters * 8 * waves_per_simd); // one ASM group of one wave, per SIMD
    printf("%-16s %6.2f cycles per group of %d VALU (%5.2f per VALU instruction) at a nominal 2.4 GHz, 4 waves per SIMD\n", name, group_ns * 2.4, nv, group_ns * 2.4 / nv);
    (void)hipFree(d);
}
#define RUN(N) run(#N, N, N##_nv)
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    RUN(pair_mm); RUN(dep_mm); RUN(pair_m_salu_m); RUN(pair_mm_salu); RUN(pair_m_2salu_m); RUN(pair_m_lds_m); RUN(pair_mm_lds);
    RUN(pair_ms); RUN(pair_sm); RUN(pair_ss); RUN(dep_ms); RUN(quad_mmss); RUN(quad_msms); RUN(quad_dep_aabb); RUN(quad_dep_abab);
    RUN(tri_dep_aab); RUN(pair_m_mix); RUN(pair_m_sgprsrc); RUN(pair_sg_sg);
    return 0;
}
