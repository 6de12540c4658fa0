// Per-instruction VALU cost table for gfx950 at 8 waves/SIMD (cycles per wave-instruction per SIMD, nominal 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float s, int si)       \
    {                                                                                         \
        float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;        \
        for (int it = 0; it < iters; ++it) {                                                  \
            asm volatile(REP16(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "s"(si) : "vcc", "s4", "s5"); \
        }                                                                                     \
        float r = a0 + a1 + a2 + a3;                                                          \
        if (r == 12345.678f) out[0] = r;                                                      \
    }

// each ASM string = 4 independent instructions (one per accumulator)
KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")
KERNEL(k_add, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
KERNEL(k_max, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n")
KERNEL(k_med3, "v_med3_f32 %0, %0, %4, 1.0\n v_med3_f32 %1, %1, %4, 1.0\n v_med3_f32 %2, %2, %4, 1.0\n v_med3_f32 %3, %3, %4, 1.0\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %4, s[4:5]\n v_cndmask_b32_e64 %1, %1, %4, s[4:5]\n v_cndmask_b32_e64 %2, %2, %4, s[4:5]\n v_cndmask_b32_e64 %3, %3, %4, s[4:5]\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL(k_floor, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n")
KERNEL(k_cvt_i32, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n")
KERNEL(k_cvt_f32u, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")
KERNEL(k_cvt_f16, "v_cvt_f16_f32 %0, %0\n v_cvt_f16_f32 %1, %1\n v_cvt_f16_f32 %2, %2\n v_cvt_f16_f32 %3, %3\n")
KERNEL(k_cvt_f32h, "v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n")
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3\n")
KERNEL(k_rsq, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
KERNEL(k_and, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n")
KERNEL(k_addu, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3\n")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n")
KERNEL(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
KERNEL(k_fmamix, "v_fma_mix_f32 %0, %0, %4, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %4, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %4, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %4, %3 op_sel_hi:[1,0,0]\n")
KERNEL(k_fma_sgpr, "v_fma_f32 %0, %0, s4, %0\n v_fma_f32 %1, %1, s4, %1\n v_fma_f32 %2, %2, s4, %2\n v_fma_f32 %3, %3, s4, %3\n")
KERNEL(k_fmac_lit, "v_fmac_f32 %0, 0x3f800123, %0\n v_fmac_f32 %1, 0x3f800123, %1\n v_fmac_f32 %2, 0x3f800123, %2\n v_fmac_f32 %3, 0x3f800123, %3\n")

typedef void (*kern_t)(float*, int, float, int);
static void run(const char* name, kern_t k)
{
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 2048, blocks = 256 * 8;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<<<blocks, 256>>>(d, 8, 1.0001f, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<<<blocks, 256>>>(d, iters, 1.0001f, 3);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double wave_instrs = (double)iters * 64 * 8; // per SIMD: 8 waves x 64 instr/iter
    printf("%-12s %.2f cycles/instr/SIMD @2.4GHz\n", name, ms * 1e6 / wave_instrs * 2.4);
    (void)hipFree(d);
}
#define RUN(K) run(#K, K)
int main()
{
    RUN(k_fma); RUN(k_mul); RUN(k_add); RUN(k_max); RUN(k_med3); RUN(k_cnd_vcc); RUN(k_cnd_sgpr); RUN(k_cmp); RUN(k_cmp_cnd);
    RUN(k_floor); RUN(k_cvt_i32); RUN(k_cvt_f32u); RUN(k_cvt_f16); RUN(k_cvt_f32h); RUN(k_cvt_ub); RUN(k_rsq); RUN(k_rcp); RUN(k_sqrt); RUN(k_exp);
    RUN(k_and); RUN(k_lshl); RUN(k_addu); RUN(k_mad24); RUN(k_mullo); RUN(k_lshladd); RUN(k_bfe); RUN(k_mov); RUN(k_fmamix); RUN(k_fma_sgpr); RUN(k_fmac_lit);
    return 0;
}
