// Extended per-instruction VALU issue-cost table for gfx950: cycles per wave-instruction per SIMD at 8 / 4 / 2 / 1 waves
// per SIMD (occupancy throttled with dynamic LDS). Four independent dependency chains per wave.
//   hipcc -O3 --offload-arch=gfx950 valu_rate3.hip -o valu_rate3 && ./valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float s, int si)       \
    {                                                                                         \
        extern __shared__ float pad[];                                                        \
        float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;        \
        float t = s * 1.5f + threadIdx.x;                                                     \
        for (int it = 0; it < iters; ++it) {                                                  \
            asm volatile(REP16(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "s"(si) : "vcc", "s4", "s5"); \
        }                                                                                     \
        float r = a0 + a1 + a2 + a3;                                                          \
        if (r == 12345.678f) out[0] = r + pad[0];                                             \
    }
#define I4(OP, ARGS0, ARGS1, ARGS2, ARGS3) OP " " ARGS0 "\n " OP " " ARGS1 "\n " OP " " ARGS2 "\n " OP " " ARGS3 "\n"
// unary: OP %i, %i ; binary with the shared VGPR operand %4
#define UN(OP) I4(OP, "%0, %0", "%1, %1", "%2, %2", "%3, %3")
#define BIN(OP) I4(OP, "%0, %0, %4", "%1, %1, %4", "%2, %2, %4", "%3, %3, %4")
#define TRI(OP) I4(OP, "%0, %0, %4, %0", "%1, %1, %4, %1", "%2, %2, %4, %2", "%3, %3, %4, %3")

KERNEL(k_fma, TRI("v_fma_f32"))
KERNEL(k_fmac, BIN("v_fmac_f32"))
KERNEL(k_mul, BIN("v_mul_f32"))
KERNEL(k_add, BIN("v_add_f32"))
KERNEL(k_sub, BIN("v_sub_f32"))
KERNEL(k_mov, I4("v_mov_b32", "%0, %4", "%1, %4", "%2, %4", "%3, %4"))
KERNEL(k_fma_sgpr, I4("v_fma_f32", "%0, %0, s4, %0", "%1, %1, s4, %1", "%2, %2, s4, %2", "%3, %3, s4, %3"))
KERNEL(k_mul_sgpr, I4("v_mul_f32", "%0, s4, %0", "%1, s4, %1", "%2, s4, %2", "%3, s4, %3"))
KERNEL(k_add_sgpr, I4("v_add_f32", "%0, s4, %0", "%1, s4, %1", "%2, s4, %2", "%3, s4, %3"))
KERNEL(k_fmac_sgpr, I4("v_fmac_f32", "%0, s4, %0", "%1, s4, %1", "%2, s4, %2", "%3, s4, %3"))
KERNEL(k_fma_inl, I4("v_fma_f32", "%0, %0, 0.5, %0", "%1, %1, 0.5, %1", "%2, %2, 0.5, %2", "%3, %3, 0.5, %3"))
KERNEL(k_mul_inl, I4("v_mul_f32", "%0, 0.5, %0", "%1, 0.5, %1", "%2, 0.5, %2", "%3, 0.5, %3"))
KERNEL(k_mul_lit, I4("v_mul_f32", "%0, 0x3f800123, %0", "%1, 0x3f800123, %1", "%2, 0x3f800123, %2", "%3, 0x3f800123, %3"))
KERNEL(k_fmaak, I4("v_fmaak_f32", "%0, %0, %4, 0x3f800123", "%1, %1, %4, 0x3f800123", "%2, %2, %4, 0x3f800123", "%3, %3, %4, 0x3f800123"))
KERNEL(k_fma_clamp, I4("v_fma_f32", "%0, %0, %4, 1.0 clamp", "%1, %1, %4, 1.0 clamp", "%2, %2, %4, 1.0 clamp", "%3, %3, %4, 1.0 clamp"))
KERNEL(k_fma_neg, I4("v_fma_f32", "%0, -%0, %4, %0", "%1, -%1, %4, %1", "%2, -%2, %4, %2", "%3, -%3, %4, %3"))
KERNEL(k_mul_e64_abs, I4("v_mul_f32_e64", "%0, |%0|, %4", "%1, |%1|, %4", "%2, |%2|, %4", "%3, |%3|, %4"))
KERNEL(k_max, BIN("v_max_f32"))
KERNEL(k_med3, I4("v_med3_f32", "%0, %0, 0, 1.0", "%1, %1, 0, 1.0", "%2, %2, 0, 1.0", "%3, %3, 0, 1.0"))
KERNEL(k_fract, UN("v_fract_f32"))
KERNEL(k_floor, UN("v_floor_f32"))
KERNEL(k_cvt_u32, UN("v_cvt_u32_f32"))
KERNEL(k_cvt_f32u, UN("v_cvt_f32_u32"))
KERNEL(k_cvt_f32h, UN("v_cvt_f32_f16"))
KERNEL(k_cvt_pkrtz, BIN("v_cvt_pkrtz_f16_f32"))
KERNEL(k_cvt_ub0, UN("v_cvt_f32_ubyte0"))
KERNEL(k_rcp, UN("v_rcp_f32"))
KERNEL(k_rsq, UN("v_rsq_f32"))
KERNEL(k_sqrt, UN("v_sqrt_f32"))
KERNEL(k_and, BIN("v_and_b32"))
KERNEL(k_or, BIN("v_or_b32"))
KERNEL(k_xor, BIN("v_xor_b32"))
KERNEL(k_addu, BIN("v_add_u32"))
KERNEL(k_subu, BIN("v_sub_u32"))
KERNEL(k_lshl, I4("v_lshlrev_b32", "%0, 3, %0", "%1, 3, %1", "%2, 3, %2", "%3, 3, %3"))
KERNEL(k_lshr, I4("v_lshrrev_b32", "%0, 3, %0", "%1, 3, %1", "%2, 3, %2", "%3, 3, %3"))
KERNEL(k_lshladd, I4("v_lshl_add_u32", "%0, %0, 3, %4", "%1, %1, 3, %4", "%2, %2, 3, %4", "%3, %3, 3, %4"))
KERNEL(k_addlshl, I4("v_add_lshl_u32", "%0, %0, %4, 3", "%1, %1, %4, 3", "%2, %2, %4, 3", "%3, %3, %4, 3"))
KERNEL(k_add3, TRI("v_add3_u32"))
KERNEL(k_mad24, TRI("v_mad_u32_u24"))
KERNEL(k_mullo, BIN("v_mul_lo_u32"))
KERNEL(k_bfe, I4("v_bfe_u32", "%0, %0, 8, 8", "%1, %1, 8, 8", "%2, %2, 8, 8", "%3, %3, 8, 8"))
KERNEL(k_perm, TRI("v_perm_b32"))
KERNEL(k_andor, TRI("v_and_or_b32"))
KERNEL(k_fmamix, I4("v_fma_mix_f32", "%0, %0, %4, %0 op_sel_hi:[1,0,0]", "%1, %1, %4, %1 op_sel_hi:[1,0,0]", "%2, %2, %4, %2 op_sel_hi:[1,0,0]", "%3, %3, %4, %3 op_sel_hi:[1,0,0]"))
KERNEL(k_dot2_f16, TRI("v_dot2_f32_f16"))
KERNEL(k_dot2c_f16, BIN("v_dot2c_f32_f16"))
KERNEL(k_pk_fma_f16, TRI("v_pk_fma_f16"))
KERNEL(k_pk_mul_f16, BIN("v_pk_mul_f16"))
KERNEL(k_pk_add_f16, BIN("v_pk_add_f16"))
KERNEL(k_fma_f16, TRI("v_fma_f16"))
KERNEL(k_cubeid, TRI("v_cubeid_f32"))
KERNEL(k_cubema, TRI("v_cubema_f32"))
KERNEL(k_cmp, I4("v_cmp_lt_f32", "vcc, %0, %4", "vcc, %1, %4", "vcc, %2, %4", "vcc, %3, %4"))
KERNEL(k_cnd, I4("v_cndmask_b32", "%0, %0, %4, vcc", "%1, %1, %4, vcc", "%2, %2, %4, vcc", "%3, %3, %4, vcc"))
KERNEL(k_sdwa_cvt, I4("v_cvt_f32_u32_sdwa", "%0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "%1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "%2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "%3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"))
KERNEL(k_dpp_mov, I4("v_mov_b32_dpp", "%0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "%1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "%2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "%3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
// mixes: alternate a full-rate and a half-rate instruction (do they overlap?)
KERNEL(k_mix_fma_cvt, "v_fma_f32 %0, %0, %4, %0\n v_cvt_f32_u32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_cvt_f32_u32 %3, %3\n")
KERNEL(k_mix_fma_rcp, "v_fma_f32 %0, %0, %4, %0\n v_rcp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(k_mix_fma_mixf, "v_fma_f32 %0, %0, %4, %0\n v_fma_mix_f32 %1, %1, %4, %1 op_sel_hi:[1,0,0]\n v_fma_f32 %2, %2, %4, %2\n v_fma_mix_f32 %3, %3, %4, %3 op_sel_hi:[1,0,0]\n")
KERNEL(k_mix_fma_salu, "v_fma_f32 %0, %0, %4, %0\n s_add_u32 s4, s4, 1\n v_fma_f32 %2, %2, %4, %2\n s_add_u32 s5, s5, 1\n")

typedef void (*kern_t)(float*, int, float, int);
static int g_waves[] = {8, 4, 2, 1};
static void run(const char* name, kern_t k)
{
    float* d; (void)hipMalloc(&d, 4);
    printf("%-14s", name); fflush(stdout);
    for (int wi = 0; wi < 4; ++wi) {
        const int w = g_waves[wi];           // waves per SIMD = workgroups (of 4 waves) per CU
        const int iters = 1024;
        const size_t lds = w == 8 ? 0 : (size_t)(160 * 1024 / w - 1024); // throttle workgroups per CU
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int blocks = 256 * w * 2;       // two rounds of resident workgroups
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<<<blocks, 256, lds>>>(d, 8, 1.0001f, 3);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<<<blocks, 256, lds>>>(d, iters, 1.0001f, 3);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double wave_instrs = (double)iters * 64 * w * 2; // per SIMD
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf(" [%s]", hipGetErrorString(e));
        printf("  w%d %6.2f", w, ms * 1e6 / wave_instrs * 2.4); fflush(stdout);
    }
    printf("\n");
    (void)hipFree(d);
}
#define RUN(K) run(#K, K)
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per wave-instruction per SIMD at a nominal 2.4 GHz (mix kernels: per instruction of the pair)\n");
    RUN(k_fma); RUN(k_fmac); RUN(k_mul); RUN(k_add); RUN(k_sub); RUN(k_mov);
    RUN(k_fma_sgpr); RUN(k_mul_sgpr); RUN(k_add_sgpr); RUN(k_fmac_sgpr); RUN(k_fma_inl); RUN(k_mul_inl); RUN(k_mul_lit); RUN(k_fmaak);
    RUN(k_fma_clamp); RUN(k_fma_neg); RUN(k_mul_e64_abs);
    RUN(k_max); RUN(k_med3); RUN(k_fract); RUN(k_floor); RUN(k_cvt_u32); RUN(k_cvt_f32u); RUN(k_cvt_f32h); RUN(k_cvt_pkrtz); RUN(k_cvt_ub0);
    RUN(k_rcp); RUN(k_rsq); RUN(k_sqrt);
    RUN(k_and); RUN(k_or); RUN(k_xor); RUN(k_addu); RUN(k_subu); RUN(k_lshl); RUN(k_lshr); RUN(k_lshladd); RUN(k_addlshl); RUN(k_add3); RUN(k_mad24); RUN(k_mullo);
    RUN(k_bfe); RUN(k_perm); RUN(k_andor);
    RUN(k_fmamix); RUN(k_dot2_f16); RUN(k_dot2c_f16); RUN(k_pk_fma_f16); RUN(k_pk_mul_f16); RUN(k_pk_add_f16); RUN(k_fma_f16);
    RUN(k_cubeid); RUN(k_cubema); RUN(k_cmp); RUN(k_cnd); RUN(k_sdwa_cvt); RUN(k_dpp_mov);
    RUN(k_mix_fma_cvt); RUN(k_mix_fma_rcp); RUN(k_mix_fma_mixf);
    return 0;
}
