// Pairing table: does a half-rate VALU instruction overlap with full-rate ones? (X alternating 1:1 with v_fma_f32) for gfx950: cycles per wave-instruction per SIMD at 8 / 4 / 2 / 1 waves
// per SIMD (occupancy throttled with dynamic LDS). Four independent dependency chains per wave.
//   hipcc -O3 --offload-arch=gfx950 valu_rate4.hip -o valu_rate4 && ./valu_rate4
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
            asm volatile(REP16(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "s"(si) : "vcc", "s4", "s5", "s6", "s7"); \
        }                                                                                     \
        float r = a0 + a1 + a2 + a3;                                                          \
        if (r == 12345.678f) out[0] = r + pad[0];                                             \
    }
#define I4(OP, ARGS0, ARGS1, ARGS2, ARGS3) OP " " ARGS0 "\n " OP " " ARGS1 "\n " OP " " ARGS2 "\n " OP " " ARGS3 "\n"
// unary: OP %i, %i ; binary with the shared VGPR operand %4
#define UN(OP) I4(OP, "%0, %0", "%1, %1", "%2, %2", "%3, %3")
#define BIN(OP) I4(OP, "%0, %0, %4", "%1, %1, %4", "%2, %2, %4", "%3, %3, %4")
#define TRI(OP) I4(OP, "%0, %0, %4, %0", "%1, %1, %4, %1", "%2, %2, %4, %2", "%3, %3, %4, %3")

KERNEL(p_fma, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(p_fma_sgpr, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, s4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, s4, %3\n")
KERNEL(p_mul_sgpr, "v_fma_f32 %0, %0, %4, %0\n v_mul_f32 %1, s4, %1\n v_fma_f32 %2, %2, %4, %2\n v_mul_f32 %3, s4, %3\n")
KERNEL(p_max, "v_fma_f32 %0, %0, %4, %0\n v_max_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %2\n v_max_f32 %3, %3, %4\n")
KERNEL(p_med3, "v_fma_f32 %0, %0, %4, %0\n v_med3_f32 %1, %1, 0, 1.0\n v_fma_f32 %2, %2, %4, %2\n v_med3_f32 %3, %3, 0, 1.0\n")
KERNEL(p_floor, "v_fma_f32 %0, %0, %4, %0\n v_floor_f32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_floor_f32 %3, %3\n")
KERNEL(p_fract, "v_fma_f32 %0, %0, %4, %0\n v_fract_f32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_fract_f32 %3, %3\n")
KERNEL(p_cvt_u32, "v_fma_f32 %0, %0, %4, %0\n v_cvt_u32_f32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_cvt_u32_f32 %3, %3\n")
KERNEL(p_cvt_f32u, "v_fma_f32 %0, %0, %4, %0\n v_cvt_f32_u32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_cvt_f32_u32 %3, %3\n")
KERNEL(p_cvt_f32h, "v_fma_f32 %0, %0, %4, %0\n v_cvt_f32_f16 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_cvt_f32_f16 %3, %3\n")
KERNEL(p_cvt_pkrtz, "v_fma_f32 %0, %0, %4, %0\n v_cvt_pkrtz_f16_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %2\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n")
KERNEL(p_lshl, "v_fma_f32 %0, %0, %4, %0\n v_lshlrev_b32 %1, 3, %1\n v_fma_f32 %2, %2, %4, %2\n v_lshlrev_b32 %3, 3, %3\n")
KERNEL(p_lshladd, "v_fma_f32 %0, %0, %4, %0\n v_lshl_add_u32 %1, %1, 3, %4\n v_fma_f32 %2, %2, %4, %2\n v_lshl_add_u32 %3, %3, 3, %4\n")
KERNEL(p_add3, "v_fma_f32 %0, %0, %4, %0\n v_add3_u32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_add3_u32 %3, %3, %4, %3\n")
KERNEL(p_mad24, "v_fma_f32 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3\n")
KERNEL(p_mullo, "v_fma_f32 %0, %0, %4, %0\n v_mul_lo_u32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %2\n v_mul_lo_u32 %3, %3, %4\n")
KERNEL(p_bfe, "v_fma_f32 %0, %0, %4, %0\n v_bfe_u32 %1, %1, 8, 8\n v_fma_f32 %2, %2, %4, %2\n v_bfe_u32 %3, %3, 8, 8\n")
KERNEL(p_fmamix, "v_fma_f32 %0, %0, %4, %0\n v_fma_mix_f32 %1, %1, %4, %1 op_sel_hi:[1,0,0]\n v_fma_f32 %2, %2, %4, %2\n v_fma_mix_f32 %3, %3, %4, %3 op_sel_hi:[1,0,0]\n")
KERNEL(p_dot2, "v_fma_f32 %0, %0, %4, %0\n v_dot2_f32_f16 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_dot2_f32_f16 %3, %3, %4, %3\n")
KERNEL(p_pk_fma_f16, "v_fma_f32 %0, %0, %4, %0\n v_pk_fma_f16 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_pk_fma_f16 %3, %3, %4, %3\n")
KERNEL(p_cubeid, "v_fma_f32 %0, %0, %4, %0\n v_cubeid_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_cubeid_f32 %3, %3, %4, %3\n")
KERNEL(p_cubema, "v_fma_f32 %0, %0, %4, %0\n v_cubema_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_cubema_f32 %3, %3, %4, %3\n")
KERNEL(p_cmp, "v_fma_f32 %0, %0, %4, %0\n v_cmp_lt_f32 vcc, %1, %4\n v_fma_f32 %2, %2, %4, %2\n v_cmp_lt_f32 vcc, %3, %4\n")
KERNEL(p_rcp, "v_fma_f32 %0, %0, %4, %0\n v_rcp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_rcp_f32 %3, %3\n")
KERNEL(p_rsq, "v_fma_f32 %0, %0, %4, %0\n v_rsq_f32 %1, %1\n v_fma_f32 %2, %2, %4, %2\n v_rsq_f32 %3, %3\n")
KERNEL(p_sdwa, "v_fma_f32 %0, %0, %4, %0\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_fma_f32 %2, %2, %4, %2\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
KERNEL(c_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL(c_cnd4, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL(c_cnd_e64, "v_cndmask_b32_e64 %0, %0, %4, s[6:7]\n v_cndmask_b32_e64 %1, %1, %4, s[6:7]\n v_cndmask_b32_e64 %2, %2, %4, s[6:7]\n v_cndmask_b32_e64 %3, %3, %4, s[6:7]\n")
KERNEL(t_rcp_3fma, "v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(q_cvt_3fma, "v_cvt_f32_u32 %0, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(q_sgpr_3fma, "v_fma_f32 %0, %0, s4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(h_cvt_max, "v_cvt_f32_u32 %0, %0\n v_max_f32 %1, %1, %4\n v_cvt_f32_u32 %2, %2\n v_max_f32 %3, %3, %4\n")
KERNEL(h_mix_cvt, "v_fma_mix_f32 %0, %0, %4, %0 op_sel_hi:[1,0,0]\n v_cvt_f32_u32 %1, %1\n v_fma_mix_f32 %2, %2, %4, %2 op_sel_hi:[1,0,0]\n v_cvt_f32_u32 %3, %3\n")

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
    printf("average cycles per wave-instruction per SIMD (pairs: v_fma_f32 alternating with X)\n");
    RUN(p_fma);
    RUN(p_fma_sgpr);
    RUN(p_mul_sgpr);
    RUN(p_max);
    RUN(p_med3);
    RUN(p_floor);
    RUN(p_fract);
    RUN(p_cvt_u32);
    RUN(p_cvt_f32u);
    RUN(p_cvt_f32h);
    RUN(p_cvt_pkrtz);
    RUN(p_lshl);
    RUN(p_lshladd);
    RUN(p_add3);
    RUN(p_mad24);
    RUN(p_mullo);
    RUN(p_bfe);
    RUN(p_fmamix);
    RUN(p_dot2);
    RUN(p_pk_fma_f16);
    RUN(p_cubeid);
    RUN(p_cubema);
    RUN(p_cmp);
    RUN(p_rcp);
    RUN(p_rsq);
    RUN(p_sdwa);
    RUN(c_cmp_cnd);
    RUN(c_cnd4);
    RUN(c_cnd_e64);
    RUN(t_rcp_3fma);
    RUN(q_cvt_3fma);
    RUN(q_sgpr_3fma);
    RUN(h_cvt_max);
    RUN(h_mix_cvt);
    return 0;
}
