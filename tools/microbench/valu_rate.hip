// gfx950 VALU issue-rate tables (round 1-2 microbenchmarks, one program): what a vector instruction costs the issue port, alone and
// beside its neighbours. `valu_rate <table>` runs one of them, `valu_rate` all seven:
//   1  plain v_fma_f32 / v_pk_fma_f32 / transcendental / cndmask / ... at 1..8 waves per SIMD
//   2  the same instruction classes in mixed streams
//   3  per-instruction cost of the lighting loop's instruction classes
//   4  pairing with an FMA (what shares an issue slot), SGPR operands, v_fma_mix, SDWA
//   5  dependent chains
//   6, 7  operand-bank parity of the source registers
// Results: profiles/r01_microbench_valu_rate*.txt, profiles/r02_microbench_valu_rate_packed.txt.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate [table]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// ===== table 1 =====
// VALU issue-rate microbenchmark for gfx950: plain v_fma_f32 vs v_pk_fma_f32 vs transcendental, at 1..8 waves/SIMD.
namespace table1 {
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

int run()
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
} // namespace table1

// ===== table 2 =====
// Per-instruction VALU cost table for gfx950 at 8 waves/SIMD (cycles per wave-instruction per SIMD, nominal 2.4 GHz).
namespace table2 {
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
int run()
{
    RUN(k_fma); RUN(k_mul); RUN(k_add); RUN(k_max); RUN(k_med3); RUN(k_cnd_vcc); RUN(k_cnd_sgpr); RUN(k_cmp); RUN(k_cmp_cnd);
    RUN(k_floor); RUN(k_cvt_i32); RUN(k_cvt_f32u); RUN(k_cvt_f16); RUN(k_cvt_f32h); RUN(k_cvt_ub); RUN(k_rsq); RUN(k_rcp); RUN(k_sqrt); RUN(k_exp);
    RUN(k_and); RUN(k_lshl); RUN(k_addu); RUN(k_mad24); RUN(k_mullo); RUN(k_lshladd); RUN(k_bfe); RUN(k_mov); RUN(k_fmamix); RUN(k_fma_sgpr); RUN(k_fmac_lit);
    return 0;
}
} // namespace table2

// ===== table 3 =====
// Extended per-instruction VALU issue-cost table for gfx950: cycles per wave-instruction per SIMD at 8 / 4 / 2 / 1 waves
// per SIMD (occupancy throttled with dynamic LDS). Four independent dependency chains per wave.
//   (table 3 of valu_rate)
namespace table3 {
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
int run()
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
} // namespace table3

// ===== table 4 =====
// Pairing table: does a half-rate VALU instruction overlap with full-rate ones? (X alternating 1:1 with v_fma_f32) for gfx950: cycles per wave-instruction per SIMD at 8 / 4 / 2 / 1 waves
// per SIMD (occupancy throttled with dynamic LDS). Four independent dependency chains per wave.
//   (table 4 of valu_rate)
namespace table4 {
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
int run()
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
} // namespace table4

// ===== table 5 =====
// Dependent-chain issue rate on gfx950: N independent v_fma_f32 chains per wave (1, 2, 4), at 8/4/2/1 waves per SIMD.
namespace table5 {
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
int run()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per wave-instruction per SIMD (nominal 2.4 GHz)\n");
    run("chain1", chain1); run("chain2", chain2); run("chain4", chain4); run("chain1_cvt", chain1_cvt); run("chain1_mix", chain1_mix); run("chain1_rcp", chain1_rcp);
    return 0;
}
} // namespace table5

// ===== table 6 =====
// Which operand layouts let two adjacent independent v_fma_f32 of one wave share an issue slot on gfx950?
// Two alternating chains with EXPLICIT registers: chain A = v[DA] <- v[DA] * v[SA] + v[DA], chain B likewise.
namespace table6 {
#define REP8(X) X X X X X X X X
#define STR(x) #x
#define XSTR(x) STR(x)
// D0,S0 / D1,S1 register numbers (>= 40 to stay clear of the compiler's own registers; all are clobbered)
#define KERNEL(NAME, D0, S0, D1, S1)                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float s)                                \
    {                                                                                                          \
        asm volatile("v_mov_b32 v" XSTR(D0) ", %0\n v_mov_b32 v" XSTR(D1) ", %0\n v_mov_b32 v" XSTR(S0) ", %1\n v_mov_b32 v" XSTR(S1) ", %1" \
                     :: "v"(threadIdx.x * 0.001f + 1.0f), "v"(s) : "v" XSTR(D0), "v" XSTR(D1), "v" XSTR(S0), "v" XSTR(S1)); \
        for (int it = 0; it < iters; ++it)                                                                     \
            asm volatile(REP8(REP8("v_fma_f32 v" XSTR(D0) ", v" XSTR(D0) ", v" XSTR(S0) ", v" XSTR(D0) "\n v_fma_f32 v" XSTR(D1) ", v" XSTR(D1) ", v" XSTR(S1) ", v" XSTR(D1) "\n")) \
                         ::: "v" XSTR(D0), "v" XSTR(D1), "v" XSTR(S0), "v" XSTR(S1));                           \
        float r;                                                                                               \
        asm volatile("v_add_f32 %0, v" XSTR(D0) ", v" XSTR(D1) : "=v"(r) :: "v" XSTR(D0), "v" XSTR(D1));       \
        if (r == 12345.678f) out[0] = r;                                                                       \
    }
KERNEL(d40_41_s48_49, 40, 48, 41, 49)  // dst banks 0,1  src banks 0,1
KERNEL(d40_42_s48_50, 40, 48, 42, 50)  // dst banks 0,2  src banks 0,2
KERNEL(d40_44_s48_52, 40, 48, 44, 52)  // dst banks 0,0  src banks 0,0  (everything in one bank)
KERNEL(d40_44_s49_53, 40, 49, 44, 53)  // dst 0,0  src 1,1
KERNEL(d40_41_s48_48, 40, 48, 41, 48)  // shared source register
KERNEL(d40_41_s50_51, 40, 50, 41, 51)  // dst 0,1 src 2,3
KERNEL(d40_45_s50_55, 40, 50, 45, 55)  // dst 0,1 src 2,3 (far apart)
KERNEL(d40_43_s41_42, 40, 41, 43, 42)  // dst 0,3 src 1,2
KERNEL(d40_41_s44_45, 40, 44, 41, 45)  // dst 0,1 src 0,1 (src shares the dst's bank)
KERNEL(d40_41_s45_44, 40, 45, 41, 44)  // dst 0,1 src 1,0 (crossed)
typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t k)
{
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 256, w = 8, blocks = 256 * w * 2;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) { k<<<blocks, 256>>>(d, iters, 1.0001f); }
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int rep = 0; rep < 10; ++rep) k<<<blocks, 256>>>(d, iters, 1.0001f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-18s %6.2f cycles per instruction per SIMD at 2.4 GHz nominal\n", name, ms / 10 * 1e6 / ((double)iters * 128 * w * 2) * 2.4);
    fflush(stdout);
    (void)hipFree(d);
}
#define RUN(K) run(#K, K)
int run()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    RUN(d40_41_s48_49); RUN(d40_42_s48_50); RUN(d40_44_s48_52); RUN(d40_44_s49_53); RUN(d40_41_s48_48); RUN(d40_41_s50_51);
    RUN(d40_45_s50_55); RUN(d40_43_s41_42); RUN(d40_41_s44_45); RUN(d40_41_s45_44);
    return 0;
}
} // namespace table6

// ===== table 7 =====
// Operand-parity rule of VALU slot sharing on gfx950, part 2: v_fma_f32 d, a, b, c with explicit registers, two
// alternating independent chains (each chain feeds its result back through the addend c = d).
namespace table7 {
#define REP8(X) X X X X X X X X
#define STR(x) #x
#define XSTR(x) STR(x)
#define V(n) "v" XSTR(n)
#define KERNEL(NAME, D0, A0, B0, D1, A1, B1)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float s)                                \
    {                                                                                                          \
        asm volatile("v_mov_b32 " V(D0) ", %0\n v_mov_b32 " V(D1) ", %0\n v_mov_b32 " V(A0) ", %1\n v_mov_b32 " V(A1) ", %1\n v_mov_b32 " V(B0) ", %1\n v_mov_b32 " V(B1) ", %1" \
                     :: "v"(threadIdx.x * 0.001f + 1.0f), "v"(s) : V(D0), V(D1), V(A0), V(A1), V(B0), V(B1));   \
        for (int it = 0; it < iters; ++it)                                                                     \
            asm volatile(REP8(REP8("v_fma_f32 " V(D0) ", " V(A0) ", " V(B0) ", " V(D0) "\n v_fma_f32 " V(D1) ", " V(A1) ", " V(B1) ", " V(D1) "\n")) \
                         ::: V(D0), V(D1), V(A0), V(A1), V(B0), V(B1));                                         \
        float r;                                                                                               \
        asm volatile("v_add_f32 %0, " V(D0) ", " V(D1) : "=v"(r) :: V(D0), V(D1));                             \
        if (r == 12345.678f) out[0] = r;                                                                       \
    }
// name: parities of (d a b | d a b)
KERNEL(eee_ooo, 40, 48, 50, 41, 49, 51)
KERNEL(eee_eee, 40, 48, 50, 42, 52, 54)
KERNEL(ooo_ooo, 41, 49, 51, 43, 53, 55)
KERNEL(eeo_eeo, 40, 48, 51, 42, 52, 55)
KERNEL(eoo_eoo, 40, 49, 51, 42, 53, 55)
KERNEL(eee_eeo, 40, 48, 50, 42, 52, 55)
KERNEL(eee_eoo, 40, 48, 50, 42, 53, 55)
KERNEL(oee_oee, 41, 48, 50, 43, 52, 54)
KERNEL(eeo_ooe, 40, 48, 51, 41, 53, 56)
KERNEL(eee_oee, 40, 48, 50, 41, 52, 54)
KERNEL(e024_e602, 40, 42, 44, 46, 48, 50)
typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t k)
{
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 256, w = 8, blocks = 256 * w * 2;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) k<<<blocks, 256>>>(d, iters, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int rep = 0; rep < 10; ++rep) k<<<blocks, 256>>>(d, iters, 1.0001f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-12s %6.2f\n", name, ms / 10 * 1e6 / ((double)iters * 128 * w * 2) * 2.4);
    fflush(stdout);
    (void)hipFree(d);
}
#define RUN(K) run(#K, K)
int run()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per instruction per SIMD (2.4 GHz nominal); names = register parities (d a b | d a b)\n");
    RUN(eee_ooo); RUN(eee_eee); RUN(ooo_ooo); RUN(eeo_eeo); RUN(eoo_eoo); RUN(eee_eeo); RUN(eee_eoo); RUN(oee_oee); RUN(eeo_ooe); RUN(eee_oee); RUN(e024_e602);
    return 0;
}
} // namespace table7

int main(int argc, char** argv)
{
    const int only = argc > 1 ? std::atoi(argv[1]) : 0;
    typedef int (*fn_t)();
    const fn_t tables[7] = {table1::run, table2::run, table3::run, table4::run, table5::run, table6::run, table7::run};
    for (int t = 1; t <= 7; ++t)
        if (only == 0 || only == t) {
            std::printf("===== table %d =====\n", t);
            if (tables[t - 1]() != 0) return 1;
        }
    return 0;
}
