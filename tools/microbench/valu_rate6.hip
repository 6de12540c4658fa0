// Which operand layouts let two adjacent independent v_fma_f32 of one wave share an issue slot on gfx950?
// Two alternating chains with EXPLICIT registers: chain A = v[DA] <- v[DA] * v[SA] + v[DA], chain B likewise.
#include <hip/hip_runtime.h>
#include <cstdio>
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
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    RUN(d40_41_s48_49); RUN(d40_42_s48_50); RUN(d40_44_s48_52); RUN(d40_44_s49_53); RUN(d40_41_s48_48); RUN(d40_41_s50_51);
    RUN(d40_45_s50_55); RUN(d40_43_s41_42); RUN(d40_41_s44_45); RUN(d40_41_s45_44);
    return 0;
}
