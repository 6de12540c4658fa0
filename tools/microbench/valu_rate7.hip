// Operand-parity rule of VALU slot sharing on gfx950, part 2: v_fma_f32 d, a, b, c with explicit registers, two
// alternating independent chains (each chain feeds its result back through the addend c = d).
#include <hip/hip_runtime.h>
#include <cstdio>
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
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per instruction per SIMD (2.4 GHz nominal); names = register parities (d a b | d a b)\n");
    RUN(eee_ooo); RUN(eee_eee); RUN(ooo_ooo); RUN(eeo_eeo); RUN(eoo_eoo); RUN(eee_eeo); RUN(eee_eoo); RUN(oee_oee); RUN(eeo_ooe); RUN(eee_oee); RUN(e024_e602);
    return 0;
}
