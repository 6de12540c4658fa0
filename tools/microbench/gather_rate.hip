// What does a 64-lane gather cost on the CU's texture-addresser / L1 path, by width and alignment? (gfx950)
// One 1024-thread workgroup per CU (16 waves, as the lighting kernel), every lane reads from a small table that stays in L1
// (32 KB per CU) at a per-lane pseudo-random texel; the loads of an iteration are independent and the loop waits for all of
// them once per iteration. Reported: cycles per wave-instruction per CU (shader clock from s_memtime).
//   hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));

// MODE: 0 = 16 B at 16-byte aligned addresses, 1 = 16 B at 8-byte aligned (odd multiples of 8 for half the lanes),
//       2 = 12 B at 4-byte aligned, 3 = 8 B at 8-byte aligned, 4 = 4 B, 5 = 16 B, all lanes of a quad in one 64-byte line,
//       6 = 16 B at 8-byte aligned, neighbouring lanes 8 bytes apart (overlapping footprints: a smooth surface)
template <int MODE>
__global__ __launch_bounds__(1024) void gather_kernel(const char* table, uint32_t table_bytes, int iters, unsigned long long* cycles, uint32_t* sink)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    uint32_t acc = 0;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        uint32_t off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            uint32_t o = (h >> 8) & (table_bytes - 1u); // (table_bytes is a power of two; the allocation has slack behind it)
            if (MODE == 0) o &= ~15u;
            if (MODE == 1) o = (o & ~15u) | ((lane & 1u) ? 8u : 0u);
            if (MODE == 2) o &= ~3u;
            if (MODE == 3) o &= ~7u;
            if (MODE == 4) o &= ~3u;
            if (MODE == 5) o = ((o & ~63u) | ((lane & 3u) * 16u));
            if (MODE == 6) o = ((__builtin_amdgcn_readfirstlane(o) & ~127u) + lane * 8u + wave * 64u) & (table_bytes - 1u);
            off[k] = o;
        }
        if (MODE == 5) { // one line per quad: the quad's lanes share the upper address bits
#pragma unroll
            for (int k = 0; k < 4; ++k) off[k] = (__shfl(off[k], lane & ~3u) & ~63u) | ((lane & 3u) * 16u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0 || MODE == 1 || MODE == 5 || MODE == 6) { const u32x4 v = *reinterpret_cast<const u32x4_a8*>(table + off[k]); acc += v.x ^ v.y ^ v.z ^ v.w; }
            else if (MODE == 2) { const u32x3 v = *reinterpret_cast<const u32x3_a4*>(table + off[k]); acc += v.x ^ v.y ^ v.z; }
            else if (MODE == 3) { const u32x2 v = *reinterpret_cast<const u32x2_a4*>(table + off[k]); acc += v.x ^ v.y; }
            else { acc += *reinterpret_cast<const uint32_t*>(table + off[k]); }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cycles[blockIdx.x * 16u + wave] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
double run(const char* name, const char* table, uint32_t bytes, int cus, unsigned long long* dcyc, uint32_t* sink)
{
    const int iters = 2000;
    hipLaunchKernelGGL(gather_kernel<MODE>, dim3(cus), dim3(1024), 0, 0, table, bytes, 50, dcyc, sink);
    hipLaunchKernelGGL(gather_kernel<MODE>, dim3(cus), dim3(1024), 0, 0, table, bytes, iters, dcyc, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(cus * 16);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0;
    for (auto v : c) mx += (double)v;
    mx /= c.size();
    // per CU: 16 waves x iters x 4 instructions in `mx` cycles
    const double per_instr = mx / (16.0 * iters * 4.0);
    printf("%-58s table %6u B: %6.1f cycles per wave-instruction per CU\n", name, bytes, per_instr);
    return per_instr;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    char* table;
    unsigned long long* dcyc;
    uint32_t* sink;
    hipMalloc(&table, 1 << 24);
    hipMemset(table, 1, 1 << 24);
    hipMalloc(&dcyc, cus * 16 * 8);
    hipMalloc(&sink, 64);
    for (uint32_t bytes : {8192u, 1u << 20}) { // L1-resident, L2-resident
        run<0>("16 B per lane, 16-byte aligned, random", table, bytes, cus, dcyc, sink);
        run<1>("16 B per lane, 8-byte aligned (half the lanes odd)", table, bytes, cus, dcyc, sink);
        run<6>("16 B per lane, 8-byte aligned, lanes 8 bytes apart", table, bytes, cus, dcyc, sink);
        run<5>("16 B per lane, each quad of lanes inside one 64-B line", table, bytes, cus, dcyc, sink);
        run<2>("12 B per lane, 4-byte aligned, random", table, bytes, cus, dcyc, sink);
        run<3>(" 8 B per lane, 8-byte aligned, random", table, bytes, cus, dcyc, sink);
        run<4>(" 4 B per lane, random", table, bytes, cus, dcyc, sink);
    }
    return 0;
}
