// The tail of the HZB chain as device code shared by two launch forms: `hzb_tail_kernel` (csrc/hzb.hip, one workgroup of
// its own) and the streaming lighting kernel's extra workgroup (csrc/lighting.hip) when the frame driver lets the tail ride
// along with the Lighting launch. Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ur {

__device__ __forceinline__ float hzb_min4(float a, float b, float c, float d) { return fminf(fminf(a, b), fminf(c, d)); }

// ---- the tail of the chain in ONE single-workgroup launch ----------------------------------------------------------------
// Once a mip has <= 16384 texels every remaining level fits in LDS, and the reference's further dispatches (<= 4 mips
// each) are a few microseconds of launch latency apiece for microseconds of work in total. The values are those of the
// reference's grouping all the same: a level whose index is a multiple of four is the FIRST level of a reference dispatch
// — it reads its parent through clamped 2x2 footprints (SampleDepth, BuildHZB.hlsl:34-39) — every other level takes the
// 2x2 of its parent's lanes, where an out-of-range parent lane holds 1.0 if the parent was a first level and 0.0
// otherwise (BuildHZB.hlsl:47,81,104).

// AGENT_LOADS: the parent level was written by other workgroups of the SAME launch (write-through, `sc1`): read it with
// agent-scope loads, which are served by L2 and never by this CU's L1.
template <bool AGENT_LOADS>
__device__ __forceinline__ float tail_load(const float* q)
{
    if (AGENT_LOADS) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *q;
}

template <uint32_t TRIPS, bool AGENT_LOADS, class P>
__device__ __forceinline__ void tail_first_level(const P& p, float* bufA)
{
    const uint32_t tid = threadIdx.x, W = p.W[0], n = W * p.H[0];
    const bool first = (p.first_mip & 3u) == 0u;                     // first level of a reference dispatch: clamped reads
    const float fill = ((p.first_mip - 1u) & 3u) == 0u ? 1.0f : 0.0f; // otherwise: what an out-of-range parent lane holds
    float t[TRIPS][4];
    bool in1[TRIPS], in2[TRIPS];
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        const uint32_t i = min(tid + k * 1024u, n - 1u); // clamped into the level: no branch separates the loads
        const uint32_t y = W == 1u ? i : __umulhi(i, p.magic[0]), x = i - y * W;
        in1[k] = first || 2u * x + 1u < p.SW;
        in2[k] = first || 2u * y + 1u < p.SH;
        const uint32_t x0 = min(2u * x, p.SW - 1u), x1 = min(2u * x + 1u, p.SW - 1u);
        const uint32_t y0 = min(2u * y, p.SH - 1u) * p.SW, y1 = min(2u * y + 1u, p.SH - 1u) * p.SW;
        t[k][0] = tail_load<AGENT_LOADS>(p.src + y0 + x0); t[k][1] = tail_load<AGENT_LOADS>(p.src + y0 + x1);
        t[k][2] = tail_load<AGENT_LOADS>(p.src + y1 + x0); t[k][3] = tail_load<AGENT_LOADS>(p.src + y1 + x1);
    }
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        const uint32_t i = tid + k * 1024u;
        if (i < n) {
            const float v = hzb_min4(t[k][0], in1[k] ? t[k][1] : fill, in2[k] ? t[k][2] : fill, (in1[k] && in2[k]) ? t[k][3] : fill);
            bufA[i] = v;
            p.dst[0][i] = v;
        }
    }
}

// The same first level with 16-byte loads, for the shapes every 16:9 chain has there (parent width a multiple of four, parent
// 16-byte aligned, no out-of-range tap: 2 W <= SW and 2 H <= SH): a thread takes the output PAIR (2j, 2j + 1) of a row from two
// float4 loads (parent rows 2y and 2y + 1, columns 4j .. 4j + 3) — an eighth of the vector-memory instructions of the
// tap-by-tap form (which at 8K made this single workgroup spend 9 us reading 130 KB: 1024 wave-instructions of 4 bytes per
// lane), and TRIPS is chosen by the level's size instead of padding to the largest.
typedef float tail_f32x4_t __attribute__((ext_vector_type(4)));

template <uint32_t TRIPS, bool AGENT_LOADS, class P>
__device__ __forceinline__ void tail_first_level_vec(const P& p, float* bufA)
{
    const uint32_t tid = threadIdx.x, W = p.W[0], W2 = W >> 1, npairs = W2 * p.H[0], SW = p.SW;
    const uint32_t mg = W2 > 1u ? (uint32_t)((1ull << 32) / W2 + 1ull) : 0u; // i / W2 for i < 2^16 (exact: W2 <= 2^13)
    tail_f32x4_t a[TRIPS], b[TRIPS];
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        const uint32_t i = min(tid + k * 1024u, npairs - 1u); // clamped into the level: no branch separates the loads
        const uint32_t y = W2 == 1u ? i : __umulhi(i, mg), j = i - y * W2;
        const float* q = p.src + (size_t)(2u * y) * SW + 4u * j;
        if (AGENT_LOADS) {
            // written by other workgroups of the same launch (sc1 stores, drained, one arrival each): sc1 loads, served by L2
            // (MI355X_MICROARCH.md, measured sc1 hand-offs, row 1: 16-byte loads are among the measured forms). An asm load is
            // invisible to hipcc's wait bookkeeping and its destination counts as written at the statement: the destinations
            // are the array elements themselves (no copy may sit between a load and the wait below) and nothing reads them
            // before the wait statement that names them all.
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a[k]) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(b[k]) : "v"(q + SW) : "memory");
        } else {
            a[k] = *reinterpret_cast<const tail_f32x4_t*>(q);
            b[k] = *reinterpret_cast<const tail_f32x4_t*>(q + SW);
        }
    }
    if (AGENT_LOADS) {
        if constexpr (TRIPS == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(b[0])::"memory");
        else if constexpr (TRIPS == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1])::"memory");
        else if constexpr (TRIPS == 4)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3])::"memory");
        else
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]), "+v"(a[4]), "+v"(b[4]), "+v"(a[5]),
                           "+v"(b[5]), "+v"(a[6]), "+v"(b[6]), "+v"(a[7]), "+v"(b[7])::"memory");
    }
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        const uint32_t i = tid + k * 1024u;
        if (i < npairs) {
            const float v0 = hzb_min4(a[k].x, a[k].y, b[k].x, b[k].y), v1 = hzb_min4(a[k].z, a[k].w, b[k].z, b[k].w);
            const uint32_t y = W2 == 1u ? i : __umulhi(i, mg), j = i - y * W2, o = y * W + 2u * j;
            bufA[o] = v0; bufA[o + 1u] = v1;
            p.dst[0][o] = v0; p.dst[0][o + 1u] = v1;
        }
    }
}

// 1024 threads; bufA holds kTailTexels floats, bufB half as many (LDS)
template <bool AGENT_LOADS = false, class P> // P = HzbTail in any address space
__device__ __forceinline__ void hzb_tail_run(const P& p, float* bufA, float* bufB)
{
    const uint32_t tid = threadIdx.x;
    // first level of the tail: parent in global memory. Every load of the thread is issued before the first reduction:
    // one memory latency (8 or 16 texels x 4 taps).
    const uint32_t n0 = p.W[0] * p.H[0];
    const bool vec = (p.SW & 3u) == 0u && (p.W[0] & 1u) == 0u && 2u * p.W[0] <= p.SW && 2u * p.H[0] <= p.SH &&
                     (reinterpret_cast<uintptr_t>(p.src) & 15u) == 0u; // uniform
    if (vec && n0 <= 2048u) tail_first_level_vec<1, AGENT_LOADS>(p, bufA);
    else if (vec && n0 <= 4096u) tail_first_level_vec<2, AGENT_LOADS>(p, bufA);
    else if (vec && n0 <= 8192u) tail_first_level_vec<4, AGENT_LOADS>(p, bufA);
    else if (vec) tail_first_level_vec<8, AGENT_LOADS>(p, bufA);
    else if (n0 <= kTailTexels / 2u) tail_first_level<kTailTexels / 2048u, AGENT_LOADS>(p, bufA);
    else tail_first_level<kTailTexels / 1024u, AGENT_LOADS>(p, bufA);
    __syncthreads();
    for (uint32_t l = 1; l < p.levels; ++l) { // uniform
        const float* par = (l & 1u) ? bufA : bufB;
        float* cur = (l & 1u) ? bufB : bufA;
        const uint32_t W = p.W[l], H = p.H[l], PW = p.W[l - 1], PH = p.H[l - 1];
        const uint32_t m = p.first_mip + l;
        const bool first = (m & 3u) == 0u;                     // first level of a reference dispatch: clamped reads
        const float fill = ((m - 1u) & 3u) == 0u ? 1.0f : 0.0f; // what an out-of-range parent lane holds otherwise
        const uint32_t mg = p.magic[l];
        for (uint32_t i = tid; i < W * H; i += 1024u) {
            const uint32_t y = W == 1u ? i : __umulhi(i, mg), x = i - y * W;
            float v;
            if (first) {
                const uint32_t x0 = min(2u * x, PW - 1u), x1 = min(2u * x + 1u, PW - 1u);
                const uint32_t y0 = min(2u * y, PH - 1u), y1 = min(2u * y + 1u, PH - 1u);
                v = hzb_min4(par[y0 * PW + x0], par[y0 * PW + x1], par[y1 * PW + x0], par[y1 * PW + x1]);
            } else {
                const uint32_t x0 = 2u * x, x1 = x0 + 1u, y0 = 2u * y, y1 = y0 + 1u;
                const float a = par[y0 * PW + x0]; // (2x, 2y) is always in range
                const float b = x1 < PW ? par[y0 * PW + x1] : fill;
                const float c = y1 < PH ? par[y1 * PW + x0] : fill;
                const float d = (x1 < PW && y1 < PH) ? par[y1 * PW + x1] : fill;
                v = hzb_min4(a, b, c, d);
            }
            cur[i] = v;
            p.dst[l][i] = v;
        }
        __syncthreads();
    }
}

} // namespace ur
