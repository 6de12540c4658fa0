// Device-side helpers shared by the kernels. Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ur {

// Eight bytes written ONCE by this launch and read by a later one: write-through and nontemporal (`sc1 nt`) - the line neither stays
// dirty in L2 nor takes a place there (the HDR store of the lighting loop, csrc/lighting.hip: either hint alone changes nothing).
typedef uint32_t once_u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_once_b64(void* p, once_u32x2_t v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// Touch every 64-byte line of the kernarg segment (explicit arguments of BYTES bytes plus the hidden ones behind them)
// with one batch of scalar loads and ONE wait. hipcc reads kernel parameters lazily, a few dwords at a time with a wait
// after each group; at the start of a launch every new line is a scalar-cache miss, and a kernel with a dozen dependent
// groups spends microseconds on them before its first vector load (in-kernel stamps on the lighting kernel: half of its
// prologue). After this batch they all hit. Lines behind the last one re-read the last one.
template <uint32_t BYTES>
__device__ __forceinline__ void warm_kernarg()
{
    static_assert(BYTES < 16 * 64, "one load per line, 16 lines");
    constexpr uint32_t kLast = (BYTES / 64u) * 64u;
#define UR_KA_LINE(i) ((i) * 64u < kLast ? (i) * 64u : kLast)
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12, d13, d14, d15;
    if constexpr (kLast <= 3 * 64u) {
        asm volatile("s_load_dword %0, %4, %5\n\t"
                     "s_load_dword %1, %4, %6\n\t"
                     "s_load_dword %2, %4, %7\n\t"
                     "s_load_dword %3, %4, %8\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3)
                     : "s"(k), "n"(UR_KA_LINE(0)), "n"(UR_KA_LINE(1)), "n"(UR_KA_LINE(2)), "n"(UR_KA_LINE(3))
                     : "memory");
    } else {
        asm volatile(
                     "s_load_dword %0, %16, %17\n\t"
                     "s_load_dword %1, %16, %18\n\t"
                     "s_load_dword %2, %16, %19\n\t"
                     "s_load_dword %3, %16, %20\n\t"
                     "s_load_dword %4, %16, %21\n\t"
                     "s_load_dword %5, %16, %22\n\t"
                     "s_load_dword %6, %16, %23\n\t"
                     "s_load_dword %7, %16, %24\n\t"
                     "s_load_dword %8, %16, %25\n\t"
                     "s_load_dword %9, %16, %26\n\t"
                     "s_load_dword %10, %16, %27\n\t"
                     "s_load_dword %11, %16, %28\n\t"
                     "s_load_dword %12, %16, %29\n\t"
                     "s_load_dword %13, %16, %30\n\t"
                     "s_load_dword %14, %16, %31\n\t"
                     "s_load_dword %15, %16, %32\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7), "=&s"(d8), "=&s"(d9), "=&s"(d10), "=&s"(d11), "=&s"(d12), "=&s"(d13), "=&s"(d14), "=&s"(d15)
                     : "s"(k), "n"(UR_KA_LINE(0)), "n"(UR_KA_LINE(1)), "n"(UR_KA_LINE(2)), "n"(UR_KA_LINE(3)), "n"(UR_KA_LINE(4)), "n"(UR_KA_LINE(5)), "n"(UR_KA_LINE(6)), "n"(UR_KA_LINE(7)), "n"(UR_KA_LINE(8)), "n"(UR_KA_LINE(9)), "n"(UR_KA_LINE(10)), "n"(UR_KA_LINE(11)), "n"(UR_KA_LINE(12)), "n"(UR_KA_LINE(13)), "n"(UR_KA_LINE(14)), "n"(UR_KA_LINE(15))
                     : "memory");
    }
#undef UR_KA_LINE
}

// Debug timeline (ur_debug_timeline): one lane per workgroup folds the constant 100 MHz clock into the launch's {first entry,
// last exit} pair. Nothing is executed when the pointer is null (one scalar branch).
__device__ __forceinline__ void timeline_entry(unsigned long long* pair)
{
    if (pair != nullptr && threadIdx.x == 0) atomicMin(pair, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void timeline_exit(unsigned long long* pair, bool lane_of_last_wave)
{
    if (pair != nullptr && lane_of_last_wave) atomicMax(pair + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

} // namespace ur
