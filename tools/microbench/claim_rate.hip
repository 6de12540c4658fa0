// What does a work-claim atomic cost on gfx950, by scope and by how many waves share the word?
//   hipcc -O3 --offload-arch=gfx950 claim_rate.hip -o claim_rate && timeout -k 5 60 ./claim_rate
// 256 workgroups of 1024 threads (one per CU, as the lighting kernel). Lane 0 of every wave claims from a counter with a returning
// atomic add and WAITS for the result (a dependent chain: one claim in flight per wave), `iters` times.
//   scope: "L2" = workgroup-scope atomic on global memory (no sc1: executed in the issuing XCD's L2, coherent among the workgroups of
//          THAT XCD only; the counter is keyed by HW_REG_XCC_ID), "agent" = device scope (sc1: coherent over all XCDs).
//   share: how many counters per XCD (L2) or per chip (agent) the waves spread over.
// Reported: microseconds per claim as one wave sees it (latency under that load) and claims per microsecond per counter and in total.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <bool AGENT>
__global__ __launch_bounds__(1024) void claim_kernel(uint32_t* counters, uint32_t per_group, int iters, uint32_t* sink, unsigned long long* ticks)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
    const uint32_t group = AGENT ? 0u : xcc; // agent scope: one set of counters for the chip; L2: one set per XCD
    uint32_t* c = counters + (group * 64u + ((blockIdx.x * 16u + wave) % per_group)) * 32u; // 128 bytes apart
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        for (int i = 0; i < iters; ++i) {
            uint32_t v;
            if (AGENT) v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            acc += v;
            asm volatile("" : "+v"(acc)); // the next claim waits for this one
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 16u + wave] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <bool AGENT>
static void run(const char* name, uint32_t per_group, int waves_active_note)
{
    (void)waves_active_note;
    uint32_t* d; unsigned long long* t; uint32_t* sink;
    const size_t words = 16u * 64u * 32u;
    (void)hipMalloc(&d, words * 4); (void)hipMemset(d, 0, words * 4);
    (void)hipMalloc(&t, 4096 * 8); (void)hipMalloc(&sink, 4);
    const int iters = 200;
    hipLaunchKernelGGL(claim_kernel<AGENT>, dim3(256), dim3(1024), 0, 0, d, per_group, 20, sink, t);
    (void)hipMemset(d, 0, words * 4);
    hipLaunchKernelGGL(claim_kernel<AGENT>, dim3(256), dim3(1024), 0, 0, d, per_group, iters, sink, t);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(4096);
    std::vector<uint32_t> c(words);
    (void)hipMemcpy(h.data(), t, 4096 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c.data(), d, words * 4, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (auto v : h) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
    mean /= 4096.0;
    unsigned long long total = 0; uint32_t used = 0;
    for (size_t i = 0; i < words; i += 32) { total += c[i]; used += c[i] != 0; }
    const double us_per_claim = mean * 0.01 / iters;
    const double total_rate = 4096.0 * iters / (mx * 0.01);
    printf("%-6s %3u counter(s) per %s (%3u in use): %6.3f us per claim as a wave sees it; %8.1f claims/us over the chip, %7.1f per counter; counted %llu of %d%s\n",
           name, per_group, AGENT ? "chip" : "XCD ", used, us_per_claim, total_rate, total_rate / used, total, 4096 * iters,
           total == 4096ull * iters ? "" : "  <-- LOST UPDATES");
    (void)hipFree(d); (void)hipFree(t); (void)hipFree(sink);
}

int main()
{
    for (uint32_t n : {1u, 2u, 4u, 8u, 16u, 64u}) run<false>("L2", n, 0);
    for (uint32_t n : {1u, 8u, 64u}) run<true>("agent", n, 0);
    return 0;
}
