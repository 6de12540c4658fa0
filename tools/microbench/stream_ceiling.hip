// What does a plain streaming kernel reach on this chip for the byte mix and launch length of the path's kernels?
// out[i] = in0[i] + ... + in{R-1}[i] on 16-byte elements: R read streams, one write stream, R = 1 (copy), 2 (Tonemap /
// TemporalAA: 2:1), 4 (fused Lighting: ~4:1). Launches back to back on one stream over a ring of cold buffer sets, as
// tools/bench_kernels.py and bench.py time the real kernels. Shapes: one-shot grid (U elements per thread, every load
// issued before the first add) and a persistent grid-stride form.
//   hipcc -O3 --offload-arch=gfx950 -o stream_ceiling stream_ceiling.hip && ./stream_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Ptrs { const float4* in[4]; float4* out; unsigned n; };

// input data: zeros (argv[1] absent) or pseudo-random bits (argv[1] = "random"): data-dependent toggling costs power, and
// a power-capped chip pays for it in clocks
static bool g_random = false;
__global__ void fill_random(unsigned* p, size_t n_words, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x & 0x3FFF3FFFu) | 0x30003000u; // two finite fp16 values in (0.125, 2) / a finite fp32
    }
}
static int init_buffer(void* q, size_t bytes, unsigned seed)
{
    if (!g_random) return hipMemset(q, 0, bytes) == hipSuccess ? 0 : 1;
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)q, bytes / 4, seed);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}

template <int R, int U>
__global__ __launch_bounds__(256) void oneshot(Ptrs p)
{
    const unsigned base = (blockIdx.x * U) * 256u + threadIdx.x;
    float4 v[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) v[u][r] = p.in[r][min(base + u * 256u, p.n - 1u)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float4 s = v[u][0];
#pragma unroll
        for (int r = 1; r < R; ++r) { s.x += v[u][r].x; s.y += v[u][r].y; s.z += v[u][r].z; s.w += v[u][r].w; }
        if (base + u * 256u < p.n) p.out[base + u * 256u] = s;
    }
}

template <int R, int U>
__global__ __launch_bounds__(256) void persistent(Ptrs p)
{
    for (unsigned blk = blockIdx.x; blk * (U * 256u) < p.n; blk += gridDim.x) {
        const unsigned base = (blk * U) * 256u + threadIdx.x;
        float4 v[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = p.in[r][min(base + u * 256u, p.n - 1u)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 s = v[u][0];
#pragma unroll
            for (int r = 1; r < R; ++r) { s.x += v[u][r].x; s.y += v[u][r].y; s.z += v[u][r].z; s.w += v[u][r].w; }
            if (base + u * 256u < p.n) p.out[base + u * 256u] = s;
        }
    }
}

// the same with 8-byte elements (one RGBA16F texel per lane, as TemporalAA and the one-pixel Tonemap address memory)
struct Ptrs8 { const float2* in[4]; float2* out; unsigned n; };
template <int R, int U>
__global__ __launch_bounds__(256) void oneshot8(Ptrs8 p)
{
    const unsigned base = (blockIdx.x * U) * 256u + threadIdx.x;
    float2 v[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) v[u][r] = p.in[r][min(base + u * 256u, p.n - 1u)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float2 s = v[u][0];
#pragma unroll
        for (int r = 1; r < R; ++r) { s.x += v[u][r].x; s.y += v[u][r].y; }
        if (base + u * 256u < p.n) p.out[base + u * 256u] = s;
    }
}

template <int R, int U>
int run8(const char* name, size_t total_bytes, int iters)
{
    const unsigned n = (unsigned)(total_bytes / ((R + 1) * 8));
    const int ring = 4;
    std::vector<Ptrs8> sets(ring);
    for (auto& s : sets) {
        for (int r = 0; r < R; ++r) { float2* q; CK(hipMalloc(&q, (size_t)n * 8)); if (init_buffer(q, (size_t)n * 8, 17u * r + 3u)) return 1; s.in[r] = q; }
        CK(hipMalloc(&s.out, (size_t)n * 8));
        s.n = n;
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    const unsigned blocks = (n + U * 256u - 1u) / (U * 256u);
    auto launch = [&](int k) { hipLaunchKernelGGL((oneshot8<R, U>), dim3(blocks), dim3(256), 0, st, sets[k % ring]); };
    for (int k = 0; k < 300; ++k) launch(k);
    CK(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int k = 0; k < iters; ++k) launch(k);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3f / iters);
    }
    std::sort(t.begin(), t.end());
    const double us = t[t.size() / 2], gbs = (double)n * (R + 1) * 8 / us / 1e3;
    printf("%-28s R=%d U=%d 8 B/lane    %6.1f MB/launch  %7.1f us  %6.0f GB/s  %4.1f %% of 8 TB/s\n", name, R, U, (double)n * (R + 1) * 8 / 1e6, us, gbs, gbs / 80.0);
    fflush(stdout);
    for (auto& s : sets) { for (int r = 0; r < R; ++r) (void)hipFree((void*)s.in[r]); (void)hipFree(s.out); }
    (void)hipStreamDestroy(st);
    return 0;
}

template <int R, int U, bool PERSIST>
int run(const char* name, size_t total_bytes, int iters)
{
    const unsigned n = (unsigned)(total_bytes / ((R + 1) * 16));
    const int ring = 4;
    std::vector<Ptrs> sets(ring);
    for (auto& s : sets) {
        for (int r = 0; r < R; ++r) { float4* q; CK(hipMalloc(&q, (size_t)n * 16)); if (init_buffer(q, (size_t)n * 16, 17u * r + 3u)) return 1; s.in[r] = q; }
        CK(hipMalloc(&s.out, (size_t)n * 16));
        s.n = n;
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    const unsigned blocks = PERSIST ? 256u * 8u : (n + U * 256u - 1u) / (U * 256u);
    auto launch = [&](int k) {
        if (PERSIST) hipLaunchKernelGGL((persistent<R, U>), dim3(blocks), dim3(256), 0, st, sets[k % ring]);
        else hipLaunchKernelGGL((oneshot<R, U>), dim3(blocks), dim3(256), 0, st, sets[k % ring]);
    };
    for (int k = 0; k < 300; ++k) launch(k); // sustained clocks
    CK(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int k = 0; k < iters; ++k) launch(k);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3f / iters);
    }
    std::sort(t.begin(), t.end());
    const double us = t[t.size() / 2], gbs = (double)n * (R + 1) * 16 / us / 1e3;
    printf("%-28s R=%d U=%d %s  %6.1f MB/launch  %7.1f us  %6.0f GB/s  %4.1f %% of 8 TB/s\n", name, R, U, PERSIST ? "persistent" : "one-shot  ",
           (double)n * (R + 1) * 16 / 1e6, us, gbs, gbs / 80.0);
    fflush(stdout);
    for (auto& s : sets) { for (int r = 0; r < R; ++r) (void)hipFree((void*)s.in[r]); (void)hipFree(s.out); }
    (void)hipStreamDestroy(st);
    return 0;
}

int main(int argc, char** argv)
{
    g_random = argc > 1 && argv[1][0] == 'r';
    printf("input data: %s\n", g_random ? "pseudo-random bits" : "zeros");
    const int it = 1000;
    // Tonemap at 4K: 99.5 MB (8 read : 4 written); TemporalAA at 4K: 199 MB (2:1); fused Lighting at 4K: 345 MB (~4:1)
    if (run<2, 2, false>("tonemap-sized", 99532800, it)) return 1;
    if (run<2, 4, false>("tonemap-sized", 99532800, it)) return 1;
    if (run<2, 4, true>("tonemap-sized", 99532800, it)) return 1;
    if (run<2, 2, false>("taa-sized", 199065600, it)) return 1;
    if (run<2, 4, false>("taa-sized", 199065600, it)) return 1;
    if (run<2, 4, true>("taa-sized", 199065600, it)) return 1;
    if (run8<2, 2>("taa-sized", 199065600, it)) return 1;
    if (run8<2, 4>("taa-sized", 199065600, it)) return 1;
    if (run8<2, 8>("taa-sized", 199065600, it)) return 1;
    if (run<4, 1, false>("lighting-sized", 345000000, it)) return 1;
    if (run<4, 2, false>("lighting-sized", 345000000, it)) return 1;
    if (run<4, 2, true>("lighting-sized", 345000000, it)) return 1;
    if (run<4, 4, true>("lighting-sized", 345000000, it)) return 1;
    if (run<1, 4, false>("copy 132 MB moved", 132000000, it)) return 1;
    if (run<1, 4, false>("copy 345 MB moved", 345000000, it)) return 1;
    if (run<1, 8, true>("copy 345 MB moved", 345000000, it)) return 1;
    if (run<1, 4, false>("copy 2 GB moved", 2000000000, 200)) return 1;
    return 0;
}
