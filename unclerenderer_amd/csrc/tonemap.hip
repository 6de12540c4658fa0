// Tonemap for gfx950 — the consumer of the HDR band.
//
// Reference: Shaders/Tonemap.hlsl:34-79 (fullscreen pixel shader: exposure [x 2^EV from the auto-exposure texture],
// Khronos PBR-neutral curve, saturate, gamma) writing the R8G8B8A8_UNORM back buffer, pass
// Source/Render/DeferredRenderer.cpp:1449-1513. One lane converts one pixel: 8-byte load, 4-byte store (12 B/pixel),
// a pure HBM stream. Run on a rank's band BEFORE the multi-GPU gather it halves the xGMI payload (8 -> 4 B/pixel).
// Built with -ffp-contract=off and explicit fmaf: the pixel-pair form, the one-pixel form and a band of either give the
// same bits for the same pixel (the oracle comparison allows one LSB: exp2/log2 vs powf, v_rcp vs divide).

#include "ur_internal.h"
#include "ur_device.h"

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

struct TonemapParams {
    const half4_t* hdr;
    const float* exposure_ev; // LogAverageLuminance texel (0,0), nullable
    uint32_t* out;
    uint32_t count;
    uint32_t enable_tonemap, enable_auto_exposure;
    float exposure, inv_gamma;
};

__device__ __forceinline__ float pow_pos(float x, float e) { return x > 0.0f ? __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x)) : 0.0f; }
__device__ __forceinline__ uint32_t unorm8(float x) { return (uint32_t)fmaf(fminf(fmaxf(x, 0.0f), 1.0f), 255.0f, 0.5f); }

// one pixel: RGBA16F -> packed R8G8B8A8 (Tonemap.hlsl:57-79)
__device__ __forceinline__ uint32_t tonemap_pixel(const TonemapParams& p, float finalExposure, half4_t h)
{
    {   // (block kept so that the body reads like the shader's main)
        float r = (float)h.x * finalExposure, g = (float)h.y * finalExposure, b = (float)h.z * finalExposure;
        if (p.enable_tonemap != 0) { // PBRNeutralToneMapping, Tonemap.hlsl:34-55
            const float startCompression = 0.8f - 0.04f, desaturation = 0.15f;
            const float x = fminf(r, fminf(g, b));
            const float offset = x < 0.08f ? fmaf(-6.25f * x, x, x) : 0.04f;
            r -= offset; g -= offset; b -= offset;
            const float peak = fmaxf(r, fmaxf(g, b));
            if (!(peak < startCompression)) {
                const float d = 1.0f - startCompression;
                // the three quotients through v_rcp_f32 (1 ulp): an IEEE divide is ~12 instructions each, which made this
                // stream VALU-bound; the 8-bit result moves by at most the one LSB the pow already allows
                const float newPeak = fmaf(-(d * d), __builtin_amdgcn_rcpf(peak + d - startCompression), 1.0f);
                const float s = newPeak * __builtin_amdgcn_rcpf(fmaxf(peak, 1e-4f));
                r *= s; g *= s; b *= s;
                const float gm = 1.0f - __builtin_amdgcn_rcpf(fmaf(desaturation, peak - newPeak, 1.0f));
                r = fmaf(gm, newPeak - r, r); g = fmaf(gm, newPeak - g, g); b = fmaf(gm, newPeak - b, b);
            }
        }
        r = fminf(fmaxf(r, 0.0f), 1.0f); g = fminf(fmaxf(g, 0.0f), 1.0f); b = fminf(fmaxf(b, 0.0f), 1.0f);
        r = pow_pos(r, p.inv_gamma); g = pow_pos(g, p.inv_gamma); b = pow_pos(b, p.inv_gamma);
        return unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | 0xFF000000u;
    }
}

__device__ __forceinline__ float final_exposure(const TonemapParams& p)
{
    float e = p.exposure;
    if (p.enable_auto_exposure != 0 && p.exposure_ev != nullptr) e *= __builtin_amdgcn_exp2f(p.exposure_ev[0]);
    return e;
}

// any count, any alignment: one pixel per lane and trip
__global__ __launch_bounds__(256) void tonemap_kernel(TonemapParams p)
{
    const float finalExposure = final_exposure(p);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.count; i += gridDim.x * 256u) p.out[i] = tonemap_pixel(p, finalExposure, p.hdr[i]);
}

// The streaming form (16-byte aligned buffers): a lane converts pixel PAIRS - one 16-byte load, one 8-byte store, both
// lane-consecutive (1 KB and 512 B per wave instruction) - and a wave takes TRIPS of them, 128 pixels apart, with every
// load issued before the first conversion. Pixels [0, count & ~1). Measured (4K / 8K, us): TRIPS 1: 20.5 / 75.2,
// 2: 21.0 / 72.9, 4: 22.4 / 72.3 - short waves keep loading and converting waves mixed on a CU, long ones save launches
// of waves once the grid is many rounds deep; a plain 2:1 streaming kernel of the 4K size takes 19.2 us
// (tools/microbench/stream_ceiling.hip).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <uint32_t kPairTrips>
__global__ __launch_bounds__(256) void tonemap_pairs_kernel(TonemapParams p)
{
    const float finalExposure = final_exposure(p);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // pair index of trip k: ((block * 4 + wave) * kPairTrips + k) * 64 + lane
    const uint32_t pair0 = (blockIdx.x * 4u + wave) * (kPairTrips * 64u) + lane, npairs = p.count >> 1;
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(p.hdr);
    u32x4_t v[kPairTrips];
#pragma unroll
    for (uint32_t k = 0; k < kPairTrips; ++k) v[k] = __builtin_nontemporal_load(src + min(pair0 + k * 64u, npairs - 1u)); // read once
#pragma unroll
    for (uint32_t k = 0; k < kPairTrips; ++k) {
        const uint32_t pr = pair0 + k * 64u;
        if (pr < npairs) {
            union { u32x2_t u; half4_t h; } a, b;
            a.u = u32x2_t{v[k].x, v[k].y}; b.u = u32x2_t{v[k].z, v[k].w};
            ur::store_once_b64(reinterpret_cast<u32x2_t*>(p.out) + pr, ur::once_u32x2_t{tonemap_pixel(p, finalExposure, a.h), tonemap_pixel(p, finalExposure, b.h)});
        }
    }
}

} // namespace

extern "C" int ur_tonemap(ur_ctx* ctx, const ur_tonemap_constants* constants, const ur_half4* hdr, const float* exposure_ev, uint32_t* out_rgba8,
                          uint32_t w, uint32_t rows)
{
    if (!ctx || !constants || !hdr || !out_rgba8) { ur::set_error("ur_tonemap: null argument"); return UR_EINVAL; }
    const uint64_t n = (uint64_t)w * rows;
    if (n == 0) return UR_OK;
    if (n > 0xFFFFFFFFull) { ur::set_error("ur_tonemap: band too large"); return UR_EUNSUPPORTED; }
    TonemapParams p{};
    p.hdr = reinterpret_cast<const half4_t*>(hdr);
    p.exposure_ev = exposure_ev;
    p.out = out_rgba8;
    p.count = (uint32_t)n;
    p.enable_tonemap = constants->EnableTonemap;
    p.enable_auto_exposure = constants->EnableAutoExposure;
    p.exposure = constants->Exposure;
    p.inv_gamma = 1.0f / (constants->Gamma > 1e-3f ? constants->Gamma : 1e-3f);
    const bool aligned = ((reinterpret_cast<uintptr_t>(hdr) & 15u) == 0u) && ((reinterpret_cast<uintptr_t>(out_rgba8) & 7u) == 0u);
    if (aligned && n >= 2u) {
        if (n < (24u << 20)) hipLaunchKernelGGL(tonemap_pairs_kernel<1>, dim3((uint32_t)((n / 2u * 2u + 511u) / 512u)), dim3(256), 0, ctx->stream, p);
        else hipLaunchKernelGGL(tonemap_pairs_kernel<2>, dim3((uint32_t)((n / 2u * 2u + 1023u) / 1024u)), dim3(256), 0, ctx->stream, p);
        UR_HIP_TRY(hipGetLastError());
        if ((n & 1u) == 0u) return UR_OK;
        // the odd last pixel
        p.hdr += n - 1u; p.out += n - 1u; p.count = 1u;
    }
    uint32_t blocks = (uint32_t)((p.count + 255u) / 256u);
    const uint32_t cap = (uint32_t)ctx->cu_count * 8u * 2u;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(tonemap_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}
