// Tonemap for gfx950 — the consumer of the HDR band.
//
// Reference: Shaders/Tonemap.hlsl:34-79 (fullscreen pixel shader: exposure [x 2^EV from the auto-exposure texture],
// Khronos PBR-neutral curve, saturate, gamma) writing the R8G8B8A8_UNORM back buffer, pass
// Source/Render/DeferredRenderer.cpp:1449-1513. One lane converts one pixel: 8-byte load, 4-byte store (12 B/pixel),
// a pure HBM stream. Run on a rank's band BEFORE the multi-GPU gather it halves the xGMI payload (8 -> 4 B/pixel).

#include "ur_internal.h"

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
__device__ __forceinline__ uint32_t unorm8(float x) { return (uint32_t)(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f + 0.5f); }

__global__ __launch_bounds__(256) void tonemap_kernel(TonemapParams p)
{
    float finalExposure = p.exposure;
    if (p.enable_auto_exposure != 0 && p.exposure_ev != nullptr) finalExposure *= __builtin_amdgcn_exp2f(p.exposure_ev[0]);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.count; i += gridDim.x * 256u) {
        const half4_t h = p.hdr[i];
        float r = (float)h.x * finalExposure, g = (float)h.y * finalExposure, b = (float)h.z * finalExposure;
        if (p.enable_tonemap != 0) { // PBRNeutralToneMapping, Tonemap.hlsl:34-55
            const float startCompression = 0.8f - 0.04f, desaturation = 0.15f;
            const float x = fminf(r, fminf(g, b));
            const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
            r -= offset; g -= offset; b -= offset;
            const float peak = fmaxf(r, fmaxf(g, b));
            if (!(peak < startCompression)) {
                const float d = 1.0f - startCompression;
                const float newPeak = 1.0f - d * d / (peak + d - startCompression);
                const float s = newPeak / fmaxf(peak, 1e-4f);
                r *= s; g *= s; b *= s;
                const float gm = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
                r = r + gm * (newPeak - r); g = g + gm * (newPeak - g); b = b + gm * (newPeak - b);
            }
        }
        r = fminf(fmaxf(r, 0.0f), 1.0f); g = fminf(fmaxf(g, 0.0f), 1.0f); b = fminf(fmaxf(b, 0.0f), 1.0f);
        r = pow_pos(r, p.inv_gamma); g = pow_pos(g, p.inv_gamma); b = pow_pos(b, p.inv_gamma);
        p.out[i] = unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | 0xFF000000u;
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
    uint32_t blocks = (uint32_t)((n + 255u) / 256u);
    const uint32_t cap = (uint32_t)ctx->cu_count * 8u * 2u;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(tonemap_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}
