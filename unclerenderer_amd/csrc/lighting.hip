// DeferredLighting (GGX + IBL) and SkyAtmosphere for gfx950, as one per-pixel compute kernel with three modes.
//
// Reference: Shaders/DeferredLighting.hlsl:35-94 + Shaders/PBRCommon.hlsl:1-48 (fullscreen-triangle pixel shader,
// additive ONE/ONE blend into RGBA16F, Source/Render/DeferredRenderer.cpp:1219-1255,1997-2005) and
// Shaders/SkyAtmosphere.hlsl:40-101 (inside-out sphere, depth GREATER_EQUAL, no blend, DeferredRenderer.cpp:1263-1296).
//
// There is no rasteriser and no texture unit here: the G-buffer is read with coalesced 8-byte-per-lane loads, the
// shadow PCF, the trilinear cube lookups and the BRDF LUT are filtered in ALU from plain loads. Every per-launch uniform
// the HLSL recomputes per pixel (light vector in view space, reciprocals, ViewInverse*LightViewProjection, sky densities)
// is folded on the host into LightingParams, because gfx950 has no scalar fp32 ALU and uniform math would otherwise
// run on the VALU for all 64 lanes. The view matrix is rigid (XMMatrixLookToLH, Scene/Camera.cpp:23-31), so the
// world-space IBL vectors are the view-space ones rotated by ViewInverse — no second set of normalisations.
// Tolerance against the oracle: max(1e-3, 1 ulp fp16) per channel; fp32 math, one RTE to fp16.

#include "ur_internal.h"

#include <cmath>
#include <cstring>

namespace {

struct LightingParams {
    // frame
    uint32_t W, H, row0, rows;
    float invW, invH;
    // lighting
    float invP11, invP22;
    float L[3];          // normalize(mul(float4(LightDirection,0), View).xyz)
    float R[9];          // (float3x3)ViewInverse, row-major
    float SM[16];        // ViewInverse * LightViewProjection : view-space position -> shadow clip
    float lightRGB[3];   // LightIntensity * LightColor
    float shadowStrength, shadowBias;
    float shadowW, shadowH, shadowTexelX, shadowTexelY;
    uint32_t shadowWi, shadowHi;
    float maxMip;        // max(0, EnvMapMipCount-1)
    uint32_t envBase, envMips;
    uint32_t envMipOffset[16]; // in half4 texels
    uint32_t lutW, lutH;
    // sky
    float skyRot[9];     // rows of View's 3x3: world_j = dot(skyRot[3j..3j+2], v)
    float skyInvP11, skyInvP22;
    float skyNearOverR;  // Projection[14] / World[0]
    float sunDir[3];     // normalize(LightDirection)
    float skyScatterR[3];// rayleighColor * rayleighDensity
    float skyMie[3];     // LightColor * mieDensity * 0.8
    float sunAttenuation;
    // buffers
    const ur_half4* A;
    const ur_half4* B;
    const uint32_t* C;
    const float* depth;
    const float* shadow;
    const ur_half4* env;
    const uint32_t* lut; // RG16 texel = one dword
    const float* srgb;
    ur_half4* hdr;
};

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float mix(float a, float b, float t) { return fmaf(t, b - a, a); }

__device__ __forceinline__ float h2f(uint16_t h)
{
    _Float16 x;
    __builtin_memcpy(&x, &h, 2);
    return (float)x; // v_cvt_f32_f16, exact
}
__device__ __forceinline__ uint16_t f2h(float f)
{
    const _Float16 x = (_Float16)f; // v_cvt_f16_f32, round-to-nearest-even
    uint16_t h;
    __builtin_memcpy(&h, &x, 2);
    return h;
}

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 mix(F3 a, F3 b, float t) { return {mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)}; }

// ---- bordered cube: face f of mip m is (N+2)^2 texels, border = seamless neighbours (ur_stage_env_cube) -----------
__device__ __forceinline__ void cube_face(F3 d, int& face, float& u, float& v)
{
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float ma, uc, vc;
    if (az >= ax && az >= ay) {
        face = d.z >= 0.0f ? 4 : 5; ma = az; uc = d.z >= 0.0f ? d.x : -d.x; vc = -d.y;
    } else if (ay >= ax) {
        face = d.y >= 0.0f ? 2 : 3; ma = ay; uc = d.x; vc = d.y >= 0.0f ? d.z : -d.z;
    } else {
        face = d.x >= 0.0f ? 0 : 1; ma = ax; uc = d.x >= 0.0f ? -d.z : d.z; vc = -d.y;
    }
    const float inv = 0.5f * rcp(ma);
    u = fmaf(uc, inv, 0.5f);
    v = fmaf(vc, inv, 0.5f);
}

__device__ __forceinline__ F3 cube_bilinear(const LightingParams& p, uint32_t mip, int face, float u, float v)
{
    const uint32_t N = max(1u, p.envBase >> mip), E = N + 2u;
    const float fN = (float)N;
    const float x = fmaf(u, fN, 0.5f), y = fmaf(v, fN, 0.5f); // bordered coordinates
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const uint32_t i0 = min((uint32_t)max((int)x0, 0), N), j0 = min((uint32_t)max((int)y0, 0), N);
    const ur_half4* t = p.env + p.envMipOffset[mip] + ((size_t)face * E + j0) * E + i0;
    const ur_half4 t00 = t[0], t10 = t[1], t01 = t[E], t11 = t[E + 1];
    const F3 top = mix(f3(h2f(t00.x), h2f(t00.y), h2f(t00.z)), f3(h2f(t10.x), h2f(t10.y), h2f(t10.z)), fx);
    const F3 bot = mix(f3(h2f(t01.x), h2f(t01.y), h2f(t01.z)), f3(h2f(t11.x), h2f(t11.y), h2f(t11.z)), fx);
    return mix(top, bot, fy);
}

__device__ __forceinline__ F3 cube_sample_level(const LightingParams& p, F3 dir, float level)
{
    const float l = fminf(fmaxf(level, 0.0f), (float)(p.envMips - 1u));
    const float l0 = floorf(l);
    const uint32_t m0 = (uint32_t)l0, m1 = min(m0 + 1u, p.envMips - 1u);
    const float fl = l - l0;
    int face; float u, v;
    cube_face(dir, face, u, v);
    const F3 c0 = cube_bilinear(p, m0, face, u, v);
    if (fl == 0.0f || m1 == m0) return c0;
    const F3 c1 = cube_bilinear(p, m1, face, u, v);
    return mix(c0, c1, fl);
}

__device__ __forceinline__ void lut_sample(const LightingParams& p, float u, float v, float& a, float& b)
{
    const float x = fmaf(u, (float)p.lutW, -0.5f), y = fmaf(v, (float)p.lutH, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int W1 = (int)p.lutW - 1, H1 = (int)p.lutH - 1;
    const int i0 = min(max((int)x0, 0), W1), i1 = min(max((int)x0 + 1, 0), W1);
    const int j0 = min(max((int)y0, 0), H1), j1 = min(max((int)y0 + 1, 0), H1);
    const uint32_t t00 = p.lut[j0 * p.lutW + i0], t10 = p.lut[j0 * p.lutW + i1];
    const uint32_t t01 = p.lut[j1 * p.lutW + i0], t11 = p.lut[j1 * p.lutW + i1];
    const float s = 1.0f / 65535.0f;
    a = mix(mix((float)(t00 & 0xFFFFu), (float)(t10 & 0xFFFFu), fx), mix((float)(t01 & 0xFFFFu), (float)(t11 & 0xFFFFu), fx), fy) * s;
    b = mix(mix((float)(t00 >> 16), (float)(t10 >> 16), fx), mix((float)(t01 >> 16), (float)(t11 >> 16), fx), fy) * s;
}

// SampleCmpLevelZero: bilinear blend of four LESS_EQUAL results, border = 1.0
__device__ __forceinline__ float shadow_cmp(const LightingParams& p, float u, float v, float cmp)
{
    const float x = fmaf(u, p.shadowW, -0.5f), y = fmaf(v, p.shadowH, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int i0 = (int)x0, j0 = (int)y0;
    const int W = (int)p.shadowWi, H = (int)p.shadowHi;
    float r[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int xi = i0 + i, yj = j0 + j;
            const bool in = xi >= 0 && yj >= 0 && xi < W && yj < H;
            const float t = in ? p.shadow[(size_t)yj * W + xi] : 1.0f;
            r[j][i] = cmp <= t ? 1.0f : 0.0f;
        }
    return mix(mix(r[0][0], r[0][1], fx), mix(r[1][0], r[1][1], fx), fy);
}

// ---- PBRCommon.hlsl:24-48 -------------------------------------------------------------------------------------------
__device__ __forceinline__ F3 EvaluatePBR(F3 albedo, float metallic, float roughness, F3 F0, F3 N, F3 V, F3 L, float NdotV)
{
    F3 H = f3(V.x + L.x, V.y + L.y, V.z + L.z);
    const float hr = rsq(dot(H, H));
    H = f3(H.x * hr, H.y * hr, H.z * hr);
    const float NdotL = sat(dot(N, L));
    const float NdotH = sat(dot(N, H));
    const float VdotH = sat(dot(V, H));
    const float alpha = roughness * roughness;
    const float alpha2 = alpha * alpha;
    const float denom = fmaf(NdotH * NdotH, alpha2 - 1.0f, 1.0f);
    const float D = alpha2 * rcp(fmaxf(3.14159265f * denom * denom, 1e-4f));
    float k = roughness + 1.0f;
    k = (k * k) * 0.125f;
    const float G = (NdotV * rcp(fmaf(NdotV, 1.0f - k, k))) * (NdotL * rcp(fmaf(NdotL, 1.0f - k, k)));
    const float om = 1.0f - VdotH;
    const float om2 = om * om;
    const float p5 = om2 * om2 * om;
    const F3 F = f3(fmaf(1.0f - F0.x, p5, F0.x), fmaf(1.0f - F0.y, p5, F0.y), fmaf(1.0f - F0.z, p5, F0.z));
    const float sc = (D * G) * rcp(fmaxf(4.0f * NdotL * NdotV, 1e-4f));
    const float kdm = 1.0f - metallic;
    return f3(fmaf((1.0f - F.x) * kdm, albedo.x, sc * F.x) * NdotL, fmaf((1.0f - F.y) * kdm, albedo.y, sc * F.y) * NdotL,
              fmaf((1.0f - F.z) * kdm, albedo.z, sc * F.z) * NdotL);
}

__device__ __forceinline__ F3 shade_pixel(const LightingParams& p, const float* srgb, uint32_t px, uint32_t py, ur_half4 a, ur_half4 b, uint32_t c)
{
    const float nx = h2f(a.x), ny = h2f(a.y), nz = h2f(a.z);
    const float nr = rsq(fmaf(nz, nz, fmaf(ny, ny, nx * nx))); // normalize(0) = NaN, as in the reference
    const F3 N = f3(nx * nr, ny * nr, nz * nr);
    const float viewZ = -h2f(a.w);
    const float spec0 = h2f(b.x), metallic = h2f(b.y), roughness = h2f(b.z);
    const F3 albedo = f3(srgb[c & 0xFFu], srgb[(c >> 8) & 0xFFu], srgb[(c >> 16) & 0xFFu]);
    const F3 F0 = mix(f3(spec0, spec0, spec0), albedo, metallic);

    const float ndcx = fmaf(((float)px + 0.5f) * p.invW, 2.0f, -1.0f);
    const float ndcy = fmaf(((float)py + 0.5f) * p.invH, 2.0f, -1.0f);
    const F3 viewPos = f3(ndcx * viewZ * p.invP11, -ndcy * viewZ * p.invP22, viewZ);
    const float vr = rsq(dot(viewPos, viewPos));
    const F3 V = f3(-viewPos.x * vr, -viewPos.y * vr, -viewPos.z * vr);
    const F3 L = f3(p.L[0], p.L[1], p.L[2]);

    // shadow: viewPos -> light clip through the host-composed matrix
    float shadow = 1.0f;
    if (p.shadowStrength > 0.0f) {
        const float* M = p.SM;
        const float sx = fmaf(viewPos.z, M[8], fmaf(viewPos.y, M[4], fmaf(viewPos.x, M[0], M[12])));
        const float sy = fmaf(viewPos.z, M[9], fmaf(viewPos.y, M[5], fmaf(viewPos.x, M[1], M[13])));
        const float sz = fmaf(viewPos.z, M[10], fmaf(viewPos.y, M[6], fmaf(viewPos.x, M[2], M[14])));
        const float sw = fmaf(viewPos.z, M[11], fmaf(viewPos.y, M[7], fmaf(viewPos.x, M[3], M[15])));
        const float iw = rcp(sw);
        const float su = fmaf(sx * iw, 0.5f, 0.5f), sv = fmaf(sy * iw, -0.5f, 0.5f);
        if (su >= 0.0f && sv >= 0.0f && su <= 1.0f && sv <= 1.0f) {
            const float cmp = sz * iw - p.shadowBias;
            const float s = 0.25f * (shadow_cmp(p, su, sv, cmp) + shadow_cmp(p, su + p.shadowTexelX, sv, cmp) +
                                     shadow_cmp(p, su, sv + p.shadowTexelY, cmp) + shadow_cmp(p, su + p.shadowTexelX, sv + p.shadowTexelY, cmp));
            shadow = mix(1.0f, s, p.shadowStrength);
        }
    }

    const float NdotVraw = dot(N, V);
    const float NdotV = sat(NdotVraw);
    const F3 direct = EvaluatePBR(albedo, metallic, roughness, F0, N, V, L, NdotV);

    // IBL: world vectors are the view-space ones rotated by (float3x3)ViewInverse
    const float* R = p.R;
    const F3 Nw = f3(fmaf(N.z, R[6], fmaf(N.y, R[3], N.x * R[0])), fmaf(N.z, R[7], fmaf(N.y, R[4], N.x * R[1])),
                     fmaf(N.z, R[8], fmaf(N.y, R[5], N.x * R[2])));
    // reflect(-V, N) = -V + 2 N dot(N, V)
    const float t2 = 2.0f * NdotVraw;
    const F3 Rv = f3(fmaf(t2, N.x, -V.x), fmaf(t2, N.y, -V.y), fmaf(t2, N.z, -V.z));
    const F3 Rw = f3(fmaf(Rv.z, R[6], fmaf(Rv.y, R[3], Rv.x * R[0])), fmaf(Rv.z, R[7], fmaf(Rv.y, R[4], Rv.x * R[1])),
                     fmaf(Rv.z, R[8], fmaf(Rv.y, R[5], Rv.x * R[2])));
    const F3 prefiltered = cube_sample_level(p, Rw, roughness * p.maxMip);
    float ba, bb;
    lut_sample(p, NdotV, roughness, ba, bb);
    const F3 irradiance = cube_sample_level(p, Nw, p.maxMip);
    const float kdm = 1.0f - metallic;
    F3 color;
    color.x = fmaf(direct.x * p.lightRGB[0], shadow, fmaf(irradiance.x * albedo.x, kdm, prefiltered.x * fmaf(F0.x, ba, bb)));
    color.y = fmaf(direct.y * p.lightRGB[1], shadow, fmaf(irradiance.y * albedo.y, kdm, prefiltered.y * fmaf(F0.y, ba, bb)));
    color.z = fmaf(direct.z * p.lightRGB[2], shadow, fmaf(irradiance.z * albedo.z, kdm, prefiltered.z * fmaf(F0.z, ba, bb)));
    return color;
}

// SkyAtmosphere.hlsl:58-93 with the camera-height densities and sun attenuation folded on the host.
__device__ __forceinline__ F3 sky_pixel(const LightingParams& p, F3 v /*view-space ray, z = 1*/)
{
    const float* Q = p.skyRot;
    F3 w = f3(fmaf(v.z, Q[2], fmaf(v.y, Q[1], v.x * Q[0])), fmaf(v.z, Q[5], fmaf(v.y, Q[4], v.x * Q[3])),
              fmaf(v.z, Q[8], fmaf(v.y, Q[7], v.x * Q[6])));
    const float wr = rsq(dot(w, w));
    w = f3(w.x * wr, w.y * wr, w.z * wr);
    const float h = 1.0f - sat(fmaf(w.y, 0.5f, 0.5f));
    const float falloff = sat(h * h * h);
    const float cosSunView = dot(w, f3(p.sunDir[0], p.sunDir[1], p.sunDir[2]));
    const float rayleighPhase = (3.0f / (16.0f * 3.14159265f)) * fmaf(cosSunView, cosSunView, 1.0f);
    const float g = 0.76f, g2 = g * g;
    const float mb = fmaf(-2.0f * g, cosSunView, 1.0f + g2);
    const float denom = mb * __builtin_amdgcn_sqrtf(mb); // pow(x, 1.5)
    const float miePhase = (1.0f - g2) * rcp(4.0f * 3.14159265f * fmaxf(denom, 1e-3f));
    F3 c;
    c.x = fmaf(fmaf(p.skyMie[0], miePhase, p.skyScatterR[0] * rayleighPhase), p.sunAttenuation, mix(0.05f, 0.52f, falloff));
    c.y = fmaf(fmaf(p.skyMie[1], miePhase, p.skyScatterR[1] * rayleighPhase), p.sunAttenuation, mix(0.12f, 0.68f, falloff));
    c.z = fmaf(fmaf(p.skyMie[2], miePhase, p.skyScatterR[2] * rayleighPhase), p.sunAttenuation, mix(0.22f, 0.86f, falloff));
    return c;
}

template <int MODE>
__global__ __launch_bounds__(256) void lighting_kernel(LightingParams p)
{
    __shared__ float srgb[256];
    if (MODE != ur::UR_MODE_SKY) {
        srgb[threadIdx.x] = p.srgb[threadIdx.x];
        __syncthreads();
    }
    const uint32_t total = p.W * p.rows;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t r = i / p.W, px = i - r * p.W, py = p.row0 + r;
        bool isSky = false;
        F3 sky = f3(0, 0, 0);
        if (MODE != ur::UR_MODE_LIGHTING) {
            const float vx = fmaf(((float)px + 0.5f) * p.invW, 2.0f, -1.0f) * p.skyInvP11;
            const float vy = fmaf(((float)py + 0.5f) * p.invH, -2.0f, 1.0f) * p.skyInvP22;
            const float len = __builtin_amdgcn_sqrtf(fmaf(vx, vx, fmaf(vy, vy, 1.0f)));
            const float skyDepth = p.skyNearOverR * len; // Near / (R * (1/len))
            isSky = skyDepth >= p.depth[i];
            if (isSky) sky = sky_pixel(p, f3(vx, vy, 1.0f));
        }
        if (MODE == ur::UR_MODE_SKY) {
            if (isSky) p.hdr[i] = {f2h(sky.x), f2h(sky.y), f2h(sky.z), (uint16_t)0x3C00u};
            continue;
        }
        if (MODE == ur::UR_MODE_FUSED && isSky) {
            p.hdr[i] = {f2h(sky.x), f2h(sky.y), f2h(sky.z), (uint16_t)0x3C00u};
            continue;
        }
        const ur_half4 a = p.A[i], b = p.B[i];
        const uint32_t c = p.C[i];
        const ur_half4 d = p.hdr[i];
        const F3 col = shade_pixel(p, srgb, px, py, a, b, c);
        p.hdr[i] = {f2h(h2f(d.x) + col.x), f2h(h2f(d.y) + col.y), f2h(h2f(d.z) + col.z), f2h(h2f(d.w) + 1.0f)};
    }
}

void mat4_mul(const float* a, const float* b, float* o)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            o[i * 4 + j] = s;
        }
}

} // namespace

namespace ur {

int launch_lighting(ur_ctx* ctx, const ur_scene_constants* S, const ur_sky_constants* K, const ur_half4* A, const ur_half4* B,
                    const uint32_t* C, const float* depth, const ur_lighting_tables* T, ur_half4* hdr, uint32_t w, uint32_t h,
                    uint32_t row0, uint32_t rows, int mode)
{
    LightingParams p{};
    p.W = w; p.H = h; p.row0 = row0; p.rows = rows;
    p.invW = 1.0f / (float)w; p.invH = 1.0f / (float)h;
    p.A = A; p.B = B; p.C = C; p.depth = depth; p.hdr = hdr; p.srgb = ctx->srgb_table;
    if (mode != UR_MODE_SKY) {
        // the view matrix must be rigid: rows of (float3x3)ViewInverse orthonormal
        const float* VI = S->ViewInverse;
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j) {
                const float d = VI[i * 4] * VI[j * 4] + VI[i * 4 + 1] * VI[j * 4 + 1] + VI[i * 4 + 2] * VI[j * 4 + 2];
                if (std::fabs(d - (i == j ? 1.0f : 0.0f)) > 1e-3f) {
                    set_error("ViewInverse is not a rigid transform (row %d . row %d = %g)", i, j, d);
                    return UR_EUNSUPPORTED;
                }
            }
        p.invP11 = 1.0f / S->Projection[0];
        p.invP22 = 1.0f / S->Projection[5];
        const float* V = S->View;
        const float* LD = S->LightDirection;
        float l[3];
        for (int j = 0; j < 3; ++j) l[j] = (LD[0] * V[j] + LD[1] * V[4 + j]) + LD[2] * V[8 + j];
        const float lr = 1.0f / std::sqrt((l[0] * l[0] + l[1] * l[1]) + l[2] * l[2]);
        for (int j = 0; j < 3; ++j) p.L[j] = l[j] * lr;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) p.R[i * 3 + j] = VI[i * 4 + j];
        mat4_mul(S->ViewInverse, S->LightViewProjection, p.SM);
        for (int j = 0; j < 3; ++j) p.lightRGB[j] = S->LightIntensity * S->LightColor[j];
        p.shadowStrength = S->ShadowStrength;
        p.shadowBias = S->ShadowBias;
        p.shadowW = S->ShadowMapSize[0]; p.shadowH = S->ShadowMapSize[1];
        p.shadowWi = (uint32_t)S->ShadowMapSize[0]; p.shadowHi = (uint32_t)S->ShadowMapSize[1];
        p.shadowTexelX = 1.0f / S->ShadowMapSize[0]; p.shadowTexelY = 1.0f / S->ShadowMapSize[1];
        p.shadow = T->shadow_map;
        if (p.shadowStrength > 0.0f && (p.shadow == nullptr || p.shadowWi == 0 || p.shadowHi == 0)) {
            set_error("ShadowStrength > 0 but no shadow map / ShadowMapSize");
            return UR_EINVAL;
        }
        p.maxMip = std::fmax(0.0f, S->EnvMapMipCount - 1.0f);
        p.envBase = T->env_base_size; p.envMips = T->env_mip_count;
        if (p.envMips == 0 || p.envMips > 16 || p.envBase == 0 || T->env_cube == nullptr || T->brdf_lut_rg16 == nullptr ||
            T->lut_width == 0 || T->lut_height == 0) {
            set_error("bad lighting tables");
            return UR_EINVAL;
        }
        uint32_t off = 0;
        for (uint32_t m = 0; m < p.envMips; ++m) {
            p.envMipOffset[m] = off;
            const uint32_t e = (p.envBase >> m > 1u ? p.envBase >> m : 1u) + 2u;
            off += 6u * e * e;
        }
        p.env = T->env_cube;
        p.lut = reinterpret_cast<const uint32_t*>(T->brdf_lut_rg16);
        p.lutW = T->lut_width; p.lutH = T->lut_height;
    }
    if (mode != UR_MODE_LIGHTING) {
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) p.skyRot[j * 3 + i] = K->View[j * 4 + i];
        p.skyInvP11 = 1.0f / K->Projection[0];
        p.skyInvP22 = 1.0f / K->Projection[5];
        p.skyNearOverR = K->Projection[14] / K->World[0];
        const float* LD = K->LightDirection;
        const float lr = 1.0f / std::sqrt((LD[0] * LD[0] + LD[1] * LD[1]) + LD[2] * LD[2]);
        for (int j = 0; j < 3; ++j) p.sunDir[j] = LD[j] * lr;
        const float viewHeight = std::fmax(0.0f, K->CameraPosition[1]);
        const float rayleighDensity = std::exp(-viewHeight / 8000.0f), mieDensity = std::exp(-viewHeight / 1200.0f);
        const float rayleighColor[3] = {0.650f, 0.570f, 0.475f};
        for (int j = 0; j < 3; ++j) {
            p.skyScatterR[j] = rayleighColor[j] * rayleighDensity;
            p.skyMie[j] = K->LightColor[j] * mieDensity * 0.8f;
        }
        const float cosSunUp = p.sunDir[1];
        p.sunAttenuation = std::fmin(std::fmax(std::exp(-std::fmax(0.0f, 1.0f - cosSunUp) * 2.0f), 0.0f), 1.0f);
    }
    const uint64_t total = (uint64_t)w * rows;
    if (total == 0) return UR_OK;
    if (total > 0xFFFFFFFFull) {
        set_error("band too large");
        return UR_EUNSUPPORTED;
    }
    uint32_t blocks = (uint32_t)((total + 255u) / 256u);
    const uint32_t cap = (uint32_t)ctx->cu_count * 8u * 4u;
    if (blocks > cap) blocks = cap;
    switch (mode) {
    case UR_MODE_LIGHTING: hipLaunchKernelGGL(lighting_kernel<UR_MODE_LIGHTING>, dim3(blocks), dim3(256), 0, ctx->stream, p); break;
    case UR_MODE_SKY: hipLaunchKernelGGL(lighting_kernel<UR_MODE_SKY>, dim3(blocks), dim3(256), 0, ctx->stream, p); break;
    default: hipLaunchKernelGGL(lighting_kernel<UR_MODE_FUSED>, dim3(blocks), dim3(256), 0, ctx->stream, p); break;
    }
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

} // namespace ur
