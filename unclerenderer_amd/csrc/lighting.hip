// DeferredLighting (GGX + IBL) and SkyAtmosphere for gfx950, as one per-pixel compute kernel with three modes.
//
// Reference: Shaders/DeferredLighting.hlsl:35-94 + Shaders/PBRCommon.hlsl:1-48 (fullscreen-triangle pixel shader,
// additive ONE/ONE blend into RGBA16F, Source/Render/DeferredRenderer.cpp:1219-1255,1997-2005) and
// Shaders/SkyAtmosphere.hlsl:40-101 (inside-out sphere, depth GREATER_EQUAL, no blend, DeferredRenderer.cpp:1263-1296).
//
// There is no rasteriser and no texture unit here. One lane shades one pixel; a wave64 covers a TILE_W x TILE_H pixel
// tile so that the G-buffer loads are 8 bytes per lane over contiguous row segments and the shadow / cube / LUT gathers
// of neighbouring lanes land on neighbouring texels. The PCF, the trilinear cube lookups and the BRDF LUT are filtered
// in ALU from plain loads (texels fetched as whole 8-byte half4s and fed to mixed-precision FMAs).
// Every per-launch uniform the HLSL recomputes per pixel (light vector in view space, reciprocals, the camera-ray ->
// shadow-clip matrix, sky densities) is folded on the host into LightingParams: gfx950 has no scalar fp32 ALU, so
// uniform math would otherwise run on the VALU for all 64 lanes. The view matrix is rigid (XMMatrixLookToLH,
// Scene/Camera.cpp:23-31), so the world-space IBL vectors are the view-space ones rotated by ViewInverse.
// The kernel is VALU/HBM co-limited: the instruction count per pixel is the budget that matters.
// Tolerance against the oracle: max(1e-3, 1 ulp fp16) per channel; fp32 math, one RTE to fp16.

#include "ur_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

struct LightingParams {
    // frame
    uint32_t W, H, row0, rows;
    float invW2, invH2;  // 2/W, 2/H
    // lighting
    float invP11, invP22;
    float L[3];          // normalize(mul(float4(LightDirection,0), View).xyz)
    float R[9];          // (float3x3)ViewInverse, row-major
    float SQ[12];        // rows 0..2 of (ViewInverse * LightViewProjection), columns x,y,z,w : applied to the camera ray (a,b,1)
    float ST[4];         // row 3 of the same matrix
    float lightRGB[3];   // LightIntensity * LightColor
    float shadowStrength, shadowBias;
    float shadowW, shadowH, shadowTexelX, shadowTexelY;
    int32_t shadowWi, shadowHi;
    float maxMip;        // max(0, EnvMapMipCount-1)
    uint32_t envBase, envMips;
    uint32_t envMipOffset[16]; // in half4 texels
    uint32_t irrOffset0, irrOffset1, irrN0, irrN1; // mip pair of the irradiance lookup (level == maxMip, launch-uniform)
    float irrFrac;
    uint32_t lutW, lutH;
    // sky
    float skyRot[9];     // rows of View's 3x3: world_j = dot(skyRot[3j..3j+2], v)
    float skyInvP11, skyInvP22;
    float skyNearOverR;  // Projection[14] / World[0]
    float sunDir[3];     // normalize(LightDirection)
    float skyScatterR[3];// rayleighColor * rayleighDensity * 3/(16 pi)
    float skyMie[3];     // LightColor * mieDensity * 0.8 * (1-g^2)/(4 pi)
    float sunAttenuation;
    // buffers
    const half4_t* A;
    const half4_t* B;
    const uint32_t* C;
    const float* depth;
    const float* shadow;
    const half4_t* env;
    const uint32_t* lut; // RG16 texel = one dword
    const float* srgb;
    half4_t* hdr;
};

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float mix(float a, float b, float t) { return fmaf(t, b - a, a); }

// base + 32-bit unsigned BYTE offset: lets the compiler use the SGPR-base + VGPR-offset addressing mode of global_load
// instead of 64-bit VALU address arithmetic (v_lshl_add_u64 per access).
template <class T>
__device__ __forceinline__ T ld(const void* base, uint32_t byte_offset)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_offset);
}
template <class T>
__device__ __forceinline__ void st(void* base, uint32_t byte_offset, T v)
{
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_offset) = v;
}

struct __attribute__((packed, aligned(8))) uint4u { uint32_t x, y, z, w; }; // 16 bytes, 8-byte aligned
struct __attribute__((packed, aligned(4))) float3u { float x, y, z; };      // 12 bytes, 4-byte aligned

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 mix(F3 a, F3 b, float t) { return {mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)}; }
// v * M for a row-major 3x3
__device__ __forceinline__ F3 rot(F3 v, const float* M)
{
    return f3(fmaf(v.z, M[6], fmaf(v.y, M[3], v.x * M[0])), fmaf(v.z, M[7], fmaf(v.y, M[4], v.x * M[1])),
              fmaf(v.z, M[8], fmaf(v.y, M[5], v.x * M[2])));
}

// ---- bordered cube: face f of mip m is (N+2)^2 texels, border = seamless neighbours (ur_stage_env_cube) -----------
struct CubeUV { uint32_t face; float u, v; };
// D3D cube addressing (+X,-X,+Y,-Y,+Z,-Z; ties z > y > x; uc/vc table of the oracle's SelectCubeFace) is exactly what
// gfx950's v_cubeid/v_cubesc/v_cubetc/v_cubema compute (cubema = 2 * signed major axis), four instructions instead of a
// compare/select ladder.
__device__ __forceinline__ CubeUV cube_face(F3 d)
{
    CubeUV r;
    r.face = (uint32_t)__builtin_amdgcn_cubeid(d.x, d.y, d.z);
    const float inv = rcp(fabsf(__builtin_amdgcn_cubema(d.x, d.y, d.z))); // 1 / (2 |major|)
    r.u = fmaf(__builtin_amdgcn_cubesc(d.x, d.y, d.z), inv, 0.5f);
    r.v = fmaf(__builtin_amdgcn_cubetc(d.x, d.y, d.z), inv, 0.5f);
    return r;
}

// acc += w * f16(lo/hi half of a packed dword): one mixed-precision FMA, no unpack/convert instructions
__device__ __forceinline__ float mix_lo(float acc, uint32_t packed, float w)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}
__device__ __forceinline__ float mix_hi(float acc, uint32_t packed, float w)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}
// the same without an addend (first tap of a sum: no zero-initialised accumulator register)
__device__ __forceinline__ float mul_lo(uint32_t packed, float w)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(w));
    return r;
}
__device__ __forceinline__ float mul_hi(uint32_t packed, float w)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(w));
    return r;
}

// ---- gathers are split into "issue the loads" and "filter" so one pixel has every independent gather in flight
// before the BRDF math starts (the math hides their latency; no branch separates them) -----------------------------
struct CubeTaps { uint4u r0, r1; float fx, fy; };

// u,v in [0,1] (a NaN direction gives index 0 and NaN weights, i.e. a NaN result, like the reference).
__device__ __forceinline__ CubeTaps cube_taps_load(const void* __restrict__ env, uint32_t mipOffset, uint32_t N, const CubeUV& c)
{
    const uint32_t E = N + 2u;
    const float fN = (float)N;
    const float x = fmaf(c.u, fN, 0.5f), y = fmaf(c.v, fN, 0.5f); // bordered coordinates, in [0.5, N + 0.5]
    const uint32_t i0 = (uint32_t)x, j0 = (uint32_t)y;            // truncation == floor for x >= 0; NaN -> 0
    CubeTaps t;
    t.fx = x - (float)i0;
    t.fy = y - (float)j0;
    const uint32_t off = (mipOffset + (c.face * E + j0) * E + i0) * 8u, row = E * 8u;
    // the two taps of a row are adjacent in memory: one 16-byte load per row (8-byte aligned; gfx950 loads may be unaligned)
    t.r0 = ld<uint4u>(env, off);
    t.r1 = ld<uint4u>(env, off + row);
    return t;
}

// scale * bilinear(taps) [+ r when ACC]: 12 mixed-precision FMAs straight from the packed fp16 texels
template <bool ACC>
__device__ __forceinline__ void cube_taps_filter(F3& r, const CubeTaps& t, float scale)
{
    const float wy1 = t.fy * scale, wy0 = scale - wy1;
    const float w10 = wy0 * t.fx, w00 = wy0 - w10, w11 = wy1 * t.fx, w01 = wy1 - w11;
    const float x0 = ACC ? mix_lo(r.x, t.r0.x, w00) : mul_lo(t.r0.x, w00);
    const float y0 = ACC ? mix_hi(r.y, t.r0.x, w00) : mul_hi(t.r0.x, w00);
    const float z0 = ACC ? mix_lo(r.z, t.r0.y, w00) : mul_lo(t.r0.y, w00);
    r.x = mix_lo(mix_lo(mix_lo(x0, t.r0.z, w10), t.r1.x, w01), t.r1.z, w11);
    r.y = mix_hi(mix_hi(mix_hi(y0, t.r0.z, w10), t.r1.x, w01), t.r1.z, w11);
    r.z = mix_lo(mix_lo(mix_lo(z0, t.r0.w, w10), t.r1.y, w01), t.r1.w, w11);
}

struct LutTaps { uint32_t t00, t10, t01, t11; float fx, fy; };
__device__ __forceinline__ LutTaps lut_taps_load(const LightingParams& p, float u, float v)
{
    const float x = fmaf(u, (float)p.lutW, -0.5f), y = fmaf(v, (float)p.lutH, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y);
    LutTaps t;
    t.fx = x - x0;
    t.fy = y - y0;
    const int W1 = (int)p.lutW - 1, H1 = (int)p.lutH - 1;
    const int i0 = min(max((int)x0, 0), W1), i1 = min(max((int)x0 + 1, 0), W1); // clamp addressing
    const int j0 = min(max((int)y0, 0), H1), j1 = min(max((int)y0 + 1, 0), H1);
    const uint32_t r0 = (uint32_t)j0 * p.lutW, r1 = (uint32_t)j1 * p.lutW;
    t.t00 = ld<uint32_t>(p.lut, (r0 + i0) * 4u);
    t.t10 = ld<uint32_t>(p.lut, (r0 + i1) * 4u);
    t.t01 = ld<uint32_t>(p.lut, (r1 + i0) * 4u);
    t.t11 = ld<uint32_t>(p.lut, (r1 + i1) * 4u);
    return t;
}
__device__ __forceinline__ void lut_taps_filter(const LutTaps& t, float& a, float& b)
{
    const float s = 1.0f / 65535.0f;
    const float wy1 = t.fy * s, wy0 = s - wy1;
    const float w10 = wy0 * t.fx, w00 = wy0 - w10, w11 = wy1 * t.fx, w01 = wy1 - w11;
    a = fmaf(w11, (float)(t.t11 & 0xFFFFu), fmaf(w01, (float)(t.t01 & 0xFFFFu), fmaf(w10, (float)(t.t10 & 0xFFFFu), w00 * (float)(t.t00 & 0xFFFFu))));
    b = fmaf(w11, (float)(t.t11 >> 16), fmaf(w01, (float)(t.t01 >> 16), fmaf(w10, (float)(t.t10 >> 16), w00 * (float)(t.t00 >> 16))));
}

// step(t) = (cmp <= t) as saturate((t - cmp) * 2^126 + 1): a full-rate subtract + clamped FMA instead of the half-rate
// v_cmp + v_cndmask pair; exact for every normal pair (equality gives 1, NaN gives 0 like the comparison).
__device__ __forceinline__ float step_le(float cmp, float t)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(r) : "v"(t - cmp), "v"(0x1p126f));
    return r;
}

// The four PCF samples of DeferredLighting.hlsl:62-70: SampleCmpLevelZero (bilinear blend of four LESS_EQUAL results,
// border = 1.0) at (u, u + 1 texel) x (v, v + 1 texel). The second sample's footprint is the first's shifted by exactly
// one texel, so the union is a 3x3 block and the sum of the four bilinear blends factors into separable weights
// (1-f, 1, f) per axis: 9 loads, 9 compares. (The oracle evaluates the shifted coordinate (u + 1/W) * W - 0.5 in fp32;
// its fraction differs from f by O(1e-4), i.e. O(1e-5) in the result — far inside the HDR tolerance.)
struct ShadowTaps { float3u ra, rb, rc; float fx, fy; int ia, ja; };
__device__ __forceinline__ ShadowTaps shadow_taps_load(const LightingParams& p, float su, float sv)
{
    const float xa = fmaf(su, p.shadowW, -0.5f), ya = fmaf(sv, p.shadowH, -0.5f);
    const float xa0 = floorf(xa), ya0 = floorf(ya);
    ShadowTaps t;
    t.fx = xa - xa0;
    t.fy = ya - ya0;
    t.ia = (int)xa0;
    t.ja = (int)ya0;
    // clamped block origin: always a valid address (the host rejects shadow maps smaller than 3x3)
    const uint32_t ic = (uint32_t)min(max(t.ia, 0), p.shadowWi - 3), jc = (uint32_t)min(max(t.ja, 0), p.shadowHi - 3);
    const uint32_t W = (uint32_t)p.shadowWi;
    const uint32_t o0 = (jc * W + ic) * 4u, o1 = o0 + W * 4u, o2 = o1 + W * 4u;
    t.ra = ld<float3u>(p.shadow, o0); // one 12-byte load per row
    t.rb = ld<float3u>(p.shadow, o1);
    t.rc = ld<float3u>(p.shadow, o2);
    return t;
}
__device__ __forceinline__ float shadow_taps_filter(const ShadowTaps& t, float cmp)
{
    const float wx0 = 1.0f - t.fx, wy0 = 1.0f - t.fy;
    const float r0 = fmaf(step_le(cmp, t.ra.z), t.fx, fmaf(step_le(cmp, t.ra.x), wx0, step_le(cmp, t.ra.y)));
    const float r1 = fmaf(step_le(cmp, t.rb.z), t.fx, fmaf(step_le(cmp, t.rb.x), wx0, step_le(cmp, t.rb.y)));
    const float r2 = fmaf(step_le(cmp, t.rc.z), t.fx, fmaf(step_le(cmp, t.rc.x), wx0, step_le(cmp, t.rc.y)));
    return 0.25f * fmaf(t.fy, r2, fmaf(wy0, r0, r1));
}
// footprint touches the border (or the map is tiny): out-of-range taps read the border colour 1.0
__device__ __noinline__ float shadow_pcf_border(const float* __restrict__ map, int W, int H, int ia, int ja, float fx, float fy, float cmp)
{
    float acc = 0.0f;
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        for (int c = 0; c < 3; ++c) {
            const int xi = ia + c, yj = ja + r;
            const bool in = xi >= 0 && yj >= 0 && xi < W && yj < H;
            const float t = in ? map[(uint32_t)yj * (uint32_t)W + (uint32_t)xi] : 1.0f;
            s += cmp <= t ? (c == 0 ? 1.0f - fx : (c == 1 ? 1.0f : fx)) : 0.0f;
        }
        acc = fmaf(r == 0 ? 1.0f - fy : (r == 1 ? 1.0f : fy), s, acc);
    }
    return 0.25f * acc;
}

// DeferredLighting.hlsl:35-94 for one pixel. (a,b) = camera ray (ndc.x/P11, -ndc.y/P22); viewPos = viewZ * (a, b, 1).
template <class T, class = void> struct ur_has_fence { static constexpr bool value = true; };
template <class T> struct ur_has_fence<T, std::void_t<decltype(std::decay_t<T>::kFence)>> { static constexpr bool value = std::decay_t<T>::kFence; };
struct NoPrefetch { static constexpr bool kFence = false; __device__ __forceinline__ void operator()() const {} };

// `mipOffset`: the per-mip texel offsets (LDS copy in the persistent kernel: a per-lane indexed read of the kernarg copy is
// a dependent global load in front of the cube gathers). `after_gathers` runs once every gather of this pixel has been
// issued and before their results are consumed: loads issued inside it are YOUNGER than the gathers, so the waits on the
// gathers (in-order vmcnt) do not wait for them — that is where the persistent kernel prefetches the next tile.
template <bool SHADOWS, class AfterGathers>
__device__ __forceinline__ F3 shade_pixel(const LightingParams& p, const float* srgb, const uint32_t* mipOffset, float ra, float rb, half4_t ga, half4_t gb,
                                          uint32_t gc, AfterGathers&& after_gathers)
{
    // ---- decode, view vectors ------------------------------------------------------------------------------------------
    const float nx = (float)ga.x, ny = (float)ga.y, nz = (float)ga.z;
    const float nr = rsq(fmaf(nz, nz, fmaf(ny, ny, nx * nx))); // normalize(0) = NaN, as in the reference
    const F3 N = f3(nx * nr, ny * nr, nz * nr);
    const float viewZ = -(float)ga.w;
    const float spec0 = (float)gb.x, metallic = (float)gb.y, roughness = (float)gb.z;
    // V = normalize(-viewPos) = -sign(viewZ) * (a,b,1)/|(a,b,1)|
    const float rl = rsq(fmaf(ra, ra, fmaf(rb, rb, 1.0f)));
    const float vs = viewZ > 0.0f ? -rl : (viewZ < 0.0f ? rl : __builtin_nanf("")); // normalize(0) = NaN
    const F3 V = f3(ra * vs, rb * vs, vs);
    const F3 L = f3(p.L[0], p.L[1], p.L[2]);
    const float NdotVraw = dot(N, V);
    const float NdotV = sat(NdotVraw);

    // ---- issue every gather ------------------------------------------------------------------------------------------
    // IBL: world vectors are the view-space ones rotated by (float3x3)ViewInverse; reflect(-V, N) = 2 N (N.V) - V
    const float t2 = 2.0f * NdotVraw;
    const CubeUV cr = cube_face(rot(f3(fmaf(t2, N.x, -V.x), fmaf(t2, N.y, -V.y), fmaf(t2, N.z, -V.z)), p.R));
    const CubeUV cn = cube_face(rot(N, p.R));
    const float lvl = fminf(fmaxf(roughness * p.maxMip, 0.0f), (float)(p.envMips - 1u));
    const uint32_t m0 = (uint32_t)lvl, m1 = min(m0 + 1u, p.envMips - 1u);
    const float fl = lvl - (float)m0; // m1 == m0 only when fl == 0: the second mip then carries weight 0
    const CubeTaps pre0 = cube_taps_load(p.env, mipOffset[m0], max(1u, p.envBase >> m0), cr);
    const CubeTaps pre1 = cube_taps_load(p.env, mipOffset[m1], max(1u, p.envBase >> m1), cr);
    const CubeTaps irr0 = cube_taps_load(p.env, p.irrOffset0, p.irrN0, cn);
    const LutTaps lut = lut_taps_load(p, NdotV, roughness);
    // The shadow term multiplies NdotL: a wave whose every pixel faces away from the light skips the PCF altogether
    // (same result: direct = 0). Coherent G-buffers make this common (ceilings, walls turned from the sun).
    const float NdotL = sat(dot(N, L));
    const bool wave_lit = SHADOWS && __any(NdotL > 0.0f);
    float su = 0.0f, sv = 0.0f, cmp = 0.0f;
    bool lit = false;
    ShadowTaps sh;
    if (wave_lit) {
        // shadow clip = viewZ * ((a,b,1) * M3) + M[3]
        const float qx = fmaf(rb, p.SQ[4], fmaf(ra, p.SQ[0], p.SQ[8]));
        const float qy = fmaf(rb, p.SQ[5], fmaf(ra, p.SQ[1], p.SQ[9]));
        const float qz = fmaf(rb, p.SQ[6], fmaf(ra, p.SQ[2], p.SQ[10]));
        const float qw = fmaf(rb, p.SQ[7], fmaf(ra, p.SQ[3], p.SQ[11]));
        const float iw = rcp(fmaf(viewZ, qw, p.ST[3]));
        su = fmaf(fmaf(viewZ, qx, p.ST[0]) * iw, 0.5f, 0.5f);
        sv = fmaf(fmaf(viewZ, qy, p.ST[1]) * iw, -0.5f, 0.5f);
        cmp = fmaf(viewZ, qz, p.ST[2]) * iw - p.shadowBias;
        lit = su >= 0.0f && sv >= 0.0f && su <= 1.0f && sv <= 1.0f;
        sh = shadow_taps_load(p, su, sv);
    }
    const F3 albedo = f3(srgb[gc & 0xFFu], srgb[(gc >> 8) & 0xFFu], srgb[(gc >> 16) & 0xFFu]);
    if constexpr (ur_has_fence<AfterGathers>::value) {
        __builtin_amdgcn_sched_barrier(0); // keep the prefetch loads behind the gathers in issue order
        after_gathers();
        __builtin_amdgcn_sched_barrier(0);
    } else {
        after_gathers();
    }

    // ---- EvaluatePBR, PBRCommon.hlsl:24-48 (runs while the gathers are in flight) --------------------------------------------
    const F3 F0 = mix(f3(spec0, spec0, spec0), albedo, metallic);
    F3 Hv = f3(V.x + L.x, V.y + L.y, V.z + L.z);
    const float hr = rsq(dot(Hv, Hv));
    const float NdotH = sat(dot(N, Hv) * hr);
    const float VdotH = dot(V, Hv) * hr; // = (1 + V.L)/|V + L| in [0,1]: saturate is the identity up to rounding
    const float alpha = roughness * roughness;
    const float alpha2 = alpha * alpha;
    const float denom = fmaf(NdotH * NdotH, alpha2 - 1.0f, 1.0f);
    const float D = alpha2 * rcp(fmaxf(3.14159265f * denom * denom, 1e-4f));
    float k = roughness + 1.0f;
    k = (k * k) * 0.125f;
    const float omk = 1.0f - k;
    // G / max(4 NdotL NdotV, 1e-4) * D, one reciprocal for the three denominators
    const float gv = fmaf(NdotV, omk, k), gl = fmaf(NdotL, omk, k);
    const float sc = (D * NdotV * NdotL) * rcp(gv * gl * fmaxf(4.0f * NdotL * NdotV, 1e-4f));
    const float om = 1.0f - VdotH;
    const float om2 = om * om;
    const float p5 = om2 * om2 * om;
    const float kdm = 1.0f - metallic;

    // ---- filter ---------------------------------------------------------------------------------------------------------------
    float shadow = 1.0f;
    if (wave_lit) {
        const bool fast = sh.ia >= 0 && sh.ja >= 0 && sh.ia + 2 < p.shadowWi && sh.ja + 2 < p.shadowHi;
        float s = shadow_taps_filter(sh, cmp);
        if (__builtin_expect(lit && !fast, 0)) {
            const float xa = fmaf(su, p.shadowW, -0.5f), ya = fmaf(sv, p.shadowH, -0.5f);
            const float xa0 = floorf(xa), ya0 = floorf(ya);
            s = shadow_pcf_border(p.shadow, p.shadowWi, p.shadowHi, (int)xa0, (int)ya0, xa - xa0, ya - ya0, cmp);
        }
        shadow = lit ? mix(1.0f, s, p.shadowStrength) : 1.0f;
    }
    const float sh_l = shadow * NdotL;
    F3 prefiltered, irradiance;
    cube_taps_filter<false>(prefiltered, pre0, 1.0f - fl);
    cube_taps_filter<true>(prefiltered, pre1, fl);
    cube_taps_filter<false>(irradiance, irr0, 1.0f - p.irrFrac);
    if (p.irrFrac != 0.0f) cube_taps_filter<true>(irradiance, cube_taps_load(p.env, p.irrOffset1, p.irrN1, cn), p.irrFrac); // uniform
    float ba, bb;
    lut_taps_filter(lut, ba, bb);

    F3 color;
#define UR_CHANNEL(ch, i)                                                                                     \
    {                                                                                                         \
        const float A = kdm * albedo.ch;                     /* (1 - metallic) * albedo: diffuse weight, also irradiance's */ \
        const float F = fmaf(1.0f - F0.ch, p5, F0.ch);                                                        \
        const float direct = fmaf(F, sc - A, A);             /* (1-F) A + F sc */                            \
        const float ambient = fmaf(irradiance.ch, A, prefiltered.ch * fmaf(F0.ch, ba, bb));                   \
        color.ch = fmaf(direct, p.lightRGB[i] * sh_l, ambient);                                               \
    }
    UR_CHANNEL(x, 0)
    UR_CHANNEL(y, 1)
    UR_CHANNEL(z, 2)
#undef UR_CHANNEL
    return color;
}

// SkyAtmosphere.hlsl:58-93 with the camera-height densities, phase constants and sun attenuation folded on the host.
__device__ __forceinline__ F3 sky_pixel(const LightingParams& p, float vx, float vy)
{
    const float* Q = p.skyRot;
    F3 w = f3(fmaf(vy, Q[1], fmaf(vx, Q[0], Q[2])), fmaf(vy, Q[4], fmaf(vx, Q[3], Q[5])), fmaf(vy, Q[7], fmaf(vx, Q[6], Q[8])));
    const float wr = rsq(dot(w, w));
    w = f3(w.x * wr, w.y * wr, w.z * wr);
    const float h = 1.0f - sat(fmaf(w.y, 0.5f, 0.5f));
    const float falloff = sat(h * h * h);
    const float cosSunView = dot(w, f3(p.sunDir[0], p.sunDir[1], p.sunDir[2]));
    const float rayleighPhase = fmaf(cosSunView, cosSunView, 1.0f);
    const float g = 0.76f, g2 = g * g;
    const float mb = fmaf(-2.0f * g, cosSunView, 1.0f + g2);
    const float denom = mb * __builtin_amdgcn_sqrtf(mb); // pow(x, 1.5)
    const float miePhase = rcp(fmaxf(denom, 1e-3f));
    F3 c;
    c.x = fmaf(fmaf(p.skyMie[0], miePhase, p.skyScatterR[0] * rayleighPhase), p.sunAttenuation, mix(0.05f, 0.52f, falloff));
    c.y = fmaf(fmaf(p.skyMie[1], miePhase, p.skyScatterR[1] * rayleighPhase), p.sunAttenuation, mix(0.12f, 0.68f, falloff));
    c.z = fmaf(fmaf(p.skyMie[2], miePhase, p.skyScatterR[2] * rayleighPhase), p.sunAttenuation, mix(0.22f, 0.86f, falloff));
    return c;
}

// A workgroup is 4 waves; a wave covers TW x TH pixels; the four waves sit side by side in x.
template <int MODE, bool SHADOWS, int TW, int WAVES>
__global__ __launch_bounds__(256, WAVES) void lighting_kernel(LightingParams p)
{
    constexpr int TH = 64 / TW;
    __shared__ float srgb[256];
    if (MODE != ur::UR_MODE_SKY) {
        srgb[threadIdx.x] = p.srgb[threadIdx.x];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px = (blockIdx.x * 4u + wave) * TW + (lane % TW);
    const uint32_t r = blockIdx.y * TH + (lane / TW); // row inside the band
    if (px >= p.W || r >= p.rows) return;
    const uint32_t py = p.row0 + r;
    const uint32_t i = r * p.W + px; // pixel index inside the band (< 2^29: byte offsets below stay 32-bit)
    const float ndcx = fmaf((float)px + 0.5f, p.invW2, -1.0f);
    const float ndcy = fmaf((float)py + 0.5f, p.invH2, -1.0f);

    if (MODE != ur::UR_MODE_LIGHTING) {
        const float vx = ndcx * p.skyInvP11, vy = -ndcy * p.skyInvP22;
        const float len = __builtin_amdgcn_sqrtf(fmaf(vx, vx, fmaf(vy, vy, 1.0f)));
        const float skyDepth = p.skyNearOverR * len; // Near / (R * unit_dir.z), unit_dir.z = 1/len
        if (skyDepth >= ld<float>(p.depth, i * 4u)) {
            const F3 sky = sky_pixel(p, vx, vy);
            half4_t o;
            o.x = (_Float16)sky.x; o.y = (_Float16)sky.y; o.z = (_Float16)sky.z; o.w = (_Float16)1.0f;
            st<half4_t>(p.hdr, i * 8u, o);
            return;
        }
        if (MODE == ur::UR_MODE_SKY) return;
    }
    const half4_t ga = ld<half4_t>(p.A, i * 8u), gb = ld<half4_t>(p.B, i * 8u);
    const uint32_t gc = ld<uint32_t>(p.C, i * 4u);
    const half4_t d = ld<half4_t>(p.hdr, i * 8u);
    const F3 col = shade_pixel<SHADOWS>(p, srgb, p.envMipOffset, ndcx * p.invP11, -ndcy * p.invP22, ga, gb, gc, NoPrefetch{});
    half4_t o;
    o.x = (_Float16)((float)d.x + col.x);
    o.y = (_Float16)((float)d.y + col.y);
    o.z = (_Float16)((float)d.z + col.z);
    o.w = (_Float16)((float)d.w + 1.0f);
    st<half4_t>(p.hdr, i * 8u, o);
}

// Persistent form of the same kernel: workgroups loop over 64x4-pixel tiles (stride gridDim.x) and issue the NEXT
// tile's streaming G-buffer loads before shading the current one, so HBM latency hides under ~500 VALU instructions
// instead of relying on occupancy alone; the sRGB table is staged into LDS once per workgroup instead of once per tile.
struct PixelIn {
    half4_t a, b, d;
    uint32_t c;
    float depth;
    uint32_t i, px, py;
    bool valid;
};

template <int MODE, int TW>
__device__ __forceinline__ PixelIn fetch_pixel(const LightingParams& p, uint32_t tile, uint32_t tilesX, uint32_t lane, uint32_t wave)
{
    constexpr int TH = 64 / TW;
    PixelIn q;
    const uint32_t ty = tile / tilesX, tx = tile - ty * tilesX; // uniform: scalar ALU
    q.px = (tx * 4u + wave) * TW + (lane % TW);
    const uint32_t r = ty * TH + (lane / TW);
    q.valid = q.px < p.W && r < p.rows;
    q.py = p.row0 + r;
    q.i = r * p.W + q.px;
    const uint32_t i = q.valid ? q.i : 0u;
    if (MODE != ur::UR_MODE_LIGHTING) q.depth = ld<float>(p.depth, i * 4u);
    q.a = ld<half4_t>(p.A, i * 8u);
    q.b = ld<half4_t>(p.B, i * 8u);
    q.c = ld<uint32_t>(p.C, i * 4u);
    q.d = ld<half4_t>(p.hdr, i * 8u);
    return q;
}

template <int MODE, bool SHADOWS, int TW, int WAVES>
__global__ __launch_bounds__(256, WAVES) void lighting_kernel_persistent(LightingParams p, uint32_t tilesX, uint32_t numTiles)
{
    static_assert(MODE != ur::UR_MODE_SKY, "sky-only uses the simple kernel");
    __shared__ float srgb[256];
    __shared__ uint32_t mipOffset[16];
    srgb[threadIdx.x] = p.srgb[threadIdx.x];
    if (threadIdx.x < 16u) mipOffset[threadIdx.x] = p.envMipOffset[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t tile = blockIdx.x;
    if (tile >= numTiles) return;
    PixelIn cur = fetch_pixel<MODE, TW>(p, tile, tilesX, lane, wave);
    for (;;) {
        const uint32_t next = tile + gridDim.x;
        const bool more = next < numTiles; // uniform
        // The prefetch is issued unconditionally (the last iteration re-reads its own tile): a branch around it would
        // make the compiler's s_waitcnt bookkeeping assume "no younger loads" and wait vmcnt(0) on the gathers.
        const uint32_t pf = more ? next : tile;
        PixelIn nxt;
        const float ndcx = fmaf((float)cur.px + 0.5f, p.invW2, -1.0f);
        const float ndcy = fmaf((float)cur.py + 0.5f, p.invH2, -1.0f);
        bool sky = false;
        F3 out = f3(0.0f, 0.0f, 0.0f);
        if (MODE == ur::UR_MODE_FUSED) {
            const float vx = ndcx * p.skyInvP11, vy = -ndcy * p.skyInvP22;
            const float len = __builtin_amdgcn_sqrtf(fmaf(vx, vx, fmaf(vy, vy, 1.0f)));
            sky = p.skyNearOverR * len >= cur.depth;
            if (sky) out = sky_pixel(p, vx, vy);
        }
        // Shade when any lane of the wave has geometry (wave-uniform branch, so the prefetch inside runs for every lane);
        // sky / out-of-frame lanes compute on whatever they loaded and their result is dropped.
        if (__any(cur.valid && !sky)) {
            const F3 col = shade_pixel<SHADOWS>(p, srgb, mipOffset, ndcx * p.invP11, -ndcy * p.invP22, cur.a, cur.b, cur.c,
                                                [&] { nxt = fetch_pixel<MODE, TW>(p, pf, tilesX, lane, wave); });
            if (!sky) out = f3((float)cur.d.x + col.x, (float)cur.d.y + col.y, (float)cur.d.z + col.z);
        } else {
            nxt = fetch_pixel<MODE, TW>(p, pf, tilesX, lane, wave);
        }
        if (cur.valid) {
            half4_t o;
            o.x = (_Float16)out.x; o.y = (_Float16)out.y; o.z = (_Float16)out.z;
            o.w = sky ? (_Float16)1.0f : (_Float16)((float)cur.d.w + 1.0f);
            st<half4_t>(p.hdr, cur.i * 8u, o);
        }
        if (!more) break;
        cur = nxt;
        tile = next;
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

int env_int(const char* name, int dflt)
{
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

template <int MODE, bool SHADOWS, int TW>
void launch_tile_shape(ur_ctx* ctx, const LightingParams& p)
{
    const uint32_t tilesX = (p.W + 4 * TW - 1) / (4 * TW), tilesY = (p.rows + (64 / TW) - 1) / (64 / TW);
    if constexpr (MODE != ur::UR_MODE_SKY) {
        static const int bpc = env_int("UR_LIGHTING_PERSISTENT", 0); // 0 = one workgroup per tile; N = N persistent workgroups per CU
        const uint32_t numTiles = tilesX * tilesY;
        if (bpc > 0 && numTiles > (uint32_t)(ctx->cu_count * bpc)) {
            if (bpc >= 6) hipLaunchKernelGGL((lighting_kernel_persistent<MODE, SHADOWS, TW, 6>), dim3(ctx->cu_count * bpc), dim3(256), 0, ctx->stream, p, tilesX, numTiles);
            else if (bpc == 5) hipLaunchKernelGGL((lighting_kernel_persistent<MODE, SHADOWS, TW, 5>), dim3(ctx->cu_count * bpc), dim3(256), 0, ctx->stream, p, tilesX, numTiles);
            else hipLaunchKernelGGL((lighting_kernel_persistent<MODE, SHADOWS, TW, 4>), dim3(ctx->cu_count * bpc), dim3(256), 0, ctx->stream, p, tilesX, numTiles);
            return;
        }
    }
    // register budget: waves/SIMD the kernel is compiled for (6 -> 80 VGPRs, 4 -> no cap)
    static const int waves = env_int("UR_LIGHTING_WAVES", 6); // measured T ~ 65 us + 247 us / waves-per-SIMD: 6 waves (80 VGPRs) is the most that does not spill
    static const int ldspad = env_int("UR_LIGHTING_LDSPAD", 0); // diagnostic: unused dynamic LDS to throttle workgroups per CU
    if (waves >= 8) hipLaunchKernelGGL((lighting_kernel<MODE, SHADOWS, TW, 8>), dim3(tilesX, tilesY), dim3(256), ldspad, ctx->stream, p);
    else if (waves == 7) hipLaunchKernelGGL((lighting_kernel<MODE, SHADOWS, TW, 7>), dim3(tilesX, tilesY), dim3(256), ldspad, ctx->stream, p);
    else if (waves == 6) hipLaunchKernelGGL((lighting_kernel<MODE, SHADOWS, TW, 6>), dim3(tilesX, tilesY), dim3(256), ldspad, ctx->stream, p);
    else hipLaunchKernelGGL((lighting_kernel<MODE, SHADOWS, TW, 4>), dim3(tilesX, tilesY), dim3(256), ldspad, ctx->stream, p);
}

template <int MODE, bool SHADOWS>
void launch_tiled(ur_ctx* ctx, const LightingParams& p)
{
    // pixels per wave = TW x (64/TW). 16 x 4: 128-byte G-buffer row segments and compact gather footprints.
    static const int tw = env_int("UR_LIGHTING_TW", 16);
    switch (tw) {
    case 8: launch_tile_shape<MODE, SHADOWS, 8>(ctx, p); break;
    case 32: launch_tile_shape<MODE, SHADOWS, 32>(ctx, p); break;
    case 64: launch_tile_shape<MODE, SHADOWS, 64>(ctx, p); break;
    default: launch_tile_shape<MODE, SHADOWS, 16>(ctx, p); break;
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
    p.invW2 = 2.0f / (float)w; p.invH2 = 2.0f / (float)h;
    p.A = reinterpret_cast<const half4_t*>(A);
    p.B = reinterpret_cast<const half4_t*>(B);
    p.C = C; p.depth = depth;
    p.hdr = reinterpret_cast<half4_t*>(hdr);
    p.srgb = ctx->srgb_table;
    bool shadows = false;
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
        float SM[16];
        mat4_mul(S->ViewInverse, S->LightViewProjection, SM);
        std::memcpy(p.SQ, SM, sizeof(p.SQ));
        std::memcpy(p.ST, SM + 12, sizeof(p.ST));
        for (int j = 0; j < 3; ++j) p.lightRGB[j] = S->LightIntensity * S->LightColor[j];
        p.shadowStrength = S->ShadowStrength;
        p.shadowBias = S->ShadowBias;
        p.shadowW = S->ShadowMapSize[0]; p.shadowH = S->ShadowMapSize[1];
        p.shadowWi = (int32_t)S->ShadowMapSize[0]; p.shadowHi = (int32_t)S->ShadowMapSize[1];
        p.shadowTexelX = 1.0f / S->ShadowMapSize[0]; p.shadowTexelY = 1.0f / S->ShadowMapSize[1];
        p.shadow = T->shadow_map;
        shadows = p.shadowStrength > 0.0f;
        if (shadows && (p.shadowWi < 3 || p.shadowHi < 3)) {
            set_error("shadow maps smaller than 3x3 texels are not supported");
            return UR_EUNSUPPORTED;
        }
        if (shadows && (p.shadow == nullptr || p.shadowWi <= 0 || p.shadowHi <= 0)) {
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
        {
            const float l = std::fmin(std::fmax(p.maxMip, 0.0f), (float)(p.envMips - 1u));
            const uint32_t m0 = (uint32_t)l, m1 = m0 + 1u < p.envMips ? m0 + 1u : p.envMips - 1u;
            p.irrFrac = l - (float)m0;
            p.irrOffset0 = p.envMipOffset[m0]; p.irrOffset1 = p.envMipOffset[m1];
            p.irrN0 = p.envBase >> m0 > 1u ? p.envBase >> m0 : 1u;
            p.irrN1 = p.envBase >> m1 > 1u ? p.envBase >> m1 : 1u;
        }
        p.env = reinterpret_cast<const half4_t*>(T->env_cube);
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
        const float g2 = 0.76f * 0.76f;
        for (int j = 0; j < 3; ++j) {
            p.skyScatterR[j] = rayleighColor[j] * rayleighDensity * (3.0f / (16.0f * 3.14159265f));
            p.skyMie[j] = K->LightColor[j] * mieDensity * 0.8f * ((1.0f - g2) / (4.0f * 3.14159265f));
        }
        const float cosSunUp = p.sunDir[1];
        p.sunAttenuation = std::fmin(std::fmax(std::exp(-std::fmax(0.0f, 1.0f - cosSunUp) * 2.0f), 0.0f), 1.0f);
    }
    if ((uint64_t)w * rows == 0) return UR_OK;
    if ((uint64_t)w * rows >= (1ull << 29)) {
        set_error("band of %u x %u pixels exceeds the 2^29-pixel limit of one launch", w, rows);
        return UR_EUNSUPPORTED;
    }
    switch (mode) {
    case UR_MODE_LIGHTING:
        if (shadows) launch_tiled<UR_MODE_LIGHTING, true>(ctx, p); else launch_tiled<UR_MODE_LIGHTING, false>(ctx, p);
        break;
    case UR_MODE_SKY: launch_tiled<UR_MODE_SKY, false>(ctx, p); break;
    default:
        if (shadows) launch_tiled<UR_MODE_FUSED, true>(ctx, p); else launch_tiled<UR_MODE_FUSED, false>(ctx, p);
        break;
    }
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

} // namespace ur
