// DeferredLighting (GGX + IBL) and SkyAtmosphere for gfx950: per-pixel compute kernels with three modes (lighting, sky, fused).
//
// Reference: Shaders/DeferredLighting.hlsl:35-94 + Shaders/PBRCommon.hlsl:1-48 (fullscreen-triangle pixel shader,
// additive ONE/ONE blend into RGBA16F, Source/Render/DeferredRenderer.cpp:1219-1255,1997-2005) and
// Shaders/SkyAtmosphere.hlsl:40-101 (inside-out sphere, depth GREATER_EQUAL, no blend, DeferredRenderer.cpp:1263-1296).
//
// There is no rasteriser and no texture unit here. One lane shades one pixel; a wave64 covers a 16 x 4 pixel tile so that
// the G-buffer rows are contiguous 128-byte segments and the shadow / cube gathers of neighbouring lanes land on
// neighbouring texels. The PCF, the trilinear cube lookups and the BRDF LUT are filtered in ALU.
//
// Two kernels share that arithmetic:
//   * lighting_stream_kernel (second half of this file, the default): persistent workgroups, G-buffer tiles prefetched into
//     LDS by DMA, side tables in LDS, work claimed from an LDS counter, instruction selection tuned to gfx950's VALU issue
//     rules (DESIGN.md section 3.3 has the measurements behind every choice). A launch may carry one extra workgroup
//     that runs the held-back tail of the HZB chain on a CU of its own (ur_defer_hzb_tail, csrc/hzb_tail.h);
//   * lighting_kernel (first half): one workgroup per 64 x 4 pixels, plain loads; sky-only launches and every
//     configuration the streaming kernel declines (launch_lighting() at the end decides per launch, never per row).
// Every per-launch uniform the HLSL recomputes per pixel is folded on the host: gfx950 has no scalar fp32 ALU. The view
// matrix is rigid (XMMatrixLookToLH, Scene/Camera.cpp:23-31), so world-space vectors are view-space ones rotated by
// ViewInverse. Tolerance against the oracle: max(1e-3, 1 ulp fp16) per channel; fp32 math, one RTE to fp16.

#include "ur_internal.h"
#include "ur_device.h"
#include "hzb_tail.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <type_traits>

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// The launch-uniform values of the streaming kernel's loop. The hot ones stay in SGPRs across the loop; cold paths (sky
// constants, shadow slow path, partial tiles) re-read theirs from the kernarg segment when they run (fresh_params()).
struct StreamHot {
    uint32_t groups; // lighting workgroups of the launch (a workgroup with this index, if any, runs the deferred HZB tail)
    uint32_t tilesX, numTiles, tilesXMagic, W, rows, row0, irrN0, irrRowBytes; // tilesXMagic: tile / tilesX = (tile * magic) >> 32
    uint32_t staticClaims; // a workgroup's claims c < staticClaims are dealt statically (above); from there on they index the chunks the
                           // workgroup claims at run time (Balance). 0xFFFFFFFF: every tile is dealt statically
    float invW2, invH2, invP11, nInvP22;      // ray: ra = ndc.x * invP11, rb = ndc.y * nInvP22 (= -1/P22)
    float skyInvP11, nSkyInvP22, skyNearOverR2, maxMip;
    float envMaxLevel, irrNf, irrEf, irrEEf, irrOfff; // irradiance mip: N, N+2, (N+2)^2, texel offset — as floats (exact)
    float shadowWm3, shadowHm3, shadowWf;     // W-3, H-3, W as floats
    float shadowXmax, shadowYmax, shadowStrength, shadowNegQuarterStrength; // W - 0.5, H - 0.5, s, -s/4
    uint32_t shadowRowBytes;
    int32_t shadowWi, shadowHi;
    const void* env;
    const float* shadow;
    void* hdr;
    float skyDepthMax;   // no sphere depth of the frame exceeds it
    // the cube's small mips in LDS: a pixel whose prefiltered level is >= cubeLdsLevel takes both footprints from the workgroup's
    // copy of the RGB row-pair entries of mips [cubeLdsLevel, last] (byte address = global byte offset - cubeLdsAdj)
    float cubeLdsLevel;  // (16.0: nothing is in LDS)
    uint32_t cubeLdsBase, cubeLdsBytes; // where those mips' entries start in the staged buffer (bytes), and how many bytes they are
    // read once per wave into VGPRs
    float R[9];          // (float3x3)ViewInverse, row-major
    float Lw[3];         // light direction, world space
    float WA[3], WB[3], WC[3]; // world-space camera ray through the pixel = ndc.x * WA + ndc.y * WB + WC
    float lightRGB[3];
    float shA[3], shB[3], shC[3], shT[3]; // (su * W - 0.5, sv * H - 0.5, depth - bias)[k] = viewZ * (ndc.x * shA[k] + ndc.y * shB[k] + shC[k]) + shT[k]
};

// Inter-workgroup balancing of a streaming launch (UR_OPT_LIGHTING_BALANCE). Equal static shares leave the mean wave idle for the
// last ~5 us of a 4K launch: XCDs differ by up to 8 % in speed on the same work (profiles/r03_wave_exit_stamps.txt). So only the tiles
// [0, staticTiles) are dealt statically; the rest is a pool of chunks of 2^dynShift consecutive tiles that workgroups claim at run
// time, one returning device-scope atomic per chunk, `lookahead` chunks ahead of use (the first `lookahead` of a workgroup are
// pre-assigned). The pool is cut into kClaimWords sub-pools, word q serving workgroups 8q .. 8q+7 - one per XCD under round-robin
// placement, which is what evens out the XCDs; any placement is correct. Per workgroup the claims are strictly sequential (the chunk
// of slot k is claimed only after slot k - 1 has been published in LDS), so its slots are valid up to the first failed claim and
// END from there on: exactly one failed claim per workgroup, after which it adds one to words[kClaimWords * stride]; the
// workgroup whose add comes last puts every word back to zero for the next launch (also under hipGraph replay).
struct Balance {
    uint32_t poolChunks;   // 0: off
    uint32_t staticTiles, dynShift, lookahead;
    unsigned long long poolMagic; // first chunk of the share of workgroups [0, x) = (x * poolMagic) >> 32 (= x * poolChunks / groups, rounded up)
    uint32_t* words;
    uint32_t* timedOut;    // host-visible (mapped, coherent): a wave gave up waiting for a slot of its workgroup (ur_ctx::claim_timed_out)
};

struct LightingParams {
    // frame
    uint32_t W, H, row0, rows;
    float invW2, invH2;  // 2/W, 2/H
    // lighting
    float invP11, invP22;
    float L[3];          // normalize(mul(float4(LightDirection,0), View).xyz)
    float R[9];          // (float3x3)ViewInverse, row-major
    float SQ[12];        // rows 0..2 of (ViewInverse * LightViewProjection), columns x,y,z,w : applied to the camera ray (a,b,1)
    float VIt[3], camPos[3]; // row 3 of ViewInverse, CameraPosition: the general path below (general != 0)
    uint32_t general;    // ViewInverse is not a rigid transform, or CameraPosition is not its origin: world vectors are formed literally
    uint32_t shadowSmall; // a shadow map below 3x3 texels: every pixel takes the bordered PCF
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
    StreamHot hot;       // streaming kernel: everything one loop iteration reads
    Balance bal;         // ... and what its run-time tile claims read (cold)
    unsigned long long* timeline; // debug: {first entry, last exit} of this launch (ur_debug_timeline), else null
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
// (The pointers are global memory by contract; saying so keeps pointers that were themselves loaded from memory off the
// flat_load path.)
#define UR_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T ld(const void* base, uint32_t byte_offset)
{
    return *reinterpret_cast<const UR_GLOBAL T*>((const UR_GLOBAL char*)base + byte_offset);
}
template <class T>
__device__ __forceinline__ void st(void* base, uint32_t byte_offset, T v)
{
    *reinterpret_cast<UR_GLOBAL T*>((UR_GLOBAL char*)base + byte_offset) = v;
}

// HDR out of the per-tile kernel: written once, write-through + nontemporal like the streaming kernel's store (store_hdr below)
__device__ __forceinline__ void st_hdr_once(void* base, uint32_t byte_offset, half4_t v)
{
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t u;
    __builtin_memcpy(&u, &v, 8);
    asm volatile("global_store_dwordx2 %0, %1, %2 sc1 nt" ::"v"(byte_offset), "v"(u), "s"(base) : "memory");
}

struct uint4u { uint32_t x, y, z, w; };  // 16 bytes loaded from an 8-byte-aligned address
struct float3u { float x, y, z; };       // 12 bytes loaded from a 4-byte-aligned address
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x3_t __attribute__((ext_vector_type(3)));
typedef u32x4_t u32x4_a8 __attribute__((aligned(8)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef u32x3_t u32x3_a4 __attribute__((aligned(4)));
typedef f32x3_t f32x3_a4 __attribute__((aligned(4)));
template <>
__device__ __forceinline__ uint4u ld<uint4u>(const void* base, uint32_t byte_offset)
{
    const u32x4_t v = *reinterpret_cast<const UR_GLOBAL u32x4_a8*>((const UR_GLOBAL char*)base + byte_offset);
    return {v.x, v.y, v.z, v.w};
}
template <>
__device__ __forceinline__ float3u ld<float3u>(const void* base, uint32_t byte_offset)
{
    const f32x3_t v = *reinterpret_cast<const UR_GLOBAL f32x3_a4*>((const UR_GLOBAL char*)base + byte_offset);
    return {v.x, v.y, v.z};
}

// The x, y halves of a vector as a pair. Packed fp32 arithmetic on such pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 through
// the vector type: hipcc pairs the registers and broadcasts scalar operands through op_sel without a move) was built for every
// per-channel chain of the loop in round 3 and measured: 100 plain instructions became 46 packed ones, no scratch, and the
// launch took the same time (72.0 against 72.0-72.3 us for the five groups added one at a time, profiles/r03_lighting_diet.txt):
// in this loop a packed instruction costs the issue port what its two halves would. The pairs below are plain containers.
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t f2(float x, float y) { return f2_t{x, y}; }

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

// the same, inlined (the streaming kernel keeps no call in its loop: a call pins live values to callee-saved registers)
__device__ __forceinline__ float shadow_pcf_border_inline(const float* __restrict__ map, int W, int H, int ia, int ja, float fx, float fy, float cmp)
{
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        const int yj = ja + r;
        const uint32_t rowo = (uint32_t)min(max(yj, 0), H - 1) * (uint32_t)W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int xi = ia + c;
            const bool in = (uint32_t)xi < (uint32_t)W && (uint32_t)yj < (uint32_t)H;
            const float t = ld<float>(map, (rowo + (uint32_t)min(max(xi, 0), W - 1)) * 4u);
            s += (in ? cmp <= t : true) ? (c == 0 ? 1.0f - fx : (c == 1 ? 1.0f : fx)) : 0.0f;
        }
        acc = fmaf(r == 0 ? 1.0f - fy : (r == 1 ? 1.0f : fy), s, acc);
    }
    return 0.25f * acc;
}

// DeferredLighting.hlsl:35-94 for one pixel. (a,b) = camera ray (ndc.x/P11, -ndc.y/P22); viewPos = viewZ * (a, b, 1).
template <bool SHADOWS>
__device__ __forceinline__ F3 shade_pixel(const LightingParams& p, const float* srgb, const uint32_t* mipOffset, float ra, float rb, half4_t ga, half4_t gb,
                                          uint32_t gc)
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
    // With a rigid view matrix whose origin is CameraPosition (every camera the reference builds) that rotation keeps lengths and
    // angles, worldView is the rotated V and dot(worldNormal, worldView) = N.V. Otherwise (uniform, p.general) the vectors are formed
    // as the shader writes them: worldPos = viewPos * ViewInverse, worldView = normalize(CameraPosition - worldPos),
    // worldNormal = normalize(normal * (float3x3)ViewInverse) (DeferredLighting.hlsl:55,76-78,84).
    F3 wR = f3(0.0f, 0.0f, 0.0f), wN = f3(0.0f, 0.0f, 0.0f);
    float NdotVibl = NdotV;
    if (p.general == 0u) {
        const float t2 = 2.0f * NdotVraw;
        wR = rot(f3(fmaf(t2, N.x, -V.x), fmaf(t2, N.y, -V.y), fmaf(t2, N.z, -V.z)), p.R);
        wN = rot(N, p.R);
    } else {
        const F3 wp = rot(f3(ra * viewZ, rb * viewZ, viewZ), p.R);
        F3 wv = f3(p.camPos[0] - (wp.x + p.VIt[0]), p.camPos[1] - (wp.y + p.VIt[1]), p.camPos[2] - (wp.z + p.VIt[2]));
        const float wvr = rsq(dot(wv, wv));
        wv = f3(wv.x * wvr, wv.y * wvr, wv.z * wvr);
        wN = rot(N, p.R);
        const float wnr = rsq(dot(wN, wN));
        wN = f3(wN.x * wnr, wN.y * wnr, wN.z * wnr);
        const float nv = dot(wN, wv);
        wR = f3(fmaf(2.0f * nv, wN.x, -wv.x), fmaf(2.0f * nv, wN.y, -wv.y), fmaf(2.0f * nv, wN.z, -wv.z)); // reflect(-worldView, worldNormal)
        NdotVibl = sat(nv);
    }
    const CubeUV cr = cube_face(wR);
    const CubeUV cn = cube_face(wN);
    const float lvl = fminf(fmaxf(roughness * p.maxMip, 0.0f), (float)(p.envMips - 1u));
    const uint32_t m0 = (uint32_t)lvl, m1 = min(m0 + 1u, p.envMips - 1u);
    const float fl = lvl - (float)m0; // m1 == m0 only when fl == 0: the second mip then carries weight 0
    const CubeTaps pre0 = cube_taps_load(p.env, mipOffset[m0], max(1u, p.envBase >> m0), cr);
    const CubeTaps pre1 = cube_taps_load(p.env, mipOffset[m1], max(1u, p.envBase >> m1), cr);
    const CubeTaps irr0 = cube_taps_load(p.env, p.irrOffset0, p.irrN0, cn);
    const LutTaps lut = lut_taps_load(p, NdotVibl, roughness);
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
        if (p.shadowSmall == 0u) sh = shadow_taps_load(p, su, sv); // (uniform; the 3x3 block needs a map of at least 3x3 texels)
        else sh = ShadowTaps{};
    }
    const F3 albedo = f3(srgb[gc & 0xFFu], srgb[(gc >> 8) & 0xFFu], srgb[(gc >> 16) & 0xFFu]);

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
        const bool fast = p.shadowSmall == 0u && sh.ia >= 0 && sh.ja >= 0 && sh.ia + 2 < p.shadowWi && sh.ja + 2 < p.shadowHi;
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
// P: pointer to the parameters (generic for the per-tile kernel; a re-read kernarg pointer in the streaming kernel)
template <class P>
__device__ __forceinline__ F3 sky_pixel(P p, float vx, float vy)
{
    const auto* Q = p->skyRot;
    F3 w = f3(fmaf(vy, Q[1], fmaf(vx, Q[0], Q[2])), fmaf(vy, Q[4], fmaf(vx, Q[3], Q[5])), fmaf(vy, Q[7], fmaf(vx, Q[6], Q[8])));
    const float wr = rsq(dot(w, w));
    w = f3(w.x * wr, w.y * wr, w.z * wr);
    const float h = 1.0f - sat(fmaf(w.y, 0.5f, 0.5f));
    const float falloff = sat(h * h * h);
    const float cosSunView = dot(w, f3(p->sunDir[0], p->sunDir[1], p->sunDir[2]));
    const float rayleighPhase = fmaf(cosSunView, cosSunView, 1.0f);
    const float g = 0.76f, g2 = g * g;
    const float mb = fmaf(-2.0f * g, cosSunView, 1.0f + g2);
    const float denom = mb * __builtin_amdgcn_sqrtf(mb); // pow(x, 1.5)
    const float miePhase = rcp(fmaxf(denom, 1e-3f));
    F3 c;
    c.x = fmaf(fmaf(p->skyMie[0], miePhase, p->skyScatterR[0] * rayleighPhase), p->sunAttenuation, mix(0.05f, 0.52f, falloff));
    c.y = fmaf(fmaf(p->skyMie[1], miePhase, p->skyScatterR[1] * rayleighPhase), p->sunAttenuation, mix(0.12f, 0.68f, falloff));
    c.z = fmaf(fmaf(p->skyMie[2], miePhase, p->skyScatterR[2] * rayleighPhase), p->sunAttenuation, mix(0.22f, 0.86f, falloff));
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
    // ndc.x in the streaming kernel's two-step form (16-pixel tile origin, then the column inside the tile): the same bits
    // in both kernels, so a fused streaming launch equals Lighting followed by this kernel's Sky launch bit for bit
    const float ndcx = fmaf((float)(px & ~15u), p.invW2, fmaf((float)(px & 15u), p.invW2, 0.5f * p.invW2 - 1.0f));
    const float ndcy = fmaf((float)py + 0.5f, p.invH2, -1.0f);

    if (MODE != ur::UR_MODE_LIGHTING) {
        const float vx = ndcx * p.skyInvP11, vy = -ndcy * p.skyInvP22;
        const float len = __builtin_amdgcn_sqrtf(fmaf(vx, vx, fmaf(vy, vy, 1.0f)));
        const float skyDepth = p.skyNearOverR * len; // Near / (R * unit_dir.z), unit_dir.z = 1/len
        if (skyDepth >= ld<float>(p.depth, i * 4u)) {
            F3 sky = sky_pixel(&p, vx, vy);
            // the colour is an fp32 value rounded to fp16 in a second step, as in the oracle and in the streaming kernel: kept
            // apart from the conversion, or hipcc fuses the last FMA with it (v_fma_mixlo_f16: ONE rounding, a different bit in
            // about one sky pixel in seven thousand)
            asm volatile("" : "+v"(sky.x), "+v"(sky.y), "+v"(sky.z));
            half4_t o;
            o.x = (_Float16)sky.x; o.y = (_Float16)sky.y; o.z = (_Float16)sky.z; o.w = (_Float16)1.0f;
            st_hdr_once(p.hdr, i * 8u, o);
            return;
        }
        if (MODE == ur::UR_MODE_SKY) return;
    }
    const half4_t ga = ld<half4_t>(p.A, i * 8u), gb = ld<half4_t>(p.B, i * 8u);
    const uint32_t gc = ld<uint32_t>(p.C, i * 4u);
    const half4_t d = ld<half4_t>(p.hdr, i * 8u);
    const F3 col = shade_pixel<SHADOWS>(p, srgb, p.envMipOffset, ndcx * p.invP11, -ndcy * p.invP22, ga, gb, gc);
    // blend in fp32, then ONE conversion to fp16 (not a fused mixed-precision add: see the sky branch above)
    float bx = (float)d.x + col.x, by = (float)d.y + col.y, bz = (float)d.z + col.z, bw = (float)d.w + 1.0f;
    asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz), "+v"(bw));
    half4_t o;
    o.x = (_Float16)bx;
    o.y = (_Float16)by;
    o.z = (_Float16)bz;
    o.w = (_Float16)bw;
    st_hdr_once(p.hdr, i * 8u, o);
}

// =====================================================================================================================
// Streaming form (the default for full-tile regions): persistent waves, each looping over 16x4-pixel tiles.
//
//  * G-buffer tiles arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR destination) into a wave-private double buffer,
//    two tiles ahead: the HBM latency of the streaming reads is never exposed and costs no registers. Two DMA
//    instructions move a whole tile (A|B and HDR|C|depth: the per-lane SOURCE address selects the buffer, the LDS image
//    is lane-linear), so pixel p of the tile sits at a fixed LDS offset per buffer.
//  * Ordering protocol (the DMA is invisible to hipcc's s_waitcnt bookkeeping): in iteration t the DMA for tile t+2 is
//    issued into the buffer tile t was read from, at a point where hipcc has no pending load of its own (every gather
//    result has been touched by an asm statement, so its waits sit in front of that point). The wait for iteration
//    t+1's gathers is a vmcnt(0) by construction (hipcc sees nothing younger), which also retires the DMA for tile t+2
//    one full iteration after its issue — before iteration t+2 reads it. No counted waits, no barriers in the loop.
//  * The BRDF LUT lives in LDS as bordered fp32 pairs (clamp addressing = border texels; one med3 instead of eight
//    min/max, no unpack/convert per tap), the launch-uniform irradiance mip as per-cell fp32 polynomials, the sRGB table as before.
//  * The shadow transform of the (orthographic) light is folded on the host into three affine forms of the camera ray.
// =====================================================================================================================
constexpr uint32_t kLutW = 128, kLutH = 32, kLutE = kLutW + 2;    // streaming kernel: LUT dimensions are compile-time
constexpr uint32_t kChunkShift = 2;                                 // static deal: chunks of 4 consecutive tiles (4K, round 1: chunks of 16 / 4 / 1 tiles -> 75.4 / 74.6 / 79.1 us)
constexpr uint32_t kLdsSrgb = 0;                                    // 256 floats
constexpr uint32_t kLdsIrrBytes = 6 * 9 * 64;                       // 54 cells x 64 B
constexpr uint32_t kLdsWork = 1024 + 17 * 32 + kLdsIrrBytes;        // [0] the workgroup's tile counter, [1] waves that left the loop, [2] HZB walkers done,
                                                                    // [4] first run-time chunk of the workgroup's claim word, [5] how many follow (32 bytes reserved)
constexpr uint32_t kDynSlots = 256;                                 // run-time chunk slots of a workgroup: tile index of the chunk's first tile, or:
constexpr uint32_t kDynNotReady = 0xFFFFFFFFu, kDynEnd = 0xFFFF0000u; // not published yet / nothing left to claim (any value >= kDynEnd)
constexpr uint32_t kLdsDyn = kLdsWork + 32;
constexpr uint32_t kLdsMip = 1024;                                  // 17 x 16 B: per-mip cube constants (MipEntry)
constexpr uint32_t kLdsIrr = 1024 + 17 * 32;                        // irradiance mip (N <= 2) as per-cell polynomials: up to 6 * 3 * 3 cells of 64 B
constexpr uint32_t kLdsHzb = kLdsDyn + kDynSlots * 4;               // 80 floats per wave: mip-2 / mip-3 scratch of the waves that walk HZB pieces
constexpr uint32_t kLdsLut = kLdsHzb + 16 * 80 * 4;                 // (kLutW + 2) x (kLutH + 2) float2
constexpr uint32_t kLdsTiles = kLdsLut + kLutE * (kLutH + 2) * 8;   // per wave: 2 x 2 KB
constexpr uint32_t kTileBytes = 2048;                               // A 512 | B 512 | HDR 512 | C 256 | depth 256
constexpr uint32_t kLdsCubeBytes = 32768;                           // behind the waves' tile buffers: RGB row-pair entries of the cube's small mips
                                                                    // (the shipped 256^2 cube: mips 4..8 = 31 968 bytes)
static_assert(kLdsTiles % 16 == 0, "tile buffers are 16-byte aligned");

__device__ __forceinline__ uint32_t lds_address(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// One LDS-DMA wave-instruction: lane l's 16 bytes at gsrc land at lds_dst + 16 l. M0 is saved and restored (it is
// compiler-reserved and not preserved around asm statements).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst_uniform)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}

struct TileSrc { const char* p1; const char* p2; uint32_t mul2; }; // per-lane source bases of the two DMA instructions

// Kernel parameters re-read from the kernarg segment (scalar loads) behind an opaque pointer: hipcc would otherwise hoist
// every parameter out of the persistent loop and spill the ~150 live scalars to VGPR lanes.
typedef const __attribute__((address_space(4))) LightingParams* KParams;
__device__ __forceinline__ KParams fresh_params()
{
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return (KParams)k; // LightingParams is the kernel's only argument: offset 0
}

// maxRow = last valid row of the tile (3 for a whole tile): the rows of a partial bottom tile re-read the last valid one
template <int MODE, class P>
__device__ __forceinline__ TileSrc tile_src(P p, uint32_t lane, uint32_t maxRow)
{
    TileSrc s;
    const uint32_t W = p->W;
    const uint32_t q = lane & 31u, r8 = min(q >> 3, maxRow), c8 = (q & 7u) * 2u;     // 8-byte-per-pixel buffers: 2 pixels per lane
    const uint32_t q4 = lane & 15u, r4 = min(q4 >> 2, maxRow), c4 = (q4 & 3u) * 4u;  // 4-byte-per-pixel buffers: 4 pixels per lane
    const uint64_t o8 = (uint64_t)(r8 * W + c8) * 8u, o4 = (uint64_t)(r4 * W + c4) * 4u;
    const char* A = reinterpret_cast<const char*>(p->A);
    const char* B = reinterpret_cast<const char*>(p->B);
    const char* C = reinterpret_cast<const char*>(p->C);
    const char* D = reinterpret_cast<const char*>(p->depth);
    const char* H = reinterpret_cast<const char*>(p->hdr);
    s.p1 = (lane < 32u ? A : B) + o8;
    if (lane < 32u) { s.p2 = H + o8; s.mul2 = 8u; }
    else if (lane < 48u || MODE == ur::UR_MODE_LIGHTING) { s.p2 = C + o4; s.mul2 = 4u; }
    else { s.p2 = D + o4; s.mul2 = 4u; }
    return s;
}

// Both DMA instructions of a tile in one statement, one M0 set-up: the instruction offset (1024) moves the global AND the
// LDS address of the second one, so its source pointer is biased by -1024.
__device__ __forceinline__ void tile_dma_at(const char* g1, const char* g2 /* biased by -1024 */, uint32_t lds_dst)
{
    uint32_t keep;
    // The A|B instruction carries the nontemporal hint: those 16 bytes per pixel are read exactly once (4K: 73.9 -> 72.5 us). The
    // other one does not: it carries the depth rows, which a riding Build HZB reads a second time in the same launch (with the
    // hint on it that second read comes from memory again: frame 84.4-85.8 instead of 80.5-81.1 us).
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g1), "v"(g2), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void tile_dma(const TileSrc& s, uint32_t origin /*pixel index of the tile's first pixel*/, uint32_t lds_dst)
{
    const char* g1 = s.p1 + (uint64_t)origin * 8u;
    const char* g2 = s.p2 + (uint64_t)origin * s.mul2 - 1024;
    tile_dma_at(g1, g2, lds_dst);
}

// tile (tx, ty) -> LDS; `full` = the precomputed per-lane sources of a whole tile
template <int MODE, class P>
__device__ __forceinline__ void tile_prefetch(P p, uint32_t W, uint32_t rows, const TileSrc& full, uint32_t lane, uint32_t tx, uint32_t ty, uint32_t lds_dst)
{
    const uint32_t origin = (ty * 4u) * W + tx * 16u, rowsLeft = rows - ty * 4u; // uniform
    if (rowsLeft >= 4u) tile_dma(full, origin, lds_dst);
    else {
        // cold path (the one partial tile row of a band): the lane index goes through an opaque move so that nothing derived
        // from it is hoisted out of the persistent loop and kept live (it would be the loop's one spilled value)
        uint32_t l = lane;
        asm volatile("" : "+v"(l));
        tile_dma(tile_src<MODE>(p, l, rowsLeft - 1u), origin, lds_dst);
    }
}

struct __attribute__((aligned(16))) float4a { float x, y, z, w; };

// A launch-uniform value pinned in a VGPR (a VOP3 instruction reads at most one SGPR: the second uniform operand of an FMA
// would otherwise cost a v_mov per use and per iteration).
__device__ __forceinline__ float vreg(float uniform)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
    return v;
}

// fp16 halves of a packed dword -> fp32 behind an opaque conversion: hipcc would otherwise fold the conversions into
// v_fma_mix_f32 / SDWA forms, which hold the VALU issue port for 4-6 cycles on gfx950 (tools/microbench/valu_rate.hip, table 4)
// where a plain v_cvt on the side pipe overlaps with the FMAs around it.
__device__ __forceinline__ float h2f_lo(uint32_t w)
{
    float r;
    asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float h2f_hi(uint32_t w) { return h2f_lo(w >> 16); }

// g = (cmp > t) as clamp(cmp * 2^126 - t * 2^126): one clamped FMA per tap (cmpBig = cmp * 2^126 is exact), exact for every
// finite pair; equality gives 0 (LESS_EQUAL passes).
__device__ __forceinline__ float gt_step(float cmpBig, float t, float negBig /* -2^126 in a VGPR */)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(t), "v"(negBig), "v"(cmpBig));
    return r;
}

// Wave-uniform conditions as SGPR integers. A `bool` that is defined in one basic block and tested or negated in another is a
// lane mask to hipcc, and every such use goes through a VGPR: `v_cndmask_b32 v, 0, 1, mask` + `v_cmp_ne_u32 mask', 1, v` to
// negate a mask that `s_not_b64` would negate — two vector instructions of the 4-cycle class that cannot share an issue slot
// with a neighbour (tools/microbench/valu_rate.hip, tables 3 and 4). The loop had four such pairs per iteration; with the ballot taken where
// the comparison is made and a 32-bit scalar crossing the blocks there are none (profiles/r03_lighting_diet.txt: -1.2 us).
__device__ __forceinline__ uint32_t flag_any(bool pred)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(pred);
    uint32_t r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return r;
}
__device__ __forceinline__ uint32_t flag_all(bool pred) // (every lane of the wave is active where this is used)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(pred);
    uint32_t r;
    asm("s_cmp_eq_u64 %1, -1\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return r;
}

// Per-mip constants of the cube's RGB ROW-PAIR section (ur_stage_env_cube: 12-byte entries): N, and - in BYTES from the start of the
// buffer, as floats (exact below 2^24) - a pair-row (12 E, E = N + 2), a face (12 E (E - 1)) and the mip's first entry.
struct __attribute__((aligned(16))) MipEntry { float Nf, rowBf, faceBf, offBf; };

// input-only "these registers are needed here": hipcc puts the s_waitcnt of pending loads in front of the statement
template <class V>
__device__ __forceinline__ void need(const V& a, const V& b, const V& c, const V& d)
{
    asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d));
}

// ---- run-time tile claims (struct Balance) -----------------------------------------------------------------------------------------
// One word of the workgroup's slot table, read by every lane (same address: a broadcast), as a scalar.
__device__ __forceinline__ uint32_t dyn_slot_load(const uint32_t* slot)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// ... waiting until it is published. The publisher is another wave of this workgroup that waits for nothing but memory, so the
// wait ends; it is bounded all the same (~0.25 s of s_sleep), and giving up is reported (ur_ctx::claim_timed_out -> UR_ETIMEOUT).
template <class P>
__device__ __forceinline__ uint32_t dyn_slot_wait(const uint32_t* slot, P kp)
{
    uint32_t e = dyn_slot_load(slot), spins = 0;
    while (e == kDynNotReady) {
        __builtin_amdgcn_s_sleep(8);
        e = dyn_slot_load(slot);
        if (++spins > (1u << 21)) {
            __hip_atomic_store(kp->bal.timedOut, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            e = kDynEnd;
            break;
        }
    }
    return e;
}
// Claim c >= staticClaims of the workgroup -> tile index (0xFFFFFFFF: nothing left). The wave that draws the first claim of slot s
// also claims the chunk of slot s + lookahead from the workgroup's word and publishes it. It blocks for that one atomic (~1 us under
// the launch's stream; its next tile is already in LDS), which happens once per 2^dynShift tiles of the WORKGROUP and only in the
// last part of the launch. Called at the prefetch point: the wave has no vector-memory operation in flight.
template <class P>
__device__ __forceinline__ uint32_t dyn_claim(P kp, uint32_t* work, uint32_t* dynT, uint32_t c, uint32_t lane)
{
    const uint32_t d = c - kp->hot.staticClaims;
    const uint32_t s = min(d >> kp->bal.dynShift, kDynSlots - 1u);
    const uint32_t e = dyn_slot_wait(dynT + s, kp);
    if (e >= kDynEnd) return 0xFFFFFFFFu;
    const uint32_t idx = d & ((1u << kp->bal.dynShift) - 1u);
    if (idx == 0u) { // uniform
        const uint32_t k = s + kp->bal.lookahead;
        if (k < kDynSlots) {
            uint32_t out = kDynEnd;
            // strictly one claim after the other: the slot in front must have been published (and still have had a chunk)
            if (dyn_slot_wait(dynT + (k - 1u), kp) < kDynEnd && k + 1u < kDynSlots) {
                uint32_t v = 0;
                if (lane == 0) v = __hip_atomic_fetch_add(kp->bal.words + (blockIdx.x >> 3) * ur::kClaimWordStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = __builtin_amdgcn_readfirstlane(v);
                if (v < dyn_slot_load(work + 5)) out = kp->bal.staticTiles + ((dyn_slot_load(work + 4) + v) << kp->bal.dynShift);
                else {
                    // this workgroup's last claim (every claim in front of it has returned): count it; the workgroup that
                    // counts last puts the words back to zero, write-through, for the next launch
                    uint32_t done = 0;
                    if (lane == 0) done = __hip_atomic_fetch_add(kp->bal.words + ur::kClaimWords * ur::kClaimWordStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_readfirstlane(done) == kp->hot.groups - 1u && lane <= ur::kClaimWords)
                        __hip_atomic_store(kp->bal.words + lane * ur::kClaimWordStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (lane == 0) __hip_atomic_store(dynT + k, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    return e + idx;
}

// WPB waves per workgroup, one workgroup per CU (two of 10 waves at 96 VGPRs were tried: the loop spills).
// HDR stores are write-through AND nontemporal (sc1 nt). Write-through alone (round 2): the launch takes the same time as with plain
// or nontemporal stores, but nothing of it is left dirty in L2 for the end of the launch to write back, and the NEXT launch of the frame
// starts 0.6-0.9 us earlier. Both hints together (round 4; either alone changes nothing): the 66 MB written once neither stay dirty nor
// take cache lines from the gathers' working set - 4K alone 68.8 -> 66.6 us, the dispatch inside the frame 72.3 -> 70.8, frame 76.7 -> 75.4
// (profiles/r04_priority.txt; the plain streaming kernel gains the same way from the hints: 64.8 -> 60.8 us).
__device__ __forceinline__ void store_hdr(void* base, uint32_t byte_offset, uint32_t lo, uint32_t hi)
{
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {lo, hi};
    asm volatile("global_store_dwordx2 %0, %1, %2 sc1 nt" ::"v"(byte_offset), "v"(v), "s"(base) : "memory");
}

// The wide launch of a held-back HZB chain, taken along by the lighting workgroups (ur_defer_hzb_tail(ctx, 2)): workgroup g's
// last wave walks the 128x32 source pieces g, g + groups, ... before it joins the tile loop (the other waves take up its
// share of tiles through the LDS work counter), then signals `done`; the riding tail workgroup waits for `groups` arrivals.
struct HzbRide {
    ur::HzbDispatch d;
    uint32_t grid_x, pieces; // pieces == 0: nothing rides
    uint32_t walkers;        // how many waves of a workgroup walk pieces: 1 (the last one) when the band is long - the others shade
                             // meanwhile -, 16 when it is short - the chain must not outlast the shading (informative: the kernel
                             // instantiation, RIDE_ALL, carries the choice)
    uint32_t want;           // arrivals the riding tail waits for: the launch's lighting workgroups (one more under UR_OPT_DEBUG_HZB_RIDE_STALL)
    uint32_t spin_limit;     // polls (s_sleep 32 each) before it gives up: 2^22 ~ 4 s
    uint32_t pad;
    uint32_t* done;          // [0] arrivals of this launch's lighting workgroups (reset by the tail workgroup once it has seen them all);
                             // [1] sticky: a tail gave up, the host has not reset the words yet (check_hzb_timeout)
    uint32_t* timed_out;     // host-visible (mapped, coherent) flag: the tail gave up waiting (ur_ctx::hzb_timed_out)
};

template <int MODE, bool SHADOWS, bool IRR_LDS, int WPB, bool RIDE_ALL>
__global__ __launch_bounds__(64 * WPB, 1) void lighting_stream_kernel(LightingParams p, ur::HzbTail hzbTail, HzbRide hzbRide)
{
    static_assert(MODE != ur::UR_MODE_SKY, "sky-only uses the per-tile kernel");
    ur::warm_kernarg<sizeof(LightingParams)>(); // (ur_device.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ur::timeline_entry(p.timeline);
    if (blockIdx.x >= p.hot.groups) { // uniform: the one extra workgroup of a launch that carries a deferred HZB tail (ur_defer_hzb_tail)
        if constexpr (WPB == 16) {
            static_assert(kLdsTiles + 16u * 2u * kTileBytes >= (ur::kTailTexels + ur::kTailTexels / 2u) * sizeof(float), "the tail's two level buffers fit the launch's LDS");
            float* bufA = reinterpret_cast<float*>(smem);
            // read through the kernarg segment behind an opaque pointer: nothing of the 264-byte argument is live outside this branch
            auto ka = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            typedef const __attribute__((address_space(4))) ur::HzbTail* KTail;
            typedef const __attribute__((address_space(4))) HzbRide* KRide;
            static_assert(sizeof(LightingParams) % 8 == 0 && sizeof(ur::HzbTail) % 8 == 0, "the kernel arguments follow one another without padding");
            const KRide ride = (KRide)((const __attribute__((address_space(4))) char*)ka + sizeof(LightingParams) + sizeof(ur::HzbTail));
            if (ride->pieces != 0u) {
                // The tail's parent level is written by the lighting workgroups of THIS launch: wait for all of them (a relaxed
                // agent-scope poll by one lane, then acquire, then the barrier: MI355X_MICROARCH.md, inter-workgroup visibility,
                // valid consumer form). The wait cannot deadlock as launched: this is the launch's HIGHEST-indexed workgroup, the
                // grid is one workgroup per CU, workgroups are dispatched in index order, so every producer is resident before
                // this one starts. It is bounded all the same (~4 s of s_sleep): a lost arrival must not hang the chip. Giving up
                // is NOT silent: the flag in host-visible memory makes the next ur_flush / ur_build_hzb / ur_cull_indirect_args* /
                // ur_frame_render on the context return UR_ETIMEOUT (the levels below are then built from a stale parent), and
                // the arrival counter is left as it is — stragglers may still add to it; the host resets it when it reports.
                if (threadIdx.x == 0) {
                    uint32_t* done = ride->done;
                    const uint32_t want = ride->want, limit = ride->spin_limit;
                    // done[1] != 0: an earlier launch on this context gave up and the host has not put the counter back yet (it does at
                    // its next entry point, stream-ordered): whatever this launch reads in done[0] may be that launch's leftover, so it
                    // reports too and leaves the words alone - no frame between a time-out and the host's reset goes unreported
                    const bool stale = __hip_atomic_load(done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                    uint32_t spins = 0;
                    bool gave_up = false;
                    while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(32);
                        if (++spins > limit) { gave_up = true; break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (gave_up || stale) {
                        __hip_atomic_store(ride->timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (gave_up) __hip_atomic_store(done + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the next launch starts from zero
                    }
                }
                __syncthreads();
            }
            if (ride->pieces != 0u) ur::hzb_tail_run<true>(*(KTail)((const __attribute__((address_space(4))) char*)ka + sizeof(LightingParams)), bufA, bufA + ur::kTailTexels);
            else ur::hzb_tail_run<false>(*(KTail)((const __attribute__((address_space(4))) char*)ka + sizeof(LightingParams)), bufA, bufA + ur::kTailTexels);
        }
        ur::timeline_exit(p.timeline, threadIdx.x == 0);
        return;
    }
    float* srgb = reinterpret_cast<float*>(smem + kLdsSrgb);
    MipEntry* mipT = reinterpret_cast<MipEntry*>(smem + kLdsMip);
    float4a* irrT = reinterpret_cast<float4a*>(smem + kLdsIrr);
    float2* lut = reinterpret_cast<float2*>(smem + kLdsLut);
    uint32_t* work = reinterpret_cast<uint32_t*>(smem + kLdsWork);
    uint32_t* dynT = reinterpret_cast<uint32_t*>(smem + kLdsDyn);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // uniform by construction: keeps the tile walk on the scalar ALU
    // ---- Prologue order (in-kernel stamps, tools/stamps_lighting.py): the table loads go first - the barrier below waits
    //      for their conversion, and vector memory returns in order, so behind the tile DMAs they would wait for 64 KB of
    //      tiles per CU -, then the wave's two static tiles, then the constants only the loop needs. One memory latency
    //      for everything.
    constexpr uint32_t T = 64 * WPB, kLutN = kLutE * (kLutH + 2u), kLutTrips = (kLutN + T - 1) / T;
    uint32_t lt[kLutTrips];
#pragma unroll
    for (uint32_t k = 0; k < kLutTrips; ++k) {
        const uint32_t i = min(threadIdx.x + k * T, kLutN - 1u);
        const uint32_t by = i / kLutE, bx = i - by * kLutE;
        const uint32_t sx = min(max(bx, 1u), kLutW) - 1u, sy = min(max(by, 1u), kLutH) - 1u; // border = clamp addressing
        lt[k] = p.lut[sy * kLutW + sx];
    }
    const float sv = threadIdx.x < 256u ? p.srgb[threadIdx.x] : 0.0f;
    half4_t ih = {}, ih10 = {}, ih01 = {}, ih11 = {};
    const uint32_t irrE = p.irrN0 + 2u;
    const uint32_t irrC = p.irrN0 + 1u, irrCount = IRR_LDS ? 6u * irrC * irrC : 0u; // cells: <= 54
    if (threadIdx.x < irrCount) {
        const uint32_t f = threadIdx.x / (irrC * irrC), r = threadIdx.x - f * irrC * irrC, cj = r / irrC, ci = r - cj * irrC;
        const uint32_t t0 = p.irrOffset0 + (f * irrE + cj) * irrE + ci;
        ih = p.env[t0]; ih10 = p.env[t0 + 1u]; ih01 = p.env[t0 + irrE]; ih11 = p.env[t0 + irrE + 1u];
    }
    // Tile schedule: chunks of 2^cs consecutive tiles are dealt round-robin to the workgroups (chunks of 4: the per-workgroup
    // work then differs by +-3 %; with 16 the image content makes it +-9 %, with 1 the DRAM locality of a row is lost),
    // dynamically inside a workgroup: a wave takes its next tile from a counter in LDS. The SIMD's oldest-first
    // arbitration lets some waves of a workgroup run up to twice as fast as others (measured with in-kernel stamps); with
    // a static split the slow ones set the kernel's duration, with the counter all of them finish together.
    // Claims c = wave and c = WPB + wave are static (the two tiles of the prologue).
    constexpr uint32_t cs = kChunkShift, cmask = (1u << cs) - 1u;
    const uint32_t chunkStride = p.hot.groups << cs, base = blockIdx.x << cs;
    // claim c -> tile (c >> cs) * chunkStride + base + (c & cmask): chunks of 2^cs consecutive tiles, dealt round-robin
    uint32_t tile = (wave >> cs) * chunkStride + base + (wave & cmask);
    uint32_t tile1 = ((wave + WPB) >> cs) * chunkStride + base + ((wave + WPB) & cmask);
    const bool have0 = tile < p.hot.numTiles; // (a grid larger than the band: some waves have no tile at all)
    uint32_t ty = (uint32_t)(((uint64_t)tile * p.hot.tilesXMagic) >> 32), tx = tile - ty * p.hot.tilesX; // (exact for tile < numTiles)
    uint32_t ty1 = (uint32_t)(((uint64_t)tile1 * p.hot.tilesXMagic) >> 32), tx1 = tile1 - ty1 * p.hot.tilesX;
    const uint32_t col = lane & 15u, row = lane >> 4;
    const TileSrc src = tile_src<MODE>(&p, lane, 3u);
    const uint32_t bufBase = __builtin_amdgcn_readfirstlane(lds_address(smem + kLdsTiles + wave * (2u * kTileBytes)));
    const unsigned char* myTiles = smem + kLdsTiles + wave * (2u * kTileBytes);
    // this lane's pixel inside a tile, as a byte offset into the HDR band (the tile origin is added per iteration)
    const uint32_t laneHdr = (row * p.hot.W + col) * 8u;


    if (have0) tile_prefetch<MODE>(&p, p.hot.W, p.hot.rows, src, lane, tx, ty, bufBase);
    // The cube's small mips -> LDS by DMA, 1-KB pieces dealt over the waves (the shipped cube: 32 pieces, two per wave), behind the
    // first tile: rough pixels then take their two prefiltered footprints from LDS instead of through the L1 (four gathers fewer).
    const uint32_t cubeLds = __builtin_amdgcn_readfirstlane(lds_address(smem + kLdsTiles + WPB * (2u * kTileBytes)));
    const uint32_t cubeLdsAdj = p.hot.cubeLdsBase - cubeLds; // LDS byte address of an entry = its byte offset in the staged buffer - this
    {
        const uint32_t nb = p.hot.cubeLdsBytes; // uniform; 0: off
        for (uint32_t pc = wave; pc * 1024u < nb; pc += WPB) { // uniform trip count per wave
            const uint32_t at = min(pc * 1024u + lane * 16u, nb - 16u); // (the last piece re-reads the section's last 16 bytes past its end)
            dma16(reinterpret_cast<const char*>(p.hot.env) + p.hot.cubeLdsBase + at, cubeLds + pc * 1024u);
        }
    }
    // per-lane part of the pixel's NDC (the tile origin is added per iteration), and the few uniforms that appear as the
    // SECOND scalar operand of an FMA
    const float ndcxL = fmaf((float)col, p.invW2, 0.5f * p.invW2 - 1.0f);
    const float rowh = (float)row + 0.5f; // ndc.y = (py + 0.5) * 2/H - 1 with py = row0 + 4 ty + row summed exactly: a band and the whole frame agree bit for bit
    const float negBig = vreg(-0x1p126f);
    // Launch constants of the FMA-dense parts as VGPR operands: two neighbouring VALU instructions that both read an SGPR
    // cannot share an issue slot (tools/microbench/valu_rate.hip, table 4), and a VOP3 reads one SGPR at most.
    float R[9], WC[3], shC[3], shT[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = vreg(p.hot.R[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        WC[k] = vreg(p.hot.WC[k]); shC[k] = vreg(p.hot.shC[k]); shT[k] = vreg(p.hot.shT[k]);
    }
    // ---- the tables: converted and written to LDS once per workgroup ------------------------------------------------------
    {
        if (threadIdx.x == 0) { work[0] = 2u * WPB; work[1] = 0u; work[2] = 0u; work[3] = 0u; } // [0] next tile claim, [1] waves that have left the loop (debug timeline), [2] waves done with their HZB pieces, [3] waves whose pieces of the cube's small mips have landed in LDS
        if (p.bal.poolChunks != 0u && threadIdx.x < kDynSlots) { // uniform: the run-time part of the tile schedule (struct Balance)
            // this workgroup's claim word q serves workgroups [8q, 8q + 8): its share of the pool is chunks [P0, P1); the first
            // `lookahead` chunks of each of its nq workgroups are pre-assigned, the rest is claimed
            const uint32_t q8 = blockIdx.x & ~7u, nq = min(8u, p.hot.groups - q8), j = blockIdx.x - q8, la = p.bal.lookahead;
            const uint32_t P0 = (uint32_t)(((unsigned long long)q8 * p.bal.poolMagic) >> 32);
            const uint32_t P1 = (uint32_t)(((unsigned long long)(q8 + nq) * p.bal.poolMagic) >> 32);
            const uint32_t first = P0 + la * nq; // (the host has checked P1 - P0 >= la * nq for every word)
            dynT[threadIdx.x] = threadIdx.x < la ? p.bal.staticTiles + ((P0 + j * la + threadIdx.x) << p.bal.dynShift) : kDynNotReady;
            if (threadIdx.x == 0) { work[4] = first; work[5] = P1 - first; }
        }
        if (threadIdx.x < 256u) srgb[threadIdx.x] = sv;
        if (threadIdx.x < irrCount) {
            // value(fx, fy) = t00 + (t10 - t00) fx + (t01 - t00) fy + (t11 - t10 - t01 + t00) fx fy, per channel
            const float a[3] = {(float)ih.x, (float)ih.y, (float)ih.z}, b[3] = {(float)ih10.x, (float)ih10.y, (float)ih10.z};
            const float c[3] = {(float)ih01.x, (float)ih01.y, (float)ih01.z}, d[3] = {(float)ih11.x, (float)ih11.y, (float)ih11.z};
            // one 64-byte entry per cell: (a.x a.y b.x b.y | c.x c.y d.x d.y | a.z b.z c.z d.z): x and y are evaluated as a packed pair
            float A_[3], B_[3], C_[3], D_[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { A_[k] = a[k]; B_[k] = b[k] - a[k]; C_[k] = c[k] - a[k]; D_[k] = (d[k] - b[k]) - (c[k] - a[k]); }
            irrT[threadIdx.x * 4u + 0u] = float4a{A_[0], A_[1], B_[0], B_[1]};
            irrT[threadIdx.x * 4u + 1u] = float4a{C_[0], C_[1], D_[0], D_[1]};
            irrT[threadIdx.x * 4u + 2u] = float4a{A_[2], B_[2], C_[2], D_[2]};
        }
        if (threadIdx.x < 17u) {
            const uint32_t m = min(threadIdx.x, p.envMips - 1u); // entry [envMips] repeats the last mip (weight 0 when it is read)
            const uint32_t N = max(1u, p.envBase >> m), E = N + 2u;
            uint32_t bordered = 0u, before = 0u; // texels of all bordered mips; pair entries of the mips in front of this one
            for (uint32_t k = 0; k < p.envMips; ++k) {
                const uint32_t e = max(1u, p.envBase >> k) + 2u;
                bordered += 6u * e * e;
                if (k < m) before += 6u * e * (e - 1u);
            }
            mipT[threadIdx.x] = MipEntry{(float)N, (float)(12u * E), (float)(12u * E * (E - 1u)), (float)(bordered * 8u + before * 12u)};
        }
#pragma unroll
        for (uint32_t k = 0; k < kLutTrips; ++k) {
            const uint32_t i = threadIdx.x + k * T;
            // unorm16 -> float by the reciprocal: within one ulp of the oracle's quotient, a tenth of the instructions
            if (i < kLutN) lut[i] = float2{(float)(lt[k] & 0xFFFFu) * (1.0f / 65535.0f), (float)(lt[k] >> 16) * (1.0f / 65535.0f)};
        }
    }
    __syncthreads();
    // The second static tile goes out behind the barrier, i.e. behind every first tile of the workgroup: the start-up burst
    // (2 KB per wave and tile, 8 MB over the chip) that the first iteration has to wait for is halved, the other half
    // lands under the first iteration's arithmetic. Its two DMA instructions may stay in flight here.
    if (have0 && tile1 < p.hot.numTiles) {
        tile_prefetch<MODE>(&p, p.hot.W, p.hot.rows, src, lane, tx1, ty1, bufBase + kTileBytes);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (that wait covered this wave's pieces of the cube's small mips, issued ahead of the barrier: count the wave in; a pixel takes
    // the LDS copy only once all WPB waves are counted, the global section - the same bytes - until then)
    if (lane == 0) __hip_atomic_fetch_add(work + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

    // ---- a held-back HZB chain's wide launch rides along: this workgroup's pieces g, g + groups, ... before the tile loop,
    //      dealt over its last `walkers` waves. The waves that walk none join the tile loop at once and the LDS work counter
    //      hands the walkers' share of tiles to them. ONE walker when the band is long (a whole 4K frame: fifteen waves shade
    //      beside it, frame 78.8 us; with all sixteen walking 81.0: the chain's loads then compete with every wave's first
    //      tiles), all of them when it is short (a 1/8 band: 22.8 against 37.8 us with one walker).
    //      (RIDE_ALL is a template parameter: with the number of walkers a run-time value the loop spills.)
    if constexpr (WPB == 16) {
        constexpr uint32_t kWalkers = WPB;
        if (RIDE_ALL || wave == kWalkers - 1u) { // uniform
            auto ka = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            typedef const __attribute__((address_space(4))) HzbRide* KRide;
            const KRide ride = (KRide)((const __attribute__((address_space(4))) char*)ka + sizeof(LightingParams) + sizeof(ur::HzbTail));
            const uint32_t pieces = ride->pieces;
            if (pieces != 0u) {
                const uint32_t gx = ride->grid_x, groups = p.hot.groups;
                float* sh2 = reinterpret_cast<float*>(smem + kLdsHzb) + (RIDE_ALL ? wave * 80u : 0u);
                const uint32_t first = RIDE_ALL ? blockIdx.x + (kWalkers - 1u - wave) * groups : blockIdx.x, step = RIDE_ALL ? groups * kWalkers : groups;
                // issue priority 3 for the walk (see the loop below): the sooner its loads leave, the sooner this wave shades again
                __builtin_amdgcn_s_setprio(3);
                for (uint32_t piece = first; piece < pieces; piece += step) { // uniform
                    const uint32_t by = piece / gx, bx = piece - by * gx;
                    ur::hzb_wide_piece_by_one_wave<true>(ride->d, bx, by + ride->d.by0, lane, sh2, sh2 + 64);
                }
                // producer side of the hand-off: mip 4 was stored write-through (sc1); the wave's stores drained, then ONE
                // arrival per workgroup. (A release fence instead would write back everything the lighting waves have dirtied
                // in this XCD's L2: measured, it made the launch 10 us longer.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_setprio(0);
                bool signals = true;
                if (RIDE_ALL) { // every wave counts itself in LDS behind its drained stores; the one that completes the count signals
                    uint32_t before = 0;
                    if (lane == 0) before = __hip_atomic_fetch_add(work + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    signals = __builtin_amdgcn_readfirstlane(before) == kWalkers - 1u;
                }
                // A single-lane device-scope add (agent scope is the default for global atomics on gfx950), written as an
                // instruction with EXEC narrowed to lane 0: the same statement as C++ under `if (lane == 0)` makes hipcc keep the
                // loop's LDS-DMA destination (an SGPR operand of inline asm) in a VGPR.
                uint32_t* done = ride->done; // null: nobody inside this launch waits for the pieces (a band-sharded chain: ur_build_hzb_band)
                if (signals && done != nullptr) { // uniform
                    uint64_t keep_exec;
                    const uint32_t zero = 0u, one = 1u;
                    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %1, %2, %3\n\ts_mov_b64 exec, %0"
                                 : "=&s"(keep_exec)
                                 : "v"(zero), "v"(one), "s"(done)
                                 : "memory");
                }
            }
        }
    }

    uint32_t parity = 0;
// (macro: the statement appears in the shading path and in the all-sky path)
#define UR_PREFETCH_POINT()                                                                                              \
    do {                                                                                                                 \
        const uint32_t c = __builtin_amdgcn_readfirstlane(claim); /* 0 when nothing was claimed (more1 == 0) */            \
        if (__builtin_expect(c >= p.hot.staticClaims, 0)) tile2 = dyn_claim(kp, work, dynT, c, lane); /* (uniform) */      \
        else tile2 = (c >> cs) * chunkStride + base + (c & cmask);                                                       \
        uint32_t in_band;                                                                                                \
        asm("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(in_band) : "s"(tile2), "s"(p.hot.numTiles) : "scc"); \
        more2 = in_band & more1; /* scalar AND: no branch on more1 here */                                                \
        if (more2) {                                                                                                     \
            ty2 = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)tile2 * p.hot.tilesXMagic) >> 32));                  \
            tx2 = tile2 - ty2 * p.hot.tilesX;                                                                               \
            tile_prefetch<MODE>(kp, p.hot.W, p.hot.rows, src, lane, tx2, ty2, bufBase + parity * kTileBytes);                  \
        } else {                                                                                                         \
            tile2 = 0xFFFFFFFFu; /* nothing was left to claim */                                                          \
        }                                                                                                                \
    } while (0)
    while (have0) {
        const KParams kp = fresh_params(); // cold paths re-read what they need (sky constants, shadow slow path, partial tiles)
        // the tile two steps ahead: claimed here (LDS atomic, long back when the prefetch point needs it)
        uint32_t tile2 = 0xFFFFFFFFu, tx2 = 0, ty2 = 0;
        uint32_t more2 = 0u;
        uint32_t more1; // uniform, kept as a scalar integer (see flag_any)
        asm("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(more1) : "s"(__builtin_amdgcn_readfirstlane(tile1)), "s"(p.hot.numTiles) : "scc");
        uint32_t claim = 0;
        if (more1 && lane == 0) claim = __hip_atomic_fetch_add(work, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // Wave issue priority follows the iteration's phase: 0 from here to the gather wait, 2 behind it (below). The SIMD then serves the
        // wave whose gathers have landed first: it issues the DMA for its tile two steps ahead, filters, stores and frees its slot
        // in the iteration sooner, and the others wait on memory anyway (4K alone 69.1-69.4 -> 68.0-68.6 us, in the frame
        // 72.2-72.6 -> 71.6-71.9, profiles/r04_priority.txt; the opposite order - decode and gather issue first - loses 0.3 us at 4K).
        __builtin_amdgcn_s_setprio(0);
        const unsigned char* buf = myTiles + parity * kTileBytes;
        const uint2 ga = *reinterpret_cast<const uint2*>(buf + lane * 8u);          // (nx, ny), (nz, -viewZ)
        const uint2 gb = *reinterpret_cast<const uint2*>(buf + 512u + lane * 8u);   // (specular, metallic), (roughness, 1)
        const uint2 gd = *reinterpret_cast<const uint2*>(buf + 1024u + lane * 8u);  // HDR in
        const uint32_t gc = *reinterpret_cast<const uint32_t*>(buf + 1536u + lane * 4u);
        const float ndcx = fmaf((float)(tx * 16u), p.hot.invW2, ndcxL), ndcy = fmaf((float)(p.hot.row0 + ty * 4u) + rowh, p.hot.invH2, -1.0f);

        bool sky = false;
        uint32_t all_sky = 0u; // uniform: every pixel of the tile is sky
        F3 out = f3(0.0f, 0.0f, 0.0f);
        if (MODE == ur::UR_MODE_FUSED) {
            const float depth = *reinterpret_cast<const float*>(buf + 1792u + lane * 4u);
            // no pixel of the frame has a sphere depth above skyDepthMax: a wave of nearer geometry skips the per-pixel test
            if (__any(!(depth > p.hot.skyDepthMax))) {
                const float vx = ndcx * p.hot.skyInvP11, vy = ndcy * p.hot.nSkyInvP22;
                // sphere depth (Near/R) * |(vx, vy, 1)| >= depth, compared squared (depth is in [0,1]): no square root
                sky = p.hot.skyNearOverR2 * fmaf(vx, vx, fmaf(vy, vy, 1.0f)) >= depth * depth;
                all_sky = flag_all(sky);
                if (sky) out = sky_pixel(kp, vx, vy);
            }
        }
        float outw = 1.0f;
        if (!all_sky) { // wave-uniform: sky lanes shade whatever they loaded and drop the result
            // ---- decode; every vector in WORLD space (the view matrix is rigid): the camera ray through the pixel is affine in
            //      ndc, the normal is rotated once, and the reflection vector needs no rotation of its own. x and y of a vector
            //      travel as a packed pair wherever both take the same operation -----------------------------------------------
            const float nx = h2f_lo(ga.x), ny = h2f_hi(ga.x), nz = h2f_lo(ga.y), wv = h2f_hi(ga.y);
            const float nr = rsq(fmaf(nz, nz, fmaf(ny, ny, nx * nx))); // normalize(0) = NaN, as in the reference
            f2_t Nxy;
            float Nz;
            const F3 N = rot(f3(nx * nr, ny * nr, nz * nr), R);
            Nxy = f2(N.x, N.y); Nz = N.z;
            const float viewZ = -wv;
            const float spec0 = h2f_lo(gb.x), metallic = h2f_hi(gb.x), roughness = h2f_lo(gb.y);
            const F3 L = f3(p.hot.Lw[0], p.hot.Lw[1], p.hot.Lw[2]);
            const float NdotLraw = fmaf(Nz, L.z, fmaf(Nxy.y, L.y, Nxy.x * L.x));
            const float NdotL = sat(NdotLraw);
            f2_t Wxy;
            Wxy = f2(fmaf(ndcx, p.hot.WA[0], fmaf(ndcy, p.hot.WB[0], WC[0])), fmaf(ndcx, p.hot.WA[1], fmaf(ndcy, p.hot.WB[1], WC[1])));
            const float Wz = fmaf(ndcx, p.hot.WA[2], fmaf(ndcy, p.hot.WB[2], WC[2])); // (ra, rb, 1) * ViewInverse3x3
            // V = normalize(-viewPos) = -sign(viewZ) Wd / |Wd|; sign(-viewZ) is the stored sign of A.w
            const float vs = __builtin_copysignf(rsq(fmaf(Wz, Wz, fmaf(Wxy.y, Wxy.y, Wxy.x * Wxy.x))), wv);
            const f2_t Vxy = f2(Wxy.x * vs, Wxy.y * vs);
            const float Vz = Wz * vs;
            const float NdotVraw = fmaf(Nz, Vz, fmaf(Nxy.y, Vxy.y, Nxy.x * Vxy.x));
            const float NdotV = sat(NdotVraw);
            // ---- global gathers: the two prefiltered mips (bordered cube, addresses in fp32: every integer multiply would hold
            //      the issue port), then the shadow block ------------------------------------------------------------------------
            const float t2 = 2.0f * NdotVraw;
            const f2_t Rxyw = f2(fmaf(t2, Nxy.x, -Vxy.x), fmaf(t2, Nxy.y, -Vxy.y));
            const F3 Rw = f3(Rxyw.x, Rxyw.y, fmaf(t2, Nz, -Vz));
            const float lvl = __builtin_amdgcn_fmed3f(roughness * p.hot.maxMip, 0.0f, p.hot.envMaxLevel);
            const float fl = __builtin_amdgcn_fractf(lvl);
            const MipEntry* me = mipT + (uint32_t)lvl;
            const float4a e0 = *reinterpret_cast<const float4a*>(me), e1 = *reinterpret_cast<const float4a*>(me + 1);
            const float faceR = __builtin_amdgcn_cubeid(Rw.x, Rw.y, Rw.z);
            const float invR = rcp(fabsf(__builtin_amdgcn_cubema(Rw.x, Rw.y, Rw.z)));
            f2_t uvR;
            uvR = f2(fmaf(__builtin_amdgcn_cubesc(Rw.x, Rw.y, Rw.z), invR, 0.5f), fmaf(__builtin_amdgcn_cubetc(Rw.x, Rw.y, Rw.z), invR, 0.5f));
            const void* env = p.hot.env;
            u32x3_t p0a, p0b, p1a, p1b;
            f2_t f0, f1; // (fx, fy) of the two mips
            uint32_t o0, o1; // byte offsets of the two footprints in the staged cube's RGB row-pair section
            {
                const f2_t xy = f2(fmaf(uvR.x, e0.x, 0.5f), fmaf(uvR.y, e0.x, 0.5f)); // bordered coordinates in [0.5, N + 0.5]
                const float i0 = floorf(xy.x), j0 = floorf(xy.y);
                f0 = f2(xy.x - i0, xy.y - j0);
                // the footprint's 24 contiguous bytes {(i0, j0), (i0, j0 + 1)} {(i0 + 1, j0), (i0 + 1, j0 + 1)}
                o0 = (uint32_t)fmaf(faceR, e0.z, fmaf(j0, e0.y, fmaf(i0, 12.0f, e0.w)));
            }
            {
                const f2_t xy = f2(fmaf(uvR.x, e1.x, 0.5f), fmaf(uvR.y, e1.x, 0.5f));
                const float i0 = floorf(xy.x), j0 = floorf(xy.y);
                f1 = f2(xy.x - i0, xy.y - j0);
                o1 = (uint32_t)fmaf(faceR, e1.z, fmaf(j0, e1.y, fmaf(i0, 12.0f, e1.w)));
            }
            // a footprint whose mip is among the cube's small ones is read from the workgroup's LDS copy once every wave's pieces have
            // landed (the same bytes as the global section): the upper mip of the pair from level cubeLdsLevel - 1 on, both from
            // cubeLdsLevel on
            {
                typedef __attribute__((address_space(3))) const u32x3_a4* LdsRgb;
                const float lv = __hip_atomic_load(work + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (uint32_t)WPB ? lvl : -2.0f;
                if (lv >= p.hot.cubeLdsLevel) {
                    const uint32_t l0 = o0 - cubeLdsAdj;
                    p0a = *(LdsRgb)(uintptr_t)l0; p0b = *(LdsRgb)(uintptr_t)(l0 + 12u);
                } else {
                    p0a = *reinterpret_cast<const UR_GLOBAL u32x3_a4*>((const UR_GLOBAL char*)env + o0);
                    p0b = *reinterpret_cast<const UR_GLOBAL u32x3_a4*>((const UR_GLOBAL char*)env + (o0 + 12u));
                }
                if (lv + 1.0f >= p.hot.cubeLdsLevel) { // (the pair's upper mip is floor(level) + 1, or the last one: in LDS either way)
                    const uint32_t l1 = o1 - cubeLdsAdj;
                    p1a = *(LdsRgb)(uintptr_t)l1; p1b = *(LdsRgb)(uintptr_t)(l1 + 12u);
                } else {
                    p1a = *reinterpret_cast<const UR_GLOBAL u32x3_a4*>((const UR_GLOBAL char*)env + o1);
                    p1b = *reinterpret_cast<const UR_GLOBAL u32x3_a4*>((const UR_GLOBAL char*)env + (o1 + 12u));
                }
            }
            const float faceN = __builtin_amdgcn_cubeid(Nxy.x, Nxy.y, Nz);
            const float invN = rcp(fabsf(__builtin_amdgcn_cubema(Nxy.x, Nxy.y, Nz)));
            f2_t uvN;
            uvN = f2(fmaf(__builtin_amdgcn_cubesc(Nxy.x, Nxy.y, Nz), invN, 0.5f), fmaf(__builtin_amdgcn_cubetc(Nxy.x, Nxy.y, Nz), invN, 0.5f));
            u32x4_t pia = {0, 0, 0, 0}, pib = {0, 0, 0, 0};
            float fxi = 0.0f, fyi = 0.0f;
            if (!IRR_LDS) {
                const float x = fmaf(uvN.x, p.hot.irrNf, 0.5f), y = fmaf(uvN.y, p.hot.irrNf, 0.5f);
                const float i0 = floorf(x), j0 = floorf(y);
                fxi = x - i0; fyi = y - j0;
                const uint32_t o = (uint32_t)(fmaf(faceN, p.hot.irrEEf, fmaf(j0, p.hot.irrEf, i0)) + p.hot.irrOfff) * 8u;
                pia = *reinterpret_cast<const UR_GLOBAL u32x4_a8*>((const UR_GLOBAL char*)env + o);
                pib = *reinterpret_cast<const UR_GLOBAL u32x4_a8*>((const UR_GLOBAL char*)env + (o + p.hot.irrRowBytes));
            }
            // The shadow term multiplies NdotL: a wave whose every pixel faces away from the light skips the PCF (direct = 0).
            const uint32_t wave_direct = flag_any(NdotL > 0.0f);
            f32x3_t sa = {0, 0, 0}, sb = {0, 0, 0}, sc3 = {0, 0, 0};
            f2_t xya = f2(0.0f, 0.0f), sf = f2(0.0f, 0.0f); // (su * W - 0.5, sv * H - 0.5) and its fraction
            float cmp = 0.0f;
            uint32_t any_slow = 0u; // uniform: some pixel's 3x3 block touches the border of the map (or lies outside it)
            float scs = 0.0f, p5 = 0.0f;
            // ONE uniform region for everything only a lit wave needs ahead of the gather wait: the shadow taps' issue (the last
            // gathers of the iteration) and EvaluatePBR's D, G, F terms (PBRCommon.hlsl:24-48), which run while the gathers are
            // in flight. A wave without a lit pixel has direct = 0 whatever they are.
            if (wave_direct) {
                if (SHADOWS) {
                    // orthographic light: (su * W - 0.5, sv * H - 0.5, z - bias) = viewZ * (affine in ndc) + constant
                    xya = f2(fmaf(viewZ, fmaf(ndcx, p.hot.shA[0], fmaf(ndcy, p.hot.shB[0], shC[0])), shT[0]), fmaf(viewZ, fmaf(ndcx, p.hot.shA[1], fmaf(ndcy, p.hot.shB[1], shC[1])), shT[1]));
                    cmp = fmaf(viewZ, fmaf(ndcx, p.hot.shA[2], fmaf(ndcy, p.hot.shB[2], shC[2])), shT[2]);
                    const float xa0 = floorf(xya.x), ya0 = floorf(xya.y);
                    sf = f2(xya.x - xa0, xya.y - ya0);
                    // 3x3 block origin clamped into the map (always a valid address); unclamped <=> no tap touches the border
                    const float ic = __builtin_amdgcn_fmed3f(xa0, 0.0f, p.hot.shadowWm3), jc = __builtin_amdgcn_fmed3f(ya0, 0.0f, p.hot.shadowHm3);
                    any_slow = flag_any(ic != xa0) | flag_any(jc != ya0); // (each ballot straight off its comparison)
                    // Only lanes that face the light need their block: the shadow term multiplies N.L, so a lane with N.L = 0 gets
                    // direct = 0 whatever its taps say. Such lanes all fetch the map's first block instead of their own - one cache line
                    // per row for the lot of them instead of up to three lines EACH (independent-per-pixel G-buffers: half the lanes of
                    // every wave, 207 -> 160 us at 4K). A select on the address, not on EXEC: the three loads stay straight-line code.
                    const uint32_t o0 = NdotL > 0.0f ? (uint32_t)fmaf(jc, p.hot.shadowWf, ic) * 4u : 0u, o1 = o0 + p.hot.shadowRowBytes, o2 = o1 + p.hot.shadowRowBytes;
                    const float* smap = p.hot.shadow;
                    sa = *reinterpret_cast<const UR_GLOBAL f32x3_a4*>((const UR_GLOBAL char*)smap + o0);
                    sb = *reinterpret_cast<const UR_GLOBAL f32x3_a4*>((const UR_GLOBAL char*)smap + o1);
                    sc3 = *reinterpret_cast<const UR_GLOBAL f32x3_a4*>((const UR_GLOBAL char*)smap + o2);
                }
                float NdotH, om;
                // V and L are unit vectors: |V + L|^2 = 2 + 2 V.L, N.(V + L) = N.V + N.L, V.(V + L) = 1 + V.L
                const float VL = fmaf(Vz, L.z, fmaf(Vxy.y, L.y, Vxy.x * L.x));
                float hr = rsq(fmaf(VL, 2.0f, 2.0f));
                // ... which cancels when the view ray runs along the light (V.L -> -1: 2 + 2 V.L keeps 1e-7 / (1 + V.L) of relative
                // error). A wave that has such a pixel (a camera looking into the light; uniform, rare) takes |V + L| from the components,
                // as the shader writes it (PBRCommon.hlsl via DeferredLighting.hlsl:73; tests: test_lighting_view_ray_along_the_light)
                if (__builtin_expect(flag_any(VL < -0.98f), 0)) {
                    const float hx = Vxy.x + L.x, hy = Vxy.y + L.y, hz = Vz + L.z;
                    hr = rsq(fmaf(hz, hz, fmaf(hy, hy, hx * hx)));
                }
                NdotH = sat((NdotVraw + NdotLraw) * hr);
                om = fmaf(-VL, hr, 1.0f - hr); // 1 - VdotH, VdotH = (1 + V.L) / |V + L| in [0,1]: saturate is the identity up to rounding
                const float alpha = roughness * roughness;
                const float alpha2 = alpha * alpha;
                const float denom = fmaf(NdotH * NdotH, alpha2 - 1.0f, 1.0f);
                float k = roughness + 1.0f;
                k = (k * k) * 0.125f;
                const float omk = 1.0f - k;
                const float gv = fmaf(NdotV, omk, k), gl = fmaf(NdotL, omk, k);
                // max(4 x, 1e-4) = 4 max(x, 1e-4 / 4) exactly (power-of-two scaling)
                const float nvl = NdotV * NdotL;
                // D G / max(4 NdotL NdotV, 1e-4) with ONE reciprocal: every factor of the denominator is >= 1e-4 and <= ~40, their
                // product stays far inside the fp32 range
                scs = (alpha2 * nvl) * rcp((fmaxf(3.14159265f * denom * denom, 1e-4f) * (gv * gl)) * (4.0f * fmaxf(nvl, 1e-4f * 0.25f)));
                const float om2 = om * om;
                p5 = om2 * om2 * om;
            }
            // ---- LDS lookups: sRGB, BRDF LUT, irradiance -----------------------------------------------------------------------
            // table byte offsets straight from the packed texel: (c << 2) & 0x3FC, (c >> 6) & 0x3FC, (c >> 14) & 0x3FC
            const unsigned char* srgbB = reinterpret_cast<const unsigned char*>(srgb);
            const f2_t albxy = f2(*reinterpret_cast<const float*>(srgbB + ((gc << 2) & 0x3FCu)), *reinterpret_cast<const float*>(srgbB + ((gc >> 6) & 0x3FCu)));
            const float albz = *reinterpret_cast<const float*>(srgbB + ((gc >> 14) & 0x3FCu));
            f2_t bab; // (brdf.x, brdf.y)
            {
                // bordered coordinates: x in [0.5, W + 0.5] (NdotV is saturated), y clamped likewise (roughness is not)
                const float x = fmaf(NdotV, (float)kLutW, 0.5f);
                const float y = __builtin_amdgcn_fmed3f(fmaf(roughness, (float)kLutH, 0.5f), 0.5f, (float)kLutH + 0.5f);
                const float i0 = floorf(x), j0 = floorf(y);
                const f2_t fxy = f2(x - i0, y - j0);
                const f2_t* t = reinterpret_cast<const f2_t*>(lut) + (uint32_t)fmaf(j0, (float)kLutE, i0);
                const f2_t t00 = t[0], t10 = t[1], t01 = t[kLutE], t11 = t[kLutE + 1];
                const float fx = fxy.x, fy = fxy.y;
                const float wy0 = 1.0f - fy;
                const float w10 = wy0 * fx, w00 = wy0 - w10, w11 = fy * fx, w01 = fy - w11;
                bab = f2(fmaf(w11, t11.x, fmaf(w01, t01.x, fmaf(w10, t10.x, w00 * t00.x))), fmaf(w11, t11.y, fmaf(w01, t01.y, fmaf(w10, t10.y, w00 * t00.y))));
            }
            f2_t irrxy = f2(0.0f, 0.0f);
            float irrz = 0.0f;
            if (IRR_LDS) {
                const f2_t xy = f2(fmaf(uvN.x, p.hot.irrNf, 0.5f), fmaf(uvN.y, p.hot.irrNf, 0.5f));
                const float i0 = floorf(xy.x), j0 = floorf(xy.y);
                const f2_t fxy2 = f2(xy.x - i0, xy.y - j0);
                const float fx = fxy2.x, fy = fxy2.y;
                // bilinear CELL (i0, j0) of the face (irrEf / irrEEf hold N + 1 and (N + 1)^2 when the table is in LDS): the polynomial
                // a + b fx + c fy + d fx fy of each channel - one 64-byte entry, three ds_read_b128, laid out as
                // (a.x a.y b.x b.y | c.x c.y d.x d.y | a.z b.z c.z d.z): x and y are evaluated as a packed pair
                const float4a* t = irrT + 4u * (uint32_t)fmaf(faceN, p.hot.irrEEf, fmaf(j0, p.hot.irrEf, i0));
                const float4a c0 = t[0], c1 = t[1], c2 = t[2];
                const float fxy = fx * fy;
                irrxy = f2(fmaf(c1.z, fxy, fmaf(c1.x, fy, fmaf(c0.z, fx, c0.x))), fmaf(c1.w, fxy, fmaf(c1.y, fy, fmaf(c0.w, fx, c0.y))));
                irrz = fmaf(c2.w, fxy, fmaf(c2.z, fy, fmaf(c2.y, fx, c2.x)));
            }
            // ---- EvaluatePBR, PBRCommon.hlsl:24-48 (runs while the gathers are in flight) ----------------------------------------
            float kdm = 1.0f - metallic;
            f2_t F0xy;
            float F0z;
            // lerp(spec0, albedo, metallic) = albedo * metallic + spec0 * (1 - metallic)
            const float s0k = spec0 * kdm;
            F0xy = f2(fmaf(albxy.x, metallic, s0k), fmaf(albxy.y, metallic, s0k));
            F0z = fmaf(albz, metallic, s0k);
            // the math above is wanted BEFORE the first wait on a gather, not sunk behind it
            asm volatile("" : "+v"(scs), "+v"(p5), "+v"(kdm), "+v"(bab), "+v"(irrxy), "+v"(irrz));
            __builtin_amdgcn_sched_barrier(0);
            need(p0a, p0b, p1a, p1b);
            if (!IRR_LDS) need(pia, pib, pia, pib);
            __builtin_amdgcn_sched_barrier(0);
            // hipcc has no load of its own in flight here: the DMA for the tile two steps ahead goes into the buffer just read.
            // The vmcnt(0) retires every older vector-memory operation of the wave, in particular the DMA issued at the previous
            // iteration's prefetch point: the tile the NEXT iteration reads is in LDS from here on.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(2);
            UR_PREFETCH_POINT();
            __builtin_amdgcn_sched_barrier(0);
            // ---- filter, combine ---------------------------------------------------------------------------------------------------
            f2_t prexy;
            float prez;
            {
                // The 24 mixed-precision FMAs below cannot share an issue slot with anything (tools/microbench/valu_rate.hip, table 4):
                // a weight computed BETWEEN two of them costs a slot of its own, computed next to another weight half of one.
                const float s0 = 1.0f - fl;
                float a00, a10, a01, a11, b00, b10, b01, b11;
                const float fx0 = f0.x, fy0 = f0.y, fx1 = f1.x, fy1 = f1.y;
                float a1 = fy0 * s0, a0 = s0 - a1, b1 = fy1 * fl, b0 = fl - b1;
                a10 = a0 * fx0; a00 = a0 - a10; a11 = a1 * fx0; a01 = a1 - a11;
                b10 = b0 * fx1; b00 = b0 - b10; b11 = b1 * fx1; b01 = b1 - b11;
                asm volatile("" : "+v"(a00), "+v"(a10), "+v"(a01), "+v"(a11), "+v"(b00), "+v"(b10), "+v"(b01), "+v"(b11));
                float x, y, z;
                // (p.a = {R G | B R' | G' B'} of column i0: texel (i0, j0) then texel (i0, j0 + 1); p.b the same of column i0 + 1; summed
                // in the order 00, 10, 01, 11)
                x = mul_lo(p0a.x, a00); y = mul_hi(p0a.x, a00); z = mul_lo(p0a.y, a00);
                x = mix_lo(x, p0b.x, a10); y = mix_hi(y, p0b.x, a10); z = mix_lo(z, p0b.y, a10);
                x = mix_hi(x, p0a.y, a01); y = mix_lo(y, p0a.z, a01); z = mix_hi(z, p0a.z, a01);
                x = mix_hi(x, p0b.y, a11); y = mix_lo(y, p0b.z, a11); z = mix_hi(z, p0b.z, a11);
                x = mix_lo(x, p1a.x, b00); y = mix_hi(y, p1a.x, b00); z = mix_lo(z, p1a.y, b00);
                x = mix_lo(x, p1b.x, b10); y = mix_hi(y, p1b.x, b10); z = mix_lo(z, p1b.y, b10);
                x = mix_hi(x, p1a.y, b01); y = mix_lo(y, p1a.z, b01); z = mix_hi(z, p1a.z, b01);
                x = mix_hi(x, p1b.y, b11); y = mix_lo(y, p1b.z, b11); z = mix_hi(z, p1b.z, b11);
                prexy = f2(x, y); prez = z;
                if (!IRR_LDS) {
                    CubeTaps t;
                    F3 irradiance;
                    t.r0 = uint4u{pia.x, pia.y, pia.z, pia.w}; t.r1 = uint4u{pib.x, pib.y, pib.z, pib.w}; t.fx = fxi; t.fy = fyi;
                    cube_taps_filter<false>(irradiance, t, 1.0f); // irrFrac == 0 in this kernel
                    irrxy = f2(irradiance.x, irradiance.y); irrz = irradiance.z;
                }
            }
            // ambient = irradiance * (1 - metallic) * albedo + prefiltered * (F0 * brdf.x + brdf.y)
            f2_t Axy, colxy;
            float Az, colz;
            Axy = f2(kdm * albxy.x, kdm * albxy.y);
            colxy = f2(fmaf(irrxy.x, Axy.x, prexy.x * fmaf(F0xy.x, bab.x, bab.y)), fmaf(irrxy.y, Axy.y, prexy.y * fmaf(F0xy.y, bab.x, bab.y)));
            Az = kdm * albz;
            colz = fmaf(irrz, Az, prez * fmaf(F0z, bab.x, bab.y));
            if (wave_direct) { // + ((1 - F) A + F sc) * light * shadow * N.L, which is zero in every lane of an unlit wave
                // ---- shadow filter first (the same uniform region: the explicit vmcnt(0) at the prefetch point has retired its taps) ----
                float shadow = 1.0f;
                if (SHADOWS) {
                    // PCF = 1 - 0.25 sum w (cmp > t) with separable weights (1-f, 1, f); then lerp(1, pcf, strength)
                    const float sfx = sf.x, sfy = sf.y;
                    const float cb = cmp * 0x1p126f;
                    const float wx0 = 1.0f - sfx, wy0 = 1.0f - sfy;
                    const float r0 = fmaf(gt_step(cb, sa.z, negBig), sfx, fmaf(gt_step(cb, sa.x, negBig), wx0, gt_step(cb, sa.y, negBig)));
                    const float r1 = fmaf(gt_step(cb, sb.z, negBig), sfx, fmaf(gt_step(cb, sb.x, negBig), wx0, gt_step(cb, sb.y, negBig)));
                    const float r2 = fmaf(gt_step(cb, sc3.z, negBig), sfx, fmaf(gt_step(cb, sc3.x, negBig), wx0, gt_step(cb, sc3.y, negBig)));
                    shadow = fmaf(fmaf(sfy, r2, fmaf(wy0, r0, r1)), p.hot.shadowNegQuarterStrength, 1.0f);
                    if (__builtin_expect(any_slow != 0u, 0)) { // some pixel's footprint touches the border (or lies outside the map)
                        const float xa = xya.x, ya = xya.y;
                        const bool lit = xa >= -0.5f && ya >= -0.5f && xa <= kp->hot.shadowXmax && ya <= kp->hot.shadowYmax;
                        const float xa0 = floorf(xa), ya0 = floorf(ya);
                        if (!(__builtin_amdgcn_fmed3f(xa0, 0.0f, kp->hot.shadowWm3) == xa0 && __builtin_amdgcn_fmed3f(ya0, 0.0f, kp->hot.shadowHm3) == ya0)) {
                            const float s = shadow_pcf_border_inline(p.hot.shadow, kp->hot.shadowWi, kp->hot.shadowHi, (int)xa0, (int)ya0, sfx, sfy, cmp);
                            shadow = mix(1.0f, s, kp->hot.shadowStrength);
                        }
                        if (!lit) shadow = 1.0f;
                    }
                }
                const float sh_l = shadow * NdotL;
#define UR_CHANNEL(F0c, Ac, colc, i)                                                                          \
{                                                                                                         \
    const float F = fmaf(1.0f - F0c, p5, F0c);                                                            \
    const float direct = fmaf(F, scs - Ac, Ac);                                                           \
    colc = fmaf(direct, p.hot.lightRGB[i] * sh_l, colc); /* (an SGPR operand: three registers the claim path needs) */ \
}
                UR_CHANNEL(F0xy.x, Axy.x, colxy.x, 0)
                UR_CHANNEL(F0xy.y, Axy.y, colxy.y, 1)
                UR_CHANNEL(F0z, Az, colz, 2)
#undef UR_CHANNEL
            }
            if (!sky) {
                out = f3(h2f_lo(gd.x) + colxy.x, h2f_hi(gd.x) + colxy.y, h2f_lo(gd.y) + colz);
                outw = h2f_hi(gd.y) + 1.0f;
            }
        } else {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            UR_PREFETCH_POINT();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ty * 4u + row < p.hot.rows) { // false only in the rows a partial bottom tile hangs over the band
            half4_t o;
            o.x = (_Float16)out.x; o.y = (_Float16)out.y; o.z = (_Float16)out.z; o.w = (_Float16)outw;
            uint2 ob;
            __builtin_memcpy(&ob, &o, 8);
            store_hdr(p.hot.hdr, ((ty * 4u) * p.hot.W + tx * 16u) * 8u + laneHdr, ob.x, ob.y);
        }
        if (!more1) break;
        tile = tile1; tile1 = tile2; // (0xFFFFFFFF when nothing was left to claim)
        tx = tx1; ty = ty1; tx1 = tx2; ty1 = ty2;
        parity ^= 1u;
    }
#undef UR_PREFETCH_POINT
    // debug timeline: the workgroup's LAST wave to leave the loop stamps the exit (uniform branch). The pointer is re-read from the
    // kernarg segment: kept in SGPRs across the loop it was two of the loop's eight spilled scalars.
    unsigned long long* const tl = fresh_params()->timeline;
    if (tl != nullptr) {
        uint32_t left = 0;
        if (lane == 0) left = __hip_atomic_fetch_add(work + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ur::timeline_exit(tl, lane == 0 && left == WPB - 1u);
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

// One Lighting launch. With a pair of events waiting on the context (ur_time_next_lighting) the dispatch itself carries
// them (hipExtLaunchKernelGGL): their distance is the kernel's own begin -> end interval, no event record in the queue.
template <class K, class... Args>
void launch_timed(ur_ctx* ctx, K kern, dim3 grid, dim3 block, uint32_t lds, Args... args)
{
    if (ctx->time_stop != nullptr) {
        hipExtLaunchKernelGGL(kern, grid, block, lds, ctx->stream, ctx->time_start, ctx->time_stop, 0, args...);
        ctx->time_start = ctx->time_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, ctx->stream, args...);
    }
}

// Per-tile kernel (one 64x4-pixel workgroup per tile): partial tiles, sky-only launches and configurations the streaming
// kernel does not cover.
template <int MODE, bool SHADOWS>
void launch_tiled(ur_ctx* ctx, const LightingParams& p)
{
    constexpr int TW = 16; // pixels per wave = 16 x 4: 128-byte G-buffer row segments and compact gather footprints
    const uint32_t tilesX = (p.W + 4 * TW - 1) / (4 * TW), tilesY = (p.rows + (64 / TW) - 1) / (64 / TW);
    // register budget: waves/SIMD the kernel is compiled for (6 -> 80 VGPRs, the most that does not spill; 4 -> no cap)
    if (ctx->opt.tiled_waves >= 6) /* UR_OPT_LIGHTING_TILED_WAVES */ launch_timed(ctx, lighting_kernel<MODE, SHADOWS, TW, 6>, dim3(tilesX, tilesY), dim3(256), 0u, p);
    else launch_timed(ctx, lighting_kernel<MODE, SHADOWS, TW, 4>, dim3(tilesX, tilesY), dim3(256), 0u, p);
}


template <int MODE, bool SHADOWS, bool IRR_LDS, int WPB>
int launch_stream_wpb(ur_ctx* ctx, LightingParams p /* by value: the tile walk is filled in here */)
{
    typedef void (*kernel_t)(LightingParams, ur::HzbTail, HzbRide);
    constexpr uint32_t lds = kLdsTiles + WPB * 2u * kTileBytes + kLdsCubeBytes;
    p.timeline = ur::next_timeline_pair(ctx);
    // MaxDynamicSharedMemorySize is a per-DEVICE attribute of the function: one flag per instantiation and device
    static bool attr_set[2][64] = {};
    const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : -1;
    StreamHot& h = p.hot;
    h.tilesX = p.W / 16u;
    h.numTiles = h.tilesX * ((p.rows + 3u) / 4u);
    // A deferred HZB tail (ur_defer_hzb_tail) rides along as one extra 1024-thread workgroup on a CU of its own: the
    // lighting workgroups give up one CU (0.4 % of their throughput) and the frame saves a ~5 us single-workgroup launch.
    ur::HzbTail tail{};
    // UR_OPT_LIGHTING_LEAVE_CUS = n leaves n CUs to kernels of other streams (the graph's async-compute passes): the persistent
    // workgroups otherwise fill every CU's register file and nothing runs beside them. Never below one lighting workgroup
    // (a CPX partition reports 32 CUs), and the tail is carried only when that still leaves the lighting a CU of its own.
    const int leave_env = ctx->opt.leave_cus;
    const int cus = std::max(ctx->cu_count, 1);
    const int leave_cus = std::min(leave_env, cus - 1);
    const bool carry_tail = ctx->hzb_tail_pending && WPB == 16 && cus >= 16 && cus - leave_cus >= 2;
    HzbRide ride{};
    if (carry_tail) {
        tail = ctx->pending_tail;
        ctx->hzb_tail_pending = false;
        if (ctx->hzb_wide_pending) { // the whole chain rides: the lighting workgroups take the wide launch's pieces along
            ride.d = ctx->pending_wide;
            ride.grid_x = ctx->pending_wide_grid_x;
            ride.pieces = ctx->pending_wide_grid_x * ctx->pending_wide_grid_y;
            ride.done = ctx->hzb_done;
            ride.timed_out = ctx->hzb_timed_out_dev;
            ride.spin_limit = ctx->opt.debug_hzb_ride_stall != 0 ? (1u << 9) : (1u << 22);
            // walkers: the chain should be done within about a quarter of the shading (a piece is ~3 us of one wave's time, a
            // tile ~1.75 us): walkers >= 7 x pieces-per-workgroup / tiles-per-wave, rounded up to a power of two
            {
                const int forced = ctx->opt.ride_walkers; // UR_OPT_RIDE_WALKERS
                const uint32_t lighting_groups = std::max(1, cus - 1 - leave_cus);
                const double per_group = (double)ride.pieces / lighting_groups, tiles_per_wave = (double)h.numTiles / (lighting_groups * WPB);
                uint32_t wk = 1;
                while (wk < (uint32_t)WPB && (double)wk * tiles_per_wave < 7.0 * per_group) wk *= 2;
                if (forced >= 1) wk = (uint32_t)forced;
                ride.walkers = wk >= 4u ? (uint32_t)WPB : 1u; // two instantiations: the last wave alone, or all of them
            }
            ctx->hzb_wide_pending = false;
        }
    } else if (ctx->hzb_wide_pending && !ctx->hzb_tail_pending && WPB == 16 && cus >= 16) {
        // a band-sharded chain's pieces (ur_build_hzb_band) ride without a tail: the tail waits for the ranks' gather. Nothing inside
        // the launch consumes the pieces, so no arrival is signalled (done stays null) and no CU is set aside.
        ride.d = ctx->pending_wide;
        ride.grid_x = ctx->pending_wide_grid_x;
        ride.pieces = ctx->pending_wide_grid_x * ctx->pending_wide_grid_y;
        const uint32_t lighting_groups = std::max(1, cus - leave_cus);
        const double per_group = (double)ride.pieces / lighting_groups, tiles_per_wave = (double)h.numTiles / (lighting_groups * WPB);
        uint32_t wk = 1;
        while (wk < (uint32_t)WPB && (double)wk * tiles_per_wave < 7.0 * per_group) wk *= 2;
        if (ctx->opt.ride_walkers >= 1) wk = (uint32_t)ctx->opt.ride_walkers;
        ride.walkers = wk >= 4u ? (uint32_t)WPB : 1u;
        ctx->hzb_wide_pending = false;
    } else if (ctx->hzb_wide_pending) { // cannot ride (12-wave build, tiny device): the ordinary launches, in front
        const int frc = ur::flush_hzb_tail(ctx);
        if (frc != UR_OK) return frc;
    }
    const uint32_t groups = std::min<uint32_t>((uint32_t)std::max(1, cus - (carry_tail ? 1 : 0) - leave_cus), (h.numTiles + WPB - 1) / WPB);
    h.groups = groups;
    ride.want = groups + (ctx->opt.debug_hzb_ride_stall != 0 ? 1u : 0u);
    // tile / tilesX by multiplication: exact while (magic * tilesX - 2^32) * tile < 2^32 (checked by the caller)
    h.tilesXMagic = (uint32_t)((1ull << 32) / h.tilesX + 1ull);
    // ---- the run-time part of the schedule (struct Balance): whole rounds of the static deal in front, a pool of chunks behind
    h.staticClaims = 0xFFFFFFFFu;
    p.bal = Balance{};
    if (ctx->opt.balance != 0 && ctx->claim_words != nullptr && groups >= 16u && groups <= 8u * ur::kClaimWords) {
        constexpr uint32_t cs = kChunkShift;
        const uint32_t round = groups << cs;
        const uint32_t want_pool = (uint32_t)((uint64_t)h.numTiles * (uint32_t)ctx->opt.balance_pool_16ths / 16u);
        // Chunks of 16 tiles (one tile per wave of a workgroup) and nothing smaller by default: a claim blocks its wave for ~1 us, and
        // with chunks of 4 tiles a workgroup needs one every 0.5 us - measured, that LOSES 1.5 us at 1080p and 2.2 us on a 540-row
        // band of a 4K frame (profiles/r04_balance.txt). A launch too short for `lookahead + 2` such chunks per workgroup is dealt
        // statically as a whole. (UR_OPT_BALANCE_CHUNK_SHIFT below 4 exists for the tests, which drive the claim path hard with it.)
        do {
            const uint32_t sh = (uint32_t)ctx->opt.balance_chunk_shift;
            const uint32_t la = sh >= 4u ? 2u : (sh == 3u ? 3u : 4u); // chunks claimed ahead of use
            const uint32_t rounds = (h.numTiles - want_pool) / round;
            if ((rounds << cs) < 2u * (uint32_t)WPB) break; // (the two tiles of a wave's prologue are static claims)
            const uint32_t static_tiles = rounds * round, chunks = (h.numTiles - static_tiles + (1u << sh) - 1u) >> sh;
            if (chunks < (la + 2u) * groups) break;
            if ((uint64_t)chunks * 8u / groups + la + 8u > kDynSlots) break; // a workgroup's slot table would not hold its word's share
            const unsigned long long magic = (((unsigned long long)chunks << 32) + groups - 1u) / groups;
            bool ok = true;
            for (uint32_t q8 = 0; q8 < groups && ok; q8 += 8u) {
                const uint32_t nq = std::min(8u, groups - q8);
                const uint32_t P0 = (uint32_t)((q8 * magic) >> 32), P1 = (uint32_t)(((q8 + nq) * magic) >> 32);
                ok = P1 >= P0 + la * nq && P1 <= chunks;
            }
            if (!ok) break;
            h.staticClaims = rounds << cs;
            p.bal.poolChunks = chunks; p.bal.staticTiles = static_tiles; p.bal.dynShift = sh; p.bal.lookahead = la;
            p.bal.poolMagic = magic;
            p.bal.words = ctx->claim_words;
            p.bal.timedOut = ctx->claim_timed_out_dev;
        } while (false);
    }
    const bool ride_all = ride.pieces != 0u && ride.walkers > 1u;
    const kernel_t kern = ride_all ? static_cast<kernel_t>(lighting_stream_kernel<MODE, SHADOWS, IRR_LDS, WPB, true>)
                                   : static_cast<kernel_t>(lighting_stream_kernel<MODE, SHADOWS, IRR_LDS, WPB, false>);
    if (dev < 0 || !attr_set[ride_all][dev]) {
        UR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0) attr_set[ride_all][dev] = true;
    }
    {
        const uint32_t sched[8] = {groups, h.numTiles, p.bal.poolChunks != 0u ? p.bal.staticTiles : h.numTiles, p.bal.poolChunks, p.bal.dynShift, p.bal.lookahead,
                                   (uint32_t)WPB, ride.pieces};
        std::memcpy(ctx->last_schedule, sched, sizeof(sched));
    }
    launch_timed(ctx, kern, dim3(groups + (carry_tail ? 1u : 0u)), dim3(64 * WPB), lds, p, tail, ride);
    return UR_OK;
}

// One persistent workgroup per CU; waves per SIMD = WPB / 4. Measured at 4K (sustained clocks), round 1: 16 -> 76.6 us, 12 -> 80.6 us,
// 8 -> ~95 us; round 2: 16 -> 71.8 us, 12 -> 75.1 us (same box). Two workgroups of 10 waves per CU (96 VGPRs) spill.
template <int MODE, bool SHADOWS, bool IRR_LDS>
int launch_stream(ur_ctx* ctx, const LightingParams& p)
{
    if (ctx->opt.lighting_wpb == 12) /* UR_OPT_LIGHTING_WAVES_PER_WG */ return launch_stream_wpb<MODE, SHADOWS, IRR_LDS, 12>(ctx, p);
    return launch_stream_wpb<MODE, SHADOWS, IRR_LDS, 16>(ctx, p);
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
    float ortho_err = 0.0f; // departure of (float3x3)ViewInverse from an orthonormal matrix
    if (mode != UR_MODE_SKY) {
        // the view matrix must be rigid: rows of (float3x3)ViewInverse orthonormal
        const float* VI = S->ViewInverse;
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j) {
                const float d = VI[i * 4] * VI[j * 4] + VI[i * 4 + 1] * VI[j * 4 + 1] + VI[i * 4 + 2] * VI[j * 4 + 2];
                ortho_err = std::fmax(ortho_err, std::fabs(d - (i == j ? 1.0f : 0.0f)));
            }
        // Every camera the reference builds is rigid with CameraPosition as its origin (RendererUtils.cpp: View from LookTo, its
        // inverse, the same position). Anything else takes the per-tile kernel's literal world-space vectors.
        float cam_err = 0.0f;
        for (int j = 0; j < 3; ++j) {
            p.VIt[j] = VI[12 + j]; p.camPos[j] = S->CameraPosition[j];
            cam_err = std::fmax(cam_err, std::fabs(VI[12 + j] - S->CameraPosition[j]) / std::fmax(1.0f, std::fabs(VI[12 + j])));
        }
        p.general = (!(ortho_err <= 1e-3f) || !(cam_err <= 1e-5f)) ? 1u : 0u;
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
        p.shadowSmall = (shadows && (p.shadowWi < 3 || p.shadowHi < 3)) ? 1u : 0u; // per-tile kernel, every tap through the bordered PCF
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
        if (T->env_cube_texels != (uint64_t)ur_env_cube_texels(p.envBase, p.envMips)) {
            set_error("ur_lighting_tables.env_cube_texels = %llu, but this version's ur_stage_env_cube writes %llu texels for a %u^2 cube of %u mips: "
                      "the buffer was sized or staged for another layout", (unsigned long long)T->env_cube_texels,
                      (unsigned long long)ur_env_cube_texels(p.envBase, p.envMips), p.envBase, p.envMips);
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
    // ---- streaming kernel when the band is a whole number of 16-pixel tile columns; the per-tile kernel otherwise -----------
    const int use_stream = ctx->opt.lighting_stream; // UR_OPT_LIGHTING_STREAM
    bool streamed = false;
    // (the streaming kernel addresses the staged cube's RGB row-pair section in fp32 BYTE offsets, which must stay below 2^24: base
    // sizes up to 256; bigger cubes take the per-tile kernel)
    uint64_t env_texels = 0;
    if (mode != UR_MODE_SKY)
        for (uint32_t m = 0; m < p.envMips; ++m) {
            const uint64_t e = (uint64_t)std::max(1u, p.envBase >> m) + 2u;
            env_texels += 6u * e * e * 8u + 6u * e * (e - 1u) * 12u;
        }
    const uint64_t n_tiles = (uint64_t)(w / 16u) * ((rows + 3u) / 4u);
    const uint64_t magic_err = w >= 16u ? ((1ull << 32) / (w / 16u) + 1ull) * (w / 16u) - (1ull << 32) : 0;
    // (the streaming kernel takes its dot products in world space: the rotation must be orthonormal to rounding; its tile DMA
    // moves 16 bytes per lane: 16-byte-aligned band buffers)
    const uintptr_t align_bits = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                                 reinterpret_cast<uintptr_t>(p.depth) | reinterpret_cast<uintptr_t>(p.hdr);
    if (use_stream && mode != UR_MODE_SKY && (align_bits & 15u) == 0 && ortho_err <= 1e-5f && p.general == 0u && p.shadowSmall == 0u && w % 16u == 0 && w >= 32u /* the magic of one tile per row does not fit 32 bits */ && magic_err * n_tiles < (1ull << 32) && p.lutW == kLutW && p.lutH == kLutH && p.irrFrac == 0.0f && env_texels < (1ull << 24)) {
        bool ok = true;
        StreamHot& h = p.hot;
        if (shadows) {
            // orthographic light (BuildDirectionalLightViewProjection, RendererUtils.cpp:1117-1137): clip.w == 1, so
            // su * W - 0.5, sv * H - 0.5 and depth - bias are affine in viewZ * (ra, rb, 1)
            ok = p.SQ[3] == 0.0f && p.SQ[7] == 0.0f && p.SQ[11] == 0.0f && p.ST[3] == 1.0f;
            const double hw = 0.5 * p.shadowW, hh = 0.5 * p.shadowH;
            const double sc[3] = {hw, -hh, 1.0}; // clip -> (texel x, texel y, depth)
            for (int k = 0; k < 3; ++k) {
                // clip[k] = viewZ * (ra * SQ[k] + rb * SQ[4 + k] + SQ[8 + k]) + ST[k], ra = ndc.x / P11, rb = -ndc.y / P22
                h.shA[k] = (float)(p.SQ[k] * sc[k] * p.invP11);
                h.shB[k] = (float)(p.SQ[4 + k] * sc[k] * -(double)p.invP22);
                h.shC[k] = (float)(p.SQ[8 + k] * sc[k]);
            }
            h.shT[0] = (float)(p.ST[0] * hw + hw - 0.5);
            h.shT[1] = (float)(p.ST[1] * -hh + hh - 0.5);
            h.shT[2] = p.ST[2] - p.shadowBias;
            h.shadowXmax = p.shadowW - 0.5f;
            h.shadowYmax = p.shadowH - 0.5f;
            h.shadowWi = p.shadowWi; h.shadowHi = p.shadowHi;
            h.shadowWm3 = (float)(p.shadowWi - 3); h.shadowHm3 = (float)(p.shadowHi - 3); h.shadowWf = (float)p.shadowWi;
            h.shadowRowBytes = (uint32_t)p.shadowWi * 4u;
            h.shadowStrength = p.shadowStrength;
            h.shadowNegQuarterStrength = -0.25f * p.shadowStrength;
            h.shadow = p.shadow;
            ok = ok && (uint64_t)p.shadowWi * (uint64_t)p.shadowHi < (1ull << 24); // texel indices are computed in fp32 (exact below 2^24)
        }
        if (ok) {
            h.W = p.W; h.rows = p.rows; h.row0 = p.row0;
            h.invW2 = p.invW2; h.invH2 = p.invH2;
            h.invP11 = p.invP11; h.nInvP22 = -p.invP22;
            h.skyInvP11 = p.skyInvP11; h.nSkyInvP22 = -p.skyInvP22;
            h.skyNearOverR2 = p.skyNearOverR * p.skyNearOverR;
            h.maxMip = p.maxMip;
            h.envMaxLevel = (float)(p.envMips - 1u);
            const uint32_t iE = p.irrN0 + 2u;
            h.irrN0 = p.irrN0; h.irrNf = (float)p.irrN0; h.irrEf = (float)iE; h.irrEEf = (float)(iE * iE);
            if (p.irrN0 <= 2u) { h.irrEf = (float)(p.irrN0 + 1u); h.irrEEf = (float)((p.irrN0 + 1u) * (p.irrN0 + 1u)); } // LDS table of cells
            h.irrOfff = (float)p.irrOffset0; h.irrRowBytes = iE * 8u;
            h.env = p.env; h.hdr = p.hdr;
            for (int k = 0; k < 9; ++k) h.R[k] = p.R[k];
            for (int k = 0; k < 3; ++k) {
                h.Lw[k] = (p.L[0] * p.R[k] + p.L[1] * p.R[3 + k]) + p.L[2] * p.R[6 + k]; // view-space L rotated like every other vector
                h.WA[k] = p.invP11 * p.R[k];
                h.WB[k] = -p.invP22 * p.R[3 + k];
                h.WC[k] = p.R[6 + k];
                h.lightRGB[k] = p.lightRGB[k];
            }
            {   // largest sphere depth of the frame: (Near/R) * |(vx, vy, 1)| at the ndc corner, with a margin of a few ulp
                const double vx = p.skyInvP11, vy = p.skyInvP22;
                h.skyDepthMax = (float)(p.skyNearOverR * std::sqrt(vx * vx + vy * vy + 1.0) * (1.0 + 1e-5));
            }
            {   // the cube's smallest mips whose RGB row-pair entries fit the workgroup's LDS copy (the shipped cube: mips 4..8)
                uint32_t first = p.envMips;
                uint64_t bytes = 0, bordered = 0, before = 0;
                for (uint32_t m = p.envMips; m-- > 0;) {
                    const uint64_t e = (uint64_t)std::max(1u, p.envBase >> m) + 2u, b = 6u * e * (e - 1u) * 12u;
                    if (bytes + b > kLdsCubeBytes) break;
                    bytes += b;
                    first = m;
                }
                for (uint32_t m = 0; m < p.envMips; ++m) {
                    const uint64_t e = (uint64_t)std::max(1u, p.envBase >> m) + 2u;
                    bordered += 6u * e * e * 8u;
                    if (m < first) before += 6u * e * (e - 1u) * 12u;
                }
                h.cubeLdsLevel = first < p.envMips ? (float)first : 16.0f;
                h.cubeLdsBase = (uint32_t)(bordered + before);
                h.cubeLdsBytes = first < p.envMips ? (uint32_t)bytes : 0u;
            }
            streamed = true;
            const LightingParams& q = p;
            const bool irr_lds = p.irrN0 <= 2u;
            int rc;
            if (mode == UR_MODE_LIGHTING) {
                if (shadows) rc = irr_lds ? launch_stream<UR_MODE_LIGHTING, true, true>(ctx, q) : launch_stream<UR_MODE_LIGHTING, true, false>(ctx, q);
                else rc = irr_lds ? launch_stream<UR_MODE_LIGHTING, false, true>(ctx, q) : launch_stream<UR_MODE_LIGHTING, false, false>(ctx, q);
            } else {
                if (shadows) rc = irr_lds ? launch_stream<UR_MODE_FUSED, true, true>(ctx, q) : launch_stream<UR_MODE_FUSED, true, false>(ctx, q);
                else rc = irr_lds ? launch_stream<UR_MODE_FUSED, false, true>(ctx, q) : launch_stream<UR_MODE_FUSED, false, false>(ctx, q);
            }
            if (rc != UR_OK) return rc;
        }
    }
    if (!streamed) {
        // the per-tile kernel cannot carry a held-back HZB tail: it goes out on its own, in front (ur_defer_hzb_tail's contract:
        // every Lighting launch on the context completes the chain)
        const int frc = flush_hzb_tail(ctx);
        if (frc != UR_OK) return frc;
        switch (mode) {
        case UR_MODE_LIGHTING:
            if (shadows) launch_tiled<UR_MODE_LIGHTING, true>(ctx, p); else launch_tiled<UR_MODE_LIGHTING, false>(ctx, p);
            break;
        case UR_MODE_SKY: launch_tiled<UR_MODE_SKY, false>(ctx, p); break;
        default:
            if (shadows) launch_tiled<UR_MODE_FUSED, true>(ctx, p); else launch_tiled<UR_MODE_FUSED, false>(ctx, p);
            break;
        }
    }
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

} // namespace ur
