// ur_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar CPU restatement of the reference's GPU-driven visibility + deferred-shading path, one function
// per HLSL function, same names, same operation order, built with -O2 -ffp-contract=off (no fast-math).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (unclerenderer_amd/) never does.
//
// PARITY UNPINNED BY THE REFERENCE: /root/reference holds no test, golden image or known-answer vector for
// this path and its D3D12/DXC/Windows sources cannot be built or run here. This file is
// therefore pinned only by (a) hand-derived known answers in tests/test_oracle_*.py, (b) the internal
// cross-check GPU-form vs CPU-form frustum test (CullIndirectArgs.hlsl:24-41 vs RendererUtils.cpp:1192-1218),
// and (c) the committed fixtures under tests/golden/ that this file itself minted.
//
// Where the reference leaves arithmetic to fixed-function hardware or to the HLSL compiler, this file fixes
// one definition and both sides (oracle, HIP kernels) follow it:
//   * mul(float4(p,1), M): ((p.x*M[0][j] + p.y*M[1][j]) + p.z*M[2][j]) + M[3][j], no contraction;
//   * dot(a,b) for float3: (a.x*b.x + a.y*b.y) + a.z*b.z;
//   * floor(log2(x)) for x > 1: the IEEE exponent of x (exact; GPU log2 is approximate) — SURVEY.md H3;
//   * texture filtering: exact fp32 weights, seamless cube edges by folding the one-texel overshoot onto the
//     adjacent face (corner taps clamp the second coordinate first) — SURVEY.md H5;
//   * RGBA16F blend: fp32 add of (decoded dst + src), one round-to-nearest-even to fp16 — SURVEY.md H6;
//   * sky sphere: analytic sphere of radius World[0] instead of the 64x32 tessellation.

#include "../include/ur_hotpath.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <thread>
#include <vector>

namespace {

int g_threads = 1;

void parallel_rows(uint32_t rows, const std::function<void(uint32_t, uint32_t)>& fn)
{
    const int n = std::max(1, std::min<int>(g_threads, static_cast<int>(rows)));
    if (n == 1) {
        fn(0, rows);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(n);
    for (int t = 0; t < n; ++t) {
        const uint32_t r0 = static_cast<uint32_t>((uint64_t)rows * t / n);
        const uint32_t r1 = static_cast<uint32_t>((uint64_t)rows * (t + 1) / n);
        pool.emplace_back([=, &fn] { fn(r0, r1); });
    }
    for (auto& th : pool) th.join();
}

// ---- fp16 <-> fp32, bit-exact, round-to-nearest-even --------------------------------------------------
float h2f(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

uint16_t f2h(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) { // inf / nan
        return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? (0x200u | ((x >> 13) & 0x3FFu)) : 0u));
    }
    if (x >= 0x477FF000u) { // rounds to >= 65520 -> inf
        return (uint16_t)(sign | 0x7C00u);
    }
    if (x < 0x38800000u) { // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign; // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
        const int e = (int)(x >> 23);               // biased fp32 exponent, 102..112
        const uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;                  // 14..24 : result = m >> shift (units of 2^-24)
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (x - 0x38000000u) >> 13; // rebias exponent, truncate mantissa
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

// ---- HLSL intrinsics ----------------------------------------------------------------------------------
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

inline float saturate(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3 lerp(float3 a, float3 b, float t) { return {lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)}; }
inline float3 lerp(float3 a, float3 b, float3 t) { return {lerp(a.x, b.x, t.x), lerp(a.y, b.y, t.y), lerp(a.z, b.z, t.z)}; }
// normalize(v) = v * rsqrt(dot(v,v)); NaN for the zero vector, as in HLSL.
inline float3 normalize(float3 v)
{
    const float r = 1.0f / std::sqrt(dot(v, v));
    return {v.x * r, v.y * r, v.z * r};
}
inline float3 reflect(float3 i, float3 n) { return i - (2.0f * dot(n, i)) * n; }

// mul(float4(v, w), M).xyz / full, M row-major 4x4, row-vector convention.
inline float4 mul4(float3 v, float w, const float* M)
{
    float4 r;
    r.x = ((v.x * M[0] + v.y * M[4]) + v.z * M[8]) + w * M[12];
    r.y = ((v.x * M[1] + v.y * M[5]) + v.z * M[9]) + w * M[13];
    r.z = ((v.x * M[2] + v.y * M[6]) + v.z * M[10]) + w * M[14];
    r.w = ((v.x * M[3] + v.y * M[7]) + v.z * M[11]) + w * M[15];
    return r;
}
// mul(v, (float3x3)M)
inline float3 mul3(float3 v, const float* M)
{
    return {(v.x * M[0] + v.y * M[4]) + v.z * M[8], (v.x * M[1] + v.y * M[5]) + v.z * M[9],
            (v.x * M[2] + v.y * M[6]) + v.z * M[10]};
}

// =======================================================================================================
// BuildHZB — Shaders/BuildHZB.hlsl:34-126, one call == one Dispatch(ceil(W0/8), ceil(H0/8), 1).
// =======================================================================================================
struct HZBConstants { // BuildHZB.hlsl:5-18
    uint32_t SourceWidth, SourceHeight, DestWidth, DestHeight, DestWidth1, DestHeight1, DestWidth2, DestHeight2,
        DestWidth3, DestHeight3, SourceMip;
};

void BuildHZB_dispatch(const HZBConstants& C, const float* Source, float* Dest0, float* Dest1, float* Dest2,
                       float* Dest3, int MipsPerDispatch)
{
    const uint32_t GroupsX = (C.DestWidth + 7) / 8, GroupsY = (C.DestHeight + 7) / 8;
    auto SampleDepth = [&](uint32_t cx, uint32_t cy) { // BuildHZB.hlsl:34-39
        const uint32_t x = std::min(cx, C.SourceWidth - 1);
        const uint32_t y = std::min(cy, C.SourceHeight - 1);
        return Source[(size_t)y * C.SourceWidth + x];
    };
    for (uint32_t gy = 0; gy < GroupsY; ++gy)
        for (uint32_t gx = 0; gx < GroupsX; ++gx) {
            float SharedDepth[8][8], SharedDepth1[4][4], SharedDepth2[2][2];
            for (uint32_t ty = 0; ty < 8; ++ty)
                for (uint32_t tx = 0; tx < 8; ++tx) { // BuildHZB.hlsl:47-61
                    const uint32_t dx = gx * 8 + tx, dy = gy * 8 + ty;
                    float minDepth = 1.0f;
                    if (dx < C.DestWidth && dy < C.DestHeight) {
                        const uint32_t bx = dx * 2, by = dy * 2;
                        const float d0 = SampleDepth(bx, by), d1 = SampleDepth(bx + 1, by);
                        const float d2 = SampleDepth(bx, by + 1), d3 = SampleDepth(bx + 1, by + 1);
                        minDepth = std::fmin(std::fmin(d0, d1), std::fmin(d2, d3));
                        Dest0[(size_t)dy * C.DestWidth + dx] = minDepth;
                    }
                    SharedDepth[ty][tx] = minDepth;
                }
            if (MipsPerDispatch < 2) continue;
            for (uint32_t ty = 0; ty < 4; ++ty)
                for (uint32_t tx = 0; tx < 4; ++tx) { // BuildHZB.hlsl:64-83
                    const uint32_t x1 = gx * 4 + tx, y1 = gy * 4 + ty;
                    if (x1 < C.DestWidth1 && y1 < C.DestHeight1) {
                        const uint32_t bx = tx * 2, by = ty * 2;
                        const float m = std::fmin(std::fmin(SharedDepth[by][bx], SharedDepth[by][bx + 1]),
                                                  std::fmin(SharedDepth[by + 1][bx], SharedDepth[by + 1][bx + 1]));
                        Dest1[(size_t)y1 * C.DestWidth1 + x1] = m;
                        SharedDepth1[ty][tx] = m;
                    } else {
                        SharedDepth1[ty][tx] = 0.0f;
                    }
                }
            if (MipsPerDispatch < 3) continue;
            for (uint32_t ty = 0; ty < 2; ++ty)
                for (uint32_t tx = 0; tx < 2; ++tx) { // BuildHZB.hlsl:85-106
                    const uint32_t x2 = gx * 2 + tx, y2 = gy * 2 + ty;
                    if (x2 < C.DestWidth2 && y2 < C.DestHeight2) {
                        const uint32_t bx = tx * 2, by = ty * 2;
                        const float m = std::fmin(std::fmin(SharedDepth1[by][bx], SharedDepth1[by][bx + 1]),
                                                  std::fmin(SharedDepth1[by + 1][bx], SharedDepth1[by + 1][bx + 1]));
                        Dest2[(size_t)y2 * C.DestWidth2 + x2] = m;
                        SharedDepth2[ty][tx] = m;
                    } else {
                        SharedDepth2[ty][tx] = 0.0f;
                    }
                }
            if (MipsPerDispatch < 4) continue;
            if (gx < C.DestWidth3 && gy < C.DestHeight3) { // BuildHZB.hlsl:108-122
                Dest3[(size_t)gy * C.DestWidth3 + gx] =
                    std::fmin(std::fmin(SharedDepth2[0][0], SharedDepth2[0][1]), std::fmin(SharedDepth2[1][0], SharedDepth2[1][1]));
            }
        }
}

// =======================================================================================================
// CullIndirectArgs — Shaders/CullIndirectArgs.hlsl
// =======================================================================================================
struct CullingConstants { // CullIndirectArgs.hlsl:1-11 == 46 root constants, Renderer.cpp:411-429
    float4 FrustumPlanes[6];
    float ViewProjection[16];
    uint32_t ModelCount, HZBEnabled, HZBMipCount, HZBWidth, HZBHeight, DebugPrintEnabled;
};
static_assert(sizeof(CullingConstants) == 46 * 4, "46 dwords");

bool IsAabbVisible(const CullingConstants& C, float3 boundsMin, float3 boundsMax) // :24-41
{
    for (uint32_t i = 0; i < 6; ++i) {
        const float4 plane = C.FrustumPlanes[i];
        const float3 positiveVertex = {plane.x >= 0.0f ? boundsMax.x : boundsMin.x, plane.y >= 0.0f ? boundsMax.y : boundsMin.y,
                                       plane.z >= 0.0f ? boundsMax.z : boundsMin.z};
        if (dot(float3{plane.x, plane.y, plane.z}, positiveVertex) + plane.w < 0.0f) return false;
    }
    return true;
}

// RendererUtils::IsAabbInCameraFrustum (RendererUtils.cpp:1192-1218): XMVector4Dot(plane, (X,Y,Z,1)).
// XMVector4Dot sums x*x' + y*y' + z*z' + w*1 — as a 4-term dot; kept as a separate restatement so the
// test-suite can assert both forms agree.
bool IsAabbInCameraFrustum(const float4 Planes[6], float3 BoundsMin, float3 BoundsMax)
{
    for (int PlaneIndex = 0; PlaneIndex < 6; ++PlaneIndex) {
        const float4 P = Planes[PlaneIndex];
        const float X = P.x >= 0.0f ? BoundsMax.x : BoundsMin.x;
        const float Y = P.y >= 0.0f ? BoundsMax.y : BoundsMin.y;
        const float Z = P.z >= 0.0f ? BoundsMax.z : BoundsMin.z;
        const float d = ((P.x * X + P.y * Y) + P.z * Z) + P.w * 1.0f;
        if (d < 0.0f) return false;
    }
    return true;
}

inline uint32_t FloorLog2(float x) // x > 1, finite: IEEE exponent == floor(log2 x)
{
    uint32_t b;
    std::memcpy(&b, &x, 4);
    return ((b >> 23) & 0xFFu) - 127u;
}

bool IsOccluded(const CullingConstants& C, const float* hzb, const ur_mip_desc* mips, float3 boundsMin, float3 boundsMax) // :48-130
{
    if (C.HZBEnabled == 0 || C.HZBWidth == 0 || C.HZBHeight == 0 || C.HZBMipCount == 0) return false;
    const float3 corners[8] = {
        {boundsMin.x, boundsMin.y, boundsMin.z}, {boundsMax.x, boundsMin.y, boundsMin.z}, {boundsMin.x, boundsMax.y, boundsMin.z},
        {boundsMax.x, boundsMax.y, boundsMin.z}, {boundsMin.x, boundsMin.y, boundsMax.z}, {boundsMax.x, boundsMin.y, boundsMax.z},
        {boundsMin.x, boundsMax.y, boundsMax.z}, {boundsMax.x, boundsMax.y, boundsMax.z}};
    float2 minUv = {1.0f, 1.0f}, maxUv = {0.0f, 0.0f};
    float maxDepth = 0.0f;
    for (uint32_t i = 0; i < 8; ++i) {
        const float4 clip = mul4(corners[i], 1.0f, C.ViewProjection); // ProjectToClip :43-46
        if (clip.w <= 0.0f) return false;                             // anyBehind :77-81,93-96
        const float3 ndc = {clip.x / clip.w, clip.y / clip.w, clip.z / clip.w};
        float2 uv;
        uv.x = ndc.x * 0.5f + 0.5f;
        uv.y = 1 - (ndc.y * 0.5f + 0.5f);
        minUv = {std::fmin(minUv.x, uv.x), std::fmin(minUv.y, uv.y)};
        maxUv = {std::fmax(maxUv.x, uv.x), std::fmax(maxUv.y, uv.y)};
        maxDepth = std::fmax(maxDepth, ndc.z);
    }
    if (maxUv.x < 0.0f || maxUv.y < 0.0f || minUv.x > 1.0f || minUv.y > 1.0f) return false;
    minUv = {saturate(minUv.x), saturate(minUv.y)};
    maxUv = {saturate(maxUv.x), saturate(maxUv.y)};
    const float2 extent = {maxUv.x - minUv.x, maxUv.y - minUv.y};
    const float2 pixelSize = {extent.x * (float)C.HZBWidth, extent.y * (float)C.HZBHeight};
    const float maxDim = std::fmax(pixelSize.x, pixelSize.y);
    uint32_t mipLevel = 0;
    if (maxDim > 1.0f) {
        const float l = std::fmin(std::fmax((float)FloorLog2(maxDim), 0.0f), (float)(C.HZBMipCount - 1));
        mipLevel = (uint32_t)l;
    }
    const uint32_t mipWidth = std::max(1u, C.HZBWidth >> mipLevel);
    const uint32_t mipHeight = std::max(1u, C.HZBHeight >> mipLevel);
    uint32_t minX = (uint32_t)(minUv.x * (float)mipWidth), minY = (uint32_t)(minUv.y * (float)mipHeight);
    uint32_t maxX = (uint32_t)(maxUv.x * (float)mipWidth), maxY = (uint32_t)(maxUv.y * (float)mipHeight);
    minX = std::min(minX, mipWidth - 1);
    minY = std::min(minY, mipHeight - 1);
    maxX = std::min(maxX, mipWidth - 1);
    maxY = std::min(maxY, mipHeight - 1);
    // Texture2D.Load(int3(x, y, mip)). The D3D mip has exactly (mipWidth x mipHeight) == HZB sizing (a10).
    const ur_mip_desc& M = mips[mipLevel];
    auto Load = [&](uint32_t x, uint32_t y) { return hzb[M.offset + (size_t)y * M.width + x]; };
    float hzbDepth = 1.0f;
    hzbDepth = std::fmin(hzbDepth, Load(minX, minY));
    hzbDepth = std::fmin(hzbDepth, Load(maxX, minY));
    hzbDepth = std::fmin(hzbDepth, Load(minX, maxY));
    hzbDepth = std::fmin(hzbDepth, Load(maxX, maxY));
    return maxDepth < hzbDepth;
}

// =======================================================================================================
// Texture units in software — definitions shared with the kernels by specification only.
// =======================================================================================================
struct CubeFaceUV { int face; float u, v; };

// D3D cube addressing: +X,-X,+Y,-Y,+Z,-Z. Ties: z wins over y wins over x.
CubeFaceUV SelectCubeFace(float3 d)
{
    const float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    CubeFaceUV r;
    float ma, uc, vc;
    if (az >= ax && az >= ay) {
        r.face = d.z >= 0.0f ? 4 : 5;
        ma = az;
        uc = d.z >= 0.0f ? d.x : -d.x;
        vc = -d.y;
    } else if (ay >= ax) {
        r.face = d.y >= 0.0f ? 2 : 3;
        ma = ay;
        uc = d.x;
        vc = d.y >= 0.0f ? d.z : -d.z;
    } else {
        r.face = d.x >= 0.0f ? 0 : 1;
        ma = ax;
        uc = d.x >= 0.0f ? -d.z : d.z;
        vc = -d.y;
    }
    r.u = (uc / ma + 1.0f) * 0.5f;
    r.v = (vc / ma + 1.0f) * 0.5f;
    return r;
}

// Point on the unit cube for face coordinates (s,t) in [-1,1]^2 — inverse of SelectCubeFace.
void CubeFacePoint(int face, double s, double t, double p[3])
{
    switch (face) {
    case 0: p[0] = 1; p[1] = -t; p[2] = -s; break;
    case 1: p[0] = -1; p[1] = -t; p[2] = s; break;
    case 2: p[0] = s; p[1] = 1; p[2] = t; break;
    case 3: p[0] = s; p[1] = -1; p[2] = -t; break;
    case 4: p[0] = s; p[1] = -t; p[2] = 1; break;
    default: p[0] = -s; p[1] = -t; p[2] = -1; break;
    }
}

struct EnvCube { // DDS order: face-major, mips inner (TextureLoader.cpp:276-315)
    const ur_half4* texels;
    uint32_t base, mipCount;
    size_t faceStride; // texels per face (all mips)
    size_t mipOffset[16];
};

EnvCube MakeEnvCube(const ur_half4* texels, uint32_t base, uint32_t mipCount)
{
    EnvCube c{texels, base, mipCount, 0, {}};
    size_t off = 0;
    for (uint32_t m = 0; m < mipCount; ++m) {
        c.mipOffset[m] = off;
        const uint32_t n = std::max(1u, base >> m);
        off += (size_t)n * n;
    }
    c.faceStride = off;
    return c;
}

// Texel (i,j) of a face mip with i,j in [-1,N]: out-of-face taps fold onto the adjacent face.
float4 FetchCubeTexel(const EnvCube& c, uint32_t mip, int face, int i, int j)
{
    const int N = (int)std::max(1u, c.base >> mip);
    const bool iOut = i < 0 || i >= N, jOut = j < 0 || j >= N;
    if (iOut || jOut) {
        if (iOut && jOut) j = std::min(std::max(j, 0), N - 1); // corner: clamp the second coordinate first
        const double s = 2.0 * (i + 0.5) / N - 1.0, t = 2.0 * (j + 0.5) / N - 1.0;
        double p[3];
        CubeFacePoint(face, s, t, p);
        const int major = face >> 1; // axis index of the face normal
        const double over = (iOut ? std::fabs(s) : std::fabs(t)) - 1.0;
        for (int a = 0; a < 3; ++a) {
            if (a == major) p[a] *= (1.0 - over);
            else if (std::fabs(p[a]) > 1.0) p[a] = p[a] > 0 ? 1.0 : -1.0;
        }
        const CubeFaceUV f = SelectCubeFace(float3{(float)p[0], (float)p[1], (float)p[2]});
        face = f.face;
        i = std::min(std::max((int)std::floor(f.u * N), 0), N - 1);
        j = std::min(std::max((int)std::floor(f.v * N), 0), N - 1);
    }
    const ur_half4 h = c.texels[(size_t)face * c.faceStride + c.mipOffset[mip] + (size_t)j * N + i];
    return {h2f(h.x), h2f(h.y), h2f(h.z), h2f(h.w)};
}

float3 SampleCubeBilinear(const EnvCube& c, uint32_t mip, const CubeFaceUV& f)
{
    const float N = (float)std::max(1u, c.base >> mip);
    if (!(f.u == f.u) || !(f.v == f.v)) { // NaN direction (cleared G-buffer pixel): every filter weight is NaN
        const float n = std::numeric_limits<float>::quiet_NaN();
        return {n, n, n};
    }
    const float x = f.u * N - 0.5f, y = f.v * N - 0.5f;
    const float x0 = std::floor(x), y0 = std::floor(y);
    const float fx = x - x0, fy = y - y0;
    const int i0 = (int)x0, j0 = (int)y0;
    const float4 t00 = FetchCubeTexel(c, mip, f.face, i0, j0), t10 = FetchCubeTexel(c, mip, f.face, i0 + 1, j0);
    const float4 t01 = FetchCubeTexel(c, mip, f.face, i0, j0 + 1), t11 = FetchCubeTexel(c, mip, f.face, i0 + 1, j0 + 1);
    const float3 top = lerp(float3{t00.x, t00.y, t00.z}, float3{t10.x, t10.y, t10.z}, fx);
    const float3 bot = lerp(float3{t01.x, t01.y, t01.z}, float3{t11.x, t11.y, t11.z}, fx);
    return lerp(top, bot, fy);
}

// TextureCube.SampleLevel with MIN_MAG_MIP_LINEAR (DeferredRenderer.cpp:1735-1738)
float3 SampleCubeLevel(const EnvCube& c, float3 dir, float level)
{
    const float maxLevel = (float)(c.mipCount - 1);
    const float l = std::fmin(std::fmax(level, 0.0f), maxLevel);
    const float l0 = std::floor(l);
    const uint32_t m0 = (uint32_t)l0, m1 = std::min(m0 + 1, c.mipCount - 1);
    const float fl = l - l0;
    const CubeFaceUV f = SelectCubeFace(dir);
    const float3 c0 = SampleCubeBilinear(c, m0, f);
    if (fl == 0.0f || m1 == m0) return c0;
    const float3 c1 = SampleCubeBilinear(c, m1, f);
    return lerp(c0, c1, fl);
}

// Texture2D.Sample, RG16_UNORM, bilinear, clamp (BrdfLut with IblSampler)
float2 SampleLutBilinear(const uint16_t* lut, uint32_t W, uint32_t H, float u, float v)
{
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0 = std::floor(x), y0 = std::floor(y);
    const float fx = x - x0, fy = y - y0;
    auto cl = [](int a, int hi) { return std::min(std::max(a, 0), hi); };
    const int i0 = cl((int)x0, (int)W - 1), i1 = cl((int)x0 + 1, (int)W - 1);
    const int j0 = cl((int)y0, (int)H - 1), j1 = cl((int)y0 + 1, (int)H - 1);
    auto tex = [&](int i, int j) {
        const uint16_t* p = lut + 2 * ((size_t)j * W + i);
        return float2{(float)p[0] / 65535.0f, (float)p[1] / 65535.0f};
    };
    const float2 t00 = tex(i0, j0), t10 = tex(i1, j0), t01 = tex(i0, j1), t11 = tex(i1, j1);
    return {lerp(lerp(t00.x, t10.x, fx), lerp(t01.x, t11.x, fx), fy), lerp(lerp(t00.y, t10.y, fx), lerp(t01.y, t11.y, fx), fy)};
}

// How a comparison within 1e-5 of flipping is resolved: 0 = as computed (the reference's answer), +1 = every such tie
// passes, -1 = every such tie fails. The two forced modes bracket what ANY correctly rounded evaluation of the shadow
// coordinate can return for such a pixel (the shaded colour is monotone in each tap): the parity tests hold those pixels
// to that interval instead of skipping them (uro_set_shadow_tie_mode; tests/util.py:hdr_mismatch).
std::atomic<int> g_tie_mode{0};

// Texture2D.SampleCmpLevelZero, COMPARISON_MIN_MAG_LINEAR_MIP_POINT, BORDER opaque white, LESS_EQUAL
// (DeferredRenderer.cpp:1723-1728). `tie` is set when a comparison is within 1e-5 of flipping.
float SampleCmpLevelZero(const float* map, uint32_t W, uint32_t H, float u, float v, float cmp, bool* tie)
{
    const int tie_mode = g_tie_mode.load(std::memory_order_relaxed);
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0 = std::floor(x), y0 = std::floor(y);
    const float fx = x - x0, fy = y - y0;
    const int i0 = (int)x0, j0 = (int)y0;
    auto tap = [&](int i, int j) {
        const float t = (i < 0 || j < 0 || i >= (int)W || j >= (int)H) ? 1.0f : map[(size_t)j * W + i];
        if (std::fabs(cmp - t) <= 1e-5f) {
            *tie = true;
            if (tie_mode != 0) return tie_mode > 0 ? 1.0f : 0.0f;
        }
        return cmp <= t ? 1.0f : 0.0f;
    };
    const float r00 = tap(i0, j0), r10 = tap(i0 + 1, j0), r01 = tap(i0, j0 + 1), r11 = tap(i0 + 1, j0 + 1);
    return lerp(lerp(r00, r10, fx), lerp(r01, r11, fx), fy);
}

float g_srgb[256];
bool g_srgb_init = false;
void InitSrgb()
{
    if (g_srgb_init) return;
    for (int i = 0; i < 256; ++i) {
        const double c = i / 255.0;
        g_srgb[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    g_srgb_init = true;
}

// =======================================================================================================
// PBRCommon.hlsl:1-48
// =======================================================================================================
const float PI = 3.14159265f;

float DistributionGGX(float NdotH, float alpha)
{
    const float alpha2 = alpha * alpha;
    const float denom = (NdotH * NdotH) * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / std::fmax(PI * denom * denom, 1e-4f);
}
float GeometrySchlickGGX(float NdotX, float k) { return NdotX / (NdotX * (1.0f - k) + k); }
float3 FresnelSchlick(float VdotH, float3 F0)
{
    const float p = std::pow(1.0f - VdotH, 5.0f);
    return F0 + (float3{1.0f, 1.0f, 1.0f} - F0) * p;
}
float3 EvaluatePBR(float3 albedo, float metallic, float roughness, float3 F0, float3 N, float3 V, float3 L)
{
    const float3 H = normalize(V + L);
    const float NdotL = saturate(dot(N, L));
    const float NdotV = saturate(dot(N, V));
    const float NdotH = saturate(dot(N, H));
    const float VdotH = saturate(dot(V, H));
    const float alpha = roughness * roughness;
    const float D = DistributionGGX(NdotH, alpha);
    float k = (roughness + 1.0f);
    k = (k * k) / 8.0f;
    const float G = GeometrySchlickGGX(NdotV, k) * GeometrySchlickGGX(NdotL, k);
    const float3 F = FresnelSchlick(VdotH, F0);
    const float3 specular = ((D * G) * F) / std::fmax(4.0f * NdotL * NdotV, 1e-4f);
    const float3 kd = (float3{1.0f, 1.0f, 1.0f} - F) * (1.0f - metallic);
    const float3 diffuse = kd * albedo; // "/ PI" is commented out in the reference (PBRCommon.hlsl:45)
    return (diffuse + specular) * NdotL;
}

// =======================================================================================================
// DeferredLighting.hlsl:35-94 (PSMain) for pixel (px,py) of a W x H frame; returns (color, 1).
// =======================================================================================================
struct LightingInputs {
    const ur_scene_constants* S;
    const float* shadow;
    EnvCube env;
    const uint16_t* lut;
    uint32_t lutW, lutH;
};

float4 DeferredLighting_PSMain(const LightingInputs& in, uint32_t W, uint32_t H, uint32_t px, uint32_t py, ur_half4 a, ur_half4 b,
                               uint32_t c, bool* fragile)
{
    const ur_scene_constants& S = *in.S;
    // Fullscreen triangle: UV interpolates to the pixel centre; the point sampler then reads texel (px,py).
    const float2 UV = {((float)px + 0.5f) / (float)W, ((float)py + 0.5f) / (float)H};
    const float4 normalDepth = {h2f(a.x), h2f(a.y), h2f(a.z), h2f(a.w)};
    const float3 normal = normalize(float3{normalDepth.x, normalDepth.y, normalDepth.z});
    const float depth = normalDepth.w;
    const float4 smr = {h2f(b.x), h2f(b.y), h2f(b.z), h2f(b.w)};
    const float3 albedo = {g_srgb[c & 0xFF], g_srgb[(c >> 8) & 0xFF], g_srgb[(c >> 16) & 0xFF]};

    const float roughness = smr.z;
    const float metallic = smr.y;
    const float3 F0 = lerp(float3{smr.x, smr.x, smr.x}, albedo, metallic);

    const float2 ndc = {UV.x * 2.0f - 1.0f, UV.y * 2.0f - 1.0f};
    const float viewZ = -depth;
    const float viewX = ndc.x * viewZ / S.Projection[0];
    const float viewY = -ndc.y * viewZ / S.Projection[5];
    const float3 viewPos = {viewX, viewY, viewZ};

    const float3 V = normalize(-viewPos);
    const float4 Lv = mul4(float3{S.LightDirection[0], S.LightDirection[1], S.LightDirection[2]}, 0.0f, S.View);
    const float3 L = normalize(float3{Lv.x, Lv.y, Lv.z});

    const float4 wp4 = mul4(viewPos, 1.0f, S.ViewInverse);
    const float3 worldPos = {wp4.x, wp4.y, wp4.z};
    const float4 shadowPosition = mul4(worldPos, 1.0f, S.LightViewProjection);
    const float3 shadowCoord = {shadowPosition.x / shadowPosition.w, shadowPosition.y / shadowPosition.w, shadowPosition.z / shadowPosition.w};
    const float2 shadowUV = {shadowCoord.x * 0.5f + 0.5f, shadowCoord.y * -0.5f + 0.5f};
    const float shadowDepth = shadowCoord.z;
    float shadow = 1.0f;
    if (S.ShadowStrength > 0.0f && shadowUV.x >= 0.0f && shadowUV.y >= 0.0f && shadowUV.x <= 1.0f && shadowUV.y <= 1.0f) {
        const float2 shadowTexel = {1.0f / S.ShadowMapSize[0], 1.0f / S.ShadowMapSize[1]};
        const float shadowCompare = shadowDepth - S.ShadowBias;
        const uint32_t SW = (uint32_t)S.ShadowMapSize[0], SH = (uint32_t)S.ShadowMapSize[1];
        shadow = 0.25f * (((SampleCmpLevelZero(in.shadow, SW, SH, shadowUV.x, shadowUV.y, shadowCompare, fragile) +
                            SampleCmpLevelZero(in.shadow, SW, SH, shadowUV.x + shadowTexel.x, shadowUV.y, shadowCompare, fragile)) +
                           SampleCmpLevelZero(in.shadow, SW, SH, shadowUV.x, shadowUV.y + shadowTexel.y, shadowCompare, fragile)) +
                          SampleCmpLevelZero(in.shadow, SW, SH, shadowUV.x + shadowTexel.x, shadowUV.y + shadowTexel.y, shadowCompare, fragile));
        shadow = lerp(1.0f, shadow, S.ShadowStrength);
    }

    const float3 lightColor = {S.LightColor[0], S.LightColor[1], S.LightColor[2]};
    const float3 lighting = ((EvaluatePBR(albedo, metallic, roughness, F0, normal, V, L) * S.LightIntensity) * lightColor) * shadow;

    const float3 worldNormal = normalize(mul3(normal, S.ViewInverse));
    const float3 worldView = normalize(float3{S.CameraPosition[0], S.CameraPosition[1], S.CameraPosition[2]} - worldPos);
    const float3 reflection = reflect(-worldView, worldNormal);

    const float maxMip = std::fmax(0.0f, S.EnvMapMipCount - 1.0f);
    const float mipLevel = roughness * maxMip;
    const float3 prefilteredColor = SampleCubeLevel(in.env, reflection, mipLevel);

    const float NdotV = saturate(dot(worldNormal, worldView));
    const float2 brdf = SampleLutBilinear(in.lut, in.lutW, in.lutH, NdotV, roughness);
    const float3 specularIbl = prefilteredColor * (F0 * brdf.x + float3{brdf.y, brdf.y, brdf.y});

    const float3 irradiance = SampleCubeLevel(in.env, worldNormal, maxMip);
    const float3 diffuseIbl = (irradiance * albedo) * (1.0f - metallic);

    const float3 ambient = diffuseIbl + specularIbl;
    const float3 color = lighting + ambient;
    return {color.x, color.y, color.z, 1.0f};
}

// =======================================================================================================
// SkyAtmosphere.hlsl:40-101
// =======================================================================================================
float RayleighPhase(float cosTheta)
{
    const float k = 3.0f / (16.0f * 3.14159265f);
    return k * (1.0f + cosTheta * cosTheta);
}
float MiePhase(float cosTheta, float g)
{
    const float g2 = g * g;
    const float denom = std::pow(1.0f + g2 - 2.0f * g * cosTheta, 1.5f);
    return (1.0f - g2) / (4.0f * 3.14159265f * std::fmax(denom, 1e-3f));
}
float3 ApplyAtmosphere(const ur_sky_constants& K, float3 viewDir)
{
    const float horizonFalloff = saturate(std::pow(1.0f - saturate(viewDir.y * 0.5f + 0.5f), 3.0f));
    const float3 zenithColor = {0.05f, 0.12f, 0.22f};
    const float3 horizonColor = {0.52f, 0.68f, 0.86f};
    const float3 baseSky = lerp(zenithColor, horizonColor, horizonFalloff);

    const float3 up = {0.0f, 1.0f, 0.0f};
    const float3 Ln = normalize(float3{K.LightDirection[0], K.LightDirection[1], K.LightDirection[2]});
    const float cosSunView = dot(viewDir, Ln);
    const float cosSunUp = dot(Ln, up);

    const float rayleighScaleHeight = 8000.0f;
    const float mieScaleHeight = 1200.0f;
    const float viewHeight = std::fmax(0.0f, K.CameraPosition[1]);
    const float rayleighDensity = std::exp(-viewHeight / rayleighScaleHeight);
    const float mieDensity = std::exp(-viewHeight / mieScaleHeight);

    const float rayleighPhase = RayleighPhase(cosSunView);
    const float miePhase = MiePhase(cosSunView, 0.76f);

    const float3 rayleighColor = {0.650f, 0.570f, 0.475f};
    float3 scattered = (rayleighColor * rayleighDensity) * rayleighPhase;
    scattered = scattered + ((float3{K.LightColor[0], K.LightColor[1], K.LightColor[2]} * mieDensity) * miePhase) * 0.8f;

    const float sunAttenuation = saturate(std::exp(-std::fmax(0.0f, 1.0f - cosSunUp) * 2.0f));
    const float3 atmospheric = scattered * sunAttenuation;
    return baseSky + atmospheric;
}

// Camera ray through the centre of pixel (px,py): view-space (ndc.x/P11, ndc.y/P22, 1); world = v * R^-1
// where R^-1 = transpose of View's rotation block. Returns the sphere's depth along that ray.
float3 SkyViewDir(const ur_sky_constants& K, uint32_t W, uint32_t H, uint32_t px, uint32_t py, float* skyDepth)
{
    const float2 UV = {((float)px + 0.5f) / (float)W, ((float)py + 0.5f) / (float)H};
    const float3 v = {(UV.x * 2.0f - 1.0f) / K.Projection[0], (1.0f - UV.y * 2.0f) / K.Projection[5], 1.0f};
    const float3 w = {(v.x * K.View[0] + v.y * K.View[1]) + v.z * K.View[2], (v.x * K.View[4] + v.y * K.View[5]) + v.z * K.View[6],
                      (v.x * K.View[8] + v.y * K.View[9]) + v.z * K.View[10]};
    const float invLen = 1.0f / std::sqrt(dot(v, v));
    const float zView = K.World[0] * (v.z * invLen); // R * unit_dir.z
    *skyDepth = K.Projection[14] / zView;            // Near / z_view (Camera.cpp:33-47)
    return normalize(w);
}

inline ur_half4 BlendAdd(ur_half4 dst, float4 src)
{
    return {f2h(h2f(dst.x) + src.x), f2h(h2f(dst.y) + src.y), f2h(h2f(dst.z) + src.z), f2h(h2f(dst.w) + src.w)};
}

} // namespace

// =======================================================================================================
// C entry points (ctypes)
// =======================================================================================================
extern "C" {

void uro_set_threads(int n) { g_threads = std::max(1, n); }
int uro_hardware_threads() { return (int)std::max(1u, std::thread::hardware_concurrency()); }

uint16_t uro_f2h(float f) { return f2h(f); }
float uro_h2f(uint16_t h) { return h2f(h); }

// CreateHZBResources sizing, DeferredRenderer.cpp:2801-2835
uint32_t uro_hzb_layout(uint32_t w, uint32_t h, ur_mip_desc* mips, uint32_t* mip_count)
{
    uint32_t mw = std::max(1u, (w + 1) / 2), mh = std::max(1u, (h + 1) / 2);
    uint32_t n = 0, off = 0;
    for (;;) {
        mips[n] = {off, mw, mh};
        off += mw * mh;
        ++n;
        if (!(mw > 1 || mh > 1)) break;
        mw = std::max(1u, mw / 2);
        mh = std::max(1u, mh / 2);
    }
    *mip_count = n;
    return off;
}

// The dispatch loop of the "Build HZB" pass, DeferredRenderer.cpp:1046-1207.
void uro_build_hzb(const float* depth, uint32_t src_w, uint32_t src_h, float* hzb, const ur_mip_desc* mips, uint32_t MipCount)
{
    uint32_t CurrentWidth = mips[0].width, CurrentHeight = mips[0].height;
    uint32_t MipIndex = 0;
    while (MipIndex < MipCount) {
        const uint32_t MipsThisDispatch = std::min(4u, MipCount - MipIndex);
        const bool b2 = MipsThisDispatch > 1, b3 = MipsThisDispatch > 2, b4 = MipsThisDispatch > 3;
        HZBConstants C{};
        C.SourceWidth = (MipIndex == 0) ? src_w : std::max(1u, CurrentWidth);
        C.SourceHeight = (MipIndex == 0) ? src_h : std::max(1u, CurrentHeight);
        C.DestWidth = (MipIndex == 0) ? CurrentWidth : std::max(1u, CurrentWidth / 2);
        C.DestHeight = (MipIndex == 0) ? CurrentHeight : std::max(1u, CurrentHeight / 2);
        C.DestWidth1 = b2 ? std::max(1u, C.DestWidth / 2) : 0u;
        C.DestHeight1 = b2 ? std::max(1u, C.DestHeight / 2) : 0u;
        C.DestWidth2 = b3 ? std::max(1u, C.DestWidth1 / 2) : 0u;
        C.DestHeight2 = b3 ? std::max(1u, C.DestHeight1 / 2) : 0u;
        C.DestWidth3 = b4 ? std::max(1u, C.DestWidth2 / 2) : 0u;
        C.DestHeight3 = b4 ? std::max(1u, C.DestHeight2 / 2) : 0u;
        C.SourceMip = 0;
        const float* Source = (MipIndex == 0) ? depth : hzb + mips[MipIndex - 1].offset;
        float* D[4] = {nullptr, nullptr, nullptr, nullptr};
        for (uint32_t k = 0; k < MipsThisDispatch; ++k) D[k] = hzb + mips[MipIndex + k].offset;
        BuildHZB_dispatch(C, Source, D[0], D[1], D[2], D[3], (int)MipsThisDispatch);
        if (b4) { CurrentWidth = C.DestWidth3; CurrentHeight = C.DestHeight3; }
        else if (b3) { CurrentWidth = C.DestWidth2; CurrentHeight = C.DestHeight2; }
        else if (b2) { CurrentWidth = C.DestWidth1; CurrentHeight = C.DestHeight1; }
        else { CurrentWidth = C.DestWidth; CurrentHeight = C.DestHeight; }
        MipIndex += MipsThisDispatch;
    }
}

// CSMain, CullIndirectArgs.hlsl:132-167, for every index < ModelCount; then the derived ascending list.
void uro_cull_indirect_args(const uint32_t* constants, const ur_float4* bounds, const float* hzb, const ur_mip_desc* mips,
                            void* indirect_args, uint32_t* stats2, uint32_t* visible_idx, uint32_t* visible_count,
                            uint32_t index_base)
{
    CullingConstants C;
    std::memcpy(&C, constants, sizeof(C));
    uint8_t* args = static_cast<uint8_t*>(indirect_args);
    uint32_t nvis = 0;
    for (uint32_t index = 0; index < C.ModelCount; ++index) {
        const ur_float4 mn = bounds[index * 2], mx = bounds[index * 2 + 1];
        const float3 boundsMin = {mn.x, mn.y, mn.z}, boundsMax = {mx.x, mx.y, mx.z};
        const bool frustumVisible = IsAabbVisible(C, boundsMin, boundsMax);
        bool visible = frustumVisible, occluded = false;
        if (visible && C.HZBEnabled != 0) {
            occluded = IsOccluded(C, hzb, mips, boundsMin, boundsMax);
            visible = !occluded;
        }
        const uint32_t v = visible ? 1u : 0u;
        std::memcpy(args + (size_t)index * UR_INDIRECT_COMMAND_STRIDE + UR_INDIRECT_INSTANCE_COUNT_OFFSET, &v, 4);
        if (C.DebugPrintEnabled != 0 && !visible && stats2) {
            if (!frustumVisible) stats2[0] += 1;
            else if (occluded) stats2[1] += 1;
        }
        if (visible && visible_idx) visible_idx[nvis] = index + index_base;
        nvis += v;
    }
    if (visible_count) *visible_count = nvis;
}

// CPU-side frustum test of the reference (RendererUtils.cpp:830-843,1192-1218), for the cross-check.
void uro_cpu_frustum(const float* planes24, const ur_float4* bounds, uint32_t n, uint8_t* out)
{
    float4 P[6];
    std::memcpy(P, planes24, sizeof(P));
    for (uint32_t i = 0; i < n; ++i) {
        const ur_float4 mn = bounds[i * 2], mx = bounds[i * 2 + 1];
        out[i] = IsAabbInCameraFrustum(P, float3{mn.x, mn.y, mn.z}, float3{mx.x, mx.y, mx.z}) ? 1 : 0;
    }
}

void uro_set_shadow_tie_mode(int mode) { g_tie_mode.store(mode > 0 ? 1 : (mode < 0 ? -1 : 0)); }

// Lighting pass over band rows [row0,row0+rows) (band-local buffers), additive ONE/ONE blend.
// env_cube: DDS order, unbordered. fragile (nullable): per band pixel, 1 when a shadow compare is within
// 1e-5 of flipping (tests exclude those pixels from the strict tolerance and bound their fraction).
void uro_deferred_lighting(const ur_scene_constants* scene, const ur_half4* gbuf_a, const ur_half4* gbuf_b, const uint32_t* gbuf_c,
                           const float* shadow, const ur_half4* env_cube, uint32_t env_base, uint32_t env_mips,
                           const uint16_t* lut, uint32_t lut_w, uint32_t lut_h, ur_half4* hdr, uint32_t w, uint32_t h,
                           uint32_t row0, uint32_t rows, uint8_t* fragile)
{
    InitSrgb();
    LightingInputs in{scene, shadow, MakeEnvCube(env_cube, env_base, env_mips), lut, lut_w, lut_h};
    parallel_rows(rows, [&](uint32_t r0, uint32_t r1) {
        for (uint32_t r = r0; r < r1; ++r)
            for (uint32_t x = 0; x < w; ++x) {
                const size_t i = (size_t)r * w + x;
                bool fr = false;
                const float4 c = DeferredLighting_PSMain(in, w, h, x, row0 + r, gbuf_a[i], gbuf_b[i], gbuf_c[i], &fr);
                hdr[i] = BlendAdd(hdr[i], c);
                if (fragile) fragile[i] = fr ? 1 : 0;
            }
    });
}

// Sky pass over the band: write (sky,1) where sphere depth >= stored depth.
void uro_sky_atmosphere(const ur_sky_constants* sky, const float* depth, ur_half4* hdr, uint32_t w, uint32_t h, uint32_t row0,
                        uint32_t rows)
{
    parallel_rows(rows, [&](uint32_t r0, uint32_t r1) {
        for (uint32_t r = r0; r < r1; ++r)
            for (uint32_t x = 0; x < w; ++x) {
                const size_t i = (size_t)r * w + x;
                float skyDepth;
                const float3 viewDir = SkyViewDir(*sky, w, h, x, row0 + r, &skyDepth);
                if (!(skyDepth >= depth[i])) continue;
                const float3 c = ApplyAtmosphere(*sky, viewDir);
                hdr[i] = {f2h(c.x), f2h(c.y), f2h(c.z), f2h(1.0f)};
            }
    });
}

// Bordered-cube reference layout, computed from the folding rule: the FIRST section of what ur_stage_env_cube writes (the product
// appends the same faces as row pairs behind it and sizes the whole with ur_env_cube_texels; tests compare section by section).
size_t uro_env_cube_texels(uint32_t base, uint32_t mips)
{
    size_t n = 0;
    for (uint32_t m = 0; m < mips; ++m) {
        const size_t e = std::max(1u, base >> m) + 2;
        n += 6 * e * e;
    }
    return n;
}
void uro_stage_env_cube(const ur_half4* src, uint32_t base, uint32_t mips, ur_half4* dst)
{
    const EnvCube c = MakeEnvCube(src, base, mips);
    size_t off = 0;
    for (uint32_t m = 0; m < mips; ++m) {
        const int N = (int)std::max(1u, base >> m), E = N + 2;
        for (int f = 0; f < 6; ++f)
            for (int j = -1; j <= N; ++j)
                for (int i = -1; i <= N; ++i) {
                    const float4 t = FetchCubeTexel(c, m, f, i, j);
                    dst[off + ((size_t)f * E + (j + 1)) * E + (i + 1)] = {f2h(t.x), f2h(t.y), f2h(t.z), f2h(t.w)};
                }
        off += (size_t)6 * E * E;
    }
}

// Tonemap.hlsl:34-79 over a band. exposure_ev: pointer to the auto-exposure texel or null.
void uro_tonemap(const ur_tonemap_constants* K, const ur_half4* hdr, const float* exposure_ev, uint32_t* out, uint32_t count)
{
    float finalExposure = K->Exposure;
    if (K->EnableAutoExposure != 0 && exposure_ev) finalExposure *= std::exp2(exposure_ev[0]);
    const float invGamma = 1.0f / std::fmax(K->Gamma, 1e-3f);
    for (uint32_t i = 0; i < count; ++i) {
        float3 color = float3{h2f(hdr[i].x), h2f(hdr[i].y), h2f(hdr[i].z)} * finalExposure;
        if (K->EnableTonemap != 0) { // PBRNeutralToneMapping
            const float startCompression = 0.8f - 0.04f;
            const float desaturation = 0.15f;
            const float x = std::fmin(color.x, std::fmin(color.y, color.z));
            const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
            color = color - float3{offset, offset, offset};
            const float peak = std::fmax(color.x, std::fmax(color.y, color.z));
            if (!(peak < startCompression)) {
                const float d = 1.0f - startCompression;
                const float newPeak = 1.0f - d * d / (peak + d - startCompression);
                color = color * (newPeak / std::fmax(peak, 1e-4f));
                const float g = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
                color = lerp(color, float3{newPeak, newPeak, newPeak}, g);
            }
        }
        color = {saturate(color.x), saturate(color.y), saturate(color.z)};
        color = {std::pow(color.x, invGamma), std::pow(color.y, invGamma), std::pow(color.z, invGamma)};
        auto unorm8 = [](float v) { return (uint32_t)(saturate(v) * 255.0f + 0.5f); }; // D3D float -> UNORM: round to nearest
        out[i] = unorm8(color.x) | (unorm8(color.y) << 8) | (unorm8(color.z) << 16) | 0xFF000000u;
    }
}

// TemporalAA.hlsl:12-50 for band rows [row0,row0+rows); current = full frame, history/output band-local.
void uro_temporal_aa(const ur_half4* current, const ur_half4* history, ur_half4* output, float HistoryWeight, uint32_t UseHistory, uint32_t W,
                     uint32_t H, uint32_t row0, uint32_t rows)
{
    const float w = saturate(HistoryWeight);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t x = 0; x < W; ++x) {
            const int px = (int)x, py = (int)(row0 + r);
            const ur_half4 Current = current[(size_t)py * W + px];
            const size_t bi = (size_t)r * W + x;
            if (UseHistory == 0) { output[bi] = Current; continue; }
            float3 MinColor = {h2f(Current.x), h2f(Current.y), h2f(Current.z)}, MaxColor = MinColor;
            for (int oy = -1; oy <= 1; ++oy)
                for (int ox = -1; ox <= 1; ++ox) {
                    const int sx = std::min(std::max(px + ox, 0), (int)W - 1), sy = std::min(std::max(py + oy, 0), (int)H - 1);
                    const ur_half4 s = current[(size_t)sy * W + sx];
                    const float3 c = {h2f(s.x), h2f(s.y), h2f(s.z)};
                    MinColor = {std::fmin(MinColor.x, c.x), std::fmin(MinColor.y, c.y), std::fmin(MinColor.z, c.z)};
                    MaxColor = {std::fmax(MaxColor.x, c.x), std::fmax(MaxColor.y, c.y), std::fmax(MaxColor.z, c.z)};
                }
            const ur_half4 hh = history[bi];
            float3 Hist = {h2f(hh.x), h2f(hh.y), h2f(hh.z)};
            Hist = {std::fmin(std::fmax(Hist.x, MinColor.x), MaxColor.x), std::fmin(std::fmax(Hist.y, MinColor.y), MaxColor.y),
                    std::fmin(std::fmax(Hist.z, MinColor.z), MaxColor.z)};
            const float3 Cur = {h2f(Current.x), h2f(Current.y), h2f(Current.z)};
            const float3 Blended = lerp(Cur, Hist, w);
            output[bi] = {f2h(Blended.x), f2h(Blended.y), f2h(Blended.z), Current.w};
        }
}

// Point probes used by the hand-derived known-answer tests.
void uro_evaluate_pbr(const float* albedo, float metallic, float roughness, const float* F0, const float* N, const float* V,
                      const float* L, float* out3)
{
    const float3 r = EvaluatePBR({albedo[0], albedo[1], albedo[2]}, metallic, roughness, {F0[0], F0[1], F0[2]}, {N[0], N[1], N[2]},
                                 {V[0], V[1], V[2]}, {L[0], L[1], L[2]});
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void uro_apply_atmosphere(const ur_sky_constants* sky, const float* viewDir, float* out3)
{
    const float3 r = ApplyAtmosphere(*sky, {viewDir[0], viewDir[1], viewDir[2]});
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void uro_sample_cube_level(const ur_half4* env_cube, uint32_t base, uint32_t mips, const float* dir, float level, float* out3)
{
    const float3 r = SampleCubeLevel(MakeEnvCube(env_cube, base, mips), {dir[0], dir[1], dir[2]}, level);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void uro_select_cube_face(const float* dir, int* face, float* uv)
{
    const CubeFaceUV f = SelectCubeFace({dir[0], dir[1], dir[2]});
    *face = f.face; uv[0] = f.u; uv[1] = f.v;
}
float uro_sample_cmp(const float* map, uint32_t w, uint32_t h, float u, float v, float cmp)
{
    bool tie = false;
    return SampleCmpLevelZero(map, w, h, u, v, cmp, &tie);
}
void uro_sample_lut(const uint16_t* lut, uint32_t w, uint32_t h, float u, float v, float* out2)
{
    const float2 r = SampleLutBilinear(lut, w, h, u, v);
    out2[0] = r.x; out2[1] = r.y;
}
float uro_srgb_to_linear(uint32_t byte)
{
    InitSrgb();
    return g_srgb[byte & 0xFF];
}

} // extern "C"
