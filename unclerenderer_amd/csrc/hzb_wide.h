// The wide launch of the HZB chain (depth -> mips 0..4) as device code shared by two launch forms: `hzb_reduce4_kernel`
// (csrc/hzb.hip: one 256-thread workgroup per 128x32 source tile) and the streaming lighting kernel (csrc/lighting.hip), whose
// workgroups take the same 128x32 pieces along — ONE wave walks a piece in four passes of 128x8 source texels — when the
// frame driver lets the whole chain ride with the Lighting launch (ur_defer_hzb_tail(ctx, 2)). Not installed.
//
// Reference: Shaders/BuildHZB.hlsl:34-126, dispatch loop Source/Render/DeferredRenderer.cpp:1046-1207. Values are those of
// the reference's grouping (<= 4 mips per dispatch, out-of-range lanes 1.0 under the first mip of a dispatch and 0.0 under
// the next two; mip 4 = first level of the NEXT reference dispatch: clamped 2x2 footprints of mip 3); the only arithmetic
// is fminf, so the two forms give the same bits whatever order the lanes are walked in.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ur {

struct HzbDispatch {
    const float* src;
    float* dst[5];
    uint32_t SW, SH;
    uint32_t W[5], H[5];
    uint32_t mips; // 1..4 as the reference dispatches; 5 = also the first level of the NEXT reference dispatch
    uint32_t vec4_ok; // SW % 4 == 0 and src 16-byte aligned
    uint32_t pair_ok; // W[0] even and dst[0] 8-byte aligned
    uint32_t by0;     // first 128x32 piece row of this dispatch (0 for the whole frame; a rank's first piece row when the launch is
                      // band-sharded: ur_build_hzb_band)
};

__device__ __forceinline__ float hzbw_min4(float a, float b, float c, float d) { return fminf(fminf(a, b), fminf(c, d)); }

// One 128x32 source piece (bx, by) of a five-level dispatch (p.mips == 5), walked by ONE wave: pass q covers the rows the
// workgroup form gives to its wave q. sh2: 64 floats, sh3: 16 floats of LDS private to the wave. P: HzbDispatch in any
// address space (the lighting kernel reads it from its kernarg segment).
// HAND_OFF: mip 4 is read by another workgroup of the SAME launch (the riding tail): its texels are stored write-through
// (agent scope, `sc1`) so that a drained vmcnt plus one arrival is the whole producer side - no L2 write-back fence, which
// would have to sweep everything the lighting waves have dirtied (MI355X_MICROARCH.md, inter-workgroup visibility, row 1 of
// the measured sc1 hand-offs: 4-byte sc1 stores, one arrival per storing workgroup, sc1 loads behind the consumer's barrier).
template <bool HAND_OFF, class P>
__device__ __forceinline__ void hzb_wide_piece_by_one_wave(const P& p, uint32_t bx, uint32_t by, uint32_t lane, float* sh2, float* sh3)
{
    const uint32_t tx = lane & 31u, tyw = lane >> 5; // the lane's place in a 32x2 slice of 4x4 source blocks
    const uint32_t SW = p.SW, SH = p.SH, W0 = p.W[0], H0 = p.H[0], W1 = p.W[1], H1 = p.H[1], W2 = p.W[2], H2 = p.H[2];
    const float* __restrict__ src = p.src;
    const uint32_t x1 = bx * 32u + tx, sx = x1 * 4u;
    const bool vec_ok = p.vec4_ok != 0u, pair_ok = p.pair_ok != 0u;
    float s[4][4][4];
    // ---- every load of the piece first: one memory latency for its 16 KB
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t y1 = by * 8u + q * 2u + tyw, sy = y1 * 4u;
        const bool any0 = (x1 * 2u < W0) && (y1 * 2u < H0);
        if (any0) {
            if (vec_ok && sx + 3u < SW && sy + 3u < SH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(sy + r) * SW + sx);
                    s[q][r][0] = v.x; s[q][r][1] = v.y; s[q][r][2] = v.z; s[q][r][3] = v.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t yy = min(sy + r, SH - 1u);
#pragma unroll
                    for (int c = 0; c < 4; ++c) s[q][r][c] = src[(size_t)yy * SW + min(sx + c, SW - 1u)];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) s[q][r][c] = 1.0f;
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t ty = q * 2u + tyw, y1 = by * 8u + ty;
        // ---- mip 0: four texels (2x1+i, 2y1+j); out-of-range lanes hold 1.0 (BuildHZB.hlsl:47)
        float v0[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
        const bool any0 = (x1 * 2u < W0) && (y1 * 2u < H0);
        if (any0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t x0 = x1 * 2u + i, y0 = y1 * 2u + j;
                    if (x0 < W0 && y0 < H0) v0[j][i] = hzbw_min4(s[q][2 * j][2 * i], s[q][2 * j][2 * i + 1], s[q][2 * j + 1][2 * i], s[q][2 * j + 1][2 * i + 1]);
                }
            float* d0 = p.dst[0];
            const uint32_t x0 = x1 * 2u, y0 = y1 * 2u;
            if (pair_ok && x0 + 1u < W0) {
                *reinterpret_cast<float2*>(d0 + (size_t)y0 * W0 + x0) = make_float2(v0[0][0], v0[0][1]);
                if (y0 + 1u < H0) *reinterpret_cast<float2*>(d0 + (size_t)(y0 + 1u) * W0 + x0) = make_float2(v0[1][0], v0[1][1]);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (x0 + i < W0 && y0 + j < H0) d0[(size_t)(y0 + j) * W0 + x0 + i] = v0[j][i];
            }
        }
        // ---- mip 1: this lane's texel; out-of-range lanes hold 0.0 (:81)
        float v1 = 0.0f;
        if (x1 < W1 && y1 < H1) {
            v1 = hzbw_min4(v0[0][0], v0[0][1], v0[1][0], v0[1][1]);
            p.dst[1][(size_t)y1 * W1 + x1] = v1;
        }
        // ---- mip 2: the 2x2 of v1 lives in lanes {l, l^1, l^32, l^33}
        float m = fminf(v1, __shfl_xor(v1, 1));
        m = fminf(m, __shfl_xor(m, 32));
        const uint32_t x2 = x1 >> 1, y2 = y1 >> 1;
        float v2 = 0.0f; // out-of-range holds 0.0 (:104)
        if (x2 < W2 && y2 < H2) v2 = m;
        if (((tx | ty) & 1u) == 0u) {
            if (x2 < W2 && y2 < H2) p.dst[2][(size_t)y2 * W2 + x2] = v2;
            sh2[(ty >> 1) * 16u + (tx >> 1)] = v2;
        }
    }
    // ---- mip 3: 2x2 of v2 through the wave's LDS scratch (sixteen texels: 8 x 2)
    if (lane < 16u) {
        const uint32_t tx4 = lane & 7u, ty4 = lane >> 3;
        const uint32_t x3 = bx * 8u + tx4, y3 = by * 2u + ty4;
        if (x3 < p.W[3] && y3 < p.H[3]) {
            const uint32_t cx = tx4 * 2u, cy = ty4 * 2u;
            const float v3 = hzbw_min4(sh2[cy * 16u + cx], sh2[cy * 16u + cx + 1u], sh2[(cy + 1u) * 16u + cx], sh2[(cy + 1u) * 16u + cx + 1u]);
            p.dst[3][(size_t)y3 * p.W[3] + x3] = v3;
            sh3[ty4 * 8u + tx4] = v3;
        }
    }
    // ---- mip 4 = the first level of the reference's next dispatch: clamped 2x2 footprints of mip 3, all inside the piece
    if (lane < 4u) {
        const uint32_t x4 = bx * 4u + lane, y4 = by;
        if (x4 < p.W[4] && y4 < p.H[4]) {
            const uint32_t c0 = min(2u * x4, p.W[3] - 1u) & 7u, c1 = min(2u * x4 + 1u, p.W[3] - 1u) & 7u;
            const uint32_t r0 = min(2u * y4, p.H[3] - 1u) & 1u, r1 = min(2u * y4 + 1u, p.H[3] - 1u) & 1u;
            const float v4 = hzbw_min4(sh3[r0 * 8u + c0], sh3[r0 * 8u + c1], sh3[r1 * 8u + c0], sh3[r1 * 8u + c1]);
            float* d4 = p.dst[4] + (size_t)y4 * p.W[4] + x4;
            if (HAND_OFF) __hip_atomic_store(reinterpret_cast<uint32_t*>(d4), __float_as_uint(v4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *d4 = v4;
        }
    }
}

} // namespace ur
