// TemporalAA resolve for gfx950: 3x3 neighbourhood min/max of the current HDR frame, history clamped
// into that box, blended by HistoryWeight.
//
// Reference: Shaders/TemporalAA.hlsl:12-50 ([numthreads(8,8,1)], nine Texture2D.Load per pixel, neighbour coordinates
// clamped to the frame), pass Source/Render/DeferredRenderer.cpp:1308-1361. Here a wave owns a 64-column x 8-row strip
// and keeps it in registers (see taa_strip_kernel below): no LDS, every row access 512 contiguous aligned bytes per wave,
// 2 KB per workgroup. HBM: 8 B current + 8 B history read, 8 B written per pixel (24 B/pixel; the two halo rows of a
// strip are L2 hits while the neighbouring strip is in flight).
// Built with -ffp-contract=off: min/max/clamp are exact and lerp is a + t*(b-a) in fp32 with one RTE to fp16, so the
// output is bit-identical to the oracle.
// History of the shape (4K, 199 MB per launch; a plain 2-reads-1-write streaming kernel of that size takes 35.3 us,
// tools/microbench/stream_ceiling.hip): 64x8 tiles through LDS 43.8 us; 16-row strips walked four rows at a time 43.8 us
// (a chain of five dependent memory round trips per wave: 8 us of fixed cost); 8-row strips loaded in one phase 42.5 us
// (68 VGPRs: 7 waves/SIMD, the grid took 2.3 rounds = 3); the halo texels of a strip in ONE load, 56 VGPRs, 8 waves/SIMD,
// two rounds: 37.0 us = 5.4 TB/s.

#include "ur_internal.h"
#include "ur_device.h"

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

struct TaaParams {
    const half4_t* current; // full frame
    const half4_t* history; // band
    half4_t* output;        // band
    uint32_t W, H, row0, rows;
    float weight;           // saturate(HistoryWeight)
    uint32_t use_history;
};

// ---- a wave owns a 64-column x kStripRows strip ------------------------------------------------------------------------
// The 3x3 box is separable: per row the horizontal min/max of (left, centre, right), then the vertical min/max of three
// consecutive rows. Left/right come from the neighbouring lanes (DPP wave shifts, no LDS); the texels left of lane 0 and
// right of lane 63 are fetched for all rows of the strip by one load and broadcast with v_readlane. min/max run on the
// packed fp16 pairs (exact: no rounding, the same ordering as the fp32 compares of the reference), the clamp and the
// lerp in fp32 as in TemporalAA.hlsl:41-49.
constexpr int kStripRows = 8;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

struct RowMinMax { half2_t mn0, mn1, mx0, mx1; }; // (R,G) and (B,A) pairs

__device__ __forceinline__ half2_t as_h2(uint32_t u) { union { uint32_t u; half2_t h; } c; c.u = u; return c.h; }
__device__ __forceinline__ uint32_t as_u(half2_t h) { union { uint32_t u; half2_t h; } c; c.h = h; return c.u; }
__device__ __forceinline__ half2_t pk_min(half2_t a, half2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ half2_t pk_max(half2_t a, half2_t b) { return __builtin_elementwise_max(a, b); }

// horizontal (left, centre, right) min/max of one row; (hl0, hl1) / (hr0, hr1) = the texel left of lane 0 / right of lane 63
__device__ __forceinline__ RowMinMax row_minmax(u32x2_t c, uint32_t hl0, uint32_t hl1, uint32_t hr0, uint32_t hr1)
{
    // wave_shr:1 (0x138): lane l reads lane l-1, lane 0 keeps `old`; wave_shl:1 (0x130): lane l reads lane l+1, lane 63 keeps `old`
    const uint32_t l0 = __builtin_amdgcn_update_dpp(hl0, c.x, 0x138, 0xF, 0xF, false), l1 = __builtin_amdgcn_update_dpp(hl1, c.y, 0x138, 0xF, 0xF, false);
    const uint32_t r0 = __builtin_amdgcn_update_dpp(hr0, c.x, 0x130, 0xF, 0xF, false), r1 = __builtin_amdgcn_update_dpp(hr1, c.y, 0x130, 0xF, 0xF, false);
    RowMinMax m;
    m.mn0 = pk_min(pk_min(as_h2(l0), as_h2(r0)), as_h2(c.x)); m.mn1 = pk_min(pk_min(as_h2(l1), as_h2(r1)), as_h2(c.y));
    m.mx0 = pk_max(pk_max(as_h2(l0), as_h2(r0)), as_h2(c.x)); m.mx1 = pk_max(pk_max(as_h2(l1), as_h2(r1)), as_h2(c.y));
    return m;
}

__global__ __launch_bounds__(256) void taa_strip_kernel(TaaParams p)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // the four waves of a workgroup sit side by side: a workgroup touches 2 KB of contiguous bytes per row and buffer
    // (with the waves stacked vertically - 512-byte segments - the kernel ran at 4.5 TB/s, as did the LDS-tiled one with
    // its 64x8 tiles; a plain streaming kernel of the same byte mix reaches 5.6: tools/microbench/stream_ceiling.hip)
    const uint32_t x0 = (blockIdx.x * 4u + wave) * 64u;
    const uint32_t r0 = blockIdx.y * (uint32_t)kStripRows; // first row of the strip inside the band
    if (x0 >= p.W) return;                                 // uniform per wave (no barrier in this kernel)
    const uint32_t maxx = p.W - 1u, maxy = p.H - 1u;
    const uint32_t px = x0 + lane, cx = min(px, maxx);                    // lanes right of the frame re-read the last column
    // halo texels: ONE load for the whole strip - lane k holds the texel left of the strip in row k, lane 32 + k the one
    // right of it (k < kStripRows + 2); row k's pair is broadcast from there when the row is reduced
    const uint32_t hx = lane < 32u ? (x0 == 0u ? 0u : x0 - 1u) : min(x0 + 64u, maxx); // clamped to the frame (:38)
    const u32x2_t* cur = reinterpret_cast<const u32x2_t*>(p.current);
    const u32x2_t* his = reinterpret_cast<const u32x2_t*>(p.history);
    u32x2_t* out = reinterpret_cast<u32x2_t*>(p.output);

    auto frame_row = [&](int band_row) -> size_t { // band row (may be -1 or rows) -> clamped frame row offset in texels
        const int fr = (int)p.row0 + band_row;
        return (size_t)(uint32_t)min(max(fr, 0), (int)maxy) * p.W;
    };
    if (p.use_history == 0) { // UseHistory == 0: the resolve is a copy of the current frame (TemporalAA.hlsl:23-27)
        for (uint32_t k = 0; k < min((uint32_t)kStripRows, p.rows - r0); ++k)
            if (px <= maxx) out[(size_t)(r0 + k) * p.W + px] = cur[frame_row((int)(r0 + k)) + px];
        return;
    }
    // One load phase per wave: the strip's kStripRows + 2 current rows (with their halo texels) and kStripRows history
    // rows are all in flight before the first is used - a wave's life is one memory round trip, the arithmetic, the
    // stores. (Walking a taller strip group by group made every wave a chain of five dependent round trips: 8 us of
    // fixed cost per launch at any frame size.) Later workgroups of the grid load while earlier ones compute and store.
    u32x2_t c[kStripRows + 2], hist[kStripRows];
    const u32x2_t halo = cur[frame_row((int)r0 - 1 + (int)min(lane & 31u, (uint32_t)kStripRows + 1u)) + hx];
#pragma unroll
    for (int k = 0; k < kStripRows + 2; ++k) c[k] = cur[frame_row((int)r0 + k - 1) + cx];
#pragma unroll
    for (int k = 0; k < kStripRows; ++k) hist[k] = __builtin_nontemporal_load(his + (size_t)min(r0 + (uint32_t)k, p.rows - 1u) * p.W + cx); // read once
    const uint32_t nrows = min((uint32_t)kStripRows, p.rows - r0);
    auto reduce_row = [&](int k) {
        return row_minmax(c[k], __builtin_amdgcn_readlane(halo.x, k), __builtin_amdgcn_readlane(halo.y, k),
                          __builtin_amdgcn_readlane(halo.x, 32 + k), __builtin_amdgcn_readlane(halo.y, 32 + k));
    };
    RowMinMax mPrev = reduce_row(0), mCur = reduce_row(1);
#pragma unroll
    for (int k = 0; k < kStripRows; ++k) {
        const uint32_t r = r0 + (uint32_t)k;
        const RowMinMax mNext = reduce_row(k + 2);
        if ((uint32_t)k < nrows && px <= maxx) {
            const half2_t mn0 = pk_min(pk_min(mPrev.mn0, mNext.mn0), mCur.mn0), mn1 = pk_min(pk_min(mPrev.mn1, mNext.mn1), mCur.mn1);
            const half2_t mx0 = pk_max(pk_max(mPrev.mx0, mNext.mx0), mCur.mx0), mx1 = pk_max(pk_max(mPrev.mx1, mNext.mx1), mCur.mx1);
            const half2_t c0 = as_h2(c[k + 1].x), c1 = as_h2(c[k + 1].y), h0 = as_h2(hist[k].x), h1 = as_h2(hist[k].y);
            const float cf[3] = {(float)c0.x, (float)c0.y, (float)c1.x};
            const float hv[3] = {(float)h0.x, (float)h0.y, (float)h1.x};
            const float mn[3] = {(float)mn0.x, (float)mn0.y, (float)mn1.x}, mx[3] = {(float)mx0.x, (float)mx0.y, (float)mx1.x};
            float b[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float hc = fminf(fmaxf(hv[ch], mn[ch]), mx[ch]); // clamp(History, Min, Max)
                b[ch] = cf[ch] + p.weight * (hc - cf[ch]);            // lerp(Current, History, w)
            }
            half2_t o0, o1;
            o0.x = (_Float16)b[0]; o0.y = (_Float16)b[1]; o1.x = (_Float16)b[2]; o1.y = c1.y; // alpha of the current texel
            ur::store_once_b64(out + (size_t)r * p.W + px, ur::once_u32x2_t{as_u(o0), as_u(o1)});
        }
        mPrev = mCur; mCur = mNext;
    }
}

} // namespace

extern "C" int ur_temporal_aa(ur_ctx* ctx, const ur_half4* current_frame, const ur_half4* history_band, ur_half4* output_band, float history_weight,
                              uint32_t use_history, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows)
{
    if (!ctx || !current_frame || !output_band || (use_history && !history_band)) { ur::set_error("ur_temporal_aa: null argument"); return UR_EINVAL; }
    if (w == 0 || h == 0 || (uint64_t)row0 + rows > h) { ur::set_error("ur_temporal_aa: bad frame/band"); return UR_EINVAL; }
    if (rows == 0) return UR_OK;
    TaaParams p{};
    p.current = reinterpret_cast<const half4_t*>(current_frame);
    p.history = reinterpret_cast<const half4_t*>(history_band);
    p.output = reinterpret_cast<half4_t*>(output_band);
    p.W = w; p.H = h; p.row0 = row0; p.rows = rows;
    p.weight = history_weight < 0.0f ? 0.0f : (history_weight > 1.0f ? 1.0f : history_weight);
    if (!(history_weight == history_weight)) p.weight = 0.0f; // saturate(NaN) = 0
    p.use_history = use_history ? 1u : 0u;
    hipLaunchKernelGGL(taa_strip_kernel, dim3((w + 255u) / 256u, (rows + kStripRows - 1u) / kStripRows), dim3(256), 0, ctx->stream, p);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}
