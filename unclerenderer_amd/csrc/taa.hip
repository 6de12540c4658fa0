// TemporalAA resolve for gfx950: 3x3 neighbourhood min/max of the current HDR frame, history clamped
// into that box, blended by HistoryWeight.
//
// Reference: Shaders/TemporalAA.hlsl:12-50 ([numthreads(8,8,1)], nine Texture2D.Load per pixel, neighbour coordinates
// clamped to the frame), pass Source/Render/DeferredRenderer.cpp:1308-1361. Here a 256-thread workgroup resolves a
// 64x8-pixel tile: the (64+2)x(8+2) RGBA16F neighbourhood is staged ONCE through LDS with coalesced 8-byte loads (the
// one place on this path where neighbouring pixels reuse each other's data), each lane then reads its 3x3 windows from
// LDS for two rows. HBM: 8 B current + 8 B history read, 8 B written per pixel (24 B/pixel; halo re-reads hit L2).
// Built with -ffp-contract=off: min/max/clamp are exact and lerp is a + t*(b-a) in fp32 with one RTE to fp16, so the
// output is bit-identical to the oracle.

#include "ur_internal.h"

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

constexpr int TW = 64, TH = 8;

struct TaaParams {
    const half4_t* current; // full frame
    const half4_t* history; // band
    half4_t* output;        // band
    uint32_t W, H, row0, rows;
    float weight;           // saturate(HistoryWeight)
    uint32_t use_history;
};

__global__ __launch_bounds__(256) void taa_kernel(TaaParams p)
{
    __shared__ half4_t tile[TH + 2][TW + 2];
    const int x0 = (int)blockIdx.x * TW, r0 = (int)blockIdx.y * TH; // tile origin: x in the frame, r inside the band
    const int maxx = (int)p.W - 1, maxy = (int)p.H - 1;
    if (p.use_history != 0) {
        for (int i = (int)threadIdx.x; i < (TH + 2) * (TW + 2); i += 256) {
            const int ty = i / (TW + 2), tx = i - ty * (TW + 2);
            const int gx = min(max(x0 + tx - 1, 0), maxx);
            const int gy = min(max((int)p.row0 + r0 + ty - 1, 0), maxy); // neighbour coordinates clamp to the FRAME (:38)
            tile[ty][tx] = p.current[(size_t)gy * p.W + gx];
        }
        __syncthreads();
    }
    const int lx = (int)(threadIdx.x & 63u), ly = (int)(threadIdx.x >> 6);
    const int px = x0 + lx;
    if (px > maxx) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int tr = ly * 2 + k;     // row inside the tile
        const int r = r0 + tr;         // row inside the band
        if (r >= (int)p.rows) continue;
        const size_t band_i = (size_t)r * p.W + px;
        if (p.use_history == 0) {
            p.output[band_i] = p.current[(size_t)(p.row0 + r) * p.W + px];
            continue;
        }
        const half4_t cur = tile[tr + 1][lx + 1];
        float mn[3] = {(float)cur.x, (float)cur.y, (float)cur.z}, mx[3] = {mn[0], mn[1], mn[2]};
#pragma unroll
        for (int oy = 0; oy < 3; ++oy)
#pragma unroll
            for (int ox = 0; ox < 3; ++ox) {
                const half4_t s = tile[tr + oy][lx + ox];
                mn[0] = fminf(mn[0], (float)s.x); mn[1] = fminf(mn[1], (float)s.y); mn[2] = fminf(mn[2], (float)s.z);
                mx[0] = fmaxf(mx[0], (float)s.x); mx[1] = fmaxf(mx[1], (float)s.y); mx[2] = fmaxf(mx[2], (float)s.z);
            }
        const half4_t h = p.history[band_i];
        const float c[3] = {(float)cur.x, (float)cur.y, (float)cur.z};
        const float hv[3] = {(float)h.x, (float)h.y, (float)h.z};
        half4_t o;
        float b[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float hc = fminf(fmaxf(hv[ch], mn[ch]), mx[ch]); // clamp(History, Min, Max)
            b[ch] = c[ch] + p.weight * (hc - c[ch]);              // lerp(Current, History, w)
        }
        o.x = (_Float16)b[0]; o.y = (_Float16)b[1]; o.z = (_Float16)b[2]; o.w = cur.w;
        p.output[band_i] = o;
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
    hipLaunchKernelGGL(taa_kernel, dim3((w + TW - 1) / TW, (rows + TH - 1) / TH), dim3(256), 0, ctx->stream, p);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}
