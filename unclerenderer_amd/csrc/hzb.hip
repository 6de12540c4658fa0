// BuildHZB for gfx950 — min-depth mip chain (reverse-Z: min == farthest).
//
// Reference: Shaders/BuildHZB.hlsl:34-126 (8x8 groups, <=4 mips per dispatch through groupshared tiles) and the
// dispatch loop Source/Render/DeferredRenderer.cpp:1046-1207. The values produced here are bit-identical to that
// chain, including its out-of-range fills (1.0 below the first mip of a dispatch, 0.0 below the second and third —
// BuildHZB.hlsl:47,81,104). What is NOT kept is the reference's launch shape: one workgroup here is
// four wave64s covering a 128x32 source tile; each lane owns a 4x4 source block in registers (four 16-byte loads,
// 512 contiguous bytes per wave row), so mips k and k+1 never touch LDS, mip k+2 is two DPP/shuffle steps inside the
// wave (lane^1, lane^32) and only mip k+3 crosses waves through a 64-float LDS tile. One barrier per workgroup instead
// of three; HBM traffic == algorithmic bytes (every source texel read once, every mip texel written once).
//
// Built with -ffp-contract=off; the only arithmetic is fminf.

#include "ur_internal.h"
#include "ur_device.h"
#include "hzb_tail.h"
#include "hzb_wide.h"

namespace {

using ur::HzbDispatch;

__device__ __forceinline__ float min4(float a, float b, float c, float d) { return fminf(fminf(a, b), fminf(c, d)); }

__global__ __launch_bounds__(256) void hzb_reduce4_kernel(HzbDispatch p)
{
    __shared__ float sh2[4][16];
    __shared__ float sh3[2][8];

    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
    const uint32_t by = blockIdx.y + p.by0; // (a band-sharded launch starts at its rank's first piece row)
    const uint32_t x1 = blockIdx.x * 32u + tx, y1 = by * 8u + ty; // coords in mip k+1 (== 4x4 source block index)
    const uint32_t sx = x1 * 4u, sy = y1 * 4u;

    // ---- mip k: four texels (2x1+i, 2y1+j) from the 4x4 source block, clamped reads (SampleDepth, :34-39)
    float v0[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}}; // out-of-range lanes hold 1.0 (:47)
    const bool any0 = (x1 * 2u < p.W[0]) && (y1 * 2u < p.H[0]);
    if (any0) {
        float s[4][4];
        if (p.vec4_ok && sx + 3u < p.SW && sy + 3u < p.SH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p.src + (size_t)(sy + r) * p.SW + sx)); // read once
                s[r][0] = q.x; s[r][1] = q.y; s[r][2] = q.z; s[r][3] = q.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t yy = min(sy + r, p.SH - 1u);
#pragma unroll
                for (int c = 0; c < 4; ++c) s[r][c] = p.src[(size_t)yy * p.SW + min(sx + c, p.SW - 1u)];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t x0 = x1 * 2u + i, y0 = y1 * 2u + j;
                if (x0 < p.W[0] && y0 < p.H[0]) v0[j][i] = min4(s[2 * j][2 * i], s[2 * j][2 * i + 1], s[2 * j + 1][2 * i], s[2 * j + 1][2 * i + 1]);
            }
        float* d0 = p.dst[0];
        const uint32_t x0 = x1 * 2u, y0 = y1 * 2u;
        if (p.pair_ok && x0 + 1u < p.W[0]) { // 8-byte aligned pair
            ur::store_once_b64(d0 + (size_t)y0 * p.W[0] + x0, ur::once_u32x2_t{__float_as_uint(v0[0][0]), __float_as_uint(v0[0][1])}); // written once, read by a later launch
            if (y0 + 1u < p.H[0]) ur::store_once_b64(d0 + (size_t)(y0 + 1u) * p.W[0] + x0, ur::once_u32x2_t{__float_as_uint(v0[1][0]), __float_as_uint(v0[1][1])});
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (x0 + i < p.W[0] && y0 + j < p.H[0]) d0[(size_t)(y0 + j) * p.W[0] + x0 + i] = v0[j][i];
        }
    }
    if (p.mips < 2u) return;

    // ---- mip k+1: this lane's texel; out-of-range lanes hold 0.0 (:81)
    float v1 = 0.0f;
    if (x1 < p.W[1] && y1 < p.H[1]) {
        v1 = min4(v0[0][0], v0[0][1], v0[1][0], v0[1][1]);
        p.dst[1][(size_t)y1 * p.W[1] + x1] = v1;
    }
    if (p.mips < 3u) return; // uniform

    // ---- mip k+2: 2x2 of v1 lives in lanes {l, l^1, l^32, l^33} of this wave (wave = rows 2w, 2w+1 of the tile)
    float m = fminf(v1, __shfl_xor(v1, 1));
    m = fminf(m, __shfl_xor(m, 32));
    const uint32_t x2 = x1 >> 1, y2 = y1 >> 1;
    float v2 = 0.0f; // out-of-range holds 0.0 (:104)
    if (x2 < p.W[2] && y2 < p.H[2]) v2 = m;
    if (((tx | ty) & 1u) == 0u) {
        if (x2 < p.W[2] && y2 < p.H[2]) p.dst[2][(size_t)y2 * p.W[2] + x2] = v2;
        sh2[ty >> 1][tx >> 1] = v2;
    }
    if (p.mips < 4u) return; // uniform
    __syncthreads();

    // ---- mip k+3: 2x2 of v2 through LDS
    if (((tx | ty) & 3u) == 0u) {
        const uint32_t x3 = x1 >> 2, y3 = y1 >> 2;
        if (x3 < p.W[3] && y3 < p.H[3]) {
            const uint32_t cx = tx >> 1, cy = ty >> 1;
            const float v3 = min4(sh2[cy][cx], sh2[cy][cx + 1], sh2[cy + 1][cx], sh2[cy + 1][cx + 1]);
            p.dst[3][(size_t)y3 * p.W[3] + x3] = v3;
            sh3[ty >> 2][tx >> 2] = v3;
        }
    }
    if (p.mips < 5u) return; // uniform
    __syncthreads();

    // ---- mip k+4 = the FIRST level of the reference's next dispatch, produced here so that the single-workgroup tail
    //      launch starts from a mip a quarter the size: clamped 2x2 footprints of mip k+3 (SampleDepth, :34-39). The
    //      workgroup's 8x2 texels of mip k+3 hold every tap: a clamped coordinate of an in-range texel stays in its pair.
    if (threadIdx.x < 4u) {
        const uint32_t x4 = blockIdx.x * 4u + threadIdx.x, y4 = by;
        if (x4 < p.W[4] && y4 < p.H[4]) {
            const uint32_t c0 = min(2u * x4, p.W[3] - 1u) & 7u, c1 = min(2u * x4 + 1u, p.W[3] - 1u) & 7u;
            const uint32_t r0 = min(2u * y4, p.H[3] - 1u) & 1u, r1 = min(2u * y4 + 1u, p.H[3] - 1u) & 1u;
            p.dst[4][(size_t)y4 * p.W[4] + x4] = min4(sh3[r0][c0], sh3[r0][c1], sh3[r1][c0], sh3[r1][c1]);
        }
    }
}

__global__ __launch_bounds__(1024) void hzb_tail_kernel(ur::HzbTail p)
{
    __shared__ float bufA[ur::kTailTexels], bufB[ur::kTailTexels / 2];
    ur::hzb_tail_run(p, bufA, bufB);
}

} // namespace

namespace ur {

static HzbTail make_tail(float* hzb, const ur_mip_desc* mips, uint32_t mip_count, uint32_t mip);

int launch_build_hzb(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb, const ur_mip_desc* mips,
                     uint32_t mip_count)
{
    // Same grouping as the reference's while-loop (DeferredRenderer.cpp:1046-1207): <=4 mips per launch, first launch
    // reads the depth buffer with clamped 2x2 footprints, later launches read the last mip of the previous launch —
    // until the remaining levels fit one workgroup's LDS: those run in a single launch with the same values.
    {
        const int rc = flush_hzb_tail(ctx); // an earlier chain's tail must not run after this chain's levels
        if (rc != UR_OK) return rc;
    }
    uint32_t mip = 0;
    while (mip < mip_count) {
        if (mip > 0 && (uint64_t)mips[mip].width * mips[mip].height <= kTailTexels && mip_count - mip <= kTailMaxLevels) {
            const HzbTail t = make_tail(hzb, mips, mip_count, mip);
            if (ctx->defer_hzb_tail) { // the next streaming Lighting launch takes it along (lighting.hip); ur_flush otherwise
                ctx->pending_tail = t;
                ctx->hzb_tail_pending = true;
                break;
            }
            hipLaunchKernelGGL(hzb_tail_kernel, dim3(1), dim3(1024), 0, ctx->stream, t);
            UR_HIP_TRY(hipGetLastError());
            break;
        }
        uint32_t n = (mip_count - mip) < 4u ? (mip_count - mip) : 4u;
        // the first launch also produces mip 4 when a tail launch follows: the tail then starts from 1/4 of the texels
        // (a single workgroup reads ~25 GB/s: 130 KB of mip 3 at 4K would be 5 us on its own)
        if (mip == 0 && mip_count > 4u && (uint64_t)mips[4].width * mips[4].height <= kTailTexels) n = 5u;
        // What has to fit the tail's LDS is ITS first level, mip 5; its parent, mip 4, is read from global memory: at 8K 130 KB
        // through one workgroup. With 4-byte taps that was a 9-us tail (slower than a third launch); with the 16-byte loads of
        // tail_first_level_vec it is two launches for every chain up to 8K.
        else if (mip == 0 && mip_count > 5u && (uint64_t)mips[5].width * mips[5].height <= kTailTexels &&
                 mip_count - 5u <= kTailMaxLevels) n = 5u;
        HzbDispatch d{};
        if (mip == 0) {
            d.src = depth;
            d.SW = src_w;
            d.SH = src_h;
        } else {
            d.src = hzb + mips[mip - 1].offset;
            d.SW = mips[mip - 1].width;
            d.SH = mips[mip - 1].height;
        }
        for (uint32_t k = 0; k < 5; ++k) {
            if (k < n) {
                d.dst[k] = hzb + mips[mip + k].offset;
                d.W[k] = mips[mip + k].width;
                d.H[k] = mips[mip + k].height;
            } else {
                d.dst[k] = nullptr;
                d.W[k] = 0;
                d.H[k] = 0;
            }
        }
        d.mips = n;
        d.vec4_ok = ((d.SW & 3u) == 0u && (reinterpret_cast<uintptr_t>(d.src) & 15u) == 0u) ? 1u : 0u;
        d.pair_ok = ((d.W[0] & 1u) == 0u && (reinterpret_cast<uintptr_t>(d.dst[0]) & 7u) == 0u) ? 1u : 0u;
        const dim3 grid((d.W[0] + 63u) / 64u, (d.H[0] + 15u) / 16u);
        // ur_defer_hzb_tail(ctx, 2): a chain that is ONE five-level launch from the depth buffer plus the single-workgroup tail
        // (1080p, 4K and 8K all are) is held back as a whole: the next streaming Lighting launch takes its 128x32 pieces
        // along (lighting.hip), ur_flush / a cull / another build launch it the ordinary way
        if (mip == 0 && n == 5u && ctx->defer_hzb_tail && ctx->defer_hzb_wide && mip_count > 5u &&
            (uint64_t)mips[5].width * mips[5].height <= kTailTexels && mip_count - 5u <= kTailMaxLevels && ctx->hzb_done != nullptr) {
            ctx->pending_wide = d;
            ctx->pending_wide_grid_x = grid.x;
            ctx->pending_wide_grid_y = grid.y;
            ctx->hzb_wide_pending = true;
            mip += n;
            continue;
        }
        hipLaunchKernelGGL(hzb_reduce4_kernel, grid, dim3(256), 0, ctx->stream, d);
        UR_HIP_TRY(hipGetLastError());
        mip += n;
    }
    return UR_OK;
}

// ---- band-sharded chain (multi-GPU, SURVEY.md section 8e row 3's alternative): a rank builds mips 0..4 for the 128x32 source pieces
// whose first row lies in its band - every value of those levels depends on its own piece only, so the slices are the whole-frame
// launch's bits -, the slices are all-gathered by the host, and the single-workgroup tail (mips 5..) runs on every rank behind it.
static bool chain_is_wide_plus_tail(const ur_mip_desc* mips, uint32_t mip_count)
{
    return mip_count > 5u && (uint64_t)mips[5].width * mips[5].height <= kTailTexels && mip_count - 5u <= kTailMaxLevels;
}

static HzbTail make_tail(float* hzb, const ur_mip_desc* mips, uint32_t mip_count, uint32_t mip)
{
    HzbTail t{};
    t.src = hzb + mips[mip - 1].offset;
    t.SW = mips[mip - 1].width;
    t.SH = mips[mip - 1].height;
    t.first_mip = mip;
    t.levels = mip_count - mip;
    for (uint32_t k = 0; k < t.levels; ++k) {
        t.dst[k] = hzb + mips[mip + k].offset;
        t.W[k] = mips[mip + k].width;
        t.H[k] = mips[mip + k].height;
        t.magic[k] = t.W[k] > 1u ? (uint32_t)((1ull << 32) / t.W[k] + 1ull) : 0u; // W == 1: y = i (handled in the kernel)
    }
    return t;
}

int launch_build_hzb_band(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb, const ur_mip_desc* mips, uint32_t mip_count,
                          uint32_t piece_row0, uint32_t piece_rows)
{
    {
        const int rc = flush_hzb_tail(ctx);
        if (rc != UR_OK) return rc;
    }
    if (!chain_is_wide_plus_tail(mips, mip_count)) {
        set_error("ur_build_hzb_band: a %u x %u frame's chain is not one five-level launch plus the tail (build it whole: ur_build_hzb)", src_w, src_h);
        return UR_EUNSUPPORTED;
    }
    HzbDispatch d{};
    d.src = depth; d.SW = src_w; d.SH = src_h;
    for (uint32_t k = 0; k < 5; ++k) { d.dst[k] = hzb + mips[k].offset; d.W[k] = mips[k].width; d.H[k] = mips[k].height; }
    d.mips = 5u;
    d.vec4_ok = ((d.SW & 3u) == 0u && (reinterpret_cast<uintptr_t>(d.src) & 15u) == 0u) ? 1u : 0u;
    d.pair_ok = ((d.W[0] & 1u) == 0u && (reinterpret_cast<uintptr_t>(d.dst[0]) & 7u) == 0u) ? 1u : 0u;
    d.by0 = piece_row0;
    const dim3 grid((d.W[0] + 63u) / 64u, piece_rows);
    if (piece_rows == 0u) return UR_OK;
    if (ctx->defer_hzb_tail && ctx->defer_hzb_wide) { // rides the next streaming Lighting launch (no tail: it waits for the gather)
        ctx->pending_wide = d;
        ctx->pending_wide_grid_x = grid.x;
        ctx->pending_wide_grid_y = grid.y;
        ctx->hzb_wide_pending = true;
        return UR_OK;
    }
    hipLaunchKernelGGL(hzb_reduce4_kernel, grid, dim3(256), 0, ctx->stream, d);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

int launch_build_hzb_tail(ur_ctx* ctx, float* hzb, const ur_mip_desc* mips, uint32_t mip_count)
{
    {
        const int rc = flush_hzb_tail(ctx);
        if (rc != UR_OK) return rc;
    }
    if (!chain_is_wide_plus_tail(mips, mip_count)) {
        set_error("ur_build_hzb_tail: the chain is not one five-level launch plus the tail");
        return UR_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(hzb_tail_kernel, dim3(1), dim3(1024), 0, ctx->stream, make_tail(hzb, mips, mip_count, 5u));
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

int flush_hzb_tail(ur_ctx* ctx)
{
    if (ctx->hzb_wide_pending) { // the held-back wide launch goes first (the tail reads what it writes)
        ctx->hzb_wide_pending = false;
        hipLaunchKernelGGL(hzb_reduce4_kernel, dim3(ctx->pending_wide_grid_x, ctx->pending_wide_grid_y), dim3(256), 0, ctx->stream, ctx->pending_wide);
        UR_HIP_TRY(hipGetLastError());
    }
    if (!ctx->hzb_tail_pending) return UR_OK;
    ctx->hzb_tail_pending = false;
    hipLaunchKernelGGL(hzb_tail_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->pending_tail);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

} // namespace ur
