// C-ABI of include/ur_hotpath.h: argument validation, context/workspace management, setup-time staging.
// The kernels live in hzb.hip, cull.hip and lighting.hip.

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <vector>

#include "ur_internal.h"

namespace ur {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

} // namespace ur

using ur::set_error;

namespace ur {

int check_hzb_timeout(ur_ctx* ctx, const char* who)
{
    if (ctx && ctx->claim_timed_out && *ctx->claim_timed_out != 0u) {
        *ctx->claim_timed_out = 0u;
        (void)hipMemsetAsync(ctx->claim_words, 0, (kClaimWords + 1u) * kClaimWordStride * sizeof(uint32_t), ctx->stream);
        set_error("%s: a wave of a balanced Lighting launch gave up waiting for a tile claim of its workgroup: tiles of that launch were not shaded — "
                  "shade the frame again (reported once; the context is usable)", who);
        return UR_ETIMEOUT;
    }
    if (!ctx || !ctx->hzb_timed_out || *ctx->hzb_timed_out == 0u) return UR_OK;
    *ctx->hzb_timed_out = 0u;
    (void)hipMemsetAsync(ctx->hzb_done, 0, 64, ctx->stream); // stragglers may have left any count behind
    set_error("%s: the tail of a Build HZB chain that rode a Lighting launch gave up waiting for its producers: the HZB's small levels are stale — "
              "build it again (reported once; the context is usable)", who);
    return UR_ETIMEOUT;
}

} // namespace ur

namespace {

// ---- bordered cube staging (host) ------------------------------------------------------------------------------------
// Face addressing is D3D's (+X,-X,+Y,-Y,+Z,-Z; ties z > y > x). A border texel is the texel of the adjacent face that the
// one-texel overshoot lands on when the face plane is folded over the shared edge; a corner border texel first clamps its
// second coordinate into the face (same rule as the oracle's FetchCubeTexel — the rule is the specification).
struct FaceAxes { int major, su, sv; double ms, us, vs; }; // p[major]=ms, p[su]=us*s, p[sv]=vs*t
const FaceAxes kFaces[6] = {
    {0, 2, 1, +1, -1, -1}, // +X: (1, -t, -s)
    {0, 2, 1, -1, +1, -1}, // -X: (-1, -t, s)
    {1, 0, 2, +1, +1, +1}, // +Y: (s, 1, t)
    {1, 0, 2, -1, +1, -1}, // -Y: (s, -1, -t)
    {2, 0, 1, +1, +1, -1}, // +Z: (s, -t, 1)
    {2, 0, 1, -1, -1, -1}, // -Z: (-s, -t, -1)
};

void resolve_border(int N, int face, int i, int j, int& oface, int& oi, int& oj)
{
    const bool iOut = i < 0 || i >= N, jOut = j < 0 || j >= N;
    if (!iOut && !jOut) { oface = face; oi = i; oj = j; return; }
    if (iOut && jOut) j = j < 0 ? 0 : N - 1;
    const double s = 2.0 * (i + 0.5) / N - 1.0, t = 2.0 * (j + 0.5) / N - 1.0;
    const FaceAxes& F = kFaces[face];
    double p[3];
    p[F.major] = F.ms; p[F.su] = F.us * s; p[F.sv] = F.vs * t;
    const double over = (iOut ? std::fabs(s) : std::fabs(t)) - 1.0;
    p[F.major] *= (1.0 - over);
    const int oa = iOut ? F.su : F.sv;
    p[oa] = p[oa] > 0 ? 1.0 : -1.0;
    // the folded point lies on the face whose axis is `oa`
    const int nf = oa * 2 + (p[oa] > 0 ? 0 : 1);
    const FaceAxes& G = kFaces[nf];
    const double ns = p[G.su] / G.us, nt = p[G.sv] / G.vs;
    oface = nf;
    oi = (int)std::floor((ns + 1.0) * 0.5 * N);
    oj = (int)std::floor((nt + 1.0) * 0.5 * N);
    oi = oi < 0 ? 0 : (oi >= N ? N - 1 : oi);
    oj = oj < 0 ? 0 : (oj >= N ? N - 1 : oj);
}

bool valid_hzb_chain(uint32_t src_w, uint32_t src_h, const ur_mip_desc* mips, uint32_t mip_count)
{
    if (!mips || mip_count == 0 || mip_count > UR_MAX_HZB_MIPS) return false;
    uint32_t w = (src_w + 1) / 2, h = (src_h + 1) / 2;
    w = w ? w : 1; h = h ? h : 1;
    for (uint32_t m = 0; m < mip_count; ++m) {
        if (mips[m].width != w || mips[m].height != h) return false;
        w = w / 2 ? w / 2 : 1; h = h / 2 ? h / 2 : 1;
    }
    return true;
}

// the levels below mips[0] halve by FLOOR (CreateHZBResources, DeferredRenderer.cpp:2801-2835); every level lies inside the
// allocation the layout describes (offsets ascending, no overlap)
bool valid_hzb_chain_below_mip0(const ur_mip_desc* mips, uint32_t mip_count)
{
    if (!mips || mip_count == 0 || mip_count > UR_MAX_HZB_MIPS) return false;
    uint32_t w = mips[0].width, h = mips[0].height;
    if (w == 0 || h == 0) return false;
    uint64_t end = 0;
    for (uint32_t m = 0; m < mip_count; ++m) {
        if (mips[m].width != w || mips[m].height != h) return false;
        if (m != 0 && mips[m].offset < end) return false;
        end = (uint64_t)mips[m].offset + (uint64_t)w * h;
        w = w / 2 ? w / 2 : 1; h = h / 2 ? h / 2 : 1;
    }
    return true;
}

typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);

} // namespace

extern "C" {

const char* ur_last_error(void) { return ur::g_error; }
const char* ur_version(void) { return "unclerenderer_amd hotpath 0.5 (gfx950)"; }

ur_ctx* ur_create(int device, void* stream)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        set_error("ur_create: no HIP device %d (count %d)", device, count);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        set_error("ur_create: hipSetDevice(%d) failed", device);
        return nullptr;
    }
    ur_ctx* ctx = new ur_ctx();
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(stream);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // sRGB8 -> linear (exact IEC 61966-2-1 curve, evaluated in double)
    float table[256];
    for (int i = 0; i < 256; ++i) {
        const double c = i / 255.0;
        table[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    if (hipMalloc(&ctx->srgb_table, sizeof(table)) != hipSuccess ||
        hipMemcpy(ctx->srgb_table, table, sizeof(table), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("ur_create: sRGB table upload failed");
        ur_destroy(ctx);
        return nullptr;
    }
    if (hipMalloc(&ctx->hzb_done, 64) != hipSuccess || hipMemset(ctx->hzb_done, 0, 64) != hipSuccess) {
        set_error("ur_create: HZB arrival counter allocation failed");
        ur_destroy(ctx);
        return nullptr;
    }
    {
        void* host = nullptr;
        void* devp = nullptr;
        if (hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&devp, host, 0) != hipSuccess) {
            if (host) (void)hipHostFree(host);
            set_error("ur_create: host-visible time-out flag allocation failed");
            ur_destroy(ctx);
            return nullptr;
        }
        ctx->hzb_timed_out = static_cast<volatile uint32_t*>(host);
        ctx->hzb_timed_out_dev = static_cast<uint32_t*>(devp);
        *ctx->hzb_timed_out = 0u;
        ctx->claim_timed_out = ctx->hzb_timed_out + 1;
        ctx->claim_timed_out_dev = ctx->hzb_timed_out_dev + 1;
        *ctx->claim_timed_out = 0u;
    }
    {
        const size_t bytes = (ur::kClaimWords + 1u) * ur::kClaimWordStride * sizeof(uint32_t);
        if (hipMalloc(&ctx->claim_words, bytes) != hipSuccess || hipMemset(ctx->claim_words, 0, bytes) != hipSuccess) {
            set_error("ur_create: tile-claim words allocation failed");
            ur_destroy(ctx);
            return nullptr;
        }
    }
    if (ur_reserve(ctx, 1u << 20) != UR_OK) {
        ur_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void ur_destroy(ur_ctx* ctx)
{
    if (!ctx) return;
    // A held-back HZB tail is DISCARDED, not launched: the caller may already have freed the HZB buffer it points into (the
    // chain is complete only after ur_flush or a streaming Lighting launch — see ur_build_hzb in the header).
    ctx->hzb_tail_pending = false;
    ctx->hzb_wide_pending = false;
    if (ctx->hzb_done) (void)hipFree(ctx->hzb_done);
    if (ctx->claim_words) (void)hipFree(ctx->claim_words);
    if (ctx->hzb_timed_out) (void)hipHostFree(const_cast<uint32_t*>(ctx->hzb_timed_out));
    if (ctx->srgb_table) (void)hipFree(ctx->srgb_table);
    if (ctx->block_counts) (void)hipFree(ctx->block_counts);
    if (ctx->wave_masks) (void)hipFree(ctx->wave_masks);
    delete ctx;
}

int ur_defer_hzb_tail(ur_ctx* ctx, int enable)
{
    if (!ctx) { set_error("ur_defer_hzb_tail: null context"); return UR_EINVAL; }
    if (enable < 0 || enable > 2) { set_error("ur_defer_hzb_tail: mode %d (0 off, 1 tail, 2 whole chain)", enable); return UR_EINVAL; }
    const bool narrower = (enable == 0) || (enable == 1 && ctx->defer_hzb_wide);
    ctx->defer_hzb_tail = enable != 0;
    ctx->defer_hzb_wide = enable == 2;
    return narrower ? ur::flush_hzb_tail(ctx) : UR_OK;
}

int ur_debug_timeline(ur_ctx* ctx, unsigned long long* device_pairs, uint32_t capacity_pairs)
{
    if (!ctx || (device_pairs != nullptr && capacity_pairs == 0)) { set_error("ur_debug_timeline: bad argument"); return UR_EINVAL; }
    ctx->timeline = device_pairs;
    ctx->timeline_cap = device_pairs ? capacity_pairs : 0u;
    ctx->timeline_pos = 0;
    return UR_OK;
}

int ur_time_next_lighting(ur_ctx* ctx, void* start_event, void* stop_event)
{
    if (!ctx || (start_event != nullptr && stop_event == nullptr)) { set_error("ur_time_next_lighting: a start event needs a stop event"); return UR_EINVAL; }
    ctx->time_start = static_cast<hipEvent_t>(start_event);
    ctx->time_stop = static_cast<hipEvent_t>(stop_event);
    return UR_OK;
}

int ur_time_next_cull(ur_ctx* ctx, void* stop_event)
{
    if (!ctx) { set_error("ur_time_next_cull: null context"); return UR_EINVAL; }
    ctx->time_cull_stop = static_cast<hipEvent_t>(stop_event);
    ctx->time_cull_carried = false; // (arming or clearing: nothing has carried THIS event)
    return UR_OK;
}

int ur_time_cull_carried(const ur_ctx* ctx) { return ctx && ctx->time_cull_carried ? 1 : 0; }

// option -> (field, lowest, highest, only these two values when `pair`)
namespace {
struct OptionSlot { int ur_ctx::Options::*field; int lo, hi; bool pair; };
bool option_slot(int option, OptionSlot& o)
{
    typedef ur_ctx::Options O;
    switch (option) {
    case UR_OPT_LIGHTING_STREAM: o = {&O::lighting_stream, 0, 1, false}; return true;
    case UR_OPT_LIGHTING_WAVES_PER_WG: o = {&O::lighting_wpb, 12, 16, true}; return true;
    case UR_OPT_LIGHTING_TILED_WAVES: o = {&O::tiled_waves, 4, 6, true}; return true;
    case UR_OPT_LIGHTING_LEAVE_CUS: o = {&O::leave_cus, 0, 128, false}; return true;
    case UR_OPT_RIDE_WALKERS: o = {&O::ride_walkers, 0, 16, false}; return true;
    case UR_OPT_CULL_STORE: o = {&O::cull_store, 0, 4, false}; return true;
    case UR_OPT_LIGHTING_BALANCE: o = {&O::balance, 0, 1, false}; return true;
    case UR_OPT_BALANCE_POOL_16THS: o = {&O::balance_pool_16ths, 1, 8, false}; return true;
    case UR_OPT_BALANCE_CHUNK_SHIFT: o = {&O::balance_chunk_shift, 2, 6, false}; return true;
    case UR_OPT_DEBUG_HZB_RIDE_STALL: o = {&O::debug_hzb_ride_stall, 0, 1, false}; return true;
    default: return false;
    }
}
} // namespace

int ur_set_option(ur_ctx* ctx, int option, int value)
{
    OptionSlot o;
    if (!ctx || !option_slot(option, o)) { set_error("ur_set_option: unknown option %d", option); return UR_EINVAL; }
    if (value < o.lo || value > o.hi || (o.pair && value != o.lo && value != o.hi)) {
        set_error("ur_set_option: option %d takes %d%s%d, not %d", option, o.lo, o.pair ? " or " : " .. ", o.hi, value);
        return UR_EINVAL;
    }
    ctx->opt.*(o.field) = value;
    if (option == UR_OPT_CULL_STORE) ctx->cull_record_args = nullptr; // (setting it - also to the value it has - forgets the record of flavour 4)
    return UR_OK;
}

int ur_get_option(const ur_ctx* ctx, int option, int* value)
{
    OptionSlot o;
    if (!ctx || !value || !option_slot(option, o)) { set_error("ur_get_option: unknown option %d", option); return UR_EINVAL; }
    *value = ctx->opt.*(o.field);
    return UR_OK;
}

int ur_flush(ur_ctx* ctx)
{
    if (!ctx) { set_error("ur_flush: null context"); return UR_EINVAL; }
    const int trc = ur::check_hzb_timeout(ctx, "ur_flush");
    if (trc != UR_OK) return trc;
    return ur::flush_hzb_tail(ctx);
}

int ur_debug_lighting_schedule(const ur_ctx* ctx, uint32_t out8[8])
{
    if (!ctx || !out8) { set_error("ur_debug_lighting_schedule: null argument"); return UR_EINVAL; }
    std::memcpy(out8, ctx->last_schedule, sizeof(ctx->last_schedule));
    return UR_OK;
}

int ur_debug_set_hzb_timeout(ur_ctx* ctx)
{
    if (!ctx || !ctx->hzb_timed_out) { set_error("ur_debug_set_hzb_timeout: null context"); return UR_EINVAL; }
    *ctx->hzb_timed_out = 1u; // what the riding tail workgroup writes when it gives up
    return UR_OK;
}

int ur_reserve(ur_ctx* ctx, uint32_t max_instances)
{
    if (!ctx) return UR_EINVAL;
    if (max_instances <= ctx->ws_instances) return UR_OK;
    UR_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->block_counts) (void)hipFree(ctx->block_counts);
    if (ctx->wave_masks) (void)hipFree(ctx->wave_masks);
    ctx->block_counts = nullptr; ctx->wave_masks = nullptr; ctx->ws_instances = 0;
    ctx->cull_record_args = nullptr;
    const size_t blocks = ((size_t)max_instances + 255u) / 256u;
    if (hipMalloc(&ctx->block_counts, blocks * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(&ctx->wave_masks, blocks * 4u * sizeof(uint64_t)) != hipSuccess) {
        set_error("ur_reserve: workspace allocation for %u instances failed", max_instances);
        return UR_ENOMEM;
    }
    ctx->ws_instances = (uint32_t)(blocks * 256u);
    return UR_OK;
}

uint32_t ur_hzb_layout(uint32_t src_w, uint32_t src_h, ur_mip_desc* mips, uint32_t* mip_count)
{
    if (!mips || !mip_count || src_w == 0 || src_h == 0) return 0;
    uint32_t w = (src_w + 1) / 2, h = (src_h + 1) / 2;
    w = w ? w : 1; h = h ? h : 1;
    uint32_t n = 0, off = 0;
    for (;;) {
        if (n >= UR_MAX_HZB_MIPS) return 0;
        mips[n].offset = off; mips[n].width = w; mips[n].height = h;
        ++n;
        off += (w * h + 63u) & ~63u; // every mip starts on a 256-byte boundary
        if (!(w > 1 || h > 1)) break;
        w = w / 2 ? w / 2 : 1; h = h / 2 ? h / 2 : 1;
    }
    *mip_count = n;
    return off;
}

int ur_build_hzb(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb_base, const ur_mip_desc* mips,
                 uint32_t mip_count)
{
    if (!ctx || !depth || !hzb_base || src_w == 0 || src_h == 0) { set_error("ur_build_hzb: null/zero argument"); return UR_EINVAL; }
    if (!valid_hzb_chain(src_w, src_h, mips, mip_count)) { set_error("ur_build_hzb: mip chain does not match CreateHZBResources sizing"); return UR_EINVAL; }
    const int trc = ur::check_hzb_timeout(ctx, "ur_build_hzb");
    if (trc != UR_OK) return trc;
    return ur::launch_build_hzb(ctx, depth, src_w, src_h, hzb_base, mips, mip_count);
}

static int cull_checked(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds, const float* hzb_base,
                        const ur_mip_desc* mips, void* indirect_args, uint32_t* stats2, uint32_t* visible_idx,
                        uint32_t* visible_count, uint32_t index_base)
{
    if (!ctx || !constants) { set_error("ur_cull_indirect_args: null ctx/constants"); return UR_EINVAL; }
    const uint32_t n = constants[40], hzb_on = constants[41], mipc = constants[42];
    if ((visible_idx == nullptr) != (visible_count == nullptr)) { set_error("ur_cull_indirect_args: visible_idx and visible_count go together"); return UR_EINVAL; }
    if (n != 0 && (!bounds || !indirect_args)) { set_error("ur_cull_indirect_args: null bounds/indirect_args"); return UR_EINVAL; }
    if (n != 0 && hzb_on != 0 && constants[43] != 0 && constants[44] != 0 && mipc != 0) {
        if (!hzb_base || !mips || mipc > UR_MAX_HZB_MIPS) { set_error("ur_cull_indirect_args: HZB enabled but hzb/mips missing"); return UR_EINVAL; }
        if (mips[0].width != constants[43] || mips[0].height != constants[44]) { set_error("ur_cull_indirect_args: HZBWidth/Height do not match mips[0]"); return UR_EINVAL; }
        // the kernel indexes hzb + mips[level].offset with pitch mips[level].width for every level up to HZBMipCount - 1
        if (!valid_hzb_chain_below_mip0(mips, mipc)) { set_error("ur_cull_indirect_args: mips[1..%u] do not halve from mips[0] / overlap", mipc - 1); return UR_EINVAL; }
    }
    const int trc = ur::check_hzb_timeout(ctx, "ur_cull_indirect_args");
    if (trc != UR_OK) return trc;
    return ur::launch_cull(ctx, constants, bounds, hzb_base, mips, indirect_args, stats2, visible_idx, visible_count, index_base);
}

int ur_cull_indirect_args_ex(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds, const float* hzb_base,
                             const ur_mip_desc* mips, void* indirect_args, uint32_t* stats2, uint32_t* visible_idx,
                             uint32_t* visible_count, uint32_t index_base)
{
    if (ctx) ctx->time_cull_carried = false;
    const int rc = cull_checked(ctx, constants, bounds, hzb_base, mips, indirect_args, stats2, visible_idx, visible_count, index_base);
    if (ctx) ctx->time_cull_stop = nullptr; // one-shot whatever the call did (ur_time_next_cull)
    return rc;
}

int ur_hzb_band_pieces(uint32_t src_h, uint32_t n_ranks, uint32_t rank, uint32_t* piece_row0, uint32_t* piece_rows)
{
    if (src_h == 0 || n_ranks == 0 || rank >= n_ranks || src_h % n_ranks != 0 || !piece_row0 || !piece_rows) { set_error("ur_hzb_band_pieces: bad argument"); return UR_EINVAL; }
    const uint32_t rows = src_h / n_ranks;
    const uint32_t first = (rank * rows + 31u) / 32u, last = ((rank + 1u) * rows + 31u) / 32u; // pieces whose first source row lies in the band
    *piece_row0 = first;
    *piece_rows = last - first;
    return UR_OK;
}

int ur_hzb_band_slices(const ur_mip_desc* mips, uint32_t mip_count, uint32_t piece_row0, uint32_t piece_rows, ur_hzb_slice* out5)
{
    if (!mips || mip_count < 5 || !out5) { set_error("ur_hzb_band_slices: bad argument"); return UR_EINVAL; }
    for (uint32_t k = 0; k < 5; ++k) {
        const uint32_t per = 16u >> k, H = mips[k].height, W = mips[k].width;
        const uint32_t r0 = std::min(piece_row0 * per, H), r1 = std::min((piece_row0 + piece_rows) * per, H);
        out5[k].offset = mips[k].offset + r0 * W;
        out5[k].count = (r1 - r0) * W;
    }
    return UR_OK;
}

int ur_build_hzb_band(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb_base, const ur_mip_desc* mips, uint32_t mip_count,
                      uint32_t piece_row0, uint32_t piece_rows)
{
    if (!ctx || !depth || !hzb_base || src_w == 0 || src_h == 0) { set_error("ur_build_hzb_band: null/zero argument"); return UR_EINVAL; }
    if (!valid_hzb_chain(src_w, src_h, mips, mip_count)) { set_error("ur_build_hzb_band: mip chain does not match CreateHZBResources sizing"); return UR_EINVAL; }
    if ((uint64_t)piece_row0 + piece_rows > (src_h + 31u) / 32u) { set_error("ur_build_hzb_band: piece rows [%u, %u) of %u", piece_row0, piece_row0 + piece_rows, (src_h + 31u) / 32u); return UR_EINVAL; }
    const int trc = ur::check_hzb_timeout(ctx, "ur_build_hzb_band");
    if (trc != UR_OK) return trc;
    return ur::launch_build_hzb_band(ctx, depth, src_w, src_h, hzb_base, mips, mip_count, piece_row0, piece_rows);
}

int ur_build_hzb_tail(ur_ctx* ctx, float* hzb_base, const ur_mip_desc* mips, uint32_t mip_count)
{
    if (!ctx || !hzb_base || !valid_hzb_chain_below_mip0(mips, mip_count)) { set_error("ur_build_hzb_tail: bad argument"); return UR_EINVAL; }
    const int trc = ur::check_hzb_timeout(ctx, "ur_build_hzb_tail");
    if (trc != UR_OK) return trc;
    return ur::launch_build_hzb_tail(ctx, hzb_base, mips, mip_count);
}

int ur_cull_indirect_args(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds, const float* hzb_base,
                          const ur_mip_desc* mips, void* indirect_args, uint32_t* stats2, uint32_t* visible_idx,
                          uint32_t* visible_count)
{
    return ur_cull_indirect_args_ex(ctx, constants, bounds, hzb_base, mips, indirect_args, stats2, visible_idx, visible_count, 0);
}

size_t ur_env_cube_texels(uint32_t base_size, uint32_t mip_count)
{
    if (base_size == 0 || mip_count == 0 || mip_count > 16) return 0;
    size_t n = 0;
    for (uint32_t m = 0; m < mip_count; ++m) {
        const size_t e = (base_size >> m > 1u ? base_size >> m : 1u) + 2u;
        n += 6u * e * e;        // the bordered faces
        n += 9u * (e - 1u) * e; // the same texels once more as RGB row pairs: 6 (e - 1) e entries of 12 bytes = 9 (e - 1) e half4 units
    }
    return n;
}

int ur_stage_env_cube(ur_ctx* ctx, const ur_half4* src, uint32_t base, uint32_t mip_count, ur_half4* dst_device)
{
    if (!ctx || !src || !dst_device || ur_env_cube_texels(base, mip_count) == 0) { set_error("ur_stage_env_cube: bad argument"); return UR_EINVAL; }
    std::vector<size_t> mip_off(mip_count);
    size_t face_stride = 0;
    for (uint32_t m = 0; m < mip_count; ++m) {
        mip_off[m] = face_stride;
        const size_t n = base >> m > 1u ? base >> m : 1u;
        face_stride += n * n;
    }
    std::vector<ur_half4> out(ur_env_cube_texels(base, mip_count));
    size_t off = 0;
    for (uint32_t m = 0; m < mip_count; ++m) {
        const int N = (int)(base >> m > 1u ? base >> m : 1u), E = N + 2;
        for (int f = 0; f < 6; ++f)
            for (int j = -1; j <= N; ++j)
                for (int i = -1; i <= N; ++i) {
                    int sf, si, sj;
                    resolve_border(N, f, i, j, sf, si, sj);
                    out[off + ((size_t)f * E + (j + 1)) * E + (i + 1)] = src[(size_t)sf * face_stride + mip_off[m] + (size_t)sj * N + si];
                }
        off += (size_t)6 * E * E;
    }
    // Second section, behind all bordered mips: every mip once more as RGB ROW PAIRS. Entry (f, j, i), j in [0, E-2], is the 12 bytes
    // {R G B of texel (i, j), R G B of texel (i, j + 1)} of the bordered face (the alpha channel is never sampled:
    // DeferredLighting.hlsl:82,86 take .rgb), entries of a pair-row contiguous: the 2x2 bilinear footprint at (i, j) is the 24
    // bytes at entry ((f (E-1) + j) E + i) - two 12-byte loads that almost always fall into ONE cache line where the bordered
    // layout's two rows are two lines and two 16-byte loads. The streaming lighting kernel gathers its prefiltered taps here.
    {
        uint16_t* rgb = reinterpret_cast<uint16_t*>(out.data() + off);
        size_t boff = 0, e = 0;
        for (uint32_t m = 0; m < mip_count; ++m) {
            const size_t E = (size_t)(base >> m > 1u ? base >> m : 1u) + 2u;
            for (size_t f = 0; f < 6; ++f)
                for (size_t j = 0; j + 1 < E; ++j)
                    for (size_t i = 0; i < E; ++i) {
                        const ur_half4& t0 = out[boff + (f * E + j) * E + i];
                        const ur_half4& t1 = out[boff + (f * E + j + 1) * E + i];
                        rgb[e++] = t0.x; rgb[e++] = t0.y; rgb[e++] = t0.z;
                        rgb[e++] = t1.x; rgb[e++] = t1.y; rgb[e++] = t1.z;
                    }
            boff += 6u * E * E;
        }
    }
    UR_HIP_TRY(hipMemcpy(dst_device, out.data(), out.size() * sizeof(ur_half4), hipMemcpyHostToDevice));
    return UR_OK;
}

static int check_band(const char* who, ur_ctx* ctx, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows)
{
    if (!ctx || w == 0 || h == 0 || (uint64_t)row0 + rows > h) {
        set_error("%s: bad frame/band (w=%u h=%u row0=%u rows=%u)", who, w, h, row0, rows);
        return UR_EINVAL;
    }
    return UR_OK;
}

int ur_deferred_lighting(ur_ctx* ctx, const ur_scene_constants* scene, const ur_half4* a, const ur_half4* b, const uint32_t* c,
                         const ur_lighting_tables* tables, ur_half4* hdr, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows)
{
    const int rc = check_band("ur_deferred_lighting", ctx, w, h, row0, rows);
    if (rc != UR_OK) return rc;
    if (!scene || !a || !b || !c || !tables || !hdr) { set_error("ur_deferred_lighting: null argument"); return UR_EINVAL; }
    const int trc = ur::check_hzb_timeout(ctx, "ur_deferred_lighting"); // (a launch behind one that gave up must not start from its leftovers)
    if (trc != UR_OK) return trc;
    return ur::launch_lighting(ctx, scene, nullptr, a, b, c, nullptr, tables, hdr, w, h, row0, rows, ur::UR_MODE_LIGHTING);
}

int ur_sky_atmosphere(ur_ctx* ctx, const ur_sky_constants* sky, const float* depth, ur_half4* hdr, uint32_t w, uint32_t h,
                      uint32_t row0, uint32_t rows)
{
    const int rc = check_band("ur_sky_atmosphere", ctx, w, h, row0, rows);
    if (rc != UR_OK) return rc;
    if (!sky || !depth || !hdr) { set_error("ur_sky_atmosphere: null argument"); return UR_EINVAL; }
    return ur::launch_lighting(ctx, nullptr, sky, nullptr, nullptr, nullptr, depth, nullptr, hdr, w, h, row0, rows, ur::UR_MODE_SKY);
}

int ur_deferred_lighting_sky(ur_ctx* ctx, const ur_scene_constants* scene, const ur_sky_constants* sky, const ur_half4* a,
                             const ur_half4* b, const uint32_t* c, const float* depth, const ur_lighting_tables* tables,
                             ur_half4* hdr, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows)
{
    const int rc = check_band("ur_deferred_lighting_sky", ctx, w, h, row0, rows);
    if (rc != UR_OK) return rc;
    if (!scene || !sky || !a || !b || !c || !depth || !tables || !hdr) { set_error("ur_deferred_lighting_sky: null argument"); return UR_EINVAL; }
    const int trc = ur::check_hzb_timeout(ctx, "ur_deferred_lighting_sky");
    if (trc != UR_OK) return trc;
    return ur::launch_lighting(ctx, scene, sky, a, b, c, depth, tables, hdr, w, h, row0, rows, ur::UR_MODE_FUSED);
}

// RCCL is resolved at run time from whatever copy the host process already loaded globally (the communicator must come
// from the same copy), falling back to the system's librccl: the library has no link-time dependency on RCCL.
static void* rccl_symbol(const char* name)
{
    void* fn = dlsym(RTLD_DEFAULT, name);
    if (!fn) {
        static void* lib = nullptr;
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) fn = dlsym(lib, name);
    }
    return fn;
}

int ur_allgather_rows_bytes_ex(ur_ctx* ctx, void* comm, void* image, uint32_t row_bytes, uint32_t h, uint32_t n_ranks, uint32_t rank, int mode)
{
    if (!ctx || !comm || !image || row_bytes == 0 || h == 0 || n_ranks == 0 || rank >= n_ranks || h % n_ranks != 0) {
        set_error("ur_allgather_rows: bad argument (row_bytes=%u h=%u ranks=%u rank=%u)", row_bytes, h, n_ranks, rank);
        return UR_EINVAL;
    }
    if (mode != UR_GATHER_RING && mode != UR_GATHER_DIRECT) { set_error("ur_allgather_rows: mode %d (0 ring, 1 direct)", mode); return UR_EINVAL; }
    const size_t band_bytes = (size_t)row_bytes * (h / n_ranks);
    char* base = reinterpret_cast<char*>(image);
    const char* send = base + band_bytes * rank;
    if (mode == UR_GATHER_RING) {
        static nccl_allgather_fn fn = nullptr;
        if (!fn) fn = reinterpret_cast<nccl_allgather_fn>(rccl_symbol("ncclAllGather"));
        if (!fn) { set_error("ur_allgather_rows: ncclAllGather not found"); return UR_EUNSUPPORTED; }
        const int rc = fn(send, image, band_bytes, /*ncclInt8*/ 0, comm, ctx->stream);
        if (rc != 0) { set_error("ncclAllGather failed (%d)", rc); return UR_EHIP; }
        return UR_OK;
    }
    // Direct form: the band goes to every peer over the xGMI link the two GPUs share (an MI355X node is fully connected,
    // 7 links per GPU), all N - 1 transfers of a rank in flight at once — one grouped call, no ring hops.
    typedef int (*group_fn)(void);
    typedef int (*send_fn)(const void*, size_t, int, int, void*, hipStream_t);
    typedef int (*recv_fn)(void*, size_t, int, int, void*, hipStream_t);
    static group_fn g_start = nullptr, g_end = nullptr;
    static send_fn f_send = nullptr;
    static recv_fn f_recv = nullptr;
    if (!g_start) {
        g_start = reinterpret_cast<group_fn>(rccl_symbol("ncclGroupStart"));
        g_end = reinterpret_cast<group_fn>(rccl_symbol("ncclGroupEnd"));
        f_send = reinterpret_cast<send_fn>(rccl_symbol("ncclSend"));
        f_recv = reinterpret_cast<recv_fn>(rccl_symbol("ncclRecv"));
    }
    if (!g_start || !g_end || !f_send || !f_recv) { g_start = nullptr; set_error("ur_allgather_rows: ncclGroupStart/End, ncclSend, ncclRecv not found"); return UR_EUNSUPPORTED; }
    int rc = g_start();
    // peers in the order rank + 1, rank + 2, ...: at any moment every rank sends to a different peer
    for (uint32_t k = 1; k < n_ranks && rc == 0; ++k) {
        const uint32_t to = (rank + k) % n_ranks, from = (rank + n_ranks - k) % n_ranks;
        rc = f_send(send, band_bytes, /*ncclInt8*/ 0, (int)to, comm, ctx->stream);
        if (rc == 0) rc = f_recv(base + band_bytes * from, band_bytes, /*ncclInt8*/ 0, (int)from, comm, ctx->stream);
    }
    const int rc_end = g_end();
    if (rc != 0 || rc_end != 0) { set_error("grouped ncclSend/ncclRecv failed (%d, %d)", rc, rc_end); return UR_EHIP; }
    return UR_OK;
}

int ur_allgather_rows_bytes(ur_ctx* ctx, void* comm, void* image, uint32_t row_bytes, uint32_t h, uint32_t n_ranks, uint32_t rank)
{
    return ur_allgather_rows_bytes_ex(ctx, comm, image, row_bytes, h, n_ranks, rank, UR_GATHER_RING);
}

int ur_allgather_rows(ur_ctx* ctx, void* comm, ur_half4* hdr_full, uint32_t w, uint32_t h, uint32_t n_ranks, uint32_t rank)
{
    if ((uint64_t)w * sizeof(ur_half4) > 0xFFFFFFFFull) { set_error("ur_allgather_rows: row of %u pixels is too wide", w); return UR_EINVAL; }
    return ur_allgather_rows_bytes(ctx, comm, hdr_full, w * (uint32_t)sizeof(ur_half4), h, n_ranks, rank);
}

} // extern "C"
