// Internal declarations shared by the C-ABI translation units. Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/ur_hotpath.h"

namespace ur {
// Arguments of the single-workgroup tail of the HZB chain (csrc/hzb_tail.h): the levels from `first_mip` on.
constexpr uint32_t kTailMaxLevels = 12, kTailTexels = 16384;
constexpr uint32_t kClaimWords = 32, kClaimWordStride = 32; // (stride in uint32: 128 bytes)
struct HzbTail {
    const float* src; // mip first_mip - 1 (global memory, written by the previous launch)
    uint32_t SW, SH, first_mip, levels;
    float* dst[kTailMaxLevels];
    uint32_t W[kTailMaxLevels], H[kTailMaxLevels];
    uint32_t magic[kTailMaxLevels]; // i / W[l] == __umulhi(i, magic[l]) for i < 16384 (magic = 2^32 / W + 1)
};
} // namespace ur
#include "hzb_wide.h"

struct ur_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // compaction workspace (grown by ur_reserve / lazily outside graph capture)
    uint32_t* block_counts = nullptr; // one per 256-instance block
    uint64_t* wave_masks = nullptr;   // one per 64 instances
    uint32_t ws_instances = 0;
    // UR_OPT_CULL_STORE = 4: wave_masks describes the InstanceCount words of this command buffer / count (the last multi-block launch's)
    const void* cull_record_args = nullptr;
    uint32_t cull_record_n = 0;
    // sRGB8 -> linear table (256 floats), uploaded once
    float* srgb_table = nullptr;
    int cu_count = 256;
    // The HZB tail may ride along with the next streaming Lighting launch as one extra workgroup (ur_defer_hzb_tail):
    // its single workgroup is ~5 us of latency during which the other 255 CUs would idle.
    bool defer_hzb_tail = false;
    bool hzb_tail_pending = false;
    ur::HzbTail pending_tail{};
    // ur_defer_hzb_tail(ctx, 2): the wide launch in front of the tail is held back too; the Lighting launch's workgroups take its
    // 128x32 pieces along (one wave of each) and signal `hzb_done`, which the riding tail workgroup waits for.
    bool defer_hzb_wide = false;
    bool hzb_wide_pending = false;
    ur::HzbDispatch pending_wide{};
    uint32_t pending_wide_grid_x = 0, pending_wide_grid_y = 0;
    // ur_debug_timeline: device array of {first entry, last exit} pairs in s_memrealtime ticks (100 MHz), one pair per cull and per
    // streaming Lighting launch, in launch order (the caller initialises every pair to {~0, 0})
    unsigned long long* timeline = nullptr;
    uint32_t timeline_cap = 0, timeline_pos = 0;
    uint32_t* hzb_done = nullptr; // device: arrivals of the current launch (reset by the tail workgroup)
    // the riding tail's time-out flag: one word of mapped, coherent host memory the kernel writes (system scope) and the host
    // reads without a synchronisation at its next entry point that depends on the HZB (check_hzb_timeout)
    volatile uint32_t* hzb_timed_out = nullptr;
    uint32_t* hzb_timed_out_dev = nullptr; // its device address
    // ur_time_next_lighting: events the next Lighting launch carries on its dispatch (hipExtLaunchKernel); consumed by it
    hipEvent_t time_start = nullptr, time_stop = nullptr;
    hipEvent_t time_cull_stop = nullptr; // ur_time_next_cull: carried by the last launch of the next cull call, cleared by that call
    bool time_cull_carried = false;      // ... and whether a dispatch of that call took it (ur_time_cull_carried)
    // ur_set_option: launch-shape choices of this context (include/ur_hotpath.h, UR_OPT_*)
    struct Options {
        int lighting_stream = 1, lighting_wpb = 16, tiled_waves = 6, leave_cus = 0, ride_walkers = 0, cull_store = 3;
        int balance = 1, balance_pool_16ths = 3, balance_chunk_shift = 4;
        int debug_hzb_ride_stall = 0;
    } opt;
    // Inter-workgroup tile claims of the streaming lighting kernel (UR_OPT_LIGHTING_BALANCE): kClaimWords words, each on a 128-byte
    // line of its own, + the count of workgroups that have made their last claim. All zero between launches: the workgroup whose
    // last claim comes last zeroes them (csrc/lighting.hip).
    uint32_t* claim_words = nullptr;
    uint32_t last_schedule[8] = {}; // ur_debug_lighting_schedule: the tile schedule of the context's last streaming Lighting launch
    // host-visible (mapped, coherent) word beside hzb_timed_out: a wave of a balanced launch gave up waiting for a claim
    volatile uint32_t* claim_timed_out = nullptr;
    uint32_t* claim_timed_out_dev = nullptr;
};

namespace ur {

void set_error(const char* fmt, ...);

#define UR_HIP_TRY(expr)                                                                          \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::ur::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return UR_EHIP;                                                                       \
        }                                                                                         \
    } while (0)

// kernels (one file each)
int launch_build_hzb(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb_base,
                     const ur_mip_desc* mips, uint32_t mip_count);
int launch_build_hzb_band(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb_base, const ur_mip_desc* mips, uint32_t mip_count,
                          uint32_t piece_row0, uint32_t piece_rows);
int launch_build_hzb_tail(ur_ctx* ctx, float* hzb_base, const ur_mip_desc* mips, uint32_t mip_count);
int launch_cull(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds, const float* hzb_base,
                const ur_mip_desc* mips, void* indirect_args, uint32_t* stats2, uint32_t* visible_idx,
                uint32_t* visible_count, uint32_t index_base);
int launch_lighting(ur_ctx* ctx, const ur_scene_constants* scene, const ur_sky_constants* sky, const ur_half4* gbuf_a,
                    const ur_half4* gbuf_b, const uint32_t* gbuf_c, const float* depth, const ur_lighting_tables* tables,
                    ur_half4* hdr, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows, int mode);
enum { UR_MODE_LIGHTING = 0, UR_MODE_SKY = 1, UR_MODE_FUSED = 2 };
// the next {entry, exit} pair of the debug timeline (nullptr when it is off)
inline unsigned long long* next_timeline_pair(ur_ctx* ctx)
{
    if (!ctx->timeline || ctx->timeline_pos >= ctx->timeline_cap) return nullptr; // full: later launches stamp nothing
    return ctx->timeline + 2u * (size_t)ctx->timeline_pos++;
}
// launches a deferred HZB tail on its own if one is pending (ur_flush and every launch that reads or rewrites the HZB)
int flush_hzb_tail(ur_ctx* ctx);
// UR_ETIMEOUT (once) if a riding tail has given up waiting since the last check: clears the flag and resets the arrival
// counter on the context's stream (the call that reports it does nothing else); UR_OK otherwise
int check_hzb_timeout(ur_ctx* ctx, const char* who);

} // namespace ur
