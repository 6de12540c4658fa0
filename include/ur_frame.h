/*
 * ur_frame.h — C face of the render-graph-driven frame (csrc/frame/HotPathRenderer): the four hot passes added to an
 * FRenderGraph in the reference's order and executed on the context's stream. This is what a host that does not link
 * C++ (the Python tests, bench.py) calls; a C++ renderer uses FRenderGraph / FHotPathRenderer directly.
 */
#ifndef UR_FRAME_H
#define UR_FRAME_H

#include "ur_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ur_frame ur_frame;

/* All device pointers; band-local images hold rows [row0,row0+rows) of the width x height frame. */
typedef struct ur_frame_resources {
    uint32_t width, height, row0, rows;
    const ur_half4* gbuffer_a;
    const ur_half4* gbuffer_b;
    const uint32_t* gbuffer_c;
    const float* depth_band;
    ur_half4* lighting_band;
    const float* depth_full;      /* src of the (replicated) HZB build */
    float* hzb;
    ur_mip_desc hzb_mips[UR_MAX_HZB_MIPS];
    uint32_t hzb_mip_count;
    ur_lighting_tables tables;
    const ur_float4* model_bounds;
    void* indirect_args;
    uint32_t indirect_command_count;
    uint32_t instance_index_base;
    uint32_t* visible_indices;    /* nullable */
    uint32_t* visible_count;      /* nullable */
    uint32_t* cull_stats;         /* nullable */
    uint32_t* tonemap_band;       /* nullable: R8G8B8A8_UNORM output of the optional Tonemap pass (UR_FRAME_TONEMAP) */
} ur_frame_resources;

#define UR_FRAME_INDIRECT_DRAW 0x1u
#define UR_FRAME_HZB 0x2u
#define UR_FRAME_DEPTH_PREPASS 0x4u
#define UR_FRAME_SHADOWS 0x8u
#define UR_FRAME_SKY 0x10u
#define UR_FRAME_FUSE_LIGHTING_SKY 0x20u
#define UR_FRAME_GPU_TIMING 0x40u
#define UR_FRAME_GRAPH_DUMP 0x80u
#define UR_FRAME_BARRIER_LOGS 0x100u
#define UR_FRAME_ASYNC_COMPUTE 0x200u /* GPU Culling + Build HZB on a second HIP stream, overlapping Lighting/Sky */
#define UR_FRAME_ASYNC_NO_JOIN 0x400u /* with ASYNC_COMPUTE: do not end the frame with a main<-async join; the caller calls ur_frame_join_async() */
#define UR_FRAME_TONEMAP 0x800u /* add the Tonemap pass after Sky (Exposure 0.9, Gamma 2.2, PBR-neutral curve) */
#define UR_FRAME_TIME_LIGHTING 0x1000u /* bracket the Lighting pass with a HIP event pair on its stream; read with ur_frame_lighting_times() */
#define UR_FRAME_HZB_TAIL_WITH_LIGHTING 0x2000u /* the single-workgroup tail of Build HZB rides along with the Lighting launch (ur_defer_hzb_tail); ignored with ASYNC_COMPUTE. The Build HZB pass then ends before the chain is complete; the frame is complete when ur_frame_render's launches are */
#define UR_FRAME_TIME_LIGHTING_RECORD_COST 0x8000u /* TIME_LIGHTING plus one more event recorded right behind the pair: its distance to the pair's closing event is what an event record costs on this queue (ur_frame_lighting_times_ex) */
#define UR_FRAME_HZB_WITH_LIGHTING 0x4000u /* the WHOLE Build HZB chain rides along with the Lighting launch (ur_defer_hzb_tail(ctx, 2)): its 128x32 pieces are walked by one wave of every lighting workgroup, its tail by an extra workgroup that waits for them; two launches per frame (cull, lighting). Ignored with ASYNC_COMPUTE */
#define UR_FRAME_TIME_LIGHTING_KERNEL 0x10000u /* time the Lighting pass by a HIP event pair carried on its kernel dispatch (ur_time_next_lighting): from the end of what precedes the kernel to the kernel's end, what rocprofv3's kernel trace reports for the dispatch; no event record behind the kernel; read with ur_frame_lighting_times() */
#define UR_FRAME_HZB_SHARD 0x20000u /* several ranks (ur_frame_create's world_size > 1): Build HZB builds only this rank's 128x32 pieces of mips 0..4 (ur_build_hzb_band; riding the Lighting launch with HZB_WITH_LIGHTING) and leaves the exchange of the slices and the tail (ur_build_hzb_tail) to the caller, who holds the communicator. One rank: the whole chain as usual */
#define UR_FRAME_DEFAULT (UR_FRAME_INDIRECT_DRAW | UR_FRAME_HZB | UR_FRAME_DEPTH_PREPASS | UR_FRAME_SHADOWS | UR_FRAME_SKY)

ur_frame* ur_frame_create(ur_ctx* ctx, void* stream, uint32_t frames_in_flight, int rank, int world_size);
void ur_frame_destroy(ur_frame* f);
/* One frame: BeginFrame, build the graph (GPU Culling, Build HZB, Lighting, Sky), Execute. culling_constants: the 46
 * dwords of DispatchGpuCulling; dwords 40-44 (ModelCount, HZBEnabled, HZBMipCount, HZBWidth, HZBHeight) are overwritten
 * from the resources and from whether last frame built an HZB (bHZBReady). */
int ur_frame_render(ur_frame* f, const ur_frame_resources* res, const uint32_t* culling_constants, const ur_scene_constants* scene,
                    const ur_sky_constants* sky, uint32_t option_flags);
/* Main stream waits for everything the async-compute stream has been given so far (see UR_FRAME_ASYNC_NO_JOIN). */
void ur_frame_join_async(ur_frame* f);
/* Elapsed milliseconds of the Lighting passes recorded since the last call (UR_FRAME_TIME_LIGHTING; up to 1024 kept).
 * Call after the stream has been synchronised. Returns how many were written. */
uint32_t ur_frame_lighting_times(ur_frame* f, float* out_ms, uint32_t cap);
/* The same plus, per sample, the time from the bracket's closing event to one more event recorded right behind it: what a
 * single event record adds to the queue (the bracket contains one such record in front of the kernel); -1 for samples taken
 * without UR_FRAME_TIME_LIGHTING_RECORD_COST. */
uint32_t ur_frame_lighting_times_ex(ur_frame* f, float* out_ms, float* out_record_ms, uint32_t cap);
int ur_frame_hzb_ready(const ur_frame* f);
void ur_frame_reset_hzb(ur_frame* f);
/* Last execution: one line per pass "name|culled(0/1)|transitions|async(0/1)|cross-stream waits". Returns bytes needed (incl. NUL). */
uint32_t ur_frame_report(const ur_frame* f, char* buf, uint32_t cap);
/* Sliding-window GPU timing (FRenderGraph::GetGpuTimingStats): "name|avg_ms|min_ms|max_ms|samples" lines. */
uint32_t ur_rg_timing_stats(char* buf, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* UR_FRAME_H */
