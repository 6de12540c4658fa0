// HotPathRenderer.cpp — pass wiring of the hot path on FRenderGraph (see HotPathRenderer.h for reference citations).

#include "HotPathRenderer.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <sstream>

#include "../../../include/ur_frame.h"

int FHotPathRenderer::RenderFrame(FHIPCommandContext& Cmd, FHotPathResources& Res, const FHotPathFrameConstants& Constants,
                                  const FHotPathOptions& Options)
{
    PassError = UR_OK;
    FRenderGraph Graph;
    Graph.SetDevice(Device);
    Graph.SetGpuTimingEnabled(Options.bGpuTiming);
    Graph.SetGraphDumpEnabled(Options.bGraphDump);
    Graph.SetResourceLifetimeLogging(Options.bGraphDump);
    Graph.SetBarrierLoggingEnabled(Options.bBarrierLogs);

    const uint32 HZBWidth = Res.HZBMipCount ? Res.HZBMips[0].width : 0, HZBHeight = Res.HZBMipCount ? Res.HZBMips[0].height : 0;

    // External resources, imported with a pointer to the owner's state variable (DeferredRenderer.cpp:437-506).
    const FRGResourceHandle DepthHandle = Graph.ImportTexture("Depth", Res.DepthFull, &Res.DepthState, {Res.Width, Res.Height, RG_FORMAT_R32_FLOAT});
    FRGResourceHandle GBufferHandles[3];
    GBufferHandles[0] = Graph.ImportTexture("GBufferA", Res.GBufferA, &Res.GBufferStates[0], {Res.Width, Res.Rows, RG_FORMAT_R16G16B16A16_FLOAT});
    GBufferHandles[1] = Graph.ImportTexture("GBufferB", Res.GBufferB, &Res.GBufferStates[1], {Res.Width, Res.Rows, RG_FORMAT_R16G16B16A16_FLOAT});
    GBufferHandles[2] = Graph.ImportTexture("GBufferC", Res.GBufferC, &Res.GBufferStates[2], {Res.Width, Res.Rows, RG_FORMAT_R8G8B8A8_UNORM_SRGB});
    const FRGResourceHandle ShadowHandle = Graph.ImportTexture("ShadowMap", const_cast<float*>(Res.Tables.shadow_map), &Res.ShadowState,
                                                               {static_cast<uint32>(Constants.Scene.ShadowMapSize[0]), static_cast<uint32>(Constants.Scene.ShadowMapSize[1]), RG_FORMAT_R32_FLOAT});
    const FRGResourceHandle LightingHandle = Graph.ImportTexture("Lighting", Res.LightingBand, &Res.LightingState, {Res.Width, Res.Rows, RG_FORMAT_R16G16B16A16_FLOAT});
    const FRGResourceHandle HZBHandle = Graph.ImportTexture("HZB", Res.HZB, &Res.HZBState, {HZBWidth, HZBHeight, RG_FORMAT_R32_FLOAT});

    const bool bHZBEnabled = Options.bHZBEnabled && Res.HZB != nullptr && Res.HZBMipCount != 0;
    if (!bHZBEnabled) bHZBReady = false; // :514-517
    const bool bUseHZBOcclusion = bHZBEnabled && bHZBReady; // ConfigureHZBOcclusion, :519-520

    // ---- GPU Culling (first pass of the frame; uses LAST frame's HZB with the current camera) --------------------
    struct FGpuCullingPassData
    {
        bool bEnabled = false;
        uint32 Constants[UR_CULL_CONSTANT_DWORDS] = {};
    };
    Graph.AddPass<FGpuCullingPassData>("GPU Culling", [&](FGpuCullingPassData& Data, FRGPassBuilder& Builder)
    {
        Data.bEnabled = Options.bEnableIndirectDraw && Res.IndirectArgs && Res.ModelBounds && Res.IndirectCommandCount != 0;
        std::memcpy(Data.Constants, Constants.CullingConstants, sizeof(Data.Constants));
        Data.Constants[40] = Res.IndirectCommandCount;
        Data.Constants[41] = bUseHZBOcclusion ? 1u : 0u;
        Data.Constants[42] = Res.HZBMipCount;
        Data.Constants[43] = HZBWidth;
        Data.Constants[44] = HZBHeight;
        if (Data.bEnabled) {
            if (bUseHZBOcclusion) Builder.ReadTexture(HZBHandle, RG_STATE_NON_PIXEL_SHADER_RESOURCE);
            Builder.KeepAlive();
            // Neither visibility pass shares a resource with Lighting/Sky inside a frame (the cull reads LAST frame's
            // HZB), so both can run beside the VALU-bound lighting kernel on the second stream.
            if (Options.bAsyncCompute) Builder.AsyncCompute();
        }
    }, [this, &Res](const FGpuCullingPassData& Data, FHIPCommandContext& Cmd)
    {
        if (!Data.bEnabled) return;
        // DispatchGpuCulling (Renderer.cpp:394-472): the UAV / INDIRECT_ARGUMENT transitions are stream order here.
        const int rc = ur_cull_indirect_args_ex(Cmd.GetContext(), Data.Constants, Res.ModelBounds, Res.HZB, Res.HZBMips, Res.IndirectArgs, Res.CullStats,
                                                Res.VisibleIndices, Res.VisibleCount, Res.InstanceIndexBase);
        if (rc != UR_OK && PassError == UR_OK) PassError = rc;
    });

    // ---- Build HZB (after the G-buffer pass; only with HZB and depth prepass enabled, :996) ------------------------
    struct FHZBPassData
    {
        uint32 Width = 0, Height = 0, MipCount = 0, SourceWidth = 0, SourceHeight = 0;
        bool bShard = false;
    };
    if (bHZBEnabled && Options.bDoDepthPrepass) {
        Graph.AddPass<FHZBPassData>("Build HZB", [&](FHZBPassData& Data, FRGPassBuilder& Builder)
        {
            Data.Width = HZBWidth;
            Data.Height = HZBHeight;
            Data.MipCount = Res.HZBMipCount;
            Data.SourceWidth = Res.Width;
            Data.SourceHeight = Res.Height;
            Data.bShard = Options.bShardHZB;
            Builder.ReadTexture(DepthHandle, RG_STATE_NON_PIXEL_SHADER_RESOURCE);
            Builder.WriteTexture(HZBHandle, RG_STATE_UNORDERED_ACCESS);
            if (Options.bAsyncCompute) Builder.AsyncCompute();
        }, [this, &Res](const FHZBPassData& Data, FHIPCommandContext& Cmd)
        {
            if (Data.MipCount == 0) return;
            int rc;
            if (Data.bShard) { // this rank's piece rows of the wide launch; the ranks' exchange and the tail are the caller's (it holds the communicator)
                uint32_t Row0 = 0, Rows = 0;
                rc = ur_hzb_band_pieces(Data.SourceHeight, static_cast<uint32_t>(Cmd.GetWorldSize()), static_cast<uint32_t>(Cmd.GetRank()), &Row0, &Rows);
                if (rc == UR_OK) rc = ur_build_hzb_band(Cmd.GetContext(), Res.DepthFull, Data.SourceWidth, Data.SourceHeight, Res.HZB, Res.HZBMips, Data.MipCount, Row0, Rows);
            } else {
                rc = ur_build_hzb(Cmd.GetContext(), Res.DepthFull, Data.SourceWidth, Data.SourceHeight, Res.HZB, Res.HZBMips, Data.MipCount);
            }
            if (rc != UR_OK && PassError == UR_OK) PassError = rc;
            Res.HZBState = RG_STATE_NON_PIXEL_SHADER_RESOURCE; // :1209
            if (rc == UR_OK) bHZBReady = true;                  // :1210
        });
    }

    const bool bSky = Options.bSkyEnabled && Res.DepthBand != nullptr;
    const bool bFused = Options.bFuseLightingAndSky && bSky;

    // ---- Lighting (fullscreen, additive) --------------------------------------------------------------------------
    struct FLightingPassData
    {
        bool bUseShadows = false;
        bool bFusedSky = false;
        ur_scene_constants Scene;
        ur_sky_constants Sky;
    };
    Graph.AddPass<FLightingPassData>("Lighting", [&](FLightingPassData& Data, FRGPassBuilder& Builder)
    {
        Data.bUseShadows = Options.bRenderShadows;
        Data.bFusedSky = bFused;
        Data.Scene = Constants.Scene;
        Data.Sky = Constants.Sky;
        if (!Data.bUseShadows) Data.Scene.ShadowStrength = 0.0f; // bShadowsEnabled ? ShadowStrength : 0 (:3777)
        Builder.ReadTexture(GBufferHandles[0], RG_STATE_PIXEL_SHADER_RESOURCE);
        Builder.ReadTexture(GBufferHandles[1], RG_STATE_PIXEL_SHADER_RESOURCE);
        Builder.ReadTexture(GBufferHandles[2], RG_STATE_PIXEL_SHADER_RESOURCE);
        if (Data.bUseShadows) Builder.ReadTexture(ShadowHandle, RG_STATE_PIXEL_SHADER_RESOURCE);
        if (Data.bFusedSky) Builder.ReadTexture(DepthHandle, RG_STATE_DEPTH_READ);
        Builder.WriteTexture(LightingHandle, RG_STATE_RENDER_TARGET);
    }, [this, &Res, &Options](const FLightingPassData& Data, FHIPCommandContext& Cmd)
    {
        const bool bTimed = Options.bTimeLighting && LightingTimer;
        if (bTimed) LightingTimer(Cmd.GetStream(), true);
        int rc;
        if (Data.bFusedSky)
            rc = ur_deferred_lighting_sky(Cmd.GetContext(), &Data.Scene, &Data.Sky, Res.GBufferA, Res.GBufferB, Res.GBufferC, Res.DepthBand, &Res.Tables,
                                          Res.LightingBand, Res.Width, Res.Height, Res.Row0, Res.Rows);
        else
            rc = ur_deferred_lighting(Cmd.GetContext(), &Data.Scene, Res.GBufferA, Res.GBufferB, Res.GBufferC, &Res.Tables, Res.LightingBand, Res.Width,
                                      Res.Height, Res.Row0, Res.Rows);
        if (bTimed) LightingTimer(Cmd.GetStream(), false);
        if (rc != UR_OK && PassError == UR_OK) PassError = rc;
    });

    // ---- Sky --------------------------------------------------------------------------------------------------------
    struct FSkyPassData
    {
        bool bEnabled = false;
        ur_sky_constants Sky;
    };
    Graph.AddPass<FSkyPassData>("Sky", [&](FSkyPassData& Data, FRGPassBuilder& Builder)
    {
        Data.bEnabled = bSky && !bFused;
        Data.Sky = Constants.Sky;
        if (Data.bEnabled) {
            Builder.ReadTexture(DepthHandle, RG_STATE_DEPTH_READ);
            Builder.WriteTexture(LightingHandle, RG_STATE_RENDER_TARGET);
        }
    }, [this, &Res](const FSkyPassData& Data, FHIPCommandContext& Cmd)
    {
        if (!Data.bEnabled) return;
        const int rc = ur_sky_atmosphere(Cmd.GetContext(), &Data.Sky, Res.DepthBand, Res.LightingBand, Res.Width, Res.Height, Res.Row0, Res.Rows);
        if (rc != UR_OK && PassError == UR_OK) PassError = rc;
    });

    // ---- Tonemap (DeferredRenderer.cpp:1449-1513 with TAA, auto exposure and CAS off): Lighting -> LDR band --------------
    struct FTonemapPassData
    {
        bool bEnabled = false;
        ur_tonemap_constants K;
    };
    if (Options.bTonemap && Res.TonemapBand) {
        const FRGResourceHandle TonemapHandle = Graph.ImportTexture("TonemapOutput", Res.TonemapBand, &Res.TonemapState, {Res.Width, Res.Rows, RG_FORMAT_R8G8B8A8_UNORM_SRGB});
        Graph.AddPass<FTonemapPassData>("Tonemap", [&](FTonemapPassData& Data, FRGPassBuilder& Builder)
        {
            Data.bEnabled = true;
            Data.K = Constants.Tonemap;
            Builder.ReadTexture(LightingHandle, RG_STATE_PIXEL_SHADER_RESOURCE);
            Builder.WriteTexture(TonemapHandle, RG_STATE_RENDER_TARGET);
        }, [this, &Res](const FTonemapPassData& Data, FHIPCommandContext& Cmd)
        {
            const int rc = ur_tonemap(Cmd.GetContext(), &Data.K, Res.LightingBand, nullptr, Res.TonemapBand, Res.Width, Res.Rows);
            if (rc != UR_OK && PassError == UR_OK) PassError = rc;
            Res.LightingState = RG_STATE_RENDER_TARGET; // the reference transitions the lighting buffer back (:1511-1512)
        });
    }

    Graph.Execute(Cmd);
    LastReport = Graph.GetLastExecutionReport();
    return PassError;
}

// ---------------------------------------------------------------------------------------------------------------------
// C face
// ---------------------------------------------------------------------------------------------------------------------
struct ur_frame
{
    FHIPDevice Device;
    FHIPCommandContext Cmd;
    FHotPathRenderer Renderer;
    FHotPathResources Res;
    hipStream_t AsyncStream = nullptr;
    ur_ctx* AsyncCtx = nullptr;
    int DeviceIndex = 0;
    struct FLightEvents { hipEvent_t first, second, after; bool has_after; bool on_dispatch; };
    std::vector<FLightEvents> LightEvents; // ring: an event pair around the Lighting pass + one more right behind it (what a record costs)
    size_t LightHead = 0, LightCount = 0;
    bool bRecordAfter = false; // this frame's bracket gets the third event (UR_FRAME_TIME_LIGHTING_RECORD_COST)
    bool bKernelEvents = false; // UR_FRAME_TIME_LIGHTING_KERNEL: the pair rides on the Lighting dispatch itself, nothing is recorded around it
    bool bStartOnCull = false;  // ... and this frame's START event was handed to the cull launch directly in front of the Lighting launch
    ur_frame(ur_ctx* Ctx, hipStream_t Stream, uint32 Frames, int Rank, int World) : Cmd(Ctx, Stream, Frames, Rank, World), Renderer(&Device) {}
};

extern "C" {

ur_frame* ur_frame_create(ur_ctx* ctx, void* stream, uint32_t frames_in_flight, int rank, int world_size)
{
    if (!ctx) return nullptr;
    ur_frame* f = new ur_frame(ctx, static_cast<hipStream_t>(stream), frames_in_flight, rank, world_size);
    f->Renderer.SetLightingTimer([f](hipStream_t s, bool begin) {
        constexpr size_t kRing = 1024;
        if (f->LightEvents.size() < kRing && begin && f->LightCount == f->LightEvents.size()) {
            hipEvent_t a = nullptr, b = nullptr, c = nullptr;
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess && hipEventCreate(&c) == hipSuccess) f->LightEvents.push_back({a, b, c, false, false});
        }
        if (f->LightEvents.empty()) return;
        if (f->bKernelEvents) {
            if (begin) {
                f->LightHead = f->LightCount % f->LightEvents.size();
                // stop = bound to the Lighting kernel's own dispatch (its completion signal's end stamp). start = the end stamp of the
                // dispatch directly in front of it when that is this frame's cull launch (ur_frame_render handed it the event: NOTHING
                // enters the queue for the measurement), else a marker the runtime puts in front of the kernel (~8 us of queue time).
                // (One event alone measures nothing on this runtime: hipEventElapsedTime(e, e) is 0.)
                // (a cull call that launched nothing took no event: the marker form then, never a stamp left over from an earlier use of the slot)
                if (f->bStartOnCull && !ur_time_cull_carried(f->Cmd.GetContext())) f->bStartOnCull = false;
                (void)ur_time_next_lighting(f->Cmd.GetContext(), f->bStartOnCull ? nullptr : f->LightEvents[f->LightHead].first, f->LightEvents[f->LightHead].second);
            } else {
                (void)ur_time_next_lighting(f->Cmd.GetContext(), nullptr, nullptr); // (a launch that failed validation consumed nothing)
                f->LightEvents[f->LightHead].has_after = false;
                f->LightEvents[f->LightHead].on_dispatch = true;
                ++f->LightCount;
            }
            return;
        }
        if (begin) {
            f->LightHead = f->LightCount % f->LightEvents.size();
            (void)hipEventRecord(f->LightEvents[f->LightHead].first, s);
        } else {
            (void)hipEventRecord(f->LightEvents[f->LightHead].second, s);
            // a third record with nothing in front of it: second -> after is what one event record adds to the bracket
            f->LightEvents[f->LightHead].has_after = f->bRecordAfter;
            f->LightEvents[f->LightHead].on_dispatch = false;
            if (f->bRecordAfter) (void)hipEventRecord(f->LightEvents[f->LightHead].after, s);
            ++f->LightCount;
        }
    });
    return f;
}

uint32_t ur_frame_lighting_times_ex(ur_frame* f, float* out_ms, float* out_record_ms, uint32_t cap)
{
    if (!f) return 0;
    const size_t n = f->LightCount < f->LightEvents.size() ? f->LightCount : f->LightEvents.size();
    uint32_t k = 0;
    for (size_t i = 0; i < n && k < cap; ++i) {
        float ms = 0.0f, rec = 0.0f;
        if (hipEventElapsedTime(&ms, f->LightEvents[i].first, f->LightEvents[i].second) != hipSuccess) continue;
        if (out_record_ms) {
            rec = -1.0f; // no third event on this sample
            if (f->LightEvents[i].has_after && hipEventElapsedTime(&rec, f->LightEvents[i].second, f->LightEvents[i].after) != hipSuccess) rec = -1.0f;
            out_record_ms[k] = rec;
        }
        out_ms[k++] = ms;
    }
    f->LightCount = 0;
    return k;
}

uint32_t ur_frame_lighting_times(ur_frame* f, float* out_ms, uint32_t cap) { return ur_frame_lighting_times_ex(f, out_ms, nullptr, cap); }

void ur_frame_destroy(ur_frame* f)
{
    if (!f) return;
    if (f->AsyncStream) (void)hipStreamSynchronize(f->AsyncStream);
    if (f->AsyncCtx) ur_destroy(f->AsyncCtx);
    if (f->AsyncStream) (void)hipStreamDestroy(f->AsyncStream);
    for (auto& e : f->LightEvents) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); (void)hipEventDestroy(e.after); }
    delete f;
}

int ur_frame_render(ur_frame* f, const ur_frame_resources* r, const uint32_t* culling_constants, const ur_scene_constants* scene,
                    const ur_sky_constants* sky, uint32_t flags)
{
    if (!f || !r || !culling_constants || !scene || !sky) return UR_EINVAL;
    FHotPathResources& R = f->Res; // resource states persist across frames, like the renderer's member variables
    R.Width = r->width; R.Height = r->height; R.Row0 = r->row0; R.Rows = r->rows;
    R.GBufferA = const_cast<ur_half4*>(r->gbuffer_a);
    R.GBufferB = const_cast<ur_half4*>(r->gbuffer_b);
    R.GBufferC = const_cast<uint32*>(r->gbuffer_c);
    R.DepthBand = const_cast<float*>(r->depth_band);
    R.LightingBand = r->lighting_band;
    R.TonemapBand = r->tonemap_band;
    R.DepthFull = const_cast<float*>(r->depth_full);
    R.HZB = r->hzb;
    std::memcpy(R.HZBMips, r->hzb_mips, sizeof(R.HZBMips));
    R.HZBMipCount = r->hzb_mip_count;
    R.Tables = r->tables;
    R.ModelBounds = const_cast<ur_float4*>(r->model_bounds);
    R.IndirectArgs = r->indirect_args;
    R.IndirectCommandCount = r->indirect_command_count;
    R.InstanceIndexBase = r->instance_index_base;
    R.VisibleIndices = r->visible_indices;
    R.VisibleCount = r->visible_count;
    R.CullStats = r->cull_stats;

    FHotPathFrameConstants K;
    std::memcpy(K.CullingConstants, culling_constants, sizeof(K.CullingConstants));
    K.Scene = *scene;
    K.Sky = *sky;
    FHotPathOptions O;
    O.bEnableIndirectDraw = (flags & UR_FRAME_INDIRECT_DRAW) != 0;
    O.bHZBEnabled = (flags & UR_FRAME_HZB) != 0;
    O.bDoDepthPrepass = (flags & UR_FRAME_DEPTH_PREPASS) != 0;
    O.bRenderShadows = (flags & UR_FRAME_SHADOWS) != 0;
    O.bSkyEnabled = (flags & UR_FRAME_SKY) != 0;
    O.bFuseLightingAndSky = (flags & UR_FRAME_FUSE_LIGHTING_SKY) != 0;
    O.bTonemap = (flags & UR_FRAME_TONEMAP) != 0;
    O.bShardHZB = (flags & UR_FRAME_HZB_SHARD) != 0 && f->Cmd.GetWorldSize() > 1;
    O.bAsyncCompute = (flags & UR_FRAME_ASYNC_COMPUTE) != 0;
    if (O.bAsyncCompute && !f->AsyncCtx) { // second stream + a context bound to it, created on first use
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipStreamCreateWithPriority(&f->AsyncStream, hipStreamNonBlocking, -1) != hipSuccess) return UR_EHIP; // high priority: its short kernels slot in beside the lighting kernel
        f->AsyncCtx = ur_create(dev, f->AsyncStream);
        if (!f->AsyncCtx) return UR_EHIP;
        f->Cmd.SetAsyncCompute(f->AsyncCtx, f->AsyncStream);
    }
    O.bTimeLighting = (flags & (UR_FRAME_TIME_LIGHTING | UR_FRAME_TIME_LIGHTING_RECORD_COST | UR_FRAME_TIME_LIGHTING_KERNEL)) != 0;
    f->bKernelEvents = (flags & UR_FRAME_TIME_LIGHTING_KERNEL) != 0;
    f->bRecordAfter = (flags & UR_FRAME_TIME_LIGHTING_RECORD_COST) != 0;
    O.bGpuTiming = (flags & UR_FRAME_GPU_TIMING) != 0;
    O.bGraphDump = (flags & UR_FRAME_GRAPH_DUMP) != 0;
    O.bBarrierLogs = (flags & UR_FRAME_BARRIER_LOGS) != 0;
    f->Cmd.SetJoinAsyncAtEnd((flags & UR_FRAME_ASYNC_NO_JOIN) == 0);
    f->Cmd.BeginFrame();
    // Launch scheduling across two passes (include/ur_hotpath.h, ur_defer_hzb_tail): only when both run on the main stream
    const bool chain_with_lighting = (flags & UR_FRAME_HZB_WITH_LIGHTING) != 0 && !O.bAsyncCompute;
    const bool tail_with_lighting = (chain_with_lighting || (flags & UR_FRAME_HZB_TAIL_WITH_LIGHTING) != 0) && !O.bAsyncCompute;
    if (tail_with_lighting) (void)ur_defer_hzb_tail(f->Cmd.GetContext(), chain_with_lighting ? 2 : 1);
    f->bStartOnCull = false;
    if (f->bKernelEvents && chain_with_lighting && O.bEnableIndirectDraw && O.bHZBEnabled && O.bDoDepthPrepass && R.IndirectArgs && R.ModelBounds &&
        R.IndirectCommandCount != 0) {
        // Two launches in this frame, the cull and the Lighting launch that carries Build HZB: the cull's own completion stamp is the
        // start of the Lighting measurement. (Events of the ring are created here if this is its first use.)
        constexpr size_t kRing = 1024;
        if (f->LightEvents.size() < kRing && f->LightCount == f->LightEvents.size()) {
            hipEvent_t a = nullptr, b = nullptr, c = nullptr;
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess && hipEventCreate(&c) == hipSuccess) f->LightEvents.push_back({a, b, c, false, false});
        }
        if (!f->LightEvents.empty() && ur_time_next_cull(f->Cmd.GetContext(), f->LightEvents[f->LightCount % f->LightEvents.size()].first) == UR_OK)
            f->bStartOnCull = true;
    }
    const int rc = f->Renderer.RenderFrame(f->Cmd, R, K, O);
    (void)ur_time_next_cull(f->Cmd.GetContext(), nullptr); // (a frame whose cull pass did not run consumed nothing)
    if (tail_with_lighting) {
        const int rc2 = ur_defer_hzb_tail(f->Cmd.GetContext(), 0); // launches the tail on its own if no Lighting launch took it
        // a riding tail that gave up waiting (a bounded wait inside an earlier Lighting launch) is reported here, once: UR_ETIMEOUT
        const int rc3 = ur_flush(f->Cmd.GetContext());
        return rc != UR_OK ? rc : (rc2 != UR_OK ? rc2 : rc3);
    }
    return rc;
}

void ur_frame_join_async(ur_frame* f) { if (f) f->Cmd.JoinAsyncCompute(); }
int ur_frame_hzb_ready(const ur_frame* f) { return f && f->Renderer.IsHZBReady() ? 1 : 0; }
void ur_frame_reset_hzb(ur_frame* f) { if (f) f->Renderer.ResetHZB(); }

static uint32_t copy_out(const std::string& s, char* buf, uint32_t cap)
{
    if (buf && cap) {
        const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return static_cast<uint32_t>(s.size() + 1);
}

uint32_t ur_frame_report(const ur_frame* f, char* buf, uint32_t cap)
{
    std::ostringstream s;
    if (f)
        for (const auto& p : f->Renderer.GetLastReport())
            s << p.Name << '|' << (p.bCulled ? 1 : 0) << '|' << p.Transitions << '|' << (p.bAsync ? 1 : 0) << '|' << p.CrossStreamWaits << '\n';
    return copy_out(s.str(), buf, cap);
}

uint32_t ur_rg_timing_stats(char* buf, uint32_t cap)
{
    std::ostringstream s;
    for (const auto& t : FRenderGraph::GetGpuTimingStats()) s << t.Name << '|' << t.AvgMs << '|' << t.MinMs << '|' << t.MaxMs << '|' << t.SampleCount << '\n';
    return copy_out(s.str(), buf, cap);
}

} // extern "C"
