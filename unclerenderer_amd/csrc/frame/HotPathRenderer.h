// HotPathRenderer — the four hot passes of FDeferredRenderer::RenderFrame wired onto the render graph.
//
// Reference wiring (Source/Render/DeferredRenderer.cpp): "GPU Culling" :508-542 (+ FRenderer::ConfigureHZBOcclusion /
// DispatchGpuCulling, Renderer.cpp:384-472), "Build HZB" :982-1212, "Lighting" :1214-1255, "Sky" :1257-1296.
// Pass names, PassData structs, declared usages/states and pass order are the reference's; the execute lambdas call
// the C-ABI (include/ur_hotpath.h) instead of recording D3D12 commands. The passes between them (shadow map, depth
// prepass, G-buffer raster, post FX) are out of scope: their outputs arrive as imported textures.
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "../../../include/ur_hotpath.h"
#include "../rg/RenderGraph.h"

// Device buffers owned by the caller (the renderer that rasterised the G-buffer). States mirror the variables the
// reference keeps next to each resource (DepthBufferState, GBufferStates[], HZBState, LightingBufferState, ...).
struct FHotPathResources
{
    uint32 Width = 0, Height = 0;      // full frame
    uint32 Row0 = 0, Rows = 0;         // screen band shaded by this rank (whole frame: 0, Height)
    // band-local images
    ur_half4* GBufferA = nullptr;
    ur_half4* GBufferB = nullptr;
    uint32* GBufferC = nullptr;
    float* DepthBand = nullptr;
    ur_half4* LightingBand = nullptr;
    uint32* TonemapBand = nullptr;    // optional: R8G8B8A8_UNORM output of the Tonemap pass for this band
    // full-frame depth for the replicated HZB build, and the HZB itself
    float* DepthFull = nullptr;
    float* HZB = nullptr;
    ur_mip_desc HZBMips[UR_MAX_HZB_MIPS] = {};
    uint32 HZBMipCount = 0;
    // lighting side tables
    ur_lighting_tables Tables = {};
    // GPU-driven draw data
    ur_float4* ModelBounds = nullptr;
    void* IndirectArgs = nullptr;
    uint32 IndirectCommandCount = 0;
    uint32 InstanceIndexBase = 0;
    uint32* VisibleIndices = nullptr; // optional (new): compacted ascending list
    uint32* VisibleCount = nullptr;
    uint32* CullStats = nullptr;      // optional: [frustum-culled, occluded]

    uint32 DepthState = RG_STATE_DEPTH_WRITE;
    uint32 GBufferStates[3] = {RG_STATE_RENDER_TARGET, RG_STATE_RENDER_TARGET, RG_STATE_RENDER_TARGET};
    uint32 ShadowState = RG_STATE_DEPTH_WRITE;
    uint32 HZBState = RG_STATE_UNORDERED_ACCESS;
    uint32 LightingState = RG_STATE_RENDER_TARGET;
    uint32 TonemapState = RG_STATE_RENDER_TARGET;
};

struct FHotPathFrameConstants
{
    uint32 CullingConstants[UR_CULL_CONSTANT_DWORDS] = {}; // packed like DispatchGpuCulling; dw 40-44 are filled per frame here
    ur_scene_constants Scene = {};
    ur_sky_constants Sky = {};
    ur_tonemap_constants Tonemap = {1u, 0u, 0.9f, 2.2f}; // bTonemapEnabled, auto exposure off, TonemapExposure, TonemapGamma (DeferredRenderer.h:193-196)
};

struct FHotPathOptions
{
    bool bEnableIndirectDraw = true;  // RendererConfig IndirectDraw
    bool bHZBEnabled = true;
    bool bShardHZB = false;           // several ranks: build only this rank's pieces of mips 0..4 (the caller gathers and runs the tail)
    bool bDoDepthPrepass = true;      // HZB is only built when the depth prepass ran (:996)
    bool bRenderShadows = true;
    bool bSkyEnabled = true;
    bool bFuseLightingAndSky = false; // MI355X fast path: one pass, same result as Lighting followed by Sky
    bool bTonemap = false;            // next row: Tonemap pass after Sky (TAA / auto exposure off)
    bool bAsyncCompute = false;       // MI355X: GPU Culling + Build HZB on the async-compute stream, overlapping Lighting
    bool bTimeLighting = false;       // HIP event pair around the Lighting pass only (bench roofline leg), see SetLightingTimer
    bool bGpuTiming = false;
    bool bGraphDump = false;
    bool bBarrierLogs = false;
};

class FHotPathRenderer
{
public:
    FHotPathRenderer(FHIPDevice* InDevice) : Device(InDevice) {}

    // Builds a fresh graph, adds the passes in the reference's order and executes it. Returns UR_OK or the first
    // error a pass reported. bHZBReady carries over between frames exactly like FDeferredRenderer::bHZBReady.
    int RenderFrame(FHIPCommandContext& Cmd, FHotPathResources& Res, const FHotPathFrameConstants& Constants, const FHotPathOptions& Options);

    // Optional hook: called right before / after the Lighting pass launches, on the pass's stream (used by the frame
    // object to bracket the dominant kernel with a HIP event pair without timing every pass).
    void SetLightingTimer(std::function<void(hipStream_t, bool /*begin*/)> Fn) { LightingTimer = std::move(Fn); }
    bool IsHZBReady() const { return bHZBReady; }
    void ResetHZB() { bHZBReady = false; }
    const std::vector<FRenderGraph::FPassReport>& GetLastReport() const { return LastReport; }

private:
    FHIPDevice* Device = nullptr;
    bool bHZBReady = false;
    int PassError = 0;
    std::vector<FRenderGraph::FPassReport> LastReport;
    std::function<void(hipStream_t, bool)> LightingTimer;
};
