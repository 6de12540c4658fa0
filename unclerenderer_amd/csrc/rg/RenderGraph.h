// RenderGraph.h — the reference's Render-Graph pass/resource API on a HIP stream backend.
//
// Public surface kept from Source/Render/RenderGraph.h:16-212 so pass code written against the reference drops in:
//   FRGTextureDesc, FRGResourceHandle, ERGResourceAccess, FRenderGraph::{SetDevice, ImportTexture, AddPass<PassData>,
//   Execute, SetDebugRecording, SetGraphDumpEnabled, SetResourceLifetimeLogging, SetBarrierLoggingEnabled,
//   SetGpuTimingEnabled, Set/GetGpuTimingWindowSeconds, Set/GetGpuTimingDisplayCount, GetGpuTimingStats,
//   AddExternalGpuTimingSample}, FRGPassBuilder::{CreateTexture, ReadTexture, WriteTexture, KeepAlive}.
// What changes is the device side of every signature:
//   ID3D12Resource*          -> FRGResourcePtr   (a HIP device pointer; "textures" are linear row-major buffers)
//   D3D12_RESOURCE_STATES    -> ERGResourceState (same bit values, so state logic reads the same)
//   DXGI_FORMAT              -> ERGFormat
//   FDX12CommandContext      -> FHIPCommandContext (stream + ur_ctx + frame slot + rank)
//   FDX12Device              -> FHIPDevice (transient allocations)
// Semantics reproduced (Source/Render/RenderGraph.cpp:214-517): insertion-order execution; backward pass culling from
// read/external resources and KeepAlive; per-usage state tracking that updates the owner's state variable; lazily
// acquired, pooled transient textures released after their last pass; one GPU timestamp pair per live pass, harvested
// once the frame slot has completed, fed into a sliding-window avg/min/max. Barriers themselves are stream order.
// Deliberate fixes: PassData destructors run (the reference placement-news into a byte vector and never destroys it,
// RenderGraph.h:66-69), and the process-wide pools live behind one accessor instead of unsynchronised statics.
#pragma once

#include <chrono>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

struct ur_ctx;
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;

using uint32 = uint32_t;
using int32 = int32_t;
using uint64 = uint64_t;

typedef void* FRGResourcePtr;

// Bit values of D3D12_RESOURCE_STATES for the states the reference's passes use.
enum ERGResourceState : uint32 {
    RG_STATE_COMMON = 0,
    RG_STATE_RENDER_TARGET = 0x4,
    RG_STATE_UNORDERED_ACCESS = 0x8,
    RG_STATE_DEPTH_WRITE = 0x10,
    RG_STATE_DEPTH_READ = 0x20,
    RG_STATE_NON_PIXEL_SHADER_RESOURCE = 0x40,
    RG_STATE_PIXEL_SHADER_RESOURCE = 0x80,
    RG_STATE_INDIRECT_ARGUMENT = 0x200,
    RG_STATE_COPY_DEST = 0x400,
    RG_STATE_COPY_SOURCE = 0x800,
};

enum ERGFormat : uint32 {
    RG_FORMAT_UNKNOWN = 0,
    RG_FORMAT_R16G16B16A16_FLOAT = 10,
    RG_FORMAT_R8G8B8A8_UNORM_SRGB = 29,
    RG_FORMAT_R32_FLOAT = 41,       // also what the depth buffer looks like through its SRV
    RG_FORMAT_D24_UNORM_S8_UINT = 45,
    RG_FORMAT_R16G16_UNORM = 35,
};
uint32 RGFormatBytesPerTexel(ERGFormat Format);
const char* RGResourceStateToString(uint32 State);

enum ERGResourceFlags : uint32 {
    RG_FLAG_NONE = 0,
    RG_FLAG_ALLOW_RENDER_TARGET = 0x1,
    RG_FLAG_ALLOW_DEPTH_STENCIL = 0x2,
    RG_FLAG_ALLOW_UNORDERED_ACCESS = 0x4,
};

// Transient-texture allocator. The default implementation is hipMalloc/hipFree; tests inject a host one.
class FHIPDevice
{
public:
    virtual ~FHIPDevice() = default;
    virtual FRGResourcePtr Allocate(size_t Bytes);
    virtual void Free(FRGResourcePtr Ptr);
};

// What a pass execute-lambda receives instead of FDX12CommandContext (RHI/DX12CommandContext.h:10-43).
class FHIPCommandContext
{
public:
    FHIPCommandContext(ur_ctx* InCtx, hipStream_t InStream, uint32 InFrameCount = 3, int InRank = 0, int InWorldSize = 1)
        : Ctx(InCtx), Stream(InStream), FrameCount(InFrameCount ? InFrameCount : 1), Rank(InRank), WorldSize(InWorldSize) {}

    // The context / stream a pass must launch on. While the graph executes a pass flagged AsyncCompute these return
    // the async-compute pair (a second HIP stream), otherwise the main pair.
    ur_ctx* GetContext() const { return bAsyncLane ? AsyncCtx : Ctx; }
    hipStream_t GetStream() const { return bAsyncLane ? AsyncStream : Stream; }
    hipStream_t GetMainStream() const { return Stream; }
    hipStream_t GetAsyncStream() const { return AsyncStream; }
    // MI355X extension: give the context a second stream (+ a ur_ctx bound to it) for passes flagged AsyncCompute.
    void SetAsyncCompute(ur_ctx* InCtx, hipStream_t InStream) { AsyncCtx = InCtx; AsyncStream = InStream; }
    bool HasAsyncCompute() const { return AsyncCtx != nullptr && AsyncStream != nullptr; }
    void SetAsyncLane(bool bAsync) { bAsyncLane = bAsync && HasAsyncCompute(); }
    // Join policy. true (default): Execute() ends with "main stream waits for the async stream", so anything enqueued on
    // the main stream afterwards sees the async passes' results. false: the caller fences explicitly with
    // JoinAsyncCompute() before main-stream work that overwrites what the async passes read or consumes what they wrote
    // (like a D3D12 async-compute fence); successive frames' async passes are still ordered among themselves.
    void SetJoinAsyncAtEnd(bool bJoin) { bJoinAsyncAtEnd = bJoin; }
    bool GetJoinAsyncAtEnd() const { return bJoinAsyncAtEnd; }
    void JoinAsyncCompute(); // main stream waits for everything submitted to the async stream so far
    bool IsAsyncLane() const { return bAsyncLane; }
    uint32 GetCurrentFrameIndex() const { return FrameIndex; }
    uint32 GetFrameCount() const { return FrameCount; }
    int GetRank() const { return Rank; }
    int GetWorldSize() const { return WorldSize; }
    // BeginFrame: advance to the next frame slot (the D3D12 context waits that slot's fence here).
    void BeginFrame() { FrameIndex = (FrameIndex + 1) % FrameCount; ++FrameNumber; }
    uint64 GetFrameNumber() const { return FrameNumber; }
    // Stream-ordered backend: a transition is bookkeeping only; the count is kept for tests/logging.
    void TransitionResources(uint32 Count) { TransitionCount += Count; }
    uint64 GetTransitionCount() const { return TransitionCount; }

private:
    ur_ctx* Ctx = nullptr;
    hipStream_t Stream = nullptr;
    ur_ctx* AsyncCtx = nullptr;
    hipStream_t AsyncStream = nullptr;
    bool bAsyncLane = false;
    bool bJoinAsyncAtEnd = true;
    hipEvent_t JoinEvent = nullptr;
    uint32 FrameCount = 3;
    uint32 FrameIndex = 0;
    uint64 FrameNumber = 0;
    int Rank = 0;
    int WorldSize = 1;
    uint64 TransitionCount = 0;
};

struct FRGTextureDesc
{
    uint32 Width = 0;
    uint32 Height = 0;
    ERGFormat Format = RG_FORMAT_UNKNOWN;
};

struct FRGResourceHandle
{
    uint32 Id = UINT32_MAX;
    explicit operator bool() const { return Id != UINT32_MAX; }
};

enum class ERGResourceAccess
{
    Read,
    Write,
};

class FRenderGraph;

class FRGPassBuilder
{
public:
    FRGPassBuilder(FRenderGraph& InGraph, void* InPass) : Graph(&InGraph), Pass(InPass) {}

    FRGResourceHandle CreateTexture(const std::string& Name, const FRGTextureDesc& Desc);
    FRGResourceHandle ReadTexture(const FRGResourceHandle& Handle, uint32 RequiredState = RG_STATE_PIXEL_SHADER_RESOURCE);
    FRGResourceHandle WriteTexture(const FRGResourceHandle& Handle, uint32 RequiredState = RG_STATE_RENDER_TARGET);
    void KeepAlive();
    // MI355X extension (no reference counterpart): run this pass on the context's async-compute stream. The graph
    // orders it against main-stream passes by the declared resource usages (RAW/WAR/WAW across streams become event
    // waits); the async stream forks from the main stream at Execute() begin and joins it at Execute() end.
    void AsyncCompute();

private:
    FRenderGraph* Graph = nullptr;
    void* Pass = nullptr;
};

class FRenderGraph
{
public:
    FRenderGraph();
    ~FRenderGraph();
    FRenderGraph(const FRenderGraph&) = delete;
    FRenderGraph& operator=(const FRenderGraph&) = delete;

    struct FGpuPassTimingStats
    {
        std::string Name;
        double AvgMs = 0.0;
        double MinMs = 0.0;
        double MaxMs = 0.0;
        uint32 SampleCount = 0;
    };

    void SetDevice(FHIPDevice* InDevice) { Device = InDevice; }

    FRGResourceHandle ImportTexture(const std::string& Name, FRGResourcePtr Resource, uint32* StatePtr, const FRGTextureDesc& Desc);

    // Setup runs immediately with (PassData&, FRGPassBuilder&); Execute is stored and later called with
    // (const PassData&, FHIPCommandContext&) — same contract as the reference (RenderGraph.h:61-80).
    template <typename PassData, typename SetupFunc, typename ExecuteFunc>
    void AddPass(const std::string& Name, SetupFunc&& Setup, ExecuteFunc&& Execute)
    {
        FPass& Pass = NewPass(Name);
        auto Data = std::make_shared<PassData>();
        FRGPassBuilder Builder = MakeBuilder(Pass);
        Setup(*Data, Builder);
        Pass.Run = [Data, Fn = std::forward<ExecuteFunc>(Execute)](FHIPCommandContext& Cmd) { Fn(static_cast<const PassData&>(*Data), Cmd); };
    }

    void Execute(FHIPCommandContext& CmdContext);

    void SetDebugRecording(bool bEnable) { bEnableDebugRecording = bEnable; }
    void SetGraphDumpEnabled(bool bEnable) { bEnableGraphDump = bEnable; }
    void SetResourceLifetimeLogging(bool bEnable) { bEnableResourceLifetimeLog = bEnable; }
    void SetBarrierLoggingEnabled(bool bEnable) { bEnableBarrierLogs = bEnable; }
    void SetGpuTimingEnabled(bool bEnable) { bEnableGpuTiming = bEnable; }

    static void SetGpuTimingWindowSeconds(double Seconds);
    static double GetGpuTimingWindowSeconds();
    static void SetGpuTimingDisplayCount(uint32 Count);
    static uint32 GetGpuTimingDisplayCount();
    static const std::vector<FGpuPassTimingStats>& GetGpuTimingStats();
    static void AddExternalGpuTimingSample(const std::string& Name, double Milliseconds);

    // ---- introspection (new; used by tests and the graph dump) ----
    struct FPassReport
    {
        std::string Name;
        bool bCulled = false;
        uint32 Transitions = 0;
        double CpuMs = 0.0;
        bool bAsync = false;        // ran on the async-compute stream
        uint32 CrossStreamWaits = 0; // event waits inserted for hazards against the other stream
    };
    const std::vector<FPassReport>& GetLastExecutionReport() const { return Report; }
    FRGResourcePtr GetResource(const FRGResourceHandle& Handle) const;
    static size_t GetPooledTextureCount();
    static void ReleaseTransientPool(FHIPDevice* Device); // frees every pooled texture (shutdown)
    static void SetLogSink(std::function<void(const std::string&)> Sink);

private:
    friend class FRGPassBuilder;

    struct FUsage
    {
        uint32 Resource = 0;
        uint32 RequiredState = RG_STATE_COMMON;
        ERGResourceAccess Access = ERGResourceAccess::Read;
    };
    struct FTexture
    {
        std::string Name;
        FRGTextureDesc Desc;
        uint32 Flags = RG_FLAG_NONE;
        FRGResourcePtr Resource = nullptr;
        uint32* ExternalState = nullptr;
        uint32 CurrentState = RG_STATE_COMMON;
        int32 FirstUsePass = -1;
        int32 LastUsePass = -1;
        int32 PoolIndex = -1;
        bool bExternal = false;
    };
    struct FPass
    {
        std::string Name;
        std::function<void(FHIPCommandContext&)> Run;
        std::vector<FUsage> Usages;
        bool bForceExecute = false;
        bool bAsync = false;
    };

    FPass& NewPass(const std::string& Name);
    FRGPassBuilder MakeBuilder(FPass& Pass);
    FRGResourceHandle RegisterTexture(const std::string& Name, const FRGTextureDesc& Desc);
    void RegisterUsage(FPass& Pass, const FRGResourceHandle& Handle, uint32 RequiredState, ERGResourceAccess Access);
    bool AcquireTransient(FTexture& Texture, uint32 InitialState);
    void ReleaseTransient(FTexture& Texture);
    void Dump(const std::vector<char>& PassLive, const std::vector<char>& ResourceLive) const;

    FHIPDevice* Device = nullptr;
    std::vector<FTexture> Textures;
    std::deque<FPass> Passes; // deque: references handed to builders stay valid
    std::vector<FPassReport> Report;

    bool bEnableDebugRecording = false;
    bool bEnableGraphDump = false;
    bool bEnableResourceLifetimeLog = false;
    bool bEnableBarrierLogs = false;
    bool bEnableGpuTiming = false;
};
