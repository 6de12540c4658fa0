// RenderGraph.cpp — HIP-stream backend of the reference's FRenderGraph (see RenderGraph.h for the API contract and
// Source/Render/RenderGraph.cpp:214-517 for the semantics this file reproduces).

#include "RenderGraph.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <sstream>

namespace {

// Process-wide state (the reference keeps these as unsynchronised class statics, RenderGraph.cpp:11-17).
struct FPooledTexture
{
    FRGTextureDesc Desc;
    uint32 Flags = RG_FLAG_NONE;
    FRGResourcePtr Resource = nullptr;
    uint32 CurrentState = RG_STATE_COMMON;
    bool bInUse = false;
};

struct FTimedPass
{
    std::string Name;
    hipEvent_t Begin = nullptr;
    hipEvent_t End = nullptr;
};

struct FSlotTimings
{
    std::vector<FTimedPass> Passes; // event pairs are created once and reused
    uint32 Used = 0;
    bool bPending = false;
};

struct FTimingSample
{
    std::chrono::steady_clock::time_point Timestamp;
    double Milliseconds = 0.0;
};

// Events used for cross-stream ordering, recycled per frame slot (a slot is reused only after FrameCount frames).
struct FSyncEvents
{
    std::vector<hipEvent_t> Events;
    uint32 Used = 0;
    hipEvent_t Next()
    {
        if (Used == Events.size()) {
            hipEvent_t E = nullptr;
            if (hipEventCreateWithFlags(&E, hipEventDisableTiming) != hipSuccess) return nullptr;
            Events.push_back(E);
        }
        return Events[Used++];
    }
};

struct FGlobals
{
    std::unordered_map<uint32, FSyncEvents> SyncEvents;
    std::vector<FPooledTexture> TexturePool;
    std::unordered_map<uint32, FSlotTimings> SlotTimings;
    std::unordered_map<std::string, std::deque<FTimingSample>> Samples;
    std::vector<FRenderGraph::FGpuPassTimingStats> CachedStats;
    double WindowSeconds = 1.0;
    uint32 DisplayCount = 3;
    std::function<void(const std::string&)> LogSink;
};

FGlobals& G()
{
    static FGlobals Instance;
    return Instance;
}

void Log(const std::string& Line)
{
    if (G().LogSink) G().LogSink(Line);
    else std::fprintf(stderr, "[RG] %s\n", Line.c_str());
}

void RefreshStats(const std::chrono::steady_clock::time_point& Now)
{
    FGlobals& S = G();
    const auto Cutoff = Now - std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(std::max(0.1, S.WindowSeconds)));
    S.CachedStats.clear();
    for (auto It = S.Samples.begin(); It != S.Samples.end();) {
        auto& Q = It->second;
        while (!Q.empty() && Q.front().Timestamp < Cutoff) Q.pop_front();
        if (Q.empty()) {
            It = S.Samples.erase(It);
            continue;
        }
        FRenderGraph::FGpuPassTimingStats St;
        St.Name = It->first;
        St.SampleCount = static_cast<uint32>(Q.size());
        St.MinMs = St.MaxMs = Q.front().Milliseconds;
        double Sum = 0.0;
        for (const FTimingSample& V : Q) {
            Sum += V.Milliseconds;
            St.MinMs = std::min(St.MinMs, V.Milliseconds);
            St.MaxMs = std::max(St.MaxMs, V.Milliseconds);
        }
        St.AvgMs = Sum / static_cast<double>(Q.size());
        S.CachedStats.push_back(std::move(St));
        ++It;
    }
    std::sort(S.CachedStats.begin(), S.CachedStats.end(), [](const auto& A, const auto& B) { return A.AvgMs > B.AvgMs; });
}

// Timestamps of a frame slot are consumed when the slot comes round again and its last event has completed
// (the reference checks the slot's fence value, RenderGraph.cpp:713-718).
void HarvestSlot(uint32 Slot, bool bTimingEnabled)
{
    auto It = G().SlotTimings.find(Slot);
    if (It == G().SlotTimings.end() || !It->second.bPending) return;
    FSlotTimings& T = It->second;
    if (!bTimingEnabled) {
        T.bPending = false;
        return;
    }
    if (T.Used == 0 || hipEventQuery(T.Passes[T.Used - 1].End) != hipSuccess) return; // not finished yet: keep pending
    const auto Now = std::chrono::steady_clock::now();
    for (uint32 I = 0; I < T.Used; ++I) {
        float Ms = 0.0f;
        if (hipEventElapsedTime(&Ms, T.Passes[I].Begin, T.Passes[I].End) == hipSuccess) G().Samples[T.Passes[I].Name].push_back({Now, static_cast<double>(Ms)});
    }
    T.bPending = false;
    RefreshStats(Now);
}

} // namespace

uint32 RGFormatBytesPerTexel(ERGFormat Format)
{
    switch (Format) {
    case RG_FORMAT_R16G16B16A16_FLOAT: return 8;
    case RG_FORMAT_R8G8B8A8_UNORM_SRGB:
    case RG_FORMAT_R32_FLOAT:
    case RG_FORMAT_D24_UNORM_S8_UINT:
    case RG_FORMAT_R16G16_UNORM: return 4;
    default: return 0;
    }
}

const char* RGResourceStateToString(uint32 State)
{
    switch (State) {
    case RG_STATE_COMMON: return "COMMON";
    case RG_STATE_RENDER_TARGET: return "RENDER_TARGET";
    case RG_STATE_UNORDERED_ACCESS: return "UNORDERED_ACCESS";
    case RG_STATE_DEPTH_WRITE: return "DEPTH_WRITE";
    case RG_STATE_DEPTH_READ: return "DEPTH_READ";
    case RG_STATE_NON_PIXEL_SHADER_RESOURCE: return "NON_PIXEL_SHADER_RESOURCE";
    case RG_STATE_PIXEL_SHADER_RESOURCE: return "PIXEL_SHADER_RESOURCE";
    case RG_STATE_INDIRECT_ARGUMENT: return "INDIRECT_ARGUMENT";
    case RG_STATE_COPY_DEST: return "COPY_DEST";
    case RG_STATE_COPY_SOURCE: return "COPY_SOURCE";
    default: return "MIXED";
    }
}

FRGResourcePtr FHIPDevice::Allocate(size_t Bytes)
{
    void* P = nullptr;
    return hipMalloc(&P, Bytes) == hipSuccess ? P : nullptr;
}

void FHIPDevice::Free(FRGResourcePtr Ptr)
{
    if (Ptr) (void)hipFree(Ptr);
}

void FHIPCommandContext::JoinAsyncCompute()
{
    if (!HasAsyncCompute()) return;
    if (!JoinEvent && hipEventCreateWithFlags(&JoinEvent, hipEventDisableTiming) != hipSuccess) return;
    if (hipEventRecord(JoinEvent, AsyncStream) == hipSuccess) (void)hipStreamWaitEvent(Stream, JoinEvent, 0);
}

FRenderGraph::FRenderGraph() = default;

FRenderGraph::~FRenderGraph()
{
    // a graph destroyed without Execute must not strand pool entries
    for (FTexture& T : Textures)
        if (!T.bExternal && T.PoolIndex >= 0) ReleaseTransient(T);
}

void FRenderGraph::SetGpuTimingWindowSeconds(double Seconds) { G().WindowSeconds = std::max(0.1, Seconds); }
double FRenderGraph::GetGpuTimingWindowSeconds() { return G().WindowSeconds; }
void FRenderGraph::SetGpuTimingDisplayCount(uint32 Count) { G().DisplayCount = std::max(1u, Count); }
uint32 FRenderGraph::GetGpuTimingDisplayCount() { return G().DisplayCount; }
const std::vector<FRenderGraph::FGpuPassTimingStats>& FRenderGraph::GetGpuTimingStats() { return G().CachedStats; }
void FRenderGraph::SetLogSink(std::function<void(const std::string&)> Sink) { G().LogSink = std::move(Sink); }
size_t FRenderGraph::GetPooledTextureCount() { return G().TexturePool.size(); }

void FRenderGraph::AddExternalGpuTimingSample(const std::string& Name, double Milliseconds)
{
    const auto Now = std::chrono::steady_clock::now();
    G().Samples[Name].push_back({Now, Milliseconds});
    RefreshStats(Now);
}

void FRenderGraph::ReleaseTransientPool(FHIPDevice* Device)
{
    for (FPooledTexture& P : G().TexturePool)
        if (Device && P.Resource) Device->Free(P.Resource);
    G().TexturePool.clear();
}

FRGResourcePtr FRenderGraph::GetResource(const FRGResourceHandle& Handle) const
{
    return (Handle && Handle.Id < Textures.size()) ? Textures[Handle.Id].Resource : nullptr;
}

FRGResourceHandle FRenderGraph::RegisterTexture(const std::string& Name, const FRGTextureDesc& Desc)
{
    FTexture T;
    T.Name = Name;
    T.Desc = Desc;
    Textures.push_back(std::move(T));
    return FRGResourceHandle{static_cast<uint32>(Textures.size() - 1)};
}

FRGResourceHandle FRenderGraph::ImportTexture(const std::string& Name, FRGResourcePtr Resource, uint32* StatePtr, const FRGTextureDesc& Desc)
{
    const FRGResourceHandle H = RegisterTexture(Name, Desc);
    FTexture& T = Textures[H.Id];
    T.Resource = Resource;
    T.ExternalState = StatePtr;
    T.bExternal = true;
    if (StatePtr) T.CurrentState = *StatePtr;
    return H;
}

FRenderGraph::FPass& FRenderGraph::NewPass(const std::string& Name)
{
    Passes.emplace_back();
    Passes.back().Name = Name;
    return Passes.back();
}

FRGPassBuilder FRenderGraph::MakeBuilder(FPass& Pass) { return FRGPassBuilder(*this, &Pass); }

void FRenderGraph::RegisterUsage(FPass& Pass, const FRGResourceHandle& Handle, uint32 RequiredState, ERGResourceAccess Access)
{
    if (!Handle || Handle.Id >= Textures.size()) return;
    FTexture& T = Textures[Handle.Id];
    if (!T.bExternal && Access == ERGResourceAccess::Write) { // creation flags accumulate from write states (:177-202)
        if (RequiredState & RG_STATE_RENDER_TARGET) T.Flags |= RG_FLAG_ALLOW_RENDER_TARGET;
        if (RequiredState & RG_STATE_DEPTH_WRITE) T.Flags |= RG_FLAG_ALLOW_DEPTH_STENCIL;
        if (RequiredState & RG_STATE_UNORDERED_ACCESS) T.Flags |= RG_FLAG_ALLOW_UNORDERED_ACCESS;
    }
    Pass.Usages.push_back({Handle.Id, RequiredState, Access});
}

FRGResourceHandle FRGPassBuilder::CreateTexture(const std::string& Name, const FRGTextureDesc& Desc) { return Graph->RegisterTexture(Name, Desc); }

FRGResourceHandle FRGPassBuilder::ReadTexture(const FRGResourceHandle& Handle, uint32 RequiredState)
{
    Graph->RegisterUsage(*static_cast<FRenderGraph::FPass*>(Pass), Handle, RequiredState, ERGResourceAccess::Read);
    return Handle;
}

FRGResourceHandle FRGPassBuilder::WriteTexture(const FRGResourceHandle& Handle, uint32 RequiredState)
{
    Graph->RegisterUsage(*static_cast<FRenderGraph::FPass*>(Pass), Handle, RequiredState, ERGResourceAccess::Write);
    return Handle;
}

void FRGPassBuilder::KeepAlive()
{
    if (Pass) static_cast<FRenderGraph::FPass*>(Pass)->bForceExecute = true;
}

void FRGPassBuilder::AsyncCompute()
{
    if (Pass) static_cast<FRenderGraph::FPass*>(Pass)->bAsync = true;
}

bool FRenderGraph::AcquireTransient(FTexture& Texture, uint32 InitialState)
{
    if (!Device) return false;
    auto& Pool = G().TexturePool;
    for (size_t I = 0; I < Pool.size(); ++I) {
        FPooledTexture& P = Pool[I];
        if (!P.bInUse && P.Desc.Width == Texture.Desc.Width && P.Desc.Height == Texture.Desc.Height && P.Desc.Format == Texture.Desc.Format &&
            P.Flags == Texture.Flags) {
            P.bInUse = true;
            Texture.Resource = P.Resource;
            Texture.CurrentState = P.CurrentState;
            Texture.PoolIndex = static_cast<int32>(I);
            return true;
        }
    }
    const size_t Bytes = static_cast<size_t>(Texture.Desc.Width) * Texture.Desc.Height * RGFormatBytesPerTexel(Texture.Desc.Format);
    FRGResourcePtr Ptr = Bytes ? Device->Allocate(Bytes) : nullptr;
    if (!Ptr) return false;
    FPooledTexture P;
    P.Desc = Texture.Desc;
    P.Flags = Texture.Flags;
    P.Resource = Ptr;
    P.CurrentState = InitialState;
    P.bInUse = true;
    Pool.push_back(P);
    Texture.Resource = Ptr;
    Texture.CurrentState = InitialState;
    Texture.PoolIndex = static_cast<int32>(Pool.size() - 1);
    return true;
}

void FRenderGraph::ReleaseTransient(FTexture& Texture)
{
    auto& Pool = G().TexturePool;
    if (Texture.PoolIndex < 0 || Texture.PoolIndex >= static_cast<int32>(Pool.size())) return;
    Pool[Texture.PoolIndex].CurrentState = Texture.CurrentState;
    Pool[Texture.PoolIndex].bInUse = false;
    Texture.Resource = nullptr;
    Texture.PoolIndex = -1;
}

void FRenderGraph::Dump(const std::vector<char>& PassLive, const std::vector<char>& ResourceLive) const
{
    Log("RenderGraph Debug Dump Begin");
    if (bEnableResourceLifetimeLog) {
        Log("Resources:");
        for (size_t I = 0; I < Textures.size(); ++I) {
            if (!ResourceLive[I]) continue;
            std::ostringstream S;
            S << " - " << Textures[I].Name << " (FirstUse: " << Textures[I].FirstUsePass << ", LastUse: " << Textures[I].LastUsePass
              << ", External: " << (Textures[I].bExternal ? "Yes" : "No") << ")";
            Log(S.str());
        }
    }
    Log("Passes:");
    for (size_t P = 0; P < Passes.size(); ++P) {
        std::ostringstream S;
        S << " - [" << P << "] " << Passes[P].Name << (PassLive[P] ? "" : " (Culled)");
        Log(S.str());
        for (const FUsage& U : Passes[P].Usages) {
            std::ostringstream L;
            L << "    * " << Textures[U.Resource].Name << " Access: " << (U.Access == ERGResourceAccess::Read ? "Read" : "Write") << " State: "
              << RGResourceStateToString(U.RequiredState);
            Log(L.str());
        }
    }
    Log("RenderGraph Debug Dump End");
}

void FRenderGraph::Execute(FHIPCommandContext& Cmd)
{
    Report.clear();
    if (!Device) { // same failure behaviour as the reference: log and return (RenderGraph.cpp:216-220)
        Log("RenderGraph Execute called without a valid device");
        return;
    }
    const uint32 Slot = Cmd.GetCurrentFrameIndex();
    HarvestSlot(Slot, bEnableGpuTiming);

    const size_t NumTex = Textures.size(), NumPass = Passes.size();

    // first/last use and "is ever read"
    std::vector<char> EverRead(NumTex, 0);
    for (FTexture& T : Textures) T.FirstUsePass = T.LastUsePass = -1;
    for (size_t P = 0; P < NumPass; ++P)
        for (const FUsage& U : Passes[P].Usages) {
            FTexture& T = Textures[U.Resource];
            if (T.FirstUsePass < 0) T.FirstUsePass = static_cast<int32>(P);
            T.LastUsePass = static_cast<int32>(P);
            if (U.Access == ERGResourceAccess::Read) EverRead[U.Resource] = 1;
        }

    // Roots: resources some pass reads, and externally tracked resources that are used at all. Then one backward sweep:
    // a pass lives if it touches a live resource or asked to be kept alive; a live pass makes all its resources live.
    std::vector<char> ResLive(NumTex, 0), PassLive(NumPass, 0);
    for (size_t I = 0; I < NumTex; ++I) ResLive[I] = EverRead[I] || (Textures[I].ExternalState && Textures[I].FirstUsePass >= 0);
    for (size_t P = NumPass; P-- > 0;) {
        const FPass& Pass = Passes[P];
        bool bLive = Pass.bForceExecute;
        for (size_t K = 0; !bLive && K < Pass.Usages.size(); ++K) bLive = ResLive[Pass.Usages[K].Resource] != 0;
        if (!bLive) continue;
        PassLive[P] = 1;
        for (const FUsage& U : Pass.Usages) ResLive[U.Resource] = 1;
    }
    if (bEnableGraphDump) Dump(PassLive, ResLive);

    // GPU timing: one event pair per live pass on this frame slot
    FSlotTimings* Timings = nullptr;
    if (bEnableGpuTiming && std::count(PassLive.begin(), PassLive.end(), 1) > 0) {
        Timings = &G().SlotTimings[Slot];
        if (Timings->bPending) Timings = nullptr; // previous use of the slot still in flight: skip this frame
        else Timings->Used = 0;
    }

    // ---- async compute: fork the async stream from the main stream; per-resource last-access events order the two
    bool bAnyAsync = false;
    for (size_t P = 0; P < NumPass; ++P) bAnyAsync = bAnyAsync || (PassLive[P] && Passes[P].bAsync);
    bAnyAsync = bAnyAsync && Cmd.HasAsyncCompute();
    FSyncEvents* Sync = nullptr;
    struct FAccess { hipEvent_t LastWrite = nullptr; int LastWriteLane = -1; hipEvent_t LastRead[2] = {nullptr, nullptr}; };
    std::vector<FAccess> Access;
    if (bAnyAsync) {
        Sync = &G().SyncEvents[Slot];
        Sync->Used = 0;
        Access.resize(NumTex);
        hipEvent_t Fork = Sync->Next();
        if (Fork && hipEventRecord(Fork, Cmd.GetMainStream()) == hipSuccess) (void)hipStreamWaitEvent(Cmd.GetAsyncStream(), Fork, 0);
        else bAnyAsync = false;
    }

    for (size_t P = 0; P < NumPass; ++P) {
        FPass& Pass = Passes[P];
        FPassReport R;
        R.Name = Pass.Name;
        R.bCulled = !PassLive[P];
        if (R.bCulled) {
            Report.push_back(std::move(R));
            continue;
        }
        const int Lane = (bAnyAsync && Pass.bAsync) ? 1 : 0;
        R.bAsync = Lane == 1;
        Cmd.SetAsyncLane(Lane == 1);
        if (bAnyAsync) { // hazards against the OTHER stream: RAW (we read, they wrote), WAW / WAR (we write, they wrote / read)
            for (const FUsage& U : Pass.Usages) {
                FAccess& A = Access[U.Resource];
                if (A.LastWrite && A.LastWriteLane != Lane) {
                    (void)hipStreamWaitEvent(Cmd.GetStream(), A.LastWrite, 0);
                    ++R.CrossStreamWaits;
                }
                if (U.Access == ERGResourceAccess::Write && A.LastRead[1 - Lane]) {
                    (void)hipStreamWaitEvent(Cmd.GetStream(), A.LastRead[1 - Lane], 0);
                    ++R.CrossStreamWaits;
                }
            }
        }

        FTimedPass* Timed = nullptr;
        if (Timings) {
            if (Timings->Used == Timings->Passes.size()) {
                FTimedPass N;
                if (hipEventCreate(&N.Begin) == hipSuccess && hipEventCreate(&N.End) == hipSuccess) Timings->Passes.push_back(N);
                else {
                    Log("GPU timing disabled for this frame due to initialization failure");
                    Timings = nullptr;
                }
            }
            if (Timings) {
                Timed = &Timings->Passes[Timings->Used++];
                Timed->Name = Pass.Name;
                (void)hipEventRecord(Timed->Begin, Cmd.GetStream());
            }
        }

        // state tracking: a usage whose tracked state differs from the required one is a transition of the whole
        // resource; it updates the owner's state variable and the graph's copy (RenderGraph.cpp:408-455)
        for (const FUsage& U : Pass.Usages) {
            FTexture& T = Textures[U.Resource];
            if (!T.Resource && !T.bExternal) AcquireTransient(T, U.RequiredState);
            if (!T.Resource) continue; // failed allocation: usage skipped, as in the reference
            uint32& State = T.ExternalState ? *T.ExternalState : T.CurrentState;
            if (State != U.RequiredState) {
                if (bEnableBarrierLogs) {
                    std::ostringstream S;
                    S << "Pass '" << Pass.Name << "' transitioning '" << (T.Name.empty() ? "<Unnamed>" : T.Name) << "': " << RGResourceStateToString(State)
                      << " -> " << RGResourceStateToString(U.RequiredState);
                    Log(S.str());
                }
                State = U.RequiredState;
                T.CurrentState = U.RequiredState;
                ++R.Transitions;
            }
        }
        Cmd.TransitionResources(R.Transitions);

        std::chrono::high_resolution_clock::time_point T0;
        if (bEnableDebugRecording) T0 = std::chrono::high_resolution_clock::now();
        if (Pass.Run) Pass.Run(Cmd);
        if (bEnableDebugRecording) R.CpuMs = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - T0).count();

        if (Timed) (void)hipEventRecord(Timed->End, Cmd.GetStream());

        // publish this pass's accesses for the other stream — only if a later live pass on the other stream touches one of
        // its resources (an event record + wait costs a barrier packet on each queue)
        bool bNeededLater = false;
        if (bAnyAsync) {
            for (size_t Q = P + 1; Q < NumPass && !bNeededLater; ++Q) {
                if (!PassLive[Q] || (Passes[Q].bAsync ? 1 : 0) == Lane) continue;
                for (const FUsage& A : Passes[Q].Usages)
                    for (const FUsage& B : Pass.Usages)
                        if (A.Resource == B.Resource && (A.Access == ERGResourceAccess::Write || B.Access == ERGResourceAccess::Write)) bNeededLater = true;
            }
        }
        if (bNeededLater) {
            hipEvent_t Done = Sync->Next();
            if (Done && hipEventRecord(Done, Cmd.GetStream()) == hipSuccess) {
                for (const FUsage& U : Pass.Usages) {
                    FAccess& A = Access[U.Resource];
                    if (U.Access == ERGResourceAccess::Write) {
                        A.LastWrite = Done;
                        A.LastWriteLane = Lane;
                        A.LastRead[0] = A.LastRead[1] = nullptr;
                    } else {
                        A.LastRead[Lane] = Done;
                    }
                }
            }
        }
        Cmd.SetAsyncLane(false);

        for (const FUsage& U : Pass.Usages) {
            FTexture& T = Textures[U.Resource];
            if (!T.bExternal && T.LastUsePass == static_cast<int32>(P)) ReleaseTransient(T);
        }
        Report.push_back(std::move(R));
    }
    if (Timings && Timings->Used > 0) Timings->bPending = true;
    // join: later main-stream work (next frame's producers, the caller's consumers) sees the async results
    if (bAnyAsync && Cmd.GetJoinAsyncAtEnd()) Cmd.JoinAsyncCompute();

    if (bEnableDebugRecording) {
        Log("RenderGraph Timing (ms):");
        for (size_t P = 0; P < Report.size(); ++P) {
            if (Report[P].bCulled) continue;
            std::ostringstream S;
            S << " - [" << P << "] " << Report[P].Name << ": " << Report[P].CpuMs;
            Log(S.str());
        }
    }
}
