// CPU-only semantics tests of the HIP-backed FRenderGraph against the behaviours of the reference's graph
// (Source/Render/RenderGraph.cpp:214-517; SURVEY.md §8b "Semantics the replacement must reproduce").
// No GPU work: passes only record their names, transient textures come from a host allocator, GPU timing stays off.

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../unclerenderer_amd/csrc/rg/RenderGraph.h"

static int g_failures = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);               \
            ++g_failures;                                                             \
        }                                                                             \
    } while (0)

struct FHostDevice : FHIPDevice
{
    int Allocs = 0, Frees = 0;
    FRGResourcePtr Allocate(size_t Bytes) override { ++Allocs; return std::malloc(Bytes ? Bytes : 1); }
    void Free(FRGResourcePtr Ptr) override { ++Frees; std::free(Ptr); }
};

struct FEmpty {};

static std::vector<std::string> g_log;

static void test_order_and_culling()
{
    FHostDevice Dev;
    FHIPCommandContext Cmd(nullptr, nullptr);
    std::vector<std::string> Ran;
    uint32 ExtState = RG_STATE_COMMON;
    int Dummy = 0;
    FRenderGraph G;
    G.SetDevice(&Dev);
    const FRGResourceHandle Ext = G.ImportTexture("Ext", &Dummy, &ExtState, {4, 4, RG_FORMAT_R32_FLOAT});
    FRGResourceHandle Tmp, Orphan;
    // A: declares nothing, no KeepAlive -> culled (how disabled passes vanish)
    G.AddPass<FEmpty>("A-nothing", [](FEmpty&, FRGPassBuilder&) {}, [&](const FEmpty&, FHIPCommandContext&) { Ran.push_back("A"); });
    // B: KeepAlive only -> runs
    G.AddPass<FEmpty>("B-keepalive", [](FEmpty&, FRGPassBuilder& B) { B.KeepAlive(); }, [&](const FEmpty&, FHIPCommandContext&) { Ran.push_back("B"); });
    // C: writes a transient that D reads -> alive through D
    G.AddPass<FEmpty>("C-producer", [&](FEmpty&, FRGPassBuilder& B) { Tmp = B.CreateTexture("Tmp", {8, 8, RG_FORMAT_R16G16B16A16_FLOAT}); B.WriteTexture(Tmp, RG_STATE_UNORDERED_ACCESS); },
                      [&](const FEmpty&, FHIPCommandContext&) { Ran.push_back("C"); });
    // O: writes a transient nobody reads -> culled
    G.AddPass<FEmpty>("O-orphan", [&](FEmpty&, FRGPassBuilder& B) { Orphan = B.CreateTexture("Orphan", {8, 8, RG_FORMAT_R32_FLOAT}); B.WriteTexture(Orphan); },
                      [&](const FEmpty&, FHIPCommandContext&) { Ran.push_back("O"); });
    // D: reads Tmp, writes the external resource (external + used -> required)
    G.AddPass<FEmpty>("D-consumer", [&](FEmpty&, FRGPassBuilder& B) { B.ReadTexture(Tmp, RG_STATE_NON_PIXEL_SHADER_RESOURCE); B.WriteTexture(Ext, RG_STATE_RENDER_TARGET); },
                      [&](const FEmpty&, FHIPCommandContext&) { Ran.push_back("D"); });
    G.Execute(Cmd);
    CHECK((Ran == std::vector<std::string>{"B", "C", "D"}));
    const auto& R = G.GetLastExecutionReport();
    CHECK(R.size() == 5);
    CHECK(R[0].bCulled && !R[1].bCulled && !R[2].bCulled && R[3].bCulled && !R[4].bCulled);
    CHECK(ExtState == RG_STATE_RENDER_TARGET);        // owner's state variable was updated
    CHECK(R[4].Transitions == 2);                     // Tmp UAV->SRV, Ext COMMON->RT
    CHECK(R[2].Transitions == 0);                     // transient acquired directly in its first required state
    CHECK(Cmd.GetTransitionCount() == 2);
    CHECK(Dev.Allocs == 1);                           // the orphan was never allocated
    CHECK(G.GetResource(Tmp) == nullptr);             // released after its last pass
    CHECK(FRenderGraph::GetPooledTextureCount() == 1);
}

static void test_pool_reuse_and_flags()
{
    FHostDevice Dev;
    FHIPCommandContext Cmd(nullptr, nullptr);
    const size_t Before = FRenderGraph::GetPooledTextureCount();
    for (int Frame = 0; Frame < 3; ++Frame) {
        FRenderGraph G;
        G.SetDevice(&Dev);
        FRGResourceHandle T1, T2;
        G.AddPass<FEmpty>("P1", [&](FEmpty&, FRGPassBuilder& B) {
            T1 = B.CreateTexture("T1", {16, 16, RG_FORMAT_R32_FLOAT}); B.WriteTexture(T1, RG_STATE_UNORDERED_ACCESS);
            T2 = B.CreateTexture("T2", {16, 16, RG_FORMAT_R32_FLOAT}); B.WriteTexture(T2, RG_STATE_RENDER_TARGET); }, [](const FEmpty&, FHIPCommandContext&) {});
        G.AddPass<FEmpty>("P2", [&](FEmpty&, FRGPassBuilder& B) { B.ReadTexture(T1); B.ReadTexture(T2); B.KeepAlive(); }, [](const FEmpty&, FHIPCommandContext&) {});
        G.Execute(Cmd);
    }
    // same desc but different creation flags (UAV vs RT) -> two pool entries, both reused every frame
    CHECK(FRenderGraph::GetPooledTextureCount() == Before + 2);
    CHECK(Dev.Allocs == 2);
    FRenderGraph::ReleaseTransientPool(&Dev);
    CHECK(FRenderGraph::GetPooledTextureCount() == 0);
}

static void test_no_device_and_destructors()
{
    static int Alive = 0;
    struct FCounted { FCounted() { ++Alive; } ~FCounted() { --Alive; } FCounted(const FCounted&) = delete; };
    FHIPCommandContext Cmd(nullptr, nullptr);
    int Ran = 0;
    FRenderGraph::SetLogSink([](const std::string& L) { g_log.push_back(L); });
    {
        FRenderGraph G; // no SetDevice
        G.AddPass<FCounted>("P", [](FCounted&, FRGPassBuilder& B) { B.KeepAlive(); }, [&](const FCounted&, FHIPCommandContext&) { ++Ran; });
        CHECK(Alive == 1);
        G.Execute(Cmd);
        CHECK(Ran == 0); // Execute logs and returns without a device (RenderGraph.cpp:216-220)
        CHECK(!g_log.empty() && g_log.back().find("without a valid device") != std::string::npos);
    }
    CHECK(Alive == 0); // PassData destructor ran (the reference leaks it)
    FRenderGraph::SetLogSink(nullptr);
}

static void test_state_tracking_across_graphs()
{
    FHostDevice Dev;
    FHIPCommandContext Cmd(nullptr, nullptr, 3);
    uint32 HzbState = RG_STATE_UNORDERED_ACCESS, DepthState = RG_STATE_DEPTH_WRITE;
    int A = 0, B = 0;
    for (int Frame = 0; Frame < 2; ++Frame) {
        Cmd.BeginFrame();
        FRenderGraph G;
        G.SetDevice(&Dev);
        const auto Depth = G.ImportTexture("Depth", &A, &DepthState, {4, 4, RG_FORMAT_R32_FLOAT});
        const auto Hzb = G.ImportTexture("HZB", &B, &HzbState, {2, 2, RG_FORMAT_R32_FLOAT});
        G.AddPass<FEmpty>("Cull", [&](FEmpty&, FRGPassBuilder& Bd) { if (Frame > 0) Bd.ReadTexture(Hzb, RG_STATE_NON_PIXEL_SHADER_RESOURCE); Bd.KeepAlive(); }, [](const FEmpty&, FHIPCommandContext&) {});
        G.AddPass<FEmpty>("Build HZB", [&](FEmpty&, FRGPassBuilder& Bd) { Bd.ReadTexture(Depth, RG_STATE_NON_PIXEL_SHADER_RESOURCE); Bd.WriteTexture(Hzb, RG_STATE_UNORDERED_ACCESS); },
                          [](const FEmpty&, FHIPCommandContext&) {});
        G.Execute(Cmd);
        const auto& R = G.GetLastExecutionReport();
        if (Frame == 0) { CHECK(R[0].Transitions == 0); CHECK(R[1].Transitions == 1); }   // Depth DEPTH_WRITE -> SRV; HZB already UAV
        else { CHECK(R[0].Transitions == 1); CHECK(R[1].Transitions == 1); }               // HZB UAV -> SRV, then SRV -> UAV; depth stays SRV
    }
    CHECK(HzbState == RG_STATE_UNORDERED_ACCESS && DepthState == RG_STATE_NON_PIXEL_SHADER_RESOURCE);
    CHECK(Cmd.GetCurrentFrameIndex() == 2 && Cmd.GetFrameNumber() == 2);
}

static void test_timing_stats_api()
{
    FRenderGraph::SetGpuTimingWindowSeconds(0.01); // clamps to 0.1
    CHECK(FRenderGraph::GetGpuTimingWindowSeconds() == 0.1);
    FRenderGraph::SetGpuTimingWindowSeconds(5.0);
    FRenderGraph::SetGpuTimingDisplayCount(0);
    CHECK(FRenderGraph::GetGpuTimingDisplayCount() == 1);
    FRenderGraph::AddExternalGpuTimingSample("Frame", 2.0);
    FRenderGraph::AddExternalGpuTimingSample("Frame", 4.0);
    FRenderGraph::AddExternalGpuTimingSample("Lighting", 9.0);
    const auto& S = FRenderGraph::GetGpuTimingStats();
    CHECK(S.size() == 2);
    CHECK(S[0].Name == "Lighting" && S[0].AvgMs == 9.0 && S[0].SampleCount == 1); // sorted by average, descending
    CHECK(S[1].Name == "Frame" && S[1].AvgMs == 3.0 && S[1].MinMs == 2.0 && S[1].MaxMs == 4.0 && S[1].SampleCount == 2);
}

static void test_invalid_handles_are_ignored()
{
    FHostDevice Dev;
    FHIPCommandContext Cmd(nullptr, nullptr);
    FRenderGraph G;
    G.SetDevice(&Dev);
    int Ran = 0;
    G.AddPass<FEmpty>("P", [&](FEmpty&, FRGPassBuilder& B) { B.ReadTexture(FRGResourceHandle{}); B.WriteTexture(FRGResourceHandle{12345}); },
                      [&](const FEmpty&, FHIPCommandContext&) { ++Ran; });
    G.Execute(Cmd);
    CHECK(Ran == 0); // nothing valid declared and no KeepAlive -> culled
    CHECK(!static_cast<bool>(FRGResourceHandle{}));
}

int main()
{
    test_order_and_culling();
    test_pool_reuse_and_flags();
    test_no_device_and_destructors();
    test_state_tracking_across_graphs();
    test_timing_stats_api();
    test_invalid_handles_are_ignored();
    if (g_failures == 0) std::printf("OK rendergraph tests passed\n");
    return g_failures ? 1 : 0;
}
