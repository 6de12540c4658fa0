// CullIndirectArgs for gfx950 — frustum + HZB occlusion cull of instance AABBs, plus the visible-list compaction the
// reference does not have.
//
// Reference: Shaders/CullIndirectArgs.hlsl:24-167 ([numthreads(64,1,1)], one thread per instance, writes the
// InstanceCount word at byte 44 of each 64-byte FIndirectDrawCommand), dispatched by FRenderer::DispatchGpuCulling
// (Source/Render/Renderer.cpp:394-472). The arithmetic below follows the HLSL statement by statement and this file is
// built with -ffp-contract=off and IEEE division so every intermediate equals the oracle's; floor(log2(x)) is taken from
// the IEEE exponent.
//
// MI355X shape: 256-thread workgroups (4 x wave64). The workgroup's 256 AABBs (8 KB) are staged through LDS with
// fully-coalesced 16-byte loads, then each lane reads its own min/max pair. Visibility is a wave ballot: lane 0 keeps
// the 64-bit mask, popcounts give per-wave and per-workgroup counts. Compaction is deterministic (ascending index, no
// atomic append) and rides in the SAME launch (cull_compact_kernel): a workgroup culls a contiguous run of 256-instance
// chunks, keeps their ballots in LDS, publishes its visible count as one 8-byte {launch epoch, count} granule, sums the
// granules of the workgroups in front of it (they were started before it: logical indices are tickets drawn at start, so
// the wait cannot deadlock whatever the dispatch order) and writes out the set bits of its masks behind that base.
// (Rounds 1-2 ran the list as a second launch: +6.4 us on a 22.5-us cull of 1 M instances; a look-back chain over 3907
// single-chunk blocks was slower still.) Up to 256 instances (Sponza 25, pica_pica 170) one workgroup does everything.

#include "ur_internal.h"
#include "ur_device.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

struct CullParams {
    // CullingConstants, CullIndirectArgs.hlsl:1-11
    float4 FrustumPlanes[6];
    float ViewProjection[16];
    uint32_t ModelCount, HZBEnabled, HZBMipCount, HZBWidth, HZBHeight, DebugPrintEnabled;
    const float4* bounds;
    const float* hzb;
    uint8_t* args;
    uint32_t* stats;
    uint32_t* visible_idx;
    uint32_t* visible_count;
    unsigned long long* aggregates; // one {epoch << 32 | count} granule per logical workgroup (cull_compact_kernel)
    uint32_t* ticket;               // device words {ticket, finished, epoch}: reset / advanced by the last workgroup of a launch to finish
    uint32_t chunks_per_group, num_chunks, cluster_offset; // cluster totals start at aggregates[cluster_offset]
    uint32_t* timed_out;            // host-visible flag (ur_ctx::hzb_timed_out_dev + 1): a bounded wait gave up
    uint32_t index_base;
    uint32_t nt_words;
    uint32_t mip_offset[UR_MAX_HZB_MIPS];
    uint32_t mip_width[UR_MAX_HZB_MIPS];
    unsigned long long* timeline; // debug: {first entry, last exit} of this launch (ur_debug_timeline), else null
};

__device__ __forceinline__ float saturate_f(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) { return (ax * bx + ay * by) + az * bz; }

__device__ __forceinline__ bool IsAabbVisible(const CullParams& C, float3 mn, float3 mx)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float4 plane = C.FrustumPlanes[i];
        const float px = plane.x >= 0.0f ? mx.x : mn.x;
        const float py = plane.y >= 0.0f ? mx.y : mn.y;
        const float pz = plane.z >= 0.0f ? mx.z : mn.z;
        if (dot3(plane.x, plane.y, plane.z, px, py, pz) + plane.w < 0.0f) return false;
    }
    return true;
}

__device__ __forceinline__ bool IsOccluded(const CullParams& C, float3 mn, float3 mx)
{
    if (C.HZBEnabled == 0 || C.HZBWidth == 0 || C.HZBHeight == 0 || C.HZBMipCount == 0) return false;
    const float* M = C.ViewProjection;
    float minUx = 1.0f, minUy = 1.0f, maxUx = 0.0f, maxUy = 0.0f, maxDepth = 0.0f;
    bool anyBehind = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float cx = (i & 1) ? mx.x : mn.x, cy = (i & 2) ? mx.y : mn.y, cz = (i & 4) ? mx.z : mn.z;
        const float clx = ((cx * M[0] + cy * M[4]) + cz * M[8]) + 1.0f * M[12];
        const float cly = ((cx * M[1] + cy * M[5]) + cz * M[9]) + 1.0f * M[13];
        const float clz = ((cx * M[2] + cy * M[6]) + cz * M[10]) + 1.0f * M[14];
        const float clw = ((cx * M[3] + cy * M[7]) + cz * M[11]) + 1.0f * M[15];
        if (clw <= 0.0f) anyBehind = true; // the HLSL breaks out here; nothing after the break feeds the result
        const float nx = clx / clw, ny = cly / clw, nz = clz / clw;
        const float ux = nx * 0.5f + 0.5f;
        const float uy = 1 - (ny * 0.5f + 0.5f);
        minUx = fminf(minUx, ux); minUy = fminf(minUy, uy);
        maxUx = fmaxf(maxUx, ux); maxUy = fmaxf(maxUy, uy);
        maxDepth = fmaxf(maxDepth, nz);
    }
    if (anyBehind) return false;
    if (maxUx < 0.0f || maxUy < 0.0f || minUx > 1.0f || minUy > 1.0f) return false;
    minUx = saturate_f(minUx); minUy = saturate_f(minUy);
    maxUx = saturate_f(maxUx); maxUy = saturate_f(maxUy);
    const float ex = maxUx - minUx, ey = maxUy - minUy;
    const float psx = ex * (float)C.HZBWidth, psy = ey * (float)C.HZBHeight;
    const float maxDim = fmaxf(psx, psy);
    uint32_t mipLevel = 0;
    if (maxDim > 1.0f) {
        const uint32_t e = ((__float_as_uint(maxDim) >> 23) & 0xFFu) - 127u; // floor(log2(maxDim)), exact
        const float l = fminf(fmaxf((float)e, 0.0f), (float)(C.HZBMipCount - 1u));
        mipLevel = (uint32_t)l;
    }
    const uint32_t mipWidth = max(1u, C.HZBWidth >> mipLevel);
    const uint32_t mipHeight = max(1u, C.HZBHeight >> mipLevel);
    uint32_t minX = (uint32_t)(minUx * (float)mipWidth), minY = (uint32_t)(minUy * (float)mipHeight);
    uint32_t maxX = (uint32_t)(maxUx * (float)mipWidth), maxY = (uint32_t)(maxUy * (float)mipHeight);
    minX = min(minX, mipWidth - 1u); minY = min(minY, mipHeight - 1u);
    maxX = min(maxX, mipWidth - 1u); maxY = min(maxY, mipHeight - 1u);
    const float* mip = C.hzb + C.mip_offset[mipLevel];
    const uint32_t pitch = C.mip_width[mipLevel];
    float hzbDepth = 1.0f;
    hzbDepth = fminf(hzbDepth, mip[(size_t)minY * pitch + minX]);
    hzbDepth = fminf(hzbDepth, mip[(size_t)minY * pitch + maxX]);
    hzbDepth = fminf(hzbDepth, mip[(size_t)maxY * pitch + minX]);
    hzbDepth = fminf(hzbDepth, mip[(size_t)maxY * pitch + maxX]);
    return maxDepth < hzbDepth;
}

// Scatter the visible indices of one 256-instance block. masks[w] = ballot of wave w; base = visible before this block.
__device__ __forceinline__ void ScatterBlock(const CullParams& C, uint32_t block, const uint64_t* masks, uint32_t base)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t wave_base = base;
    for (uint32_t w = 0; w < wave; ++w) wave_base += __popcll(masks[w]);
    const uint64_t m = masks[wave];
    if ((m >> lane) & 1ull) {
        const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
        C.visible_idx[wave_base + rank] = block * 256u + threadIdx.x + C.index_base;
    }
}

// One 256-instance chunk: stage its AABBs through LDS, test, store the InstanceCount words, count the debug statistics.
// Returns this lane's verdict (false for lanes past ModelCount). `sb` is the workgroup's 8 KB staging tile.
__device__ __forceinline__ bool cull_chunk(const CullParams& C, uint32_t chunk, float4* sb)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t first = chunk * 256u;
    const uint32_t index = first + tid;
    // stage this chunk's AABBs: 512 float4, lane-consecutive 16-byte loads
    const uint32_t nb = min(512u, (C.ModelCount - first) * 2u);
    const float4* src = C.bounds + (size_t)first * 2u;
    if (tid < nb) sb[tid] = src[tid];
    if (tid + 256u < nb) sb[tid + 256u] = src[tid + 256u];
    __syncthreads();

    bool visible = false, frustumVisible = true, occluded = false;
    const bool active = index < C.ModelCount;
    if (active) {
        const float4 bmin = sb[2u * tid], bmax = sb[2u * tid + 1u];
        const float3 mn = make_float3(bmin.x, bmin.y, bmin.z), mx = make_float3(bmax.x, bmax.y, bmax.z);
        frustumVisible = IsAabbVisible(C, mn, mx);
        visible = frustumVisible;
        if (visible && C.HZBEnabled != 0) {
            occluded = IsOccluded(C, mn, mx);
            visible = !occluded;
        }
        uint32_t* word = reinterpret_cast<uint32_t*>(C.args + (size_t)index * UR_INDIRECT_COMMAND_STRIDE + UR_INDIRECT_INSTANCE_COUNT_OFFSET);
        if (C.nt_words) __builtin_nontemporal_store(visible ? 1u : 0u, word); // (experiment: UR_CULL_NT=1)
        else *word = visible ? 1u : 0u;
    }
    if (C.DebugPrintEnabled != 0 && C.stats != nullptr) { // one atomic per wave instead of one per lane
        const uint32_t nf = __popcll(__ballot(active && !frustumVisible));
        const uint32_t no = __popcll(__ballot(active && frustumVisible && occluded));
        if (lane == 0) {
            if (nf) atomicAdd(&C.stats[0], nf);
            if (no) atomicAdd(&C.stats[1], no);
        }
    }
    return visible;
}

// Up to 256 instances (SINGLE_BLOCK: one workgroup culls and writes the list), or words only (no list asked for: one
// workgroup per chunk, nothing else).
template <bool SINGLE_BLOCK>
__global__ __launch_bounds__(256) void cull_kernel(CullParams C)
{
    __shared__ float4 sb[512];
    __shared__ uint64_t smask[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ur::timeline_entry(C.timeline);
    const bool visible = cull_chunk(C, blockIdx.x, sb);
    if (!SINGLE_BLOCK || C.visible_idx == nullptr) { // uniform
        ur::timeline_exit(C.timeline, tid == 0);
        return;
    }
    const uint64_t mask = __ballot(visible);
    if (lane == 0) smask[wave] = mask;
    __syncthreads();
    ScatterBlock(C, 0, smask, 0);
    if (tid == 0) *C.visible_count = __popcll(smask[0]) + __popcll(smask[1]) + __popcll(smask[2]) + __popcll(smask[3]);
    ur::timeline_exit(C.timeline, tid == 0);
}

// Cull + list in ONE launch for more than 256 instances. kMaxChunks chunks per workgroup at most (their ballots stay in LDS).
//
// Prefix of the visible counts, two levels (a flat "read every granule in front" made 1 M instances take 57 us: a thousand
// workgroups polling the same lines): workgroups form clusters of kCluster consecutive tickets. A workgroup publishes its
// count as one 8-byte {epoch, count} granule and arrives at its cluster's counter; the LAST arriver of a cluster sums the
// cluster's granules and publishes the cluster total the same way. A workgroup's base is then (cluster totals in front) +
// (member granules in front inside its own cluster): at most 2 x 64 granules, one per lane. Every wait is on something a
// lower ticket produces, so it cannot deadlock; every wait is bounded and a give-up is reported (UR_ETIMEOUT).
constexpr uint32_t kMaxChunks = 16, kCluster = 64;

__device__ __forceinline__ uint32_t wait_granule(const unsigned long long* g, uint32_t epoch, bool& gave_up)
{
    unsigned long long v;
    uint32_t spins = 0;
    while ((uint32_t)((v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) { gave_up = true; return 0u; }
    }
    return (uint32_t)v;
}

__global__ __launch_bounds__(256) void cull_compact_kernel(CullParams C)
{
    __shared__ float4 sb[512];
    __shared__ uint64_t smask[kMaxChunks * 4u];
    __shared__ uint32_t sred[4];
    __shared__ uint32_t sticket, sepoch, sclose;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ur::timeline_entry(C.timeline);
    // Logical index = a ticket drawn at start: every workgroup with a smaller index is already running (or done), whatever
    // order the hardware dispatches blockIdx in, so waiting for them below cannot deadlock. Ticket counter, finish counter and
    // the launch epoch live in device memory and are put back by the LAST workgroup of the launch to finish (every workgroup has
    // drawn its ticket and read the epoch by then; the next launch on the stream starts behind this one): no host bookkeeping,
    // so a captured launch replays correctly.
    if (tid == 0) { sticket = atomicAdd(C.ticket, 1u); sepoch = __hip_atomic_load(C.ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    const uint32_t g = sticket, epoch = sepoch, G = gridDim.x;
    const uint32_t c0 = g * C.chunks_per_group, c1 = min(c0 + C.chunks_per_group, C.num_chunks);
    uint32_t count = 0; // (kept by every thread: the workgroup's visible instances)
    for (uint32_t c = c0; c < c1; ++c) { // uniform
        const bool visible = cull_chunk(C, c, sb);
        const uint64_t mask = __ballot(visible);
        if (lane == 0) smask[(c - c0) * 4u + wave] = mask;
        __syncthreads(); // the masks are in LDS, and sb may be overwritten by the next chunk
        const uint32_t k = (c - c0) * 4u;
        count += __popcll(smask[k]) + __popcll(smask[k + 1u]) + __popcll(smask[k + 2u]) + __popcll(smask[k + 3u]);
    }
    unsigned long long* member = C.aggregates;                    // [G]
    unsigned long long* cluster = C.aggregates + C.cluster_offset; // [clusters]
    uint32_t* arrivals = C.ticket + 16;                           // [clusters], zero between launches (the closer puts it back)
    const uint32_t cl = g / kCluster, m = g - cl * kCluster, cl_first = cl * kCluster, cl_size = min(kCluster, G - cl_first);
    bool gave_up = false;
    // publish {epoch, count} as ONE naturally aligned 8-byte granule (a single agent-scope store: it carries its own validity, no
    // ordering is needed around it), then arrive at the cluster
    if (tid == 0) {
        __hip_atomic_store(member + g, ((unsigned long long)epoch << 32) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sclose = atomicAdd(arrivals + cl, 1u) == cl_size - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (sclose != 0u && wave == 0) { // the cluster's last arriver closes it: total = sum of its members' granules
        uint32_t t = lane < cl_size ? wait_granule(member + cl_first + lane, epoch, gave_up) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) {
            __hip_atomic_store(cluster + cl, ((unsigned long long)epoch << 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(arrivals + cl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // exclusive prefix: wave 0 sums the cluster totals in front (64 per trip), wave 1 the member granules in front inside the cluster
    uint32_t sum = 0;
    if (wave == 0) {
        for (uint32_t q = lane; q < cl; q += 64u) sum += wait_granule(cluster + q, epoch, gave_up);
    } else if (wave == 1) {
        if (lane < m) sum = wait_granule(member + cl_first + lane, epoch, gave_up);
    }
    if (gave_up) __hip_atomic_store(C.timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) sred[wave] = sum;
    __syncthreads();
    const uint32_t base = sred[0] + sred[1];
    // scatter: one thread per mask (<= 64 masks: the first wave), exclusive scan of the popcounts, set bits in ascending order
    const uint32_t nmask = (c1 - c0) * 4u;
    if (wave == 0) {
        const uint64_t mk = lane < nmask ? smask[lane] : 0ull;
        const uint32_t pc = __popcll(mk);
        uint32_t incl = pc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += up;
        }
        uint32_t at = base + incl - pc;
        uint64_t bits = mk;
        const uint32_t first = (c0 * 4u + lane) * 64u + C.index_base;
        while (bits) {
            const uint32_t b = __builtin_ctzll(bits);
            C.visible_idx[at++] = first + b;
            bits &= bits - 1ull;
        }
    }
    if (tid == 0 && c1 == C.num_chunks) *C.visible_count = base + count; // the workgroup that holds the last chunk
    if (tid == 0 && atomicAdd(C.ticket + 1, 1u) == G - 1u) { // the last one out leaves the counters ready for the next launch
        __hip_atomic_store(C.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(C.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(C.ticket + 2, epoch + 1u == 0u ? 1u : epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (0 = a fresh granule)
    }
    ur::timeline_exit(C.timeline, tid == 0);
}

__global__ void zero_count_kernel(uint32_t* p) { *p = 0; }

} // namespace

namespace ur {

int launch_cull(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds, const float* hzb, const ur_mip_desc* mips,
                void* indirect_args, uint32_t* stats2, uint32_t* visible_idx, uint32_t* visible_count, uint32_t index_base)
{
    {
        const int rc = flush_hzb_tail(ctx); // the cull reads the whole chain
        if (rc != UR_OK) return rc;
    }
    CullParams P{};
    static_assert(sizeof(float4) * 6 + sizeof(float) * 16 + 6 * 4 == UR_CULL_CONSTANT_DWORDS * 4, "46 dwords");
    std::memcpy(&P, constants, UR_CULL_CONSTANT_DWORDS * 4);
    P.bounds = reinterpret_cast<const float4*>(bounds);
    P.hzb = hzb;
    P.args = static_cast<uint8_t*>(indirect_args);
    P.stats = stats2;
    P.visible_idx = visible_idx;
    P.visible_count = visible_count;
    P.index_base = index_base;
    static const int nt_words = [] { const char* e = std::getenv("UR_CULL_NT"); return e ? std::atoi(e) : 0; }();
    P.nt_words = (uint32_t)nt_words;
    P.timeline = P.ModelCount != 0 ? next_timeline_pair(ctx) : nullptr; // (the compaction launch of a large cull is not stamped)
    if (P.HZBEnabled != 0) {
        for (uint32_t m = 0; m < P.HZBMipCount && m < UR_MAX_HZB_MIPS; ++m) {
            P.mip_offset[m] = mips[m].offset;
            P.mip_width[m] = mips[m].width;
        }
    }
    const uint32_t n = P.ModelCount;
    if (n == 0) {
        if (visible_count) {
            hipLaunchKernelGGL(zero_count_kernel, dim3(1), dim3(1), 0, ctx->stream, visible_count);
            UR_HIP_TRY(hipGetLastError());
        }
        return UR_OK;
    }
    const uint32_t blocks = (n + 255u) / 256u;
    if (blocks == 1) {
        hipLaunchKernelGGL(cull_kernel<true>, dim3(1), dim3(256), 0, ctx->stream, P);
        UR_HIP_TRY(hipGetLastError());
        return UR_OK;
    }
    if (!visible_idx) { // words only
        hipLaunchKernelGGL(cull_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, P);
        UR_HIP_TRY(hipGetLastError());
        return UR_OK;
    }
    // Cull + list in one launch: G workgroups, each a contiguous run of chunks. Eight workgroups per CU are resident at once
    // (4 waves, 8.7 KB of LDS each); the wait on the workgroups in front is deadlock-free for any G (tickets), G is only sized
    // so that a run is short (load balance, few masks per workgroup) and never longer than the LDS mask store.
    static const int per_cu = [] { const char* e = std::getenv("UR_CULL_GROUPS_PER_CU"); return e ? std::max(1, std::atoi(e)) : 32; }();
    uint32_t groups = std::min<uint32_t>(blocks, (uint32_t)std::max(ctx->cu_count, 1) * (uint32_t)per_cu);
    uint32_t cpg = (blocks + groups - 1u) / groups;
    if (cpg > kMaxChunks) { cpg = kMaxChunks; }
    groups = (blocks + cpg - 1u) / cpg;
    if (groups > ctx->cull_groups_cap) {
        const int rc = ur_reserve(ctx, n);
        if (rc != UR_OK) return rc;
        if (groups > ctx->cull_groups_cap) { set_error("cull workspace too small for %u workgroups", groups); return UR_ENOMEM; }
    }
    P.aggregates = ctx->cull_aggregates;
    P.ticket = ctx->cull_ticket;
    P.chunks_per_group = cpg;
    P.num_chunks = blocks;
    P.cluster_offset = ctx->cull_groups_cap; // (the member granules of the largest launch the workspace holds come first)
    P.timed_out = ctx->hzb_timed_out_dev + 1; // the second word of the context's host-visible flag line
    hipLaunchKernelGGL(cull_compact_kernel, dim3(groups), dim3(256), 0, ctx->stream, P);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}

} // namespace ur
