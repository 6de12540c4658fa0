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
// atomic append): pass 1 stores per-wave masks and per-workgroup counts, pass 2 (one thread per mask) takes the exclusive
// prefix of the counts and writes out the set bits of its mask. Up to 256 instances (Sponza 25, pica_pica 170) both passes
// run inside one launch.

#include "ur_internal.h"
#include "ur_device.h"

#include <hip/hip_ext.h>

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
    uint32_t* block_counts;
    uint64_t* wave_masks;
    uint32_t index_base;
    uint32_t store_flavour;
    uint32_t record_valid; // store_flavour 4: wave_masks holds the visible bits this context's previous launch left in THIS command buffer
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

template <bool SINGLE_BLOCK>
__global__ __launch_bounds__(256) void cull_kernel(CullParams C)
{
    __shared__ float4 sb[512];
    __shared__ uint64_t smask[4];
    __shared__ uint32_t scand[4];
    __shared__ uint8_t slist[256], socc[256];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t first = blockIdx.x * 256u;
    const uint32_t index = first + tid;
    ur::timeline_entry(C.timeline);

    // stage this block's AABBs: 512 float4, lane-consecutive 16-byte loads. Every load of the thread goes out before the first LDS
    // write (clamped indices instead of branches: one memory latency, not two or three in a row)
    const uint32_t nb = min(512u, (C.ModelCount - first) * 2u); // >= 2: the block has an instance
    const float4* src = C.bounds + (size_t)first * 2u;
    typedef float cf32x4_t __attribute__((ext_vector_type(4)));
    const cf32x4_t* srcv = reinterpret_cast<const cf32x4_t*>(src);
    const cf32x4_t v0 = __builtin_nontemporal_load(srcv + min(tid, nb - 1u)), v1 = __builtin_nontemporal_load(srcv + min(tid + 256u, nb - 1u)); // read once per launch: the hint keeps 32 MB of AABBs from pushing the command lines out of the caches (1 M, cold: 25.3 -> 23.7 us; blind stores 36 -> 26)
    const float4 s0 = make_float4(v0.x, v0.y, v0.z, v0.w), s1 = make_float4(v1.x, v1.y, v1.z, v1.w);

    bool visible = false, frustumVisible = true, occluded = false;
    const bool active = index < C.ModelCount;
    // UR_OPT_CULL_STORE = 3: the word's present value, fetched with the AABBs (its latency lies under the barrier and the tests)
    // UR_OPT_CULL_STORE = 4: ... taken from the context's record instead - one bit per instance, the wave masks of its previous launch on
    // this command buffer (125 KB for 1 M instances instead of 64 MB of command lines); without a valid record, as 3
    uint32_t old_word = 0xFFFFFFFFu;
    const bool from_record = !SINGLE_BLOCK && C.store_flavour == 4u && C.record_valid != 0u; // (launch-uniform)
    uint64_t old_mask = 0;
    if (from_record) old_mask = C.wave_masks[(size_t)blockIdx.x * 4u + wave];
    else if (C.store_flavour >= 3u)
        old_word = *reinterpret_cast<const uint32_t*>(C.args + (size_t)min(index, C.ModelCount - 1u) * UR_INDIRECT_COMMAND_STRIDE + UR_INDIRECT_INSTANCE_COUNT_OFFSET);
    if (tid < nb) sb[tid] = s0;
    if (tid + 256u < nb) sb[tid + 256u] = s1;
    __syncthreads();
    if (active) {
        const float4 bmin = sb[2u * tid], bmax = sb[2u * tid + 1u];
        frustumVisible = IsAabbVisible(C, make_float3(bmin.x, bmin.y, bmin.z), make_float3(bmax.x, bmax.y, bmax.z));
    }
    if (SINGLE_BLOCK) { // a few instances (the frame's own cull: 25 commands): lane by lane, no barrier in the way of the one workgroup's latency
        if (active && frustumVisible && C.HZBEnabled != 0) {
            const float4 bmin = sb[2u * tid], bmax = sb[2u * tid + 1u];
            occluded = IsOccluded(C, make_float3(bmin.x, bmin.y, bmin.z), make_float3(bmax.x, bmax.y, bmax.z));
        }
    } else if (C.HZBEnabled != 0) { // (launch-uniform)
        // The occlusion test (eight corners projected with IEEE divides, a mip choice, four taps) is several times the frustum test, and
        // only what the frustum lets through takes it - one instance in nine of the 1 M stress set: run lane by lane, every wave paid for
        // it at a ninth of its lanes. The block's survivors are packed first (ballots + popcounts, their indices in LDS) and tested
        // densely by the block's first wave(s); every instance still runs the same statements on its own bounds.
        const bool cand = active && frustumVisible;
        const uint64_t cm = __ballot(cand);
        if (lane == 0) scand[wave] = (uint32_t)__popcll(cm);
        __syncthreads();
        uint32_t at = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) { at += w < wave ? scand[w] : 0u; total += scand[w]; }
        if (cand) slist[at + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull))] = (uint8_t)tid;
        __syncthreads();
        if (tid < total) {
            const uint32_t j = slist[tid];
            const float4 bmin = sb[2u * j], bmax = sb[2u * j + 1u];
            socc[j] = IsOccluded(C, make_float3(bmin.x, bmin.y, bmin.z), make_float3(bmax.x, bmax.y, bmax.z)) ? 1u : 0u;
        }
        __syncthreads();
        occluded = cand && socc[tid] != 0u;
    }
    if (active) {
        visible = frustumVisible && !occluded;
        uint32_t* word = reinterpret_cast<uint32_t*>(C.args + (size_t)index * UR_INDIRECT_COMMAND_STRIDE + UR_INDIRECT_INSTANCE_COUNT_OFFSET);
        const uint32_t value = visible ? 1u : 0u;
        // Write-through (sc1): each word is alone in its 64-byte command, so a store is one fabric write whenever it leaves L2; leaving at
        // once means the launch ends with nothing dirty to write back (1 M instances: 22.6 -> 20.0 us words only, 29.1 -> 26.7 with
        // the list; nontemporal stores changed nothing). UR_OPT_CULL_STORE = 0 / 1 select plain / nontemporal stores for comparison.
        // UR_OPT_CULL_STORE = 3 (default): a word that already holds its value is left alone. The command buffer lives across frames
        // (the reference uploads it once per scene, Source/Render/DeferredRenderer.cpp:3397-3442, and its shader rewrites dword 11 in
        // place every frame), and from one frame to the next few instances change sides: the store - a 4-byte write into a 64-byte
        // line of its own, i.e. a read-modify-write of that line in memory - then happens for those few only, the rest costs the
        // 4-byte read. Memory ends up the same in every case (1 M instances over cold buffers: see DESIGN.md 3.2).
        if (C.store_flavour >= 3u) {
            if (from_record) old_word = (uint32_t)(old_mask >> lane) & 1u;
            if (old_word != value) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(word), "v"(value) : "memory");
        } else if (C.store_flavour == 2u) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(word), "v"(value) : "memory");
        else if (C.store_flavour == 1u) __builtin_nontemporal_store(value, word);
        else *word = value;
    }

    if (C.DebugPrintEnabled != 0 && C.stats != nullptr) { // one atomic per wave instead of one per lane
        const uint32_t nf = __popcll(__ballot(active && !frustumVisible));
        const uint32_t no = __popcll(__ballot(active && frustumVisible && occluded));
        if (lane == 0) {
            if (nf) atomicAdd(&C.stats[0], nf);
            if (no) atomicAdd(&C.stats[1], no);
        }
    }

    if (C.visible_idx == nullptr) { // uniform
        if (!SINGLE_BLOCK && C.store_flavour == 4u) { // the record of what the command buffer holds now (a launch with a list writes it below)
            const uint64_t m = __ballot(visible);
            if (lane == 0) C.wave_masks[(size_t)blockIdx.x * 4u + wave] = m;
        }
        ur::timeline_exit(C.timeline, tid == 0);
        return;
    }
    const uint64_t mask = __ballot(visible);
    if (lane == 0) smask[wave] = mask;
    __syncthreads();
    if (SINGLE_BLOCK) {
        ScatterBlock(C, 0, smask, 0);
        if (tid == 0) *C.visible_count = __popcll(smask[0]) + __popcll(smask[1]) + __popcll(smask[2]) + __popcll(smask[3]);
    } else {
        if (lane == 0) C.wave_masks[(size_t)blockIdx.x * 4u + wave] = mask;
        if (tid == 0) C.block_counts[blockIdx.x] = __popcll(smask[0]) + __popcll(smask[1]) + __popcll(smask[2]) + __popcll(smask[3]);
    }
    ur::timeline_exit(C.timeline, tid == 0);
}

// Pass 2: one thread per wave mask (64 instances), one workgroup per 256 masks = 64 cull blocks. The workgroup's base is
// the sum of the block counts in front of it (a few loads per thread), a thread's offset the exclusive scan of the mask
// popcounts inside the workgroup; the few set bits of a mask are written out in ascending order.
__global__ __launch_bounds__(256) void compact_kernel(CullParams C, uint32_t num_blocks)
{
    __shared__ uint32_t spart[4], swave[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t num_masks = num_blocks * 4u, mi = blockIdx.x * 256u + tid;
    const uint64_t m = mi < num_masks ? C.wave_masks[mi] : 0ull;
    // the counts of the blocks in front: sixteen loads per thread in flight at once (a loop of dependent-looking loads made the last
    // workgroups of a 1 M-instance cull wait for fifteen memory round trips in a row), then the rare rest
    uint32_t s = 0;
    const uint32_t limit = blockIdx.x * 64u;
    if (limit != 0u) { // uniform
        uint32_t part[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) part[k] = C.block_counts[min(tid + k * 256u, limit - 1u)];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) s += tid + k * 256u < limit ? part[k] : 0u;
        for (uint32_t b = tid + 4096u; b < limit; b += 256u) s += C.block_counts[b];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    // inclusive scan of the popcounts across the wave, then across the four waves
    const uint32_t c = __popcll(m);
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == 0) spart[wave] = s;
    if (lane == 63) swave[wave] = incl;
    __syncthreads();
    uint32_t at = spart[0] + spart[1] + spart[2] + spart[3] + incl - c;
    for (uint32_t w = 0; w < wave; ++w) at += swave[w];
    uint64_t bits = m;
    const uint32_t first = mi * 64u + C.index_base;
    while (bits) {
        const uint32_t b = __builtin_ctzll(bits);
        C.visible_idx[at++] = first + b;
        bits &= bits - 1ull;
    }
    if (mi == num_masks - 1u) *C.visible_count = at;
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
    P.store_flavour = (uint32_t)ctx->opt.cull_store; // UR_OPT_CULL_STORE
    P.timeline = P.ModelCount != 0 ? next_timeline_pair(ctx) : nullptr; // (the compaction launch of a large cull is not stamped)
    if (P.HZBEnabled != 0) {
        for (uint32_t m = 0; m < P.HZBMipCount && m < UR_MAX_HZB_MIPS; ++m) {
            P.mip_offset[m] = mips[m].offset;
            P.mip_width[m] = mips[m].width;
        }
    }
    // ur_time_next_cull: the call's LAST launch carries the event on its dispatch (its completion stamp is somebody's start time).
    // (ur_cull_indirect_args_ex clears the context's copy behind this function on every path: a raw hipEvent_t must not stay in
    // the context for a later call.)
    hipEvent_t stop = ctx->time_cull_stop;
    const uint32_t n = P.ModelCount;
    if (n == 0) {
        ctx->cull_record_args = nullptr;
        if (visible_count) {
            if (stop != nullptr) hipExtLaunchKernelGGL(zero_count_kernel, dim3(1), dim3(1), 0, ctx->stream, nullptr, stop, 0, visible_count);
            else hipLaunchKernelGGL(zero_count_kernel, dim3(1), dim3(1), 0, ctx->stream, visible_count);
            UR_HIP_TRY(hipGetLastError());
            ctx->time_cull_carried = stop != nullptr;
        }
        return UR_OK;
    }
    const uint32_t blocks = (n + 255u) / 256u;
    if (blocks == 1) {
        ctx->cull_record_args = nullptr; // (one block keeps no masks)
        if (stop != nullptr) hipExtLaunchKernelGGL(cull_kernel<true>, dim3(1), dim3(256), 0, ctx->stream, nullptr, stop, 0, P);
        else hipLaunchKernelGGL(cull_kernel<true>, dim3(1), dim3(256), 0, ctx->stream, P);
        UR_HIP_TRY(hipGetLastError());
        ctx->time_cull_carried = stop != nullptr;
        return UR_OK;
    }
    if (visible_idx || P.store_flavour == 4u) {
        if (n > ctx->ws_instances) {
            const int rc = ur_reserve(ctx, n); // (a new workspace forgets the record)
            if (rc != UR_OK) return rc;
        }
        P.block_counts = ctx->block_counts;
        P.wave_masks = ctx->wave_masks;
    }
    // UR_OPT_CULL_STORE = 4: the wave masks ARE the record of what this launch leaves in the command buffer; they describe the buffer the
    // next launch meets if that launch is on the same buffer with the same count (and the caller keeps the promise of the option)
    P.record_valid = (P.store_flavour == 4u && ctx->cull_record_args == indirect_args && ctx->cull_record_n == n) ? 1u : 0u;
    ctx->cull_record_args = P.store_flavour == 4u ? indirect_args : nullptr;
    ctx->cull_record_n = n;
    if (stop != nullptr && !visible_idx) hipExtLaunchKernelGGL(cull_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, nullptr, stop, 0, P);
    else hipLaunchKernelGGL(cull_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, P);
    UR_HIP_TRY(hipGetLastError());
    if (visible_idx) {
        if (stop != nullptr) hipExtLaunchKernelGGL(compact_kernel, dim3((blocks * 4u + 255u) / 256u), dim3(256), 0, ctx->stream, nullptr, stop, 0, P, blocks);
        else hipLaunchKernelGGL(compact_kernel, dim3((blocks * 4u + 255u) / 256u), dim3(256), 0, ctx->stream, P, blocks);
        UR_HIP_TRY(hipGetLastError());
    }
    ctx->time_cull_carried = stop != nullptr;
    return UR_OK;
}

} // namespace ur
