/*
 * ur_hotpath.h — C-ABI of the MI355X-native visibility + deferred-shading hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types. The entry points are
 * what the four pass lambdas of the reference's FDeferredRenderer::RenderFrame would call instead of
 * recording D3D12 commands (reference: Source/Render/DeferredRenderer.cpp:522-542 cull,
 * :998-1211 Build HZB, :1219-1255 Lighting, :1263-1296 Sky; Source/Render/Renderer.cpp:394-472).
 *
 * Conventions
 *  - return UR_OK (0) or a negative UR_E* code; ur_last_error() gives a thread-local message;
 *  - every pointer documented "device" is caller-owned HIP device memory; "host" is host memory
 *    read synchronously during the call (constant blocks are passed by value to the kernels, exactly
 *    like D3D12 root constants / a mapped CBV);
 *  - all launches are asynchronous on the stream given to ur_create(); no call synchronises;
 *  - images are linear row-major, pitch == width (no D3D swizzle);
 *  - matrices are row-major, row-vector convention (mul(v, M)) — the reference compiles every shader
 *    with -Zpr (Source/Render/ShaderCompiler.cpp:74);
 *  - screen-tile sharding: image buffers passed to the shading entry points are BAND-LOCAL: they hold
 *    rows [row0, row0+rows) of a frame of full size (w, h). For a whole frame pass row0=0, rows=h.
 */
#ifndef UR_HOTPATH_H
#define UR_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_OK 0
#define UR_EINVAL (-1)       /* bad argument (null pointer, zero size, inconsistent layout) */
#define UR_EHIP (-2)         /* a HIP runtime call failed */
#define UR_ENOMEM (-3)       /* workspace allocation failed */
#define UR_ENODEVICE (-4)    /* no usable gfx950 device */
#define UR_EUNSUPPORTED (-5) /* valid request outside what the kernels implement */
#define UR_ETIMEOUT (-6)     /* a riding Build HZB chain's tail gave up waiting for its producers inside a Lighting launch: the HZB's
                                levels from the tail's first one on are stale - or a wave of a balanced Lighting launch gave up waiting for a
                                tile claim of its workgroup: tiles of that launch were not shaded. Reported once by the next ur_flush /
                                ur_build_hzb* / ur_cull_indirect_args* / ur_deferred_lighting* / ur_frame_render on the context, which does
                                nothing else in that call */

#define UR_MAX_HZB_MIPS 16u
#define UR_CULL_CONSTANT_DWORDS 46u
#define UR_INDIRECT_COMMAND_STRIDE 64u        /* sizeof(FIndirectDrawCommand), RendererUtils.h:102-111 */
#define UR_INDIRECT_INSTANCE_COUNT_OFFSET 44u /* CullIndirectArgs.hlsl:21-22 */

typedef struct ur_ctx ur_ctx;

typedef struct ur_float4 { float x, y, z, w; } ur_float4;
/* four IEEE binary16 bit patterns (DXGI_FORMAT_R16G16B16A16_FLOAT texel) */
typedef struct ur_half4 { uint16_t x, y, z, w; } ur_half4;

/* One HZB mip inside a single linear R32F allocation (replaces the D3D12 subresource). */
typedef struct ur_mip_desc {
    uint32_t offset; /* in floats from hzb_base */
    uint32_t width;
    uint32_t height;
} ur_mip_desc;

/* FSceneConstants, byte-for-byte (Source/Render/RendererUtils.h:41-79 == Shaders/SceneConstants.hlsl:1-39). */
typedef struct ur_scene_constants {
    float World[16];
    float View[16];
    float ViewInverse[16];
    float Projection[16];
    float BaseColor[3];
    float LightIntensity;
    float LightDirection[3];
    float Padding1;
    float CameraPosition[3];
    float Padding2;
    float LightColor[3];
    float Padding3;
    float EmissiveFactor[3];
    float Padding4;
    float LightViewProjection[16];
    float ShadowStrength;
    float ShadowBias;
    float ShadowMapSize[2];
    float MetallicFactor;
    float RoughnessFactor;
    float BaseColorAlpha;
    float AlphaCutoff;
    uint32_t AlphaMode;
    uint32_t PaddingMaterial[3];
    float BaseColorTransformOffsetScale[4];
    float BaseColorTransformRotation[4];
    float MetallicRoughnessTransformOffsetScale[4];
    float MetallicRoughnessTransformRotation[4];
    float NormalTransformOffsetScale[4];
    float NormalTransformRotation[4];
    float EmissiveTransformOffsetScale[4];
    float EmissiveTransformRotation[4];
    float EnvMapMipCount;
    float PaddingEnvMap[3];
    uint32_t ObjectId;
    float PaddingObjectId[3];
} ur_scene_constants; /* 608 bytes */

/* FSkyAtmosphereConstants (Source/Render/RendererUtils.h:81-92 == Shaders/SkyAtmosphere.hlsl:16-27). */
typedef struct ur_sky_constants {
    float World[16]; /* scale(R) * translate(camera), DeferredRenderer.cpp:3789-3801 */
    float View[16];
    float Projection[16];
    float CameraPosition[3];
    float Padding0;
    float LightDirection[3];
    float Padding1;
    float LightColor[3];
    float Padding2;
} ur_sky_constants; /* 240 bytes */

/* Read-only side tables of the lighting pass (t3..t5 of DeferredLighting.hlsl:11-16). All device. */
typedef struct ur_lighting_tables {
    const float* shadow_map;       /* ShadowMapSize.x * ShadowMapSize.y R32F; may be NULL iff ShadowStrength <= 0 */
    const ur_half4* env_cube;      /* what ur_stage_env_cube() wrote: the bordered faces of every mip, then the same faces as RGB row pairs */
    uint32_t env_base_size;        /* edge of mip 0 (256 for Assets/Textures/output_pmrem.dds) */
    uint32_t env_mip_count;        /* mips present in env_cube (9) */
    const uint16_t* brdf_lut_rg16; /* lut_width * lut_height texels, 2 x UNORM16 each (PreintegratedGF.dds) */
    uint32_t lut_width;            /* 128: NdotV axis */
    uint32_t lut_height;           /* 32: roughness axis */
    uint64_t env_cube_texels;      /* ur_env_cube_texels(env_base_size, env_mip_count) as returned when env_cube was sized and staged: the
                                      layout's tag. A buffer staged by another version of the library (round 2's layout had the bordered
                                      faces only, a third of the size) is refused with UR_EINVAL instead of being read out of bounds */
} ur_lighting_tables;

/* ---- context ---------------------------------------------------------------------------------- */

/* Bind to HIP device `device` and launch on `stream` (a hipStream_t, NULL = default stream).
 * Replaces FDX12CommandContext as the thing a pass lambda records into (RHI/DX12CommandContext.h:10-43). */
ur_ctx* ur_create(int device, void* stream);
void ur_destroy(ur_ctx* ctx);
/* Pre-size the compaction workspace so later calls allocate nothing (graph-capture safe). */
int ur_reserve(ur_ctx* ctx, uint32_t max_instances);
/* Launch scheduling across passes (no counterpart in the reference, whose passes are separate D3D12 dispatches;
 * DeferredRenderer.cpp:1046-1207 builds the HZB, :1997-2005 lights). The last launch of ur_build_hzb is ONE workgroup
 * (the mips that fit LDS): ~5 us during which the rest of the chip idles. With ur_defer_hzb_tail(ctx, 1) that workgroup
 * is held back and rides along with the next ur_deferred_lighting / ur_deferred_lighting_sky launch on the same context
 * as an extra workgroup when that launch uses the streaming kernel, and goes out on its own in front of a launch that uses
 * the per-tile kernel: either way the HZB is complete when that Lighting launch is. Anything else that reads or rewrites
 * the HZB through this context (ur_cull_indirect_args*, ur_build_hzb), ur_flush() and ur_defer_hzb_tail(ctx, 0) launch a
 * held-back tail on its own first. ur_destroy() DISCARDS it (the HZB buffer may already be gone). While a tail is held
 * back, ur_build_hzb has returned UR_OK with the levels of the tail still unwritten: a caller that reads the HZB by other
 * means (its own kernels, a copy, another context or stream) calls ur_flush() first, and keeps the HZB buffer alive until
 * then. Off by default.
 * ur_defer_hzb_tail(ctx, 2) holds back the WHOLE chain when it is one five-level launch from the depth buffer plus the tail
 * (every frame size from a few hundred pixels up to 8K is): ur_build_hzb then launches nothing; the next streaming Lighting
 * launch walks the wide launch's 128x32 pieces with one wave of each of its workgroups (memory-bound work beside the
 * compute-bound shading) and its extra tail workgroup waits, inside the launch, for their arrival counter before it reduces
 * the rest. Same bits, same rules as above (flush / cull / rebuild / a per-tile Lighting launch send the ordinary launches
 * out first); the depth buffer must stay unchanged until then as well. */
int ur_defer_hzb_tail(ur_ctx* ctx, int mode /* 0 off, 1 tail, 2 whole chain */);
int ur_flush(ur_ctx* ctx);
/* Debug: sets the context's time-out flag as the riding tail workgroup does when it gives up waiting (a bounded wait inside
 * the Lighting launch, see UR_ETIMEOUT): the next ur_flush / ur_build_hzb / ur_cull_indirect_args* / ur_frame_render on the
 * context returns UR_ETIMEOUT once. For tests of the host's error path. */
int ur_debug_set_hzb_timeout(ur_ctx* ctx);
/* Debug: the tile schedule of the context's last streaming Lighting launch: out8 = {lighting workgroups, tiles, tiles dealt
 * statically, chunks claimed at run time (0: balancing off for that launch), log2 of the tiles per chunk, chunks claimed ahead,
 * waves per workgroup, Build HZB pieces that rode along}. All zero before the first such launch. */
int ur_debug_lighting_schedule(const ur_ctx* ctx, uint32_t out8[8]);
/* Measurement aid (no counterpart in the reference): a plain streaming kernel with the fused Lighting launch's byte mix and
 * nothing to compute, out[i] = in0[i] + in1[i] + in2[i] + in3[i] on `elements16` 16-byte elements (device pointers, 16-byte
 * aligned; four read streams, one write stream). What it sustains at a launch's byte count is the practical ceiling bench.py prints
 * beside the roofline fraction (roofline.stream_ceiling_GBps). With events (hipEvent_t, timing enabled) the dispatch carries them
 * like ur_time_next_lighting's: hipEventElapsedTime(start, stop) is the dispatch's duration. */
int ur_debug_stream_ceiling(ur_ctx* ctx, const void* in0, const void* in1, const void* in2, const void* in3, void* out, uint64_t elements16,
                            void* start_event, void* stop_event);
/* Debug: a GPU-side timeline of the context's launches. device_pairs: capacity_pairs x 2 uint64 in device memory, every pair
 * initialised by the caller to {~0, 0}. From then on each cull launch and each streaming Lighting launch on the context takes
 * the next pair (until the array is full) and folds the constant 100 MHz clock (s_memrealtime) into it: [0] = first workgroup's entry, [1] = last
 * workgroup's exit. Gaps between consecutive launches are then read off without a profiler (bench.py --timeline). NULL
 * switches it off (the default: the kernels then execute one scalar branch for it). */
int ur_debug_timeline(ur_ctx* ctx, unsigned long long* device_pairs, uint32_t capacity_pairs);
/* Timing: the NEXT ur_deferred_lighting / ur_deferred_lighting_sky launch on the context carries this pair of HIP events
 * (hipEvent_t, created by the caller with timing enabled) on the kernel dispatch (hipExtLaunchKernel): the stop event is
 * bound to the dispatch's own completion signal, the start event to a marker the runtime puts directly in front of it. After
 * the stream has been synchronised hipEventElapsedTime(start, stop) is the interval from the end of whatever preceded the
 * kernel on the stream to the kernel's end as the command processor stamps them: the dispatch's duration including its
 * launch, which is what rocprofv3's kernel trace reports for it (the profiler serialises dispatches the same way), with no
 * event record behind the kernel. (A stop event alone is accepted and attached, but measures nothing on ROCm 7.2:
 * hipEventElapsedTime(e, e) is 0.) One-shot: consumed by that launch (by the main kernel of it: a held-back HZB tail that
 * has to go out in front is not timed). The reference's counterpart is the timestamp-query pair FRenderGraph puts around a
 * pass (Source/Render/RenderGraph.cpp:402-406,475-478). NULL, NULL clears what was not consumed. */
int ur_time_next_lighting(ur_ctx* ctx, void* start_event, void* stop_event);
/* The same for the cull: the LAST launch of the next ur_cull_indirect_args* call on the context (the compaction launch when there
 * is a visible list and more than 256 instances, the cull launch otherwise, the one-thread zeroing of visible_count when there are
 * no instances) carries stop_event on its dispatch (its completion stamp). With the Lighting launch directly behind that cull on
 * the stream — the frame of ur_frame_render with UR_FRAME_HZB_WITH_LIGHTING is exactly those two launches — this event is the START
 * of the Lighting measurement and ur_time_next_lighting(ctx, NULL, stop) its end: hipEventElapsedTime(cull_stop, lighting_stop) is the
 * same interval as the marker form measures (end of what precedes the kernel -> end of the kernel) with NOTHING added to the queue.
 * One-shot: that call clears it whether or not a dispatch took it (ur_time_cull_carried tells); NULL clears it beforehand. */
int ur_time_next_cull(ur_ctx* ctx, void* stop_event);
/* 1 if the last ur_cull_indirect_args* call on the context put the event of ur_time_next_cull on one of its dispatches, 0 if it
 * launched nothing to carry it (zero instances and no visible_count). The call clears the pending event either way. */
int ur_time_cull_carried(const ur_ctx* ctx);

/* Launch-shape options, per context (two contexts of one process may differ; nothing is read from the environment). Every
 * option keeps the results bit for bit: they choose between kernels / work splits that the parity tests hold to the same values.
 * ur_set_option returns UR_EINVAL for an unknown option or a value outside the range given here. */
#define UR_OPT_LIGHTING_STREAM 1        /* [1] 1 = streaming lighting kernel where it applies, 0 = always the per-tile kernel */
#define UR_OPT_LIGHTING_WAVES_PER_WG 2  /* [16] waves per persistent workgroup of the streaming kernel: 16 (4 per SIMD) or 12 (3) */
#define UR_OPT_LIGHTING_TILED_WAVES 3   /* [6] register budget of the per-tile kernel, in waves per SIMD: 6 (80 VGPRs) or 4 (uncapped) */
#define UR_OPT_LIGHTING_LEAVE_CUS 4     /* [0] CUs the persistent lighting workgroups leave to kernels of other streams, 0..128 */
#define UR_OPT_RIDE_WALKERS 5           /* [0] riding Build HZB chain: 0 = chosen per launch, 1 = one wave per workgroup walks the pieces, 16 = all */
#define UR_OPT_CULL_STORE 7             /* [3] InstanceCount word stores: 3 = write-through (sc1), only words whose value changes (the kernel reads the present
                                           value first); 2 = write-through, every word; 1 = nontemporal; 0 = plain; 4 = as 3, but for culls of more than 256
                                           instances the present values come from the CONTEXT'S RECORD (one bit per instance) of its previous launch on the
                                           same command buffer with the same count - 1/500 of the bytes. The caller promises that nothing but this context's
                                           culls writes those words between two launches; setting the option (again) forgets the record, so does a launch on
                                           another buffer or count, a cull of <= 256 instances, ur_reserve growing the workspace. Same bytes in memory while
                                           the promise holds */
#define UR_OPT_LIGHTING_BALANCE 8       /* [1] 1 = the last part of a streaming launch's tiles is claimed by the workgroups at run time
                                           (inter-workgroup balancing), 0 = every tile dealt statically */
#define UR_OPT_BALANCE_POOL_16THS 9     /* [3] that part, in sixteenths of the launch's tiles, 1..8 */
#define UR_OPT_BALANCE_CHUNK_SHIFT 10   /* [4] log2 of the tiles per run-time claim, 2..6; a launch too short for a few such chunks per workgroup is dealt statically */
#define UR_OPT_DEBUG_HZB_RIDE_STALL 11  /* [0] debug, the one option that DOES change results: 1 = the tail workgroup of a riding Build HZB chain expects
                                           one arrival more than there are producers and gives up after ~1 ms, i.e. every riding launch takes the
                                           real time-out path (UR_ETIMEOUT at the next entry point, stale small HZB levels) - for tests of that path */
int ur_set_option(ur_ctx* ctx, int option, int value);
int ur_get_option(const ur_ctx* ctx, int option, int* value);
const char* ur_last_error(void);
const char* ur_version(void);

/* ---- BuildHZB (Shaders/BuildHZB.hlsl:34-126, DeferredRenderer.cpp:1016-1211, :2801-2835) ------- */

/* HZB sizing of CreateHZBResources: base = (max(1,(w+1)/2), max(1,(h+1)/2)), halve with max(1,d/2)
 * until 1x1. Fills mips[0..*mip_count) with packed offsets; returns the total number of floats
 * (0 on bad arguments). */
uint32_t ur_hzb_layout(uint32_t src_w, uint32_t src_h, ur_mip_desc* mips /*host, UR_MAX_HZB_MIPS*/,
                       uint32_t* mip_count);

/* depth: device, src_w*src_h floats (the depth buffer as the SRV sees it: reverse-Z in [0,1]).
 * hzb_base: device, laid out by `mips` (host). Builds every mip; values are bit-identical to the
 * reference's dispatch chain of <=4 mips per dispatch, including its out-of-range fill quirks.
 * With ur_defer_hzb_tail(ctx, 1) the chain is complete only after ur_flush() or the next Lighting launch (see there). */
int ur_build_hzb(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb_base,
                 const ur_mip_desc* mips, uint32_t mip_count);

/* Band-sharded Build HZB for row-band sharding over several GPUs (no counterpart in the reference, which has one adapter; SURVEY.md
 * section 8e). The first launch of the chain (depth -> mips 0..4) works on 128 x 32-pixel source pieces, and every value of those
 * five levels depends on its own piece only: a rank builds the piece rows whose first source row lies in its band
 * (ur_hzb_band_pieces; it needs the depth rows of those pieces, i.e. up to 31 rows below its band), the ranks exchange the
 * slices (ur_hzb_band_slices names them: one contiguous run of floats per level), and every rank runs the single-workgroup rest of
 * the chain (ur_build_hzb_tail: mips 5.. from mip 4) behind the exchange. Bit for bit ur_build_hzb's chain. Only for chains that are
 * one five-level launch plus the tail (frames of a few hundred pixels up to 8K); UR_EUNSUPPORTED otherwise (build it whole).
 * With ur_defer_hzb_tail(ctx, 2) the band's pieces ride the next streaming Lighting launch like the whole chain's do. */
typedef struct ur_hzb_slice {
    uint32_t offset; /* in floats from hzb_base */
    uint32_t count;  /* floats */
} ur_hzb_slice;
int ur_hzb_band_pieces(uint32_t src_h, uint32_t n_ranks, uint32_t rank, uint32_t* piece_row0, uint32_t* piece_rows);
int ur_hzb_band_slices(const ur_mip_desc* mips, uint32_t mip_count, uint32_t piece_row0, uint32_t piece_rows, ur_hzb_slice* out5);
int ur_build_hzb_band(ur_ctx* ctx, const float* depth, uint32_t src_w, uint32_t src_h, float* hzb_base, const ur_mip_desc* mips,
                      uint32_t mip_count, uint32_t piece_row0, uint32_t piece_rows);
int ur_build_hzb_tail(ur_ctx* ctx, float* hzb_base, const ur_mip_desc* mips, uint32_t mip_count);

/* ---- CullIndirectArgs (+ visible-list compaction) ---------------------------------------------- */

/* constants: host, the 46 root constants packed by FRenderer::DispatchGpuCulling (Renderer.cpp:411-429):
 *   dw 0-23 FrustumPlanes[6], 24-39 ViewProjection (row-major), 40 ModelCount, 41 HZBEnabled,
 *   42 HZBMipCount, 43 HZBWidth, 44 HZBHeight, 45 DebugPrintEnabled.
 * bounds: device, 2*ModelCount float4 (min.xyz,_)(max.xyz,_) (CullIndirectArgs.hlsl:13,141-143).
 * hzb_base/mips: as produced by ur_build_hzb (mips host); ignored when HZBEnabled == 0.
 * indirect_args: device, ModelCount * 64 B; only the u32 at byte 44 of each command is written (0|1).
 * stats2: device u32[2] or NULL — [0] += frustum-culled, [1] += occluded when DebugPrintEnabled != 0
 *         (DebugPrintStats byte offsets 0 and 4, CullIndirectArgs.hlsl:156-166).
 * visible_idx / visible_count: device or NULL (both or neither) — NEW: ascending list of instance
 *         indices whose InstanceCount word is 1, and its length. Deterministic (no atomics-append).
 * index_base (_ex only): added to every emitted index, for instance-range sharding across ranks. */
int ur_cull_indirect_args(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds,
                          const float* hzb_base, const ur_mip_desc* mips, void* indirect_args,
                          uint32_t* stats2, uint32_t* visible_idx, uint32_t* visible_count);
int ur_cull_indirect_args_ex(ur_ctx* ctx, const uint32_t* constants, const ur_float4* bounds,
                             const float* hzb_base, const ur_mip_desc* mips, void* indirect_args,
                             uint32_t* stats2, uint32_t* visible_idx, uint32_t* visible_count,
                             uint32_t index_base);

/* ---- DeferredLighting / SkyAtmosphere ---------------------------------------------------------- */

/* Number of half4 units (8 bytes) ur_stage_env_cube() writes for (base_size, mip_count): the bordered faces, 6 (N+2)^2 texels per mip,
 * followed by the same texels as RGB row pairs, 6 (N+2)(N+1) entries of 12 bytes per mip; 0 on bad arguments. */
size_t ur_env_cube_texels(uint32_t base_size, uint32_t mip_count);
/* Stage an RGBA16F cube in DDS order (face-major, mips inner; TextureLoader.cpp:276-315) from HOST
 * memory into the device layout the lighting kernel samples: mip-major, 6 faces per mip, each face
 * (N+2)x(N+2) with a one-texel border holding the seamless neighbours from the adjacent faces; behind all mips the same faces
 * once more as RGB ROW PAIRS (entry (f, j, i) = {R G B of texel (i, j), R G B of texel (i, j+1)}, 12 bytes, pair-rows contiguous;
 * the shader never samples the cube's alpha: DeferredLighting.hlsl:82,86), so that a 2x2 bilinear footprint is 24 contiguous
 * bytes: the layout the streaming kernel gathers its prefiltered taps from.
 * dst_device must hold ur_env_cube_texels() units. Synchronous (setup time, like the DDS upload). */
int ur_stage_env_cube(ur_ctx* ctx, const ur_half4* src_host, uint32_t base_size, uint32_t mip_count,
                      ur_half4* dst_device);

/* DeferredLighting.hlsl:35-94 over rows [row0,row0+rows) of a w x h frame, additively blended
 * (ONE/ONE, colour and alpha; DeferredRenderer.cpp:1997-2005) into hdr_inout.
 * gbuf_a: (view normal.xyz, -viewZ) RGBA16F; gbuf_b: (specular, metallic, roughness, 1) RGBA16F;
 * gbuf_c: R8G8B8A8_UNORM_SRGB (R in the low byte); hdr_inout: RGBA16F holding (emissive, 1).
 * Like the reference it shades EVERY pixel, so cleared G-buffer pixels come out NaN.
 * Numerics: within max(1e-3, one fp16 ulp) of a scalar fp32 evaluation of the HLSL per channel (tests/test_gpu_parity.py).
 * The one input the reference would shade and this returns UR_EUNSUPPORTED for is a band of 2^29 pixels or more in one call.
 * Everything else is shaded by one of two kernels (a streaming one for the common shapes, a per-tile one for the rest: a
 * width that is not a multiple of 16, a perspective light, a LUT that is not 128x32, a ViewInverse that is not a rigid
 * transform or whose origin is not CameraPosition - world-space vectors are then formed literally -, a shadow map below
 * 3x3 texels, ...) with the same values. */
int ur_deferred_lighting(ur_ctx* ctx, const ur_scene_constants* scene, const ur_half4* gbuf_a,
                         const ur_half4* gbuf_b, const uint32_t* gbuf_c, const ur_lighting_tables* tables,
                         ur_half4* hdr_inout, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows);

/* SkyAtmosphere.hlsl:29-101 restated per pixel: the inside-out sphere of radius World[0] centred on
 * the camera covers every pixel at depth Near / (R * dir_view.z); colour (sky, 1) is written, no
 * blend, where that depth >= depth[pixel] (GREATER_EQUAL, DeferredRenderer.cpp:361-365). */
int ur_sky_atmosphere(ur_ctx* ctx, const ur_sky_constants* sky, const float* depth, ur_half4* hdr_inout,
                      uint32_t w, uint32_t h, uint32_t row0, uint32_t rows);

/* Lighting + Sky in one pass over the band: pixels the sky pass would overwrite skip the lighting
 * math; every other pixel gets exactly ur_deferred_lighting's value. Result == the two calls above. */
int ur_deferred_lighting_sky(ur_ctx* ctx, const ur_scene_constants* scene, const ur_sky_constants* sky,
                             const ur_half4* gbuf_a, const ur_half4* gbuf_b, const uint32_t* gbuf_c,
                             const float* depth, const ur_lighting_tables* tables, ur_half4* hdr_inout,
                             uint32_t w, uint32_t h, uint32_t row0, uint32_t rows);

/* ---- Tonemap (next row after the path) ----------------------------------------------- */

/* TonemapParams root constants (Shaders/Tonemap.hlsl:22-28; FTonemapConstants, DeferredRenderer.cpp:1481-1495). */
typedef struct ur_tonemap_constants {
    uint32_t EnableTonemap;
    uint32_t EnableAutoExposure;
    float Exposure;
    float Gamma;
} ur_tonemap_constants;

/* Tonemap.hlsl:57-79 over a band of w x rows pixels: hdr (RGBA16F) * Exposure [* 2^exposure_ev[0] when auto exposure is
 * on; exposure_ev = device pointer to the LogAverageLuminance texel, nullable], Khronos PBR-neutral curve, saturate,
 * pow(1/max(Gamma,1e-3)), written as R8G8B8A8_UNORM (R in the low byte, A = 255). 12 B/pixel.
 * Numerics: each byte within 1 LSB of a scalar evaluation with libm's powf (the kernel raises through exp2/log2 and divides
 * through v_rcp; the reference's own pow is the D3D driver's); every launch shape gives the same bits for the same pixel. */
int ur_tonemap(ur_ctx* ctx, const ur_tonemap_constants* constants, const ur_half4* hdr, const float* exposure_ev, uint32_t* out_rgba8,
               uint32_t w, uint32_t rows);

/* ---- TemporalAA resolve (next row; Shaders/TemporalAA.hlsl:12-50, DeferredRenderer.cpp:1308-1361) ---- */

/* current_frame: device, the FULL w x h RGBA16F frame (the 3x3 neighbourhood of a band's edge rows lies outside the
 * band; with multi-GPU sharding this is the all-gathered frame). history_band / output_band: band-local rows
 * [row0,row0+rows). use_history == 0 copies current (first frame). Bit-exact against the oracle. 24 B/pixel. */
int ur_temporal_aa(ur_ctx* ctx, const ur_half4* current_frame, const ur_half4* history_band, ur_half4* output_band, float history_weight,
                   uint32_t use_history, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows);

/* ---- multi-GPU: gather the row bands of the HDR frame ------------------------------------------ */

/* comm: an ncclComm_t (RCCL). hdr_full: device, w*h half4 on every rank; rank r has already written
 * rows [r*h/n, (r+1)*h/n). In-place ncclAllGather on the ctx stream. Requires n | h. */
int ur_allgather_rows(ur_ctx* ctx, void* comm, ur_half4* hdr_full, uint32_t w, uint32_t h,
                      uint32_t n_ranks, uint32_t rank);
/* The same for any row-major image of `row_bytes` per row — the tonemapped R8G8B8A8 band (ur_tonemap ahead of the gather:
 * 4 B/pixel over xGMI instead of 8; SURVEY.md §8f-1, Shaders/Tonemap.hlsl:57-79). */
int ur_allgather_rows_bytes(ur_ctx* ctx, void* comm, void* image, uint32_t row_bytes, uint32_t h, uint32_t n_ranks,
                            uint32_t rank);
/* The same with the form of the exchange chosen by the caller. UR_GATHER_RING: one ncclAllGather (RCCL picks ring or tree).
 * UR_GATHER_DIRECT: ncclGroupStart, N - 1 ncclSend / ncclRecv pairs (this rank's band to every peer, every peer's band
 * into its rows), ncclGroupEnd — every transfer crosses exactly the one xGMI link its two GPUs share and all of a rank's
 * seven are in flight together, which is the floor SURVEY.md H1 prices (band_bytes / 153 GB/s), with no ring hops.
 * Same result bytes. */
#define UR_GATHER_RING 0
#define UR_GATHER_DIRECT 1
int ur_allgather_rows_bytes_ex(ur_ctx* ctx, void* comm, void* image, uint32_t row_bytes, uint32_t h, uint32_t n_ranks,
                               uint32_t rank, int mode);

#ifdef __cplusplus
}
#endif
#endif /* UR_HOTPATH_H */
