// The practical ceiling of a plain streaming kernel on this chip, measured next to the real kernels (bench.py, roofline.stream_ceiling_*):
// out[i] = in0[i] + in1[i] + in2[i] + in3[i] on 16-byte elements - four read streams and one write stream, the byte mix of the
// fused Lighting launch (A, B, HDR in, C + depth : HDR out) with nothing to compute. One element per lane, every load issued before
// the first add (the form that was fastest at this size in round 1: profiles/r01_stream_ceiling.txt), nontemporal loads and stores (every
// byte is touched once: 64.8 -> 60.8 us at the Lighting launch's byte count, 5.65 -> 6.0 TB/s - the ceiling is the BEST plain stream). Not a product kernel: it has no
// counterpart in the reference; it exists so that the roofline fraction can be read against what the memory system delivers to
// ANY kernel of this shape and size.

#include <hip/hip_ext.h>

#include "ur_internal.h"

namespace {

struct StreamPtrs { const float4* in[4]; float4* out; uint32_t n; };

__global__ __launch_bounds__(256) void stream4_kernel(StreamPtrs p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t j = min(i, p.n - 1u);
    typedef float sf32x4_t __attribute__((ext_vector_type(4)));
    const sf32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const sf32x4_t*>(p.in[0]) + j), b = __builtin_nontemporal_load(reinterpret_cast<const sf32x4_t*>(p.in[1]) + j),
                   c = __builtin_nontemporal_load(reinterpret_cast<const sf32x4_t*>(p.in[2]) + j), d = __builtin_nontemporal_load(reinterpret_cast<const sf32x4_t*>(p.in[3]) + j);
    if (i < p.n) __builtin_nontemporal_store(sf32x4_t{(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)}, reinterpret_cast<sf32x4_t*>(p.out) + i);
}

} // namespace

extern "C" int ur_debug_stream_ceiling(ur_ctx* ctx, const void* in0, const void* in1, const void* in2, const void* in3, void* out, uint64_t elements16,
                                       void* start_event, void* stop_event)
{
    if (!ctx || !in0 || !in1 || !in2 || !in3 || !out || elements16 == 0 || elements16 > 0xFFFFFF00ull || (start_event != nullptr && stop_event == nullptr)) {
        ur::set_error("ur_debug_stream_ceiling: bad argument");
        return UR_EINVAL;
    }
    StreamPtrs p{{static_cast<const float4*>(in0), static_cast<const float4*>(in1), static_cast<const float4*>(in2), static_cast<const float4*>(in3)},
                 static_cast<float4*>(out), (uint32_t)elements16};
    const dim3 grid((uint32_t)((elements16 + 255u) / 256u));
    if (stop_event != nullptr)
        hipExtLaunchKernelGGL(stream4_kernel, grid, dim3(256), 0, ctx->stream, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event), 0, p);
    else
        hipLaunchKernelGGL(stream4_kernel, grid, dim3(256), 0, ctx->stream, p);
    UR_HIP_TRY(hipGetLastError());
    return UR_OK;
}
