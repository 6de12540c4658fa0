"""Python face of the C-ABI for device tensors (torch is used for device memory and streams only).

Method names follow the reference's passes: CullIndirectArgs (Renderer.cpp:394 DispatchGpuCulling), BuildHZB
(DeferredRenderer.cpp:998), DeferredLighting (:1219), SkyAtmosphere (:1263).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as _lib


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA/HIP tensors"
    return C.c_void_p(t.data_ptr())


class HzbLayout:
    def __init__(self, src_w: int, src_h: int):
        self.src_w, self.src_h = src_w, src_h
        self.mips = (_lib.MipDesc * _lib.UR_MAX_HZB_MIPS)()
        n = C.c_uint32(0)
        self.total = int(_lib.load().ur_hzb_layout(src_w, src_h, self.mips, C.byref(n)))
        if self.total == 0:
            raise ValueError(f"bad HZB source size {src_w}x{src_h}")
        self.count = int(n.value)

    @property
    def width(self):
        return self.mips[0].width

    @property
    def height(self):
        return self.mips[0].height

    def as_list(self):
        return [(self.mips[i].offset, self.mips[i].width, self.mips[i].height) for i in range(self.count)]

    def mip_texels(self) -> int:
        return sum(self.mips[i].width * self.mips[i].height for i in range(self.count))

    def band_pieces(self, world: int, rank: int) -> tuple[int, int]:
        """(first piece row, piece rows) of the wide Build HZB launch that `rank` of `world` row bands builds (ur_hzb_band_pieces)."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _lib.check(_lib.load().ur_hzb_band_pieces(self.src_h, world, rank, C.byref(a), C.byref(b)), "ur_hzb_band_pieces")
        return int(a.value), int(b.value)

    def band_slices(self, piece_row0: int, piece_rows: int) -> list[tuple[int, int]]:
        """[(offset, count) in floats] of mips 0..4 for those piece rows (ur_hzb_band_slices): what a rank contributes to the exchange."""
        out = (_lib.HzbSlice * 5)()
        _lib.check(_lib.load().ur_hzb_band_slices(self.mips, self.count, piece_row0, piece_rows, out), "ur_hzb_band_slices")
        return [(int(s.offset), int(s.count)) for s in out]


class HotPath:
    """One ur_ctx bound to a device and a stream."""

    def __init__(self, device: int | None = None, stream: "torch.cuda.Stream | None" = None):
        self._L = _lib.load()  # raises if the HIP extension is not built
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the hot path has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else device
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._ctx = self._L.ur_create(self.device, C.c_void_p(self.stream.cuda_stream))
        if not self._ctx:
            raise RuntimeError("ur_create failed: " + self._L.ur_last_error().decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.ur_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        return self._ctx

    def set_option(self, option: int, value: int):
        """Launch-shape option of this context (lib.UR_OPT_*): results are the same bits under every value."""
        _lib.check(self._L.ur_set_option(self._ctx, option, value), "ur_set_option")

    def get_option(self, option: int) -> int:
        v = C.c_int(0)
        _lib.check(self._L.ur_get_option(self._ctx, option, C.byref(v)), "ur_get_option")
        return int(v.value)

    def lighting_schedule(self) -> dict:
        """Tile schedule of the last streaming Lighting launch on this context (ur_debug_lighting_schedule)."""
        out = (C.c_uint32 * 8)()
        _lib.check(self._L.ur_debug_lighting_schedule(self._ctx, out), "ur_debug_lighting_schedule")
        keys = ("groups", "tiles", "static_tiles", "pool_chunks", "chunk_shift", "lookahead", "waves_per_wg", "hzb_pieces")
        return dict(zip(keys, (int(v) for v in out)))

    def reserve(self, max_instances: int):
        _lib.check(self._L.ur_reserve(self._ctx, max_instances), "ur_reserve")

    def defer_hzb_tail(self, enable: "bool | int"):
        """1 / True: hold back the single-workgroup tail of build_hzb so that it rides along with the next streaming lighting
        launch; 2: hold back the whole chain (the lighting workgroups take the wide launch's pieces along too); 0: off."""
        _lib.check(self._L.ur_defer_hzb_tail(self._ctx, int(enable)), "ur_defer_hzb_tail")

    def debug_timeline(self, pairs: "torch.Tensor | None"):
        """pairs: (n, 2) int64 device tensor initialised to [-1, 0] rows (= {~0, 0} as uint64), or None to switch it off."""
        if pairs is None:
            _lib.check(self._L.ur_debug_timeline(self._ctx, None, 0), "ur_debug_timeline")
        else:
            assert pairs.dtype == torch.int64 and pairs.dim() == 2 and pairs.shape[1] == 2
            _lib.check(self._L.ur_debug_timeline(self._ctx, _ptr(pairs), pairs.shape[0]), "ur_debug_timeline")

    def flush(self):
        _lib.check(self._L.ur_flush(self._ctx), "ur_flush")

    def time_next_lighting(self, start: "torch.cuda.Event | None", stop: "torch.cuda.Event | None"):
        """The next Lighting launch carries this event pair on its kernel dispatch (ur_time_next_lighting): after a
        synchronise, start.elapsed_time(stop) is the dispatch's duration (launch included). The events must have been created
        with enable_timing=True and recorded once before (torch creates the HIP event lazily at its first record)."""
        if stop is None:
            _lib.check(self._L.ur_time_next_lighting(self._ctx, None, None), "ur_time_next_lighting")
        else:
            _lib.check(self._L.ur_time_next_lighting(self._ctx, C.c_void_p(start.cuda_event) if start is not None else None, C.c_void_p(stop.cuda_event)),
                       "ur_time_next_lighting")

    def stream_ceiling(self, ins, out, start: "torch.cuda.Event | None" = None, stop: "torch.cuda.Event | None" = None):
        """out = ins[0] + ins[1] + ins[2] + ins[3] on 16-byte elements (ur_debug_stream_ceiling): the plain streaming kernel whose
        rate bench.py prints as the practical ceiling. Events (recorded once before, like time_next_lighting's) ride on the dispatch."""
        assert len(ins) == 4 and all(t.numel() * t.element_size() == out.numel() * out.element_size() for t in ins)
        n16 = out.numel() * out.element_size() // 16
        _lib.check(self._L.ur_debug_stream_ceiling(self._ctx, _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]), _ptr(out), n16,
                                                   C.c_void_p(start.cuda_event) if start is not None else None,
                                                   C.c_void_p(stop.cuda_event) if stop is not None else None), "ur_debug_stream_ceiling")

    # ---- BuildHZB ----
    def build_hzb(self, depth: torch.Tensor, hzb: torch.Tensor, layout: HzbLayout):
        assert depth.dtype == torch.float32 and hzb.dtype == torch.float32 and hzb.numel() >= layout.total
        assert depth.numel() == layout.src_w * layout.src_h
        _lib.check(self._L.ur_build_hzb(self._ctx, _ptr(depth), layout.src_w, layout.src_h, _ptr(hzb), layout.mips, layout.count), "ur_build_hzb")

    def build_hzb_band(self, depth: torch.Tensor, hzb: torch.Tensor, layout: HzbLayout, piece_row0: int, piece_rows: int):
        """Mips 0..4 for the 128x32 source pieces of rows [piece_row0, piece_row0 + piece_rows) only (ur_build_hzb_band)."""
        assert depth.dtype == torch.float32 and hzb.dtype == torch.float32 and hzb.numel() >= layout.total and depth.numel() == layout.src_w * layout.src_h
        _lib.check(self._L.ur_build_hzb_band(self._ctx, _ptr(depth), layout.src_w, layout.src_h, _ptr(hzb), layout.mips, layout.count, piece_row0, piece_rows),
                   "ur_build_hzb_band")

    def build_hzb_tail(self, hzb: torch.Tensor, layout: HzbLayout):
        """The single-workgroup rest of the chain (mips 5.. from mip 4), behind the ranks' exchange of the band slices (ur_build_hzb_tail)."""
        _lib.check(self._L.ur_build_hzb_tail(self._ctx, _ptr(hzb), layout.mips, layout.count), "ur_build_hzb_tail")

    # ---- CullIndirectArgs ----
    def cull_indirect_args(self, constants: np.ndarray, bounds: torch.Tensor, hzb, layout, indirect_args: torch.Tensor,
                           stats=None, visible_idx=None, visible_count=None, index_base: int = 0):
        constants = np.ascontiguousarray(constants, np.uint32)
        assert constants.size == _lib.UR_CULL_CONSTANT_DWORDS
        cptr = constants.ctypes.data_as(C.POINTER(C.c_uint32))
        mips = layout.mips if layout is not None else None
        _lib.check(self._L.ur_cull_indirect_args_ex(self._ctx, cptr, _ptr(bounds), _ptr(hzb), mips, _ptr(indirect_args), _ptr(stats),
                                                    _ptr(visible_idx), _ptr(visible_count), index_base), "ur_cull_indirect_args")

    # ---- lighting tables ----
    def stage_env_cube(self, cube_dds_order: np.ndarray, base: int, mips: int) -> torch.Tensor:
        src = np.ascontiguousarray(cube_dds_order, np.uint16)
        n = int(self._L.ur_env_cube_texels(base, mips))
        if n == 0:
            raise ValueError("bad cube size")
        expect = 6 * sum(max(1, base >> m) ** 2 for m in range(mips))
        assert src.size == expect * 4, f"cube has {src.size // 4} texels, expected {expect}"
        dst = torch.empty((n, 4), dtype=torch.int16, device=f"cuda:{self.device}")
        _lib.check(self._L.ur_stage_env_cube(self._ctx, src.ctypes.data_as(C.c_void_p), base, mips, _ptr(dst)), "ur_stage_env_cube")
        return dst

    @staticmethod
    def make_tables(shadow, env_cube, env_base: int, env_mips: int, lut) -> _lib.LightingTables:
        t = _lib.LightingTables()
        t.shadow_map = shadow.data_ptr() if shadow is not None else None
        t.env_cube = env_cube.data_ptr()
        t.env_base_size, t.env_mip_count = env_base, env_mips
        t.env_cube_texels = env_cube.numel() * env_cube.element_size() // 8  # the staged buffer's size in half4 texels = the layout's tag
        t.brdf_lut_rg16 = lut.data_ptr()
        t.lut_height, t.lut_width = int(lut.shape[0]), int(lut.shape[1])
        t._keep = (shadow, env_cube, lut)  # keep the tensors alive
        return t

    # ---- DeferredLighting / SkyAtmosphere ----
    def deferred_lighting(self, scene, A, B, Cc, tables, hdr, w, h, row0=0, rows=None):
        rows = h - row0 if rows is None else rows
        _lib.check(self._L.ur_deferred_lighting(self._ctx, C.byref(scene), _ptr(A), _ptr(B), _ptr(Cc), C.byref(tables), _ptr(hdr), w, h, row0, rows),
                   "ur_deferred_lighting")

    def sky_atmosphere(self, sky, depth, hdr, w, h, row0=0, rows=None):
        rows = h - row0 if rows is None else rows
        _lib.check(self._L.ur_sky_atmosphere(self._ctx, C.byref(sky), _ptr(depth), _ptr(hdr), w, h, row0, rows), "ur_sky_atmosphere")

    def deferred_lighting_sky(self, scene, sky, A, B, Cc, depth, tables, hdr, w, h, row0=0, rows=None):
        rows = h - row0 if rows is None else rows
        _lib.check(self._L.ur_deferred_lighting_sky(self._ctx, C.byref(scene), C.byref(sky), _ptr(A), _ptr(B), _ptr(Cc), _ptr(depth), C.byref(tables),
                                                    _ptr(hdr), w, h, row0, rows), "ur_deferred_lighting_sky")


class Frame:
    """The render-graph-driven frame (csrc/frame/HotPathRenderer.cpp): GPU Culling -> Build HZB -> Lighting -> Sky added to
    an FRenderGraph in the reference's order and executed on the context's stream."""

    def __init__(self, hp: HotPath, frames_in_flight: int = 3, rank: int = 0, world_size: int = 1):
        self._hp = hp
        self._L = hp._L
        self._f = self._L.ur_frame_create(hp.ctx, C.c_void_p(hp.stream.cuda_stream), frames_in_flight, rank, world_size)
        if not self._f:
            raise RuntimeError("ur_frame_create failed")
        self._keep = None

    def close(self):
        if getattr(self, "_f", None):
            self._L.ur_frame_destroy(self._f)
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def resources(w, h, row0, rows, A, B, Cc, depth_band, lighting_band, depth_full, hzb, layout: HzbLayout, tables, bounds=None,
                  indirect_args=None, command_count=0, index_base=0, visible_idx=None, visible_count=None, cull_stats=None, tonemap_band=None):
        r = _lib.FrameResources()
        r.width, r.height, r.row0, r.rows = w, h, row0, rows
        dp = lambda t: t.data_ptr() if t is not None else None
        r.gbuffer_a, r.gbuffer_b, r.gbuffer_c = dp(A), dp(B), dp(Cc)
        r.depth_band, r.lighting_band, r.depth_full, r.hzb = dp(depth_band), dp(lighting_band), dp(depth_full), dp(hzb)
        if layout is not None:
            for i in range(layout.count):
                r.hzb_mips[i] = layout.mips[i]
            r.hzb_mip_count = layout.count
        r.tables = tables
        r.model_bounds, r.indirect_args = dp(bounds), dp(indirect_args)
        r.indirect_command_count, r.instance_index_base = command_count, index_base
        r.visible_indices, r.visible_count, r.cull_stats = dp(visible_idx), dp(visible_count), dp(cull_stats)
        r.tonemap_band = dp(tonemap_band)
        r._keep = (A, B, Cc, depth_band, lighting_band, depth_full, hzb, tables, bounds, indirect_args, visible_idx, visible_count, cull_stats, tonemap_band)
        return r

    def render(self, res, culling_constants: np.ndarray, scene, sky, flags: int = _lib.UR_FRAME_DEFAULT):
        cc = np.ascontiguousarray(culling_constants, np.uint32)
        _lib.check(self._L.ur_frame_render(self._f, C.byref(res), cc.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(scene), C.byref(sky), flags),
                   "ur_frame_render")

    def lighting_times_ms(self) -> np.ndarray:
        """Durations of the Lighting passes bracketed with UR_FRAME_TIME_LIGHTING since the last call (synchronise first)."""
        buf = np.zeros(1024, np.float32)
        n = self._L.ur_frame_lighting_times(self._f, _lib.fptr(buf), 1024)
        return buf[:n].copy()

    def lighting_times_and_record_cost_ms(self):
        """(bracket durations, cost of one event record behind each bracket) of the passes timed with UR_FRAME_TIME_LIGHTING."""
        buf, rec = np.zeros(1024, np.float32), np.zeros(1024, np.float32)
        n = self._L.ur_frame_lighting_times_ex(self._f, _lib.fptr(buf), _lib.fptr(rec), 1024)
        return buf[:n].copy(), rec[:n].copy()

    def join_async(self):
        self._L.ur_frame_join_async(self._f)

    @property
    def hzb_ready(self) -> bool:
        return bool(self._L.ur_frame_hzb_ready(self._f))

    def reset_hzb(self):
        self._L.ur_frame_reset_hzb(self._f)

    def report(self):
        """[(pass name, culled, transitions)] of the last executed graph."""
        n = self._L.ur_frame_report(self._f, None, 0)
        buf = C.create_string_buffer(n)
        self._L.ur_frame_report(self._f, buf, n)
        out = []
        for line in buf.value.decode().splitlines():
            name, culled, tr, *rest = line.split("|")
            out.append((name, culled == "1", int(tr)))
        return out

    def report_async(self):
        """[(pass name, ran on the async-compute stream, cross-stream waits)] of the last executed graph."""
        n = self._L.ur_frame_report(self._f, None, 0)
        buf = C.create_string_buffer(n)
        self._L.ur_frame_report(self._f, buf, n)
        return [(f[0], f[3] == "1", int(f[4])) for f in (l.split("|") for l in buf.value.decode().splitlines())]

    def timing_stats(self):
        n = self._L.ur_rg_timing_stats(None, 0)
        buf = C.create_string_buffer(n)
        self._L.ur_rg_timing_stats(buf, n)
        return [tuple(l.split("|")) for l in buf.value.decode().splitlines()]


def _tonemap(self, hdr, out_rgba8, w, rows, exposure=1.0, gamma=2.2, enable_tonemap=True, exposure_ev=None):
    """Tonemap pass (Tonemap.hlsl) over a band: RGBA16F -> R8G8B8A8_UNORM."""
    k = _lib.TonemapConstants(int(enable_tonemap), int(exposure_ev is not None), exposure, gamma)
    _lib.check(self._L.ur_tonemap(self._ctx, C.byref(k), _ptr(hdr), _ptr(exposure_ev), _ptr(out_rgba8), w, rows), "ur_tonemap")


HotPath.tonemap = _tonemap


def _temporal_aa(self, current_frame, history_band, output_band, history_weight, use_history, w, h, row0=0, rows=None):
    """TemporalAA resolve (TemporalAA.hlsl): current = full frame, history/output = band rows [row0,row0+rows)."""
    rows = h - row0 if rows is None else rows
    _lib.check(self._L.ur_temporal_aa(self._ctx, _ptr(current_frame), _ptr(history_band), _ptr(output_band), history_weight, int(use_history), w, h, row0, rows),
               "ur_temporal_aa")


HotPath.temporal_aa = _temporal_aa


def to_device(a: np.ndarray, device=0) -> torch.Tensor:
    """numpy -> device tensor, reinterpreting unsigned dtypes torch cannot hold (bit patterns are preserved)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    elif a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(f"cuda:{device}")


def to_host(t: torch.Tensor, dtype) -> np.ndarray:
    return t.detach().cpu().numpy().view(dtype)
