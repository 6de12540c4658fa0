"""Hand-derived known answers that pin the CPU oracle (the reference has no tests or golden vectors for this path —
SURVEY.md §4/§8c — so these, the internal GPU-form/CPU-form cross-check and tests/golden/ are what pins it)."""
import math

import numpy as np
import pytest

from unclerenderer_amd import hostmath, lib, synth


# ---------------------------------------------------------------------------------------------------------------------
# fp16 conversion
# ---------------------------------------------------------------------------------------------------------------------
def test_half_conversions(oracle):
    all_bits = np.arange(65536, dtype=np.uint16)
    ref = all_bits.view(np.float16).astype(np.float32)
    got = np.array([oracle.h2f(int(b)) for b in all_bits[::7]], np.float32)
    r = ref[::7]
    assert np.array_equal(np.isnan(got), np.isnan(r)) and np.array_equal(got[~np.isnan(r)].view(np.uint32), r[~np.isnan(r)].view(np.uint32))
    cases = {1.0: 0x3C00, 65504.0: 0x7BFF, 65519.99: 0x7BFF, 65520.0: 0x7C00, 1e9: 0x7C00, -2.0: 0xC000, 2.0 ** -24: 0x0001,
             2.0 ** -25: 0x0000, 1.5 * 2.0 ** -25: 0x0001, 1.0 + 2.0 ** -11: 0x3C00, 1.0 + 3 * 2.0 ** -11: 0x3C02, 0.0: 0x0000,
             2.0 ** -14: 0x0400, 2.0 ** -14 - 2.0 ** -25: 0x0400, 0.1: 0x2E66}
    for f, h in cases.items():
        assert oracle.f2h(f) == h, (f, hex(oracle.f2h(f)), hex(h))
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(size=20000) * 10.0 ** rng.integers(-9, 5, 20000), rng.random(5000) * 1e-6]).astype(np.float32)
    want = x.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f2h(float(v)) for v in x], np.uint16)
    assert np.array_equal(got, want)
    assert oracle.f2h(float("nan")) & 0x7C00 == 0x7C00 and oracle.f2h(float("nan")) & 0x3FF != 0


# ---------------------------------------------------------------------------------------------------------------------
# BuildHZB
# ---------------------------------------------------------------------------------------------------------------------
def _hzb(oracle, depth):
    h, w = depth.shape
    mips, total = oracle.hzb_layout(w, h)
    out = oracle.build_hzb(depth, mips, total)
    return [out[o:o + mw * mh].reshape(mh, mw) for (o, mw, mh) in mips]


def test_hzb_known_answers(oracle):
    d = np.array([[0.4, 0.9], [0.7, 0.2]], np.float32)
    assert _hzb(oracle, d)[0].tolist() == [[np.float32(0.2)]]
    assert _hzb(oracle, np.array([[0.37]], np.float32))[0][0, 0] == np.float32(0.37)  # 1x1: clamped 2x2 footprint of one texel
    # 3 wide x 5 high -> base 2x3 (round up) with clamped reads, then 1x1 (floor) that never sees row 2 of mip 0
    d = (np.arange(15, dtype=np.float32).reshape(5, 3) + 1) / 16
    m = _hzb(oracle, d)
    assert [x.shape for x in m] == [(3, 2), (1, 1)]
    expect0 = np.array([[min(d[0, 0], d[0, 1], d[1, 0], d[1, 1]), min(d[0, 2], d[1, 2])],
                        [min(d[2, 0], d[2, 1], d[3, 0], d[3, 1]), min(d[2, 2], d[3, 2])],
                        [min(d[4, 0], d[4, 1]), d[4, 2]]], np.float32)
    assert np.array_equal(m[0], expect0)
    assert m[1][0, 0] == expect0[:2].min()  # row 2 (from depth row 4) is dropped by the floor halving — non-conservative, as is
    # even sizes: every mip is the exact 2x2 min-pool of its parent and the top is the global min
    rng = np.random.default_rng(3)
    d = rng.random((64, 64), dtype=np.float32)
    m = _hzb(oracle, d)
    parent = d
    for level in m:
        pooled = parent.reshape(parent.shape[0] // 2, 2, parent.shape[1] // 2, 2).min(axis=(1, 3))
        assert np.array_equal(level, pooled)
        parent = level
    assert m[-1][0, 0] == d.min()


def test_hzb_h8_quirk_is_reproduced(oracle):
    """17x9 -> mips 9x5, 4x2, 2x1, 1x1 all inside ONE <=4-mip dispatch. The 1x1 mip reads the 2x2 block of the 2x1 mip;
    its out-of-range row holds 0.0 (BuildHZB.hlsl:104) and poisons the min. Reproduced, not fixed."""
    d = np.full((9, 17), 0.5, np.float32)
    m = _hzb(oracle, d)
    assert [x.shape for x in m] == [(5, 9), (2, 4), (1, 2), (1, 1)]
    assert (m[0] == 0.5).all() and (m[1] == 0.5).all() and (m[2] == 0.5).all()
    assert m[3][0, 0] == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# CullIndirectArgs
# ---------------------------------------------------------------------------------------------------------------------
def _simple_camera(w=8, h=8):
    view = np.eye(4, dtype=np.float32).ravel()
    proj = hostmath.reverse_z_projection(math.radians(90.0), 1.0, 0.1)  # xs = ys = 1
    return view, proj


def _box(cx, cy, cz, e=0.5):
    b = np.zeros((2, 4), np.float32)
    b[0, :3] = [cx - e, cy - e, cz - e]
    b[1, :3] = [cx + e, cy + e, cz + e]
    return b


def test_cull_frustum_known_answers(oracle, urlib):
    view, proj = _simple_camera()
    boxes = {
        "centre": (_box(0, 0, 5), 1), "right of frustum": (_box(10, 0, 5), 0), "left": (_box(-10, 0, 5), 0),
        "above": (_box(0, 10, 5), 0), "below": (_box(0, -10, 5), 0), "behind camera": (_box(0, 0, -5), 0),
        "straddles right plane": (_box(5, 0, 5), 1), "straddles near plane": (_box(0, 0, 0.1), 1),
        "just outside right": (_box(6.01, 0, 5.0), 0), "touching right plane": (_box(6.0, 0, 5.0), 1),
        "far away (no far plane)": (_box(0, 0, 1e6, 10.0), 1), "contains camera": (_box(0, 0, 0, 3.0), 1),
    }
    bounds = np.stack([b for b, _ in boxes.values()])
    n = len(boxes)
    consts = hostmath.pack_culling_constants(view, proj, n, False, 0, 0, 0, True)
    args, stats, vis, cnt = oracle.cull_indirect_args(consts, bounds, None, [], synth.indirect_args_initial(n))
    want = np.array([v for _, v in boxes.values()], np.uint32)
    assert args[:, 11].tolist() == want.tolist(), dict(zip(boxes, args[:, 11]))
    assert vis.tolist() == np.flatnonzero(want).tolist() and cnt == want.sum()
    assert stats.tolist() == [int((want == 0).sum()), 0]
    # only the InstanceCount dword changes
    ref = synth.indirect_args_initial(n); ref[:, 11] = want
    assert np.array_equal(args, ref)
    # CPU-side form of the same test (RendererUtils.cpp:1192-1218) agrees
    assert oracle.cpu_frustum(consts[:24].view(np.float32), bounds).tolist() == want.tolist()
    for (b, v) in boxes.values():
        assert hostmath.is_aabb_in_frustum(consts[:24].view(np.float32), b[0, :3], b[1, :3]) == bool(v)


def _hzb_const(value_per_mip, w=8, h=8):
    mips, total = [], 0
    mw, mh = (w + 1) // 2, (h + 1) // 2
    while True:
        mips.append((total, mw, mh)); total += mw * mh
        if mw == 1 and mh == 1:
            break
        mw, mh = max(1, mw // 2), max(1, mh // 2)
    buf = np.zeros(total, np.float32)
    for (o, a, b), v in zip(mips, value_per_mip):
        buf[o:o + a * b] = v
    return mips, buf


def test_cull_occlusion_known_answers(oracle, urlib):
    view, proj = _simple_camera()
    box = _box(0, 0, 5)  # nearest corner z = 4.5 -> maxDepth = 0.1 / 4.5
    max_depth = np.float32(np.float32(0.1) / np.float32(4.5))

    def run(bounds, mip_values):
        mips, hzb = _hzb_const(mip_values)
        n = bounds.shape[0]
        c = hostmath.pack_culling_constants(view, proj, n, True, len(mips), mips[0][1], mips[0][2], True)
        args, stats, vis, cnt = oracle.cull_indirect_args(c, bounds, hzb, mips, synth.indirect_args_initial(n))
        return args[:, 11].tolist(), stats.tolist()

    one = box[None]
    assert run(one, [0.5, 0.5, 0.5]) == ([0], [0, 1])            # wall at z = 0.2 in front of the box: occluded
    assert run(one, [0.01, 0.01, 0.01]) == ([1], [0, 0])         # wall at z = 10 behind it: visible
    assert run(one, [max_depth] * 3) == ([1], [0, 0])            # equality: "maxDepth < hzbDepth" is false -> visible
    assert run(one, [np.nextafter(max_depth, np.float32(1))] * 3) == ([0], [0, 1])
    # a corner behind the camera (w <= 0) is never occluded, whatever the HZB says
    span = np.zeros((1, 2, 4), np.float32); span[0, 0, :3] = [-0.5, -0.5, -1.0]; span[0, 1, :3] = [0.5, 0.5, 5.0]
    assert run(span, [1.0, 1.0, 1.0]) == ([1], [0, 0])
    # mip selection: base 4x4. The box at z=5 spans uv 0.4..0.6 -> 0.8 texels -> mip 0; a box filling the screen spans
    # 4 texels -> floor(log2(4)) = 2 -> the 1x1 mip. Mark mip 0 "never occludes" (0.0) and mip 2 "always occludes" (1.0).
    big = _box(0, 0, 5, 4.6)
    assert run(one, [0.0, 0.5, 1.0]) == ([1], [0, 0])
    assert run(big[None], [0.0, 0.5, 1.0])[0] == [0]
    mid = _box(0, 0, 5, 1.5)  # nearest z 3.5: uv half-extent 1.5/3.5/2 = .214 -> 1.71 texels -> mip 0 still (floor(log2 1.71) = 0)
    assert run(mid[None], [0.0, 1.0, 1.0])[0] == [1]
    mid2 = _box(0, 0, 5, 2.0)  # nearest z 3: half-extent .333 -> 2.67 texels -> mip 1
    assert run(mid2[None], [0.0, 1.0, 0.0])[0] == [0]
    # rectangle entirely off-screen cannot happen for a frustum survivor with all w > 0, but HZB disabled must mean visible
    c = hostmath.pack_culling_constants(view, proj, 1, False, 3, 4, 4, False)
    mips, hzb = _hzb_const([1.0, 1.0, 1.0])
    assert oracle.cull_indirect_args(c, one, hzb, mips, synth.indirect_args_initial(1))[0][0, 11] == 1


def test_cull_gpu_form_equals_cpu_form_on_random_instances(oracle, urlib):
    for scene in ("sponza", "duck", "pica_pica"):
        fc = hostmath.build_frame_constants(scene, 1280, 720)
        n = 20000
        b = synth.instances_random(n, 77, center=fc.camera_position, box=150.0)
        c = hostmath.pack_culling_constants(fc.view, fc.proj, n, False, 0, 0, 0, False)
        args, *_ = oracle.cull_indirect_args(c, b, None, [], synth.indirect_args_initial(n))
        cpu = oracle.cpu_frustum(c[:24].view(np.float32), b)
        assert np.array_equal(args[:, 11], cpu)
        assert 0 < cpu.sum() < n


# ---------------------------------------------------------------------------------------------------------------------
# PBR / sky / samplers
# ---------------------------------------------------------------------------------------------------------------------
def test_evaluate_pbr_known_answers(oracle):
    z = [0.0, 0.0, 1.0]
    # N = V = L: all dots 1 -> D = 1/(pi a^2), G = 1, F = F0; spec = F0 / (4 pi a^2); diffuse = (1-F0)(1-m) albedo (no 1/pi)
    got = oracle.evaluate_pbr([0.5] * 3, 0.0, 1.0, [0.04] * 3, z, z, z)
    np.testing.assert_allclose(got, [0.96 * 0.5 + 0.04 / (4 * math.pi)] * 3, rtol=2e-6)
    got = oracle.evaluate_pbr([0.9, 0.6, 0.3], 1.0, 0.5, [0.9, 0.6, 0.3], z, z, z)  # pure metal, a = 0.25
    np.testing.assert_allclose(got, np.array([0.9, 0.6, 0.3]) / (4 * math.pi * 0.0625), rtol=2e-6)
    # light behind the surface: NdotL = 0 -> exactly zero
    assert oracle.evaluate_pbr([0.5] * 3, 0.0, 0.5, [0.04] * 3, z, z, [0.0, 0.0, -1.0]).tolist() == [0.0, 0.0, 0.0]
    # the D clamp: a = 0.045^2 at the peak: pi d^2 < 1e-4 -> D = a^2 / 1e-4
    r = 0.045
    got = oracle.evaluate_pbr([0.0] * 3, 0.0, r, [1.0] * 3, z, z, z)
    np.testing.assert_allclose(got, [(r ** 4) / 1e-4 / 4.0] * 3, rtol=1e-5)
    # grazing half vector: V.H = 0 -> Fresnel = 1 regardless of F0
    v, l = [1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]
    n = [0.0, 0.0, 1.0]
    got = oracle.evaluate_pbr([0.5] * 3, 0.0, 0.5, [0.04] * 3, n, [0.6, 0.0, 0.8], [-0.6, 0.0, 0.8])
    assert np.isfinite(got).all() and (got > 0).all()


def _sky(light_dir=(0.0, 1.0, 0.0), cam_y=0.0, color=(1.0, 1.0, 1.0)):
    s = lib.SkyConstants()
    s.LightDirection[:] = light_dir
    s.LightColor[:] = color
    s.CameraPosition[:] = (0.0, cam_y, 0.0)
    return s


def test_sky_known_answers(oracle):
    # sun at zenith, looking straight up
    got = oracle.apply_atmosphere(_sky(), [0.0, 1.0, 0.0])
    ray = 3.0 / (16.0 * math.pi) * 2.0
    mie = (1 - 0.76 ** 2) / (4 * math.pi * (1 + 0.76 ** 2 - 2 * 0.76) ** 1.5)
    want = np.array([0.05, 0.12, 0.22]) + np.array([0.65, 0.57, 0.475]) * ray + mie * 0.8
    np.testing.assert_allclose(got, want, rtol=3e-6)
    np.testing.assert_allclose(got, [2.07282, 2.13327, 2.22193], rtol=2e-5)  # hand-computed
    # horizon: falloff = (1 - 0.5)^3 = 0.125; cos(sun, view) = 0
    got = oracle.apply_atmosphere(_sky(), [1.0, 0.0, 0.0])
    base = np.array([0.05, 0.12, 0.22]) + 0.125 * (np.array([0.52, 0.68, 0.86]) - np.array([0.05, 0.12, 0.22]))
    mie = (1 - 0.76 ** 2) / (4 * math.pi * (1 + 0.76 ** 2) ** 1.5)
    want = base + np.array([0.65, 0.57, 0.475]) * 3.0 / (16.0 * math.pi) + mie * 0.8
    np.testing.assert_allclose(got, want, rtol=3e-6)
    # sun on the horizon: attenuation exp(-2); camera 8000 m up: Rayleigh density 1/e, Mie exp(-8000/1200)
    got = oracle.apply_atmosphere(_sky((1.0, 0.0, 0.0), cam_y=8000.0), [0.0, 1.0, 0.0])
    mie = (1 - 0.76 ** 2) / (4 * math.pi * (1 + 0.76 ** 2) ** 1.5)
    want = np.array([0.05, 0.12, 0.22]) + (np.array([0.65, 0.57, 0.475]) * math.exp(-1) * 3 / (16 * math.pi) + mie * math.exp(-8000 / 1200) * 0.8) * math.exp(-2)
    np.testing.assert_allclose(got, want, rtol=3e-6)


def _const_faces_cube(base, mips, colors):
    out = []
    for f in range(6):
        for m in range(mips):
            n = max(1, base >> m)
            t = np.zeros((n * n, 4), np.float16)
            t[:, :3] = colors[f]
            t[:, 3] = 1
            out.append(t.view(np.uint16))
    return np.concatenate(out)


def test_cube_addressing_and_seams(oracle):
    # D3D face table: +X,-X,+Y,-Y,+Z,-Z
    for d, want in {(1, 0, 0): (0, .5, .5), (-1, 0, 0): (1, .5, .5), (0, 1, 0): (2, .5, .5), (0, -1, 0): (3, .5, .5), (0, 0, 1): (4, .5, .5),
                    (0, 0, -1): (5, .5, .5), (1, .5, .25): (0, .375, .25), (-2, 1, 1): (1, .75, .25), (.2, 1, -.4): (2, .6, .3),
                    (.2, -1, -.4): (3, .6, .7), (.5, .5, 2): (4, .625, .375), (.5, .5, -2): (5, .375, .375)}.items():
        f, u, v = oracle.select_cube_face(d)
        assert f == want[0] and abs(u - want[1]) < 1e-6 and abs(v - want[2]) < 1e-6, (d, f, u, v)
    colors = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1]], np.float32)
    cube = _const_faces_cube(8, 4, colors)
    for f, d in enumerate([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]):
        for level in (0.0, 1.5, 3.0, 7.0):
            np.testing.assert_allclose(oracle.sample_cube_level(cube, 8, 4, d, level), colors[f], atol=1e-6)
    # exactly on the +X/+Z edge the seamless filter averages the two faces (z wins the tie, the border tap folds onto +X)
    np.testing.assert_allclose(oracle.sample_cube_level(cube, 8, 4, (1, 0, 1), 0.0), (colors[0] + colors[4]) / 2, atol=1e-6)
    np.testing.assert_allclose(oracle.sample_cube_level(cube, 8, 4, (0, 1, 1), 0.0), (colors[2] + colors[4]) / 2, atol=1e-6)
    np.testing.assert_allclose(oracle.sample_cube_level(cube, 8, 4, (-1, -1, 0), 2.0), (colors[1] + colors[3]) / 2, atol=1e-6)
    # continuity across the edge: approaching from both sides gives the same value
    a = oracle.sample_cube_level(cube, 8, 4, (1, 0.1, 1 - 1e-4), 0.0)
    b = oracle.sample_cube_level(cube, 8, 4, (1 - 1e-4, 0.1, 1), 0.0)
    np.testing.assert_allclose(a, b, atol=1e-3)
    # a 1x1 top mip blends towards the neighbours away from the face centre, and trilinear blends mips
    cube2 = _const_faces_cube(2, 2, colors)
    np.testing.assert_allclose(oracle.sample_cube_level(cube2, 2, 2, (1, 0, 0), 1.0), colors[0], atol=1e-6)
    got = oracle.sample_cube_level(cube2, 2, 2, (1, 0, 0.5), 1.0)  # u = 0.25 -> 25% of the -u neighbour (+Z), 75% +X
    np.testing.assert_allclose(got, 0.75 * colors[0] + 0.25 * colors[4], atol=1e-6)


def test_bordered_cube_layout(oracle):
    rng = np.random.default_rng(1)
    base, mips = 4, 3
    n_tex = 6 * sum(max(1, base >> m) ** 2 for m in range(mips))
    cube = rng.random((n_tex, 4), dtype=np.float32).astype(np.float16).view(np.uint16)
    pad = oracle.stage_env_cube(cube, base, mips)
    assert pad.shape[0] == oracle.env_cube_texels(base, mips) == 6 * (36 + 16 + 9)
    # interior of face f mip 0 equals the source face
    E = base + 2
    face_stride = sum(max(1, base >> m) ** 2 for m in range(mips))
    for f in range(6):
        src = cube[f * face_stride: f * face_stride + base * base].reshape(base, base, 4)
        dst = pad[f * E * E:(f + 1) * E * E].reshape(E, E, 4)
        assert np.array_equal(dst[1:-1, 1:-1], src)
    # +X face (0), left border column (i = -1) comes from +Z face (4), its right-most column, same rows
    src_pz = cube[4 * face_stride: 4 * face_stride + base * base].reshape(base, base, 4)
    dst_px = pad[0:E * E].reshape(E, E, 4)
    assert np.array_equal(dst_px[1:-1, 0], src_pz[:, base - 1])
    # +X right border (i = N) comes from -Z face (5), left-most column
    src_nz = cube[5 * face_stride: 5 * face_stride + base * base].reshape(base, base, 4)
    assert np.array_equal(dst_px[1:-1, E - 1], src_nz[:, 0])
    # +X top border (j = -1) comes from +Y (2): its right-most column, traversed bottom-to-top
    src_py = cube[2 * face_stride: 2 * face_stride + base * base].reshape(base, base, 4)
    assert np.array_equal(dst_px[0, 1:-1], src_py[::-1, base - 1])


def test_shadow_compare_and_lut(oracle):
    m = np.full((2, 2), 0.5, np.float32)
    assert oracle.sample_cmp(m, 0.5, 0.5, 0.4) == 1.0 and oracle.sample_cmp(m, 0.5, 0.5, 0.6) == 0.0
    assert oracle.sample_cmp(m, 0.5, 0.5, 0.5) == 1.0  # LESS_EQUAL
    assert oracle.sample_cmp(m, 0.0, 0.0, 0.6) == 0.75  # three border taps (white = 1.0) pass, the texel fails
    assert oracle.sample_cmp(m, 0.0, 0.0, 1.5) == 0.0   # nothing passes against the border either
    m2 = np.array([[0.2, 0.8]], np.float32)              # bilinear blend of compare RESULTS, not of depths
    assert abs(oracle.sample_cmp(m2, 0.5, 0.5, 0.5) - 0.5) < 1e-7
    lut = np.array([[[0, 65535], [65535, 0]]], np.uint16)
    np.testing.assert_allclose(oracle.sample_lut(lut, 0.5, 0.5), [0.5, 0.5], atol=1e-7)
    np.testing.assert_allclose(oracle.sample_lut(lut, 0.0, 0.5), [0.0, 1.0], atol=1e-7)  # clamp addressing
    np.testing.assert_allclose(oracle.sample_lut(lut, 1.0, 0.5), [1.0, 0.0], atol=1e-7)
    assert oracle.srgb_to_linear(0) == 0.0 and oracle.srgb_to_linear(255) == 1.0
    assert abs(oracle.srgb_to_linear(128) - 0.21586050) < 1e-7 and abs(oracle.srgb_to_linear(10) - 10 / 255 / 12.92) < 1e-9


def test_lighting_pixel_analytic(oracle, urlib):
    """A 1x1 frame whose only pixel looks down the view axis: N = V = L, constant environment, constant LUT."""
    import ctypes as C
    w = h = 1
    view = np.eye(4, dtype=np.float32).ravel()
    proj = hostmath.reverse_z_projection(math.radians(90), 1.0, 0.1)
    sc = lib.SceneConstants()
    ident = np.eye(4, dtype=np.float32).ravel()
    urlib.ur_host_fill_scene_constants(lib.fptr(view), lib.fptr(proj), lib.fptr(np.zeros(3, np.float32)), 2.0, lib.fptr(np.array([0, 0, -1], np.float32)),
                                       lib.fptr(np.array([1.0, 0.5, 0.25], np.float32)), lib.fptr(ident), 0.0, 0.0, 0.0, 0.0, 3.0, C.byref(sc))
    E = np.array([0.25, 0.5, 1.0], np.float32)
    cube = _const_faces_cube(4, 3, np.tile(E, (6, 1)))
    lut = np.zeros((4, 4, 2), np.uint16); lut[..., 0] = 32768; lut[..., 1] = 16384
    la, lb = 32768 / 65535, 16384 / 65535
    A = np.array([[[0.0, 0.0, -1.0, -4.0]]], np.float16).view(np.uint16)          # normal (0,0,-1), viewZ = 4
    B = np.array([[[0.04, 0.0, 1.0, 1.0]]], np.float16).view(np.uint16)           # dielectric, roughness 1
    Cc = np.array([[188 | (188 << 8) | (188 << 16) | (255 << 24)]], np.uint32)
    hdr = np.array([[[0.5, 0.0, 0.0, 1.0]]], np.float16).view(np.uint16)          # emissive red 0.5
    out = oracle.deferred_lighting(sc, A, B, Cc, None, cube, 4, 3, lut, hdr, w, h)
    alb = oracle.srgb_to_linear(188)
    f0 = float(np.float16(0.04))
    direct = ((1 - f0) * alb + f0 / (4 * math.pi)) * 2.0 * np.array([1.0, 0.5, 0.25])
    ambient = E * alb + E * (f0 * la + lb)
    want = np.array([0.5, 0, 0]) + direct + ambient
    got = out.view(np.float16).astype(np.float32)[0, 0]
    np.testing.assert_allclose(got[:3], want, rtol=1.2e-3)   # one fp16 rounding
    assert got[3] == 2.0                                     # ONE/ONE blend of alpha: 1 + 1


def test_tonemap_known_answers(oracle):
    def px(r, g, b):
        return np.array([[r, g, b, 1.0]], np.float16).view(np.uint16)
    unpack = lambda v: (int(v) & 255, (int(v) >> 8) & 255, (int(v) >> 16) & 255, int(v) >> 24)
    assert unpack(oracle.tonemap(px(0, 0, 0))[0]) == (0, 0, 0, 255)
    # gamma only: 0.5^(1/2.2) = 0.72974 -> 186
    assert unpack(oracle.tonemap(px(0.5, 0.5, 0.5), enable_tonemap=False)[0]) == (186, 186, 186, 255)
    # PBR neutral on white: offset .04 -> peak .96 -> newPeak = 1 - .0576/.44 = .869091 -> ^(1/2.2) = .93822 -> 239
    assert unpack(oracle.tonemap(px(1, 1, 1))[0]) == (239, 239, 239, 255)
    # below the compression start the curve only subtracts the toe offset: x = .05 -> offset = .05 - 6.25*.0025 = .034375
    v = float(np.float16(0.05))
    off = v - 6.25 * v * v
    want = int(((v - off) ** (1 / 2.2)) * 255 + 0.5)
    assert unpack(oracle.tonemap(px(v, v, v))[0])[0] == want
    # exposure and auto exposure multiply: 0.25 * 2 * 2^1 = 1.0 -> same as white
    assert unpack(oracle.tonemap(px(0.25, 0.25, 0.25), exposure=2.0, exposure_ev=1.0)[0]) == (239, 239, 239, 255)
    # saturated channel desaturates towards the peak, never exceeds 255
    r, g, b, a = unpack(oracle.tonemap(px(8.0, 0.1, 0.1))[0])
    assert r > g == b and r <= 255 and g > 0


def test_temporal_aa_known_answers(oracle):
    h16 = lambda a: np.asarray(a, np.float16).view(np.uint16)
    cur = np.zeros((3, 3, 4), np.float16)
    cur[..., :3] = 0.5
    cur[1, 1, :3] = [1.0, 0.25, 0.5]
    cur[..., 3] = 2.0
    hist = np.zeros((3, 3, 4), np.float16)
    hist[..., :3] = [4.0, 0.0, 0.5]   # red above the box, green below it, blue inside
    out = oracle.temporal_aa(h16(cur), h16(hist), 0.9, True).view(np.float16).astype(np.float32)
    # centre pixel: box r [0.5,1], g [0.25,0.5], b [0.5,0.5]; history clamps to (1, 0.25, 0.5) = current -> unchanged
    assert out[1, 1].tolist() == [1.0, 0.25, 0.5, 2.0]
    # corner pixel (0,0): neighbourhood (clamped at the frame edge) holds 0.5 everywhere plus the centre pixel
    # -> box r [0.5,1], g [0.25,0.5], b 0.5; history -> (1, 0.25, 0.5); blend 0.5 + 0.9*(h-0.5)
    want = np.array([0.5 + 0.9 * 0.5, 0.5 + 0.9 * (0.25 - 0.5), 0.5], np.float32).astype(np.float16).astype(np.float32)
    assert np.array_equal(out[0, 0, :3], want) and out[0, 0, 3] == 2.0
    # no history: pass-through; weight saturates
    assert np.array_equal(oracle.temporal_aa(h16(cur), h16(hist), 0.9, False), h16(cur))
    assert np.array_equal(oracle.temporal_aa(h16(cur), h16(hist), 7.0, True), oracle.temporal_aa(h16(cur), h16(hist), 1.0, True))
    assert np.array_equal(oracle.temporal_aa(h16(cur), h16(hist), -1.0, True)[..., :3], h16(cur)[..., :3])
