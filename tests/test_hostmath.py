"""The C++ host math (csrc/host_math.cpp) against an independent numpy restatement of the DirectXMath definitions the
reference calls (Camera.cpp:23-47, RendererUtils.cpp:1117-1218, Renderer.cpp:411-429)."""
import math

import numpy as np

from unclerenderer_amd import hostmath


def np_look_to_lh(eye, d, up):
    eye, d, up = (np.asarray(v, np.float64) for v in (eye, d, up))
    r2 = d / np.linalg.norm(d)
    r0 = np.cross(up, r2); r0 /= np.linalg.norm(r0)
    r1 = np.cross(r2, r0)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2] = r0, r1, r2
    m[3, :3] = [-(r0 @ eye), -(r1 @ eye), -(r2 @ eye)]
    return m


def np_planes(vp):
    c = [vp[:, i] for i in range(4)]
    raw = [c[3] + c[0], c[3] - c[0], c[3] + c[1], c[3] - c[1], c[2], c[3] - c[2]]
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.stack([p / np.linalg.norm(p[:3]) for p in raw])


def test_view_projection_planes(urlib):
    p = hostmath.SCENES["sponza"]
    fc = hostmath.build_frame_constants(p, 1920, 1080)
    fwd = hostmath.camera_forward_from_euler_degrees(-12.6, 261.8)
    pr, yr = math.radians(-12.6), math.radians(261.8)
    np.testing.assert_allclose(fwd, [math.cos(pr) * math.sin(yr), -math.sin(pr), math.cos(pr) * math.cos(yr)], atol=1e-6)
    V = np_look_to_lh(p.camera_position, fwd, (0, 1, 0))
    np.testing.assert_allclose(fc.view.reshape(4, 4), V, atol=2e-6)
    ys = 1 / math.tan(math.radians(60) / 2)
    P = np.array([[ys / (1920 / 1080), 0, 0, 0], [0, ys, 0, 0], [0, 0, 0, 1], [0, 0, 0.1, 0]])
    np.testing.assert_allclose(fc.proj.reshape(4, 4), P, atol=1e-6)
    cc = hostmath.pack_culling_constants(fc.view, fc.proj, 25, True, 10, 960, 540, False)
    planes = cc[:24].view(np.float32).reshape(6, 4)
    ref = np_planes(V @ P)
    # plane 4 = column 3 alone = (0,0,0,Near): zero normal -> (NaN,NaN,NaN,+inf); it can never reject
    assert np.isnan(planes[4, :3]).all() and np.isposinf(planes[4, 3])
    keep = [0, 1, 2, 3, 5]
    np.testing.assert_allclose(planes[keep], ref[keep], atol=3e-6)
    np.testing.assert_allclose(cc[24:40].view(np.float32).reshape(4, 4), V @ P, atol=3e-6)
    assert list(cc[40:46]) == [25, 1, 10, 960, 540, 0]
    # plane 5 is the real near clip z_view >= Near
    in_front = np.array([*(np.asarray(p.camera_position) + fwd * 0.2)], np.float32)
    behind = np.array([*(np.asarray(p.camera_position) - fwd * 0.2)], np.float32)
    assert hostmath.is_aabb_in_frustum(planes.ravel(), in_front - 0.01, in_front + 0.01)
    assert not hostmath.is_aabb_in_frustum(planes.ravel(), behind - 0.01, behind + 0.01)


def test_inverse_and_light_matrices(urlib):
    fc = hostmath.build_frame_constants("duck", 512, 512)
    V = fc.view.reshape(4, 4).astype(np.float64)
    VI = np.ctypeslib.as_array(fc.scene.ViewInverse).reshape(4, 4)
    np.testing.assert_allclose(V @ VI, np.eye(4), atol=2e-6)
    np.testing.assert_allclose(VI[3, :3], fc.camera_position, atol=2e-6)  # inverse view carries the eye position
    # light: JSON direction (-.5,-1,-.3) normalised, y negated by the app's pitch/yaw round trip
    d = np.array([-0.5, -1.0, -0.3]); d /= np.linalg.norm(d)
    np.testing.assert_allclose(fc.light_direction, [d[0], -d[1], d[2]], atol=1e-6)
    # BuildDirectionalLightViewProjection: eye = centre + dir * 2.5R, ortho 2R x 2R, z in [0.1, 5R]
    R, c = fc.scene_radius, fc.scene_center.astype(np.float64)
    L = fc.light_direction.astype(np.float64)
    lv = np_look_to_lh(c + L * 2.5 * R, -L, (0, 1, 0))
    ortho = np.array([[1 / R, 0, 0, 0], [0, 1 / R, 0, 0], [0, 0, 1 / (5 * R - 0.1), 0], [0, 0, -0.1 / (5 * R - 0.1), 1]])
    np.testing.assert_allclose(np.ctypeslib.as_array(fc.scene.LightViewProjection).reshape(4, 4), lv @ ortho, atol=5e-6)
    # the scene centre projects to the middle of the shadow map at depth (2.5R - 0.1)/(5R - 0.1)
    q = np.append(c, 1.0) @ (lv @ ortho)
    np.testing.assert_allclose(q[:3] / q[3], [0, 0, (2.5 * R - 0.1) / (5 * R - 0.1)], atol=1e-5)
    assert fc.sky.World[0] == np.float32(max(5 * R, 100.0)) and fc.sky.World[12] == fc.camera_position[0]
    assert fc.scene.ShadowMapSize[0] == 2048 and fc.scene.EnvMapMipCount == 9 and fc.scene.LightIntensity == 3.0


def test_sponza_light_points_up(urlib):
    fc = hostmath.build_frame_constants("sponza", 3840, 2160)
    np.testing.assert_allclose(fc.light_direction, [0.0, 0.9659258, 0.2588190], atol=1e-6)
