"""Scene -> ModelBounds extraction (next row): the C++ extractor (csrc/scene.cpp) against an independent
numpy restatement of the reference's loader arithmetic (GltfLoader.cpp:407-593,823; RendererUtils.cpp:46-82,277-295,402-540),
on the shipped scene files (tests/golden/assets)."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from unclerenderer_amd import hostmath, scene

ASSETS = Path(__file__).parent / "golden" / "assets"


def _quat(x, y, z, w):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 0], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 0],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y), 0], [0, 0, 0, 1]], np.float64)


def _local(node):  # column-vector convention
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T
    T = np.eye(4); T[:3, 3] = node.get("translation", [0, 0, 0])
    S = np.diag(list(node.get("scale", [1, 1, 1])) + [1.0])
    return T @ _quat(*node.get("rotation", [0, 0, 0, 1])) @ S


def reference_bounds(scene_path):
    sc = json.loads(scene_path.read_text())
    Z = np.diag([1.0, 1.0, -1.0, 1.0])
    out, spheres = [], []
    for model in sc["models"]:
        g = json.loads((ASSETS / model["path"]).read_text())
        meshes = []
        for m in g["meshes"]:
            lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
            for p in m["primitives"]:
                a = g["accessors"][p["attributes"]["POSITION"]]
                mn, mx = np.array(a["min"], np.float64), np.array(a["max"], np.float64)
                mn[2], mx[2] = -a["max"][2], -a["min"][2]  # vertex z flip
                lo, hi = np.minimum(lo, mn), np.maximum(hi, mx)
            meshes.append((lo, hi, [p.get("material", -1) for p in m["primitives"]]))
        nodes = []

        def walk(i, parent):
            n = g["nodes"][i]
            world = parent @ (Z @ _local(n) @ Z)
            if "mesh" in n:
                nodes.append((n["mesh"], world))
            for c in n.get("children", []):
                walk(c, world)
        for r in g["scenes"][g.get("scene", 0)]["nodes"]:
            walk(r, np.eye(4))
        s = np.array(model.get("scale", [1, 1, 1]), np.float64)
        p, y, r = (math.radians(v) for v in model.get("rotate_euler", [0, 0, 0]))
        Rx = np.array([[1, 0, 0], [0, math.cos(p), math.sin(p)], [0, -math.sin(p), math.cos(p)]])
        Ry = np.array([[math.cos(y), 0, -math.sin(y)], [0, 1, 0], [math.sin(y), 0, math.cos(y)]])
        Rz = np.array([[math.cos(r), math.sin(r), 0], [-math.sin(r), math.cos(r), 0], [0, 0, 1]])
        R3 = Rz @ Rx @ Ry  # row-vector RotationRollPitchYaw: roll, then pitch, then yaw
        t = np.array(model.get("translate", [0, 0, 0]), np.float64)
        for order, (mi, world) in enumerate(nodes):
            lo, hi, mats = meshes[mi]
            A = world[:3, :3].T  # row-vector form of the node matrix
            lin = A @ np.diag(s) @ R3
            off = world[:3, 3] @ np.diag(s) @ R3 + t
            corners = np.array([[(hi if c & 1 else lo)[0], (hi if c & 2 else lo)[1], (hi if c & 4 else lo)[2]] for c in range(8)])
            wc = corners @ lin + off
            centre = (0.5 * (lo + hi)) @ lin + off
            radius = max(np.linalg.norm(hi - lo) * 0.5, 1.0) * np.abs(s).max() * np.linalg.norm(A, axis=0).max()
            for prim, mat in enumerate(mats):
                m = g.get("materials", [{}])[mat] if mat >= 0 else {}
                pbr = m.get("pbrMetallicRoughness", {})
                key = int("normalTexture" in m) | int("metallicRoughnessTexture" in pbr) << 1 | int("baseColorTexture" in pbr) << 2 | \
                    int("emissiveTexture" in m) << 3 | int(m.get("alphaMode") == "MASK") << 4
                out.append((key, mat, len(out), wc.min(0), wc.max(0)))
                spheres.append((centre, radius))
    out.sort(key=lambda e: (e[0], e[1] & 0xFFFFFFFF, e[2]))
    lo = np.min([c - r for c, r in spheres], 0)
    hi = np.max([c + r for c, r in spheres], 0)
    return out, 0.5 * (lo + hi), max(np.linalg.norm(hi - lo) * 0.5, 1.0)


@pytest.mark.parametrize("name,count", [("sponza", 25), ("Duck", 1), ("pica_pica", 170)])
def test_extraction_matches_restatement(urlib, name, count):
    b = scene.load_scene_bounds(ASSETS / "Scenes" / f"{name}.json")
    ref, centre, radius = reference_bounds(ASSETS / "Scenes" / f"{name}.json")
    assert b.count == count == len(ref)  # 25 / 1 / 170 draw commands
    assert b.pipeline_keys.tolist() == [e[0] for e in ref]
    tol = 2e-5 * max(1.0, radius)
    np.testing.assert_allclose(b.bounds[:, 0, :3], np.array([e[3] for e in ref]), atol=tol)
    np.testing.assert_allclose(b.bounds[:, 1, :3], np.array([e[4] for e in ref]), atol=tol)
    np.testing.assert_allclose(b.scene_center, centre, atol=tol)
    assert abs(b.scene_radius - radius) < tol
    assert (b.bounds[:, :, 3] == 0).all() and (np.diff(b.pipeline_keys.astype(np.int64)) >= 0).all()


def test_sponza_commands_share_one_box_and_presets_agree(urlib):
    b = scene.load_scene_bounds(ASSETS / "Scenes" / "sponza.json")
    assert (b.bounds == b.bounds[0]).all()  # bounds are per MESH; the 25 primitives share them
    for name, file in (("sponza", "sponza"), ("duck", "Duck"), ("pica_pica", "pica_pica")):
        sb = scene.load_scene_bounds(ASSETS / "Scenes" / f"{file}.json")
        p = hostmath.SCENES[name]
        np.testing.assert_allclose(p.scene_center, sb.scene_center, atol=1e-4)
        assert abs(p.scene_radius - sb.scene_radius) < 1e-3 and p.instance_count == sb.count
        np.testing.assert_allclose(p.model_aabb[0], sb.bounds[:, 0, :3].min(0), atol=1e-4)
        np.testing.assert_allclose(p.model_aabb[1], sb.bounds[:, 1, :3].max(0), atol=1e-4)


def test_scene_errors(urlib):
    import ctypes as C
    from unclerenderer_amd import lib
    assert urlib.ur_scene_model_count(b"{ not json") == -1
    assert urlib.ur_scene_model_count(b'{"models": []}') == 0
    s = lib.SceneSummary()
    arr = (C.c_char_p * 1)(b'{"meshes":[{"primitives":[{"attributes":{"POSITION":0}}]}],"accessors":[{}]}')
    assert urlib.ur_scene_extract(b'{"models":[{"path":"x"}]}', arr, 1, None, 0, C.byref(s)) == -4  # POSITION without min/max
    assert urlib.ur_scene_extract(b'{"models":[{"path":"x"}]}', arr, 2, None, 0, C.byref(s)) == -1
