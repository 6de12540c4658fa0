"""Scene loading for the cull pass: Assets/Scenes/<x>.json + the models' .gltf JSON -> ModelBounds in command order.
The extraction itself is C++ (csrc/scene.cpp); this module reads the files and marshals."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import lib as _lib


@dataclass
class SceneBounds:
    bounds: np.ndarray        # (n, 2, 4) float32: (min.xyz, 0), (max.xyz, 0) — the ModelBounds buffer
    pipeline_keys: np.ndarray
    materials: np.ndarray
    centers: np.ndarray
    radii: np.ndarray
    scene_center: np.ndarray
    scene_radius: float

    @property
    def count(self) -> int:
        return self.bounds.shape[0]


def load_scene_bounds(scene_json_path, assets_root=None) -> SceneBounds:
    scene_json_path = Path(scene_json_path)
    assets_root = Path(assets_root) if assets_root is not None else scene_json_path.parent.parent  # RendererUtils.cpp:326-329
    L = _lib.load()
    text = scene_json_path.read_bytes()
    n = L.ur_scene_model_count(text)
    if n <= 0:
        raise ValueError(f"{scene_json_path}: no models")
    gltfs = []
    buf = C.create_string_buffer(1024)
    for i in range(n):
        if L.ur_scene_model_path(text, i, buf, 1024) < 0:
            raise ValueError(f"model {i} has no path")
        gltfs.append((assets_root / buf.value.decode()).read_bytes())
    arr = (C.c_char_p * n)(*gltfs)
    summary = _lib.SceneSummary()
    rc = L.ur_scene_extract(text, arr, n, None, 0, C.byref(summary))
    if rc != 0:
        raise ValueError(f"ur_scene_extract failed ({rc})")
    models = (_lib.SceneModel * summary.model_count)()
    rc = L.ur_scene_extract(text, arr, n, models, summary.model_count, C.byref(summary))
    if rc != 0:
        raise ValueError(f"ur_scene_extract failed ({rc})")
    m = summary.model_count
    bounds = np.zeros((m, 2, 4), np.float32)
    for i, x in enumerate(models):
        bounds[i, 0, :3] = x.bounds_min
        bounds[i, 1, :3] = x.bounds_max
    return SceneBounds(bounds, np.array([x.pipeline_key for x in models], np.uint32), np.array([x.material_index for x in models], np.uint32),
                       np.array([list(x.center) for x in models], np.float32), np.array([x.radius for x in models], np.float32),
                       np.array(list(summary.scene_center), np.float32), float(summary.scene_radius))
