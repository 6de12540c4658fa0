"""Build the in-tree native libraries with hipcc for gfx950 (no cmake, no JIT cache).

    python -m unclerenderer_amd.build [--force]

Outputs (git-ignored, shipped to the GPU box by gpurun):
    unclerenderer_amd/csrc/_build/libur_hotpath.so   HIP kernels + C-ABI (include/ur_hotpath.h, include/ur_host.h)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "unclerenderer_amd" / "csrc"
OUT = CSRC / "_build"
LIB = OUT / "libur_hotpath.so"

ARCH = "gfx950"
COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
          f"-I{ROOT / 'include'}"]
# exact-arithmetic translation units: no FMA contraction, IEEE divide (hipcc's default)
EXACT = ["-ffp-contract=off"]

# (source relative to csrc, extra flags)
SOURCES = [
    ("ur_api.hip", []),
    ("hzb.hip", EXACT),
    ("cull.hip", EXACT),
    # packed fp32 VALU ops are not faster on gfx950 and cost v_mov traffic; the atomic optimizer would turn the one-lane LDS
    # work-counter claim into a scan + broadcast with an immediate wait
    ("lighting.hip", ["-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]),
    ("tonemap.hip", EXACT),
    ("taa.hip", EXACT),
    ("stream_ceiling.hip", []),
    ("scene.cpp", ["-x", "hip"] + EXACT),
    ("dds.cpp", ["-x", "hip"] + EXACT),
    ("host_math.cpp", ["-x", "hip"] + EXACT),
    ("rg/RenderGraph.cpp", ["-x", "hip"]),
    ("frame/HotPathRenderer.cpp", ["-x", "hip"]),
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _deps() -> list[Path]:
    hdrs = list((ROOT / "include").glob("*.h")) + list(CSRC.rglob("*.h"))
    return hdrs


def _stale(target: Path, srcs: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(s.stat().st_mtime > t for s in srcs if s.exists())


def _compile(job) -> Path:
    src, flags, force = job
    srcp = CSRC / src
    obj = OUT / (src.replace("/", "_") + ".o")
    if force or _stale(obj, [srcp] + _deps()):
        cmd = [hipcc()] + COMMON + flags + ["-c", str(srcp), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    OUT.mkdir(parents=True, exist_ok=True)
    present = [(s, f, force) for s, f in SOURCES if (CSRC / s).exists()]
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, present))
    if force or _stale(LIB, objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs] + ["-ldl", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
