#!/usr/bin/env python3
"""Diagnostic builds of libur_hotpath.so that differ from the product only in lighting.hip's compile-time switches
(UR_ABLATE, UR_HDR_STORE, ... — see the kernel source). Outputs go to unclerenderer_amd/csrc/_build/variants/ (git-ignored,
shipped to the GPU box by gpurun); select one with UR_HOTPATH_LIB=<path>.

    python tools/build_variants.py name=-DUR_ABLATE=1 name2="-DUR_HDR_STORE=1 -DUR_FOO=2" ...
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from unclerenderer_amd import build as b  # noqa: E402


def one(spec: str) -> Path:
    name, _, defs = spec.partition("=")
    out = b.OUT / "variants"
    out.mkdir(parents=True, exist_ok=True)
    flags = dict(b.SOURCES)["lighting.hip"]
    obj = out / f"lighting_{name}.o"
    lib = out / f"libur_{name}.so"
    cmd = [b.hipcc()] + b.COMMON + flags + defs.split() + ["-c", str(b.CSRC / "lighting.hip"), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{name}: {r.stderr}")
    objs = [str(b.OUT / (s.replace("/", "_") + ".o")) for s, _ in b.SOURCES if s != "lighting.hip"] + [str(obj)]
    r = subprocess.run([b.hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", str(lib)] + objs + ["-ldl", "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{name}: link: {r.stderr}")
    obj.unlink()
    return lib


def main():
    b.build()
    with ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)


if __name__ == "__main__":
    main()
