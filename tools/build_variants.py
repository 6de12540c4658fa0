#!/usr/bin/env python3
"""Diagnostic builds of libur_hotpath.so that differ from the product in the lighting kernel only. Outputs go to
unclerenderer_amd/csrc/_build/variants/ (git-ignored, shipped to the GPU box by gpurun); select one with UR_HOTPATH_LIB=<path>.

    python tools/build_variants.py name=-DFLAG=1 name2="-DA=1 -DB=2" ...          # the product source + defines
    python tools/build_variants.py --base r02 loader=-DUR_LOADER_WAVE=1 nohbm=-DUR_ABLATE=64 stamps=-DUR_STAMPS ...

--base r02 compiles the round-2 lighting kernel source, the one that carries every measured-and-rejected structure behind
compile-time switches (producer/consumer wave specialisation UR_LOADER_WAVE, ablations UR_ABLATE, in-kernel stamps UR_STAMPS,
HDR store flavours UR_HDR_STORE, DMA cache policy UR_DMA_NT, release-fence hand-off UR_RIDE_RELEASE_FENCE: DESIGN.md section 3.3,
profiles/r02_ablation.txt). It is taken from git history (`git show R02_COMMIT:unclerenderer_amd/csrc/lighting.hip`), not kept in the
tree: the product translation unit has ONE loop. Its entry points are the round-2 ones (no ur_time_next_lighting, no time-out
flag): use such a library with tools/bench_kernels.py / tools/stamps_lighting.py, not with the test suite.
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from unclerenderer_amd import build as b  # noqa: E402

R02_COMMIT = "8034e2c"  # "round 2: VERDICT + ADVICE + BENCH": the last commit whose lighting.hip holds the variant switches
BASE = None


def source_for(base: str | None) -> Path:
    if base is None:
        return b.CSRC / "lighting.hip"
    if base != "r02":
        raise SystemExit(f"unknown base {base!r} (r02)")
    out = b.CSRC / "_build" / "variants"
    out.mkdir(parents=True, exist_ok=True)
    dst = b.CSRC / "_lighting_r02_variants.hip"  # beside the headers it includes; git-ignored by name below
    text = subprocess.run(["git", "-C", str(ROOT), "show", f"{R02_COMMIT}:unclerenderer_amd/csrc/lighting.hip"], capture_output=True, text=True, check=True).stdout
    # the two context fields that changed since (the riding tail's time-out flag moved to host-visible memory)
    text = text.replace("ride.done = ctx->hzb_done;", "ride.done = ctx->hzb_done; /* [1] = the round-2 flag word, unread now */")
    dst.write_text(text)
    return dst


def one(spec: str) -> Path:
    name, _, defs = spec.partition("=")
    out = b.OUT / "variants"
    out.mkdir(parents=True, exist_ok=True)
    flags = dict(b.SOURCES)["lighting.hip"]
    obj = out / f"lighting_{name}.o"
    lib = out / f"libur_{name}.so"
    cmd = [b.hipcc()] + b.COMMON + flags + defs.split() + ["-c", str(source_for(BASE)), "-o", str(obj)]
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
    global BASE
    argv = sys.argv[1:]
    if argv[:1] == ["--base"]:
        BASE = argv[1]
        argv = argv[2:]
    b.build()
    source_for(BASE)
    with ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, argv):
            print(lib)
    tmp = b.CSRC / "_lighting_r02_variants.hip"
    if tmp.exists():
        tmp.unlink()


if __name__ == "__main__":
    main()
