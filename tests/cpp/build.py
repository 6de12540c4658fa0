"""Build the C++ render-graph test binary against the in-tree libur_hotpath.so (g++, host only)."""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
LIBDIR = ROOT / "unclerenderer_amd" / "csrc" / "_build"
OUT = HERE / "_build"


def build() -> Path:
    OUT.mkdir(exist_ok=True)
    exe = OUT / "test_rendergraph"
    src = HERE / "test_rendergraph.cpp"
    lib = LIBDIR / "libur_hotpath.so"
    if exe.exists() and exe.stat().st_mtime > max(src.stat().st_mtime, lib.stat().st_mtime):
        return exe
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", str(src), "-o", str(exe), f"-L{LIBDIR}", "-lur_hotpath",
           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"render-graph test build failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return exe


if __name__ == "__main__":
    print(build())
