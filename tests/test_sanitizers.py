"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-buildable code (SURVEY.md §5 "Race detection / sanitizers"; the
reference has only the D3D12 debug layer, Source/RHI/DX12Device.cpp:82-91). CPU box only: GPU ASan is not available.

  * the render graph (csrc/rg/RenderGraph.cpp) compiled as host C++ together with its semantics test;
  * the host-side product code (DDS/BC6H decode, scene extraction, host constant math) and the oracle, driven by
    tests/cpp/sanitize_main.cpp over the shipped fixtures, random inputs and hostile inputs in exact-size heap buffers.
Any sanitizer report aborts the binary (-fno-sanitize-recover) and fails the test."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "cpp" / "_build"
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-ffp-contract=off"]
ENV = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1", "PATH": "/usr/bin:/bin"}


def _build(exe: Path, sources: list[Path], extra: list[str]) -> Path:
    OUT.mkdir(exist_ok=True)
    deps = sources + list((ROOT / "include").glob("*.h")) + list((ROOT / "unclerenderer_amd" / "csrc" / "rg").glob("*.h"))
    if exe.exists() and exe.stat().st_mtime > max(d.stat().st_mtime for d in deps):
        return exe
    cmd = ["g++"] + SAN + [f"-I{ROOT / 'include'}"] + [str(s) for s in sources] + ["-o", str(exe), "-pthread"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stderr}"
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_and_oracle_under_asan_ubsan():
    csrc = ROOT / "unclerenderer_amd" / "csrc"
    exe = _build(OUT / "sanitize_main", [ROOT / "tests" / "cpp" / "sanitize_main.cpp", csrc / "dds.cpp", csrc / "scene.cpp", csrc / "host_math.cpp",
                                         ROOT / "oracle" / "ur_oracle.cpp"], [])
    r = subprocess.run([str(exe), str(ROOT / "tests" / "golden" / "assets")], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "OK sanitized host + oracle run clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists(), reason="needs g++ and the HIP headers")
def test_rendergraph_under_asan_ubsan():
    """RenderGraph.cpp is host C++ over the HIP runtime API: built here with g++ against libamdhip64 (no device needed: the
    semantics test never touches one) so that its own code is instrumented, which the in-tree .so's is not."""
    exe = _build(OUT / "test_rendergraph_asan", [ROOT / "tests" / "cpp" / "test_rendergraph.cpp", ROOT / "unclerenderer_amd" / "csrc" / "rg" / "RenderGraph.cpp"],
                 ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "OK rendergraph tests passed" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
