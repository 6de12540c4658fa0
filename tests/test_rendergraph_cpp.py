"""Runs the C++ render-graph semantics tests (tests/cpp/test_rendergraph.cpp) — CPU only."""
import runpy
import subprocess
from pathlib import Path


def test_rendergraph_cpp(urlib):
    mod = runpy.run_path(str(Path(__file__).parent / "cpp" / "build.py"))
    exe = mod["build"]()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK rendergraph tests passed" in r.stdout, r.stdout + r.stderr
