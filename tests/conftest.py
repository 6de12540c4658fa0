import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_spawner_proc = None


def pytest_sessionstart(session):
    """A session that may run GPU tests starts its process-spawning helper NOW, before this process has made a HIP call
    (tests/_spawner.py): the rank processes of tests/test_gpu_multirank.py are then children of a process that never
    touched the GPU. Sessions deselecting the GPU tests (-m "not gpu") start nothing."""
    global _spawner_proc
    expr = session.config.getoption("markexpr", "") or ""
    if "not gpu" in expr:
        return
    import subprocess
    _spawner_proc = subprocess.Popen([sys.executable, str(ROOT / "tests" / "_spawner.py")], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     text=True, cwd=str(ROOT))


def pytest_sessionfinish(session, exitstatus):
    global _spawner_proc
    if _spawner_proc is not None:
        try:
            _spawner_proc.stdin.close()
            _spawner_proc.wait(timeout=30)
        except Exception:
            _spawner_proc.kill()  # the one process started above
        _spawner_proc = None


@pytest.fixture(scope="session")
def spawn_ranks():
    """spawn_ranks(argv, envs, timeout) -> {"rc": [...], "tail": [...]}: runs len(envs) copies of argv as children of the
    spawner helper and waits for them."""
    import json
    if _spawner_proc is None or _spawner_proc.poll() is not None:
        pytest.skip("the process-spawning helper is not running (session started with GPU tests deselected)")

    def run(argv, envs, timeout=600):
        _spawner_proc.stdin.write(json.dumps({"argv": argv, "envs": envs, "timeout": timeout, "cwd": str(ROOT)}) + "\n")
        _spawner_proc.stdin.flush()
        return json.loads(_spawner_proc.stdout.readline())
    return run


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    o.load()
    return o


@pytest.fixture(scope="session")
def urlib():
    """The product library; built on demand here, prebuilt on the GPU box."""
    from unclerenderer_amd import build, lib
    if not lib.library_path().exists():
        build.build()
    return lib.load()


@pytest.fixture(scope="session")
def hotpath(urlib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unclerenderer_amd.hotpath import HotPath
    hp = HotPath(0)
    yield hp
    hp.close()
