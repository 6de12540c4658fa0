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
