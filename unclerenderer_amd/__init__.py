"""UncleRenderer visibility + deferred-shading hot path, MI355X-native (gfx950).

The product is csrc/_build/libur_hotpath.so (hand-written HIP kernels behind the C-ABI of include/ur_hotpath.h, the C++
render graph in csrc/rg and the pass wiring in csrc/frame). The Python in this package only marshals tensors and
constants into that library; there is no CPU fallback.
"""
from . import lib  # noqa: F401

__all__ = ["lib", "hostmath", "synth", "hotpath", "build"]
__version__ = "0.1.0"
