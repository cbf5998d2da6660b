"""st3d -- Python host side of the MI355X-native 2D->3D style-transfer step.

PyTorch-ROCm owns device memory and streams; all device arithmetic is in libst3d.so
(hand-written HIP for gfx950, C ABI in include/st3d.h).  There is no CPU fallback."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
