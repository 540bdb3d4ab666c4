"""ebfi_amd -- MI355X-native host side of the EBFI-BE frame-synthesis hot path.

The compute lives in libebfi_hip.so (hand-written gfx950 HIP, C ABI in include/ebfi_hip.h);
this package mirrors the reference's Python interfaces on top of it.  Nothing in here falls back
to CPU or to a PyTorch re-implementation: ops raise if the native library is missing.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
