"""libh264_amd -- MI355X-native H.264 bitstream parse/serialise engine behind
the libh264 C API.

The product is the C-ABI in include/ (libh264_amd/lib/*.so); this package is
the thin Python host mirror used by the tests and by bench.py: ctypes bindings
(`capi`), a convenience `Engine` wrapper and the synthetic stream generator
(`gen`).  Nothing here falls back to the CPU: without the HIP library and a
gfx950 device the engine calls raise.
"""
from . import ctypes_abi  # noqa: F401
from .capi import (Engine, Params, EngineError, engine_lib, available,  # noqa: F401
                   build_id)
from . import gen  # noqa: F401

__all__ = ["Engine", "Params", "EngineError", "engine_lib", "available", "gen",
           "ctypes_abi"]
