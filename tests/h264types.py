"""Test-side alias of the ABI mirrors that live in the package."""
from libh264_amd.ctypes_abi import *  # noqa: F401,F403
from libh264_amd.ctypes_abi import REFERENCE_SIZEOF, EV_NAMES, t2_stride  # noqa: F401
