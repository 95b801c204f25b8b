"""swarmmap_amd — MI355X (gfx950) implementation of SwarmMap's ORB front-end / matcher / BA hot path.

The product is libswarmorb.so (hand-written HIP kernels behind the C ABI of include/swarmorb.h) plus
C++ adapter classes with the reference's signatures (swarmmap_amd/host/).  This Python package is a thin
ctypes mirror of the same interface used by tests and bench.py.  There is NO CPU fallback: importing the
binding without the built library, or creating a handle without a GPU, raises.
"""
from ._lib import SwarmOrbError, build_library, device_count, library_path, load_library  # noqa: F401
from .extractor import ExtractorGroup, ORBextractor  # noqa: F401
from .dframe import DeviceFrame, DeviceMap  # noqa: F401
from .frame import FramePostProcessor  # noqa: F401
from .matcher import FrameView, ORBmatcher  # noqa: F401
from .optimizer import Optimizer  # noqa: F401

__all__ = ["ORBextractor", "ExtractorGroup", "ORBmatcher", "FrameView", "FramePostProcessor", "DeviceFrame", "DeviceMap", "Optimizer", "SwarmOrbError", "build_library", "device_count", "library_path", "load_library"]
