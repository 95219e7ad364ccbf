"""mythtracer_amd — MI355X-native MythTracer hot path.

The product is native: `lib/libmythtracer_hip.so` (HIP kernels behind the C ABI
of include/mythtracer_hip.h) and `lib/libmythtracer_host.so` (the C++ facade
with the reference's class names).  This Python package is only a ctypes
binding of both for tests, bench.py and __graft_entry__.py; it contains no
rendering code and no CPU fallback: without the libraries or without a GPU the
calls fail.
"""
from .binding import (HipAbi, MythTracer, NativeLibraryMissing, hip_abi,  # noqa: F401
                      host_lib, HIP_SYMBOLS)

__all__ = ["HipAbi", "MythTracer", "NativeLibraryMissing", "hip_abi", "host_lib",
           "HIP_SYMBOLS"]
