"""MI355X-native path-tracing core behind the CUDATracer::PathTracer API of 0x5b25/CUDAPathTracer.

The product is `libmipt.so` (C-ABI in include/mipt.h, hand-written HIP for gfx950) plus the header-only C++ classes
in cudapathtracer_amd/cpp/.  This Python package is the ctypes plumbing the tests and bench.py use.
"""
from .api import *  # noqa: F401,F403
from .api import lib, LIB_PATH  # noqa: F401
