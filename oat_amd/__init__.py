"""oat_amd -- MI355X (gfx950) implementation of Oat's per-frame image hot path.

    framefilt mog -> framefilt col (BGR2HSV) -> posidet hsv | thresh

The product is ``oat_amd/lib/liboatgpu.so`` (hand-written HIP kernels behind the
C ABI of ``include/oatgpu.h``).  This package is the thin Python binding used
by the tests and the benchmark; there is no CPU fallback -- importing
``oat_amd.ffi`` raises if the HIP library has not been built.
"""
from .ffi import OatGpuError, lib_path  # noqa: F401
from .components import (  # noqa: F401
    BackgroundSubtractorMOG, BackgroundSubtractor, Threshold, ColorConvert, HSVDetector, SimpleThreshold, DifferenceDetector, HotPath, Position2D,
)

__all__ = ["BackgroundSubtractorMOG", "BackgroundSubtractor", "Threshold", "ColorConvert", "HSVDetector", "SimpleThreshold", "DifferenceDetector", "HotPath",
           "Position2D", "OatGpuError", "lib_path"]
