"""jpeg_amd -- MI355X-native (gfx950) spectral pipeline of tayloraswift/jpeg.

Host-side mirror of the reference's types for the hot path (Spectral / Planar /
Rectangular, JPEG.Layout, the YCbCr / RGB colour targets) over the C ABI in
include/jpeg_amd.h.  All arithmetic runs in hand-written HIP kernels
(jpeg_amd/csrc); there is no CPU fallback -- a missing library raises.
"""
from ._lib import JpegAmdError, LIB_PATH  # noqa: F401
from .api import (  # noqa: F401
    RGB, YCbCr, Component, Context, Layout, Planar, Rectangular, Scan, Spectral,
    compression_quanta, default_context, inspect,
)

__all__ = ["RGB", "YCbCr", "Component", "Context", "Layout", "Planar", "Rectangular",
           "Scan", "Spectral", "JpegAmdError", "compression_quanta", "default_context", "inspect"]
