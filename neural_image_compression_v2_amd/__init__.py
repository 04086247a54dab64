"""MI355X (gfx950) native implementation of the grid + positional-encoding + MLP coordinate-network hot path
of 21K1113/Neural_Image_Compression_V2.

Modules mirror the reference's files for this path (same function names and argument meaning):
``fp_def``, ``utils``, ``models``, ``positional_encoding``, ``var2`` (settings), ``image_compression`` (driver-level
functions), plus ``fused`` (geometry descriptors + autograd glue) and ``distributed`` (sample-sharded data
parallel step over RCCL).  All arithmetic runs in libnicv2_hip.so (C ABI: include/nicv2_hip.h); there is no
CPU or eager-torch fallback - importing the package works anywhere, calling it needs a HIP device.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def library_path() -> str:
    return _lib.LIB_PATH
