"""`import cramjam` from a checkout of this repository = the MI355X engine behind the reference's names.

A drop-in shim: `cramjam.lz4`, `cramjam.snappy`, `cramjam.Buffer`, `cramjam.File`, `cramjam.CompressionError`,
`cramjam.DecompressionError` are the objects of `cramjam_amd` (native module `cramjam_amd._cramjam`, csrc/pymod.cpp),
so `cramjam.lz4.compress_block is cramjam_amd.lz4.compress_block`.  Only the hot path of SURVEY.md §8 exists: the other
codec modules of the reference (zstd, brotli, gzip, ...) are out of scope and raise AttributeError here.
Type stubs: `__init__.pyi`, `lz4.pyi`, `snappy.pyi` next to this file (`py.typed`), written from pymod.cpp's own keyword lists.
"""
import sys as _sys

import cramjam_amd as _amd
from cramjam_amd import Buffer, CompressionError, DecompressionError, File, lz4, snappy  # noqa: F401

__version__ = _amd.__version__
__all__ = ["Buffer", "File", "CompressionError", "DecompressionError", "lz4", "snappy"]

# `import cramjam.lz4` / `from cramjam.snappy import compress_raw` resolve to the native submodules
_sys.modules[__name__ + ".lz4"] = lz4
_sys.modules[__name__ + ".snappy"] = snappy
