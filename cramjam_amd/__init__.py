"""cramjam_amd — MI355X-native drop-in for cramjam's LZ4-block / Snappy-raw hot path.

    import cramjam_amd as cramjam
    cramjam.lz4.compress_block(b"...")          # -> cramjam.Buffer, computed on the GPU
    cramjam.snappy.decompress_raw(blob)

`lz4`, `snappy`, `Buffer`, `CompressionError`, `DecompressionError` mirror the reference's Python API for
that path (reference src/lz4.rs:78-229, src/snappy.rs:52-122, src/io.rs:370-684, src/exceptions.rs) and are
implemented in the native module `_cramjam` (csrc/pymod.cpp) over the C-ABI of libcramjam_hip.so.
`Engine` / `batch` are the batch extension that makes a GPU worthwhile (no reference equivalent).
There is no CPU fallback: importing works without a GPU, computing without one raises.
"""
from ._native import Engine, EngineError  # noqa: F401

__version__ = "0.1.0"          # reference: cramjam.__version__ (tests/test_variants.py:43-46)

try:
    from ._cramjam import Buffer, CompressionError, DecompressionError, File, lz4, snappy  # noqa: F401
except ImportError as _exc:          # not built yet: keep `cramjam_amd._build` importable, fail loudly on everything else
    _missing = _exc

    def __getattr__(name):
        if name not in ("Buffer", "File", "CompressionError", "DecompressionError", "lz4", "snappy", "batch"):
            raise AttributeError(name)          # lets `from cramjam_amd import _build` fall through to the submodule import
        raise ImportError("cramjam_amd: the native module is not built (%s) — run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950 + g++). There is no CPU fallback." % _missing) from _missing
else:
    from . import batch  # noqa: F401

__all__ = ["Buffer", "File", "CompressionError", "DecompressionError", "lz4", "snappy", "Engine", "EngineError", "batch"]
