"""cramjam_amd — MI355X-native drop-in for cramjam's LZ4-block / Snappy-raw hot path.

`cramjam_amd.lz4`, `cramjam_amd.snappy`, `cramjam_amd.Buffer`, `CompressionError`,
`DecompressionError` mirror the reference's Python API for that path; the batch
engine (`cramjam_amd.Engine`) is the extension that makes a GPU worthwhile.
"""
from ._native import Engine, EngineError, lib as _lib  # noqa: F401
