"""cramjam.Buffer contract, restated from the reference's black-box tests
(/root/reference/tests/test_rust_io.py:8-66, tests/test_buffer_view.py:8-175, tests/test_variants.py:292-311).
Runs on CPU: Buffer is host-side plumbing, no device involved."""
import gc

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import cramjam_amd as cramjam
from cramjam_amd import Buffer


def test_obj_api():
    buf = Buffer()
    assert buf.write(b"bytes") == 5
    assert buf.tell() == 5
    assert buf.seek(0) == 0
    assert buf.read() == b"bytes"
    assert buf.seek(-1, 2) == 4
    assert buf.read() == b"s"
    assert buf.seek(-2, whence=1) == 3
    assert buf.read() == b"es"
    with pytest.raises(ValueError):
        buf.seek(1, 3)
    for out in (b"12345", bytearray(b"12345"), Buffer()):
        buf.seek(0)
        buf.readinto(out)
        if isinstance(out, Buffer):
            out.seek(0)
            assert out.read() == b"bytes"
        else:
            assert bytes(out) == b"bytes"
    buf.set_len(2)
    buf.seek(0)
    assert buf.read() == b"by"
    buf.set_len(10)
    buf.seek(0)
    assert buf.read() == b"by" + bytes(8)
    buf.truncate()
    buf.seek(0)
    assert buf.read() == b""
    assert buf.seekable() is True


@settings(max_examples=50, deadline=None)
@given(data=st.binary())
def test_dunders(data):
    obj = Buffer()
    assert len(obj) == 0 and bool(obj) is False
    obj.write(data)
    assert len(obj) == len(data) == obj.len()
    assert bool(obj) is bool(len(data))
    assert f"len={len(data)}" in str(obj) and repr(obj) == f"cramjam.Buffer<len={len(data)}>"
    assert bytes(obj) == data and memoryview(obj).tobytes() == data
    assert np.frombuffer(obj, dtype=np.uint8).tobytes() == data
    other = Buffer(data)
    other.seek(0, 2)
    assert obj == other
    other.seek(0)
    assert (obj == other) == (len(data) == 0)


def test_buffer_protocol_is_not_writable():
    b = Buffer(b"abc")
    mv = memoryview(b)
    assert mv.format == "B" and mv.ndim == 1 and mv.shape == (3,)
    # a PyBUF_WRITABLE request is refused (reference src/io.rs:649-651); readinto() asks for one
    import io
    with pytest.raises((BufferError, TypeError)):
        io.BytesIO(b"xyz").readinto(b)
    assert bytes(b) == b"abc"


@pytest.mark.parametrize("copy", (None, True, False))
def test_buffer_view(copy):
    kwargs = {} if copy is None else {"copy": copy}
    data = bytearray(b"bytes")
    buf = Buffer(data, **kwargs)
    buf.write(b"0")
    assert data == (b"0ytes" if copy is False else b"bytes")


def test_view_write_limits():
    data = bytearray(b"bytes")
    buf = Buffer(data, copy=False)
    with pytest.raises(OSError, match="Too much to write on view"):
        buf.write(b"0" * 6)
    assert data == b"bytes"
    for _ in range(5):
        buf.write(b"0")
    with pytest.raises(OSError, match="Too much to write on view"):
        buf.write(b"0")
    assert data == b"00000"


@pytest.mark.parametrize("n", range(0, 7))
def test_view_cannot_resize(n):
    data = b"bytes"
    buf = Buffer(data, copy=False)
    with pytest.raises(OSError, match="Cannot set length on unowned buffer"):
        buf.set_len(n)
    with pytest.raises(OSError, match="Cannot truncate unowned buffer"):
        buf.truncate()
    assert data == b"bytes"


@pytest.mark.parametrize("whence", (0, 1, 2))
def test_view_bad_seek(whence):
    buf = Buffer(bytearray(b"bytes"), copy=False)
    buf.seek(2, whence=0)
    buf.seek(2, whence=1)
    buf.seek(-2, whence=2)
    buf.seek(0)
    with pytest.raises(OSError, match="Bad seek: cannot seek outside bounds of unowned buffer"):
        buf.seek(10, whence=whence)
    buf.write(b"0")


def test_view_keeps_object_alive():
    n_refs = 0

    def make():
        nonlocal n_refs
        data = bytearray(b"bytes")
        b = cramjam.Buffer(data, copy=False)
        n_refs = b.get_view_reference_count()
        return b
    buf = make()
    gc.collect()
    rc = buf.get_view_reference_count()
    assert rc is not None and 0 < rc < n_refs
    assert buf.read() == b"bytes"
    assert Buffer(b"x").get_view_reference() is None and Buffer(b"x").get_view_reference_count() is None


def test_view_follows_underlying_size():
    data = Buffer()
    data.write(b"bytes")
    buf = Buffer(data, copy=False)
    buf.write(b"12345")
    with pytest.raises(IOError, match="Too much to write on view"):
        buf.write(b"6")
    assert len(buf) == 5
    data.write(b"s")
    assert len(buf) == 6
    assert buf.tell() == 5
    buf.write(b"6")
    assert buf.tell() == 6
    data.set_len(2)
    assert buf.tell() == 2
    with pytest.raises(IOError, match="Too much to write on view"):
        buf.write(b"6")
    buf.seek(1)
    buf.write(b"1")
    assert buf.tell() == 2 and len(buf) == 2


def test_view_cannot_read_past_end():
    data = b"bytes"
    buf = cramjam.Buffer(data, copy=False)
    assert buf.read(len(data) * 2) == data
    b = b""
    buf.seek(0)
    for i in range(10):
        b += buf.read(i)
    assert b == data


def test_inputs_must_be_contiguous_bytes_like():
    with pytest.raises(TypeError):
        Buffer(12)
    arr = np.arange(16, dtype=np.uint8).reshape(4, 4)
    assert bytes(Buffer(arr)) == arr.tobytes()                 # n-dim C-contiguous numpy seen as flat bytes
    with pytest.raises(BufferError):
        Buffer(arr[:, ::2])
    assert bytes(Buffer(np.arange(4, dtype=np.uint32))) == np.arange(4, dtype=np.uint32).tobytes()


def test_errors_are_exception_subclasses_and_no_cpu_fallback():
    assert issubclass(cramjam.CompressionError, Exception) and issubclass(cramjam.DecompressionError, Exception)
    import torch
    if not torch.cuda.is_available():
        # the product path must fail loudly without a device (there is no CPU codec in the library)
        with pytest.raises(RuntimeError, match="no usable HIP device"):
            cramjam.lz4.compress_block(b"howdy neighbor")
        with pytest.raises(RuntimeError, match="no usable HIP device"):
            cramjam.snappy.decompress_raw(b"\x0e4howdy neighbor")
    # helpers that are pure arithmetic work anywhere
    assert cramjam.lz4.compress_block_bound(b"x" * 65536) == 65536 + 65536 // 255 + 16 + 4
    assert cramjam.snappy.compress_raw_max_len(b"x" * 65536) == 76490
    assert cramjam.snappy.decompress_raw_len(b"\x0e4howdy neighbor") == 14
    assert cramjam.snappy.decompress_raw_len(b"") == 0
    with pytest.raises(cramjam.DecompressionError):
        cramjam.snappy.decompress_raw_len(b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff")


# ---- cramjam.File (reference src/io.rs:30-172; tests/test_rust_io.py:8-66, tests/test_variants.py:292-311) --------------
def test_file_obj_api(tmp_path):
    File, Buffer = cramjam.File, cramjam.Buffer
    buf = File(str(tmp_path / "file.txt"))
    assert buf.write(b"bytes") == 5
    assert buf.tell() == 5
    assert buf.seek(0) == 0
    assert buf.read() == b"bytes"
    assert buf.seek(-1, 2) == 4
    assert buf.read() == b"s"
    assert buf.seek(-2, whence=1) == 3
    assert buf.read() == b"es"
    with pytest.raises(ValueError):
        buf.seek(1, 3)
    for out in (bytearray(b"12345"), File(str(tmp_path / "test.txt")), Buffer()):
        buf.seek(0)
        assert buf.readinto(out) == 5
        if isinstance(out, (File, Buffer)):
            out.seek(0)
            assert out.read() == b"bytes"
        else:
            assert out == bytearray(b"bytes")
    buf.seek(0)
    with pytest.raises(OSError):
        buf.readinto(bytearray(3))                        # output smaller than what is left: "failed to write whole buffer"
    buf.set_len(2)
    buf.seek(0)
    assert buf.read() == b"by"
    buf.set_len(10)
    buf.seek(0)
    assert buf.read() == b"by" + bytes(8)
    assert buf.seekable() and len(buf) == 10 and buf.len() == 10 and bool(buf)
    assert repr(buf) == "cramjam.File<path=%s, len=10>" % (tmp_path / "file.txt")
    buf.truncate()
    assert len(buf) == 0 and not buf
    assert buf.tell() == 10 and buf.seek(0) == 0          # set_len does not move the cursor (std::fs::File::set_len)
    # BytesType inputs of write(): Buffer is consumed from its position, another File from its position
    b = Buffer(b"0123456789"); b.seek(4)
    assert buf.write(b) == 6 and b.tell() == 10
    other = File(str(tmp_path / "other.txt")); other.write(b"abc"); other.seek(1)
    assert buf.write(other) == 2
    buf.seek(0)
    assert buf.read(3) == b"456" and buf.read() == b"789bc"
    # Buffer <-> File
    bb = Buffer()
    buf.seek(0)
    assert bb.write(buf) == 8 and bytes(bb) == b"456789bc"
    bb.seek(0)
    sink = File(str(tmp_path / "sink.txt"), truncate=True)
    assert bb.readinto(sink) == 8 and sink.seek(0) == 0 and sink.read() == b"456789bc"
    # modes
    ro = File(str(tmp_path / "sink.txt"), write=False)
    assert ro.read() == b"456789bc"
    with pytest.raises(OSError):
        ro.write(b"x")
    ap = File(str(tmp_path / "sink.txt"), append=True)
    ap.write(b"!!")
    assert File(str(tmp_path / "sink.txt")).read() == b"456789bc!!"
    with pytest.raises(TypeError):
        cramjam.lz4.compress_block(ap)                    # block functions need bytes in memory (the reference panics here)


# ---- hostile sizes must raise, never abort the interpreter (round-1 advisor finding: uncaught std::bad_alloc /
#      std::length_error used to reach std::terminate).  No GPU needed: every case fails before any device work. ----
def _run_snippet(code):
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (ROOT, code)], capture_output=True, text=True, timeout=120)
    return r


def test_hostile_sizes_raise_python_exceptions():
    code = r'''
import struct, cramjam_amd as cj
ok = 0
# LZ4 frame: FLG=0x68 (version 01, independent blocks, content size present), BD=0x40, content_size=2**46, valid header checksum
import ctypes
hdr = bytes([0x04, 0x22, 0x4D, 0x18, 0x68, 0x40]) + struct.pack("<Q", 1 << 46)
from cramjam_amd import _native as N
try:
    cj.lz4.decompress(hdr + b"\x00" + b"\x00\x00\x00\x00")
except Exception as e:
    ok += 1
try:
    cj.lz4.decompress_block(b"abc", output_len=2**63)
except (MemoryError, ValueError, OverflowError, RuntimeError, cj.DecompressionError) as e:
    ok += 1
try:
    cj.Buffer().set_len(2**62)
except (MemoryError, ValueError, OverflowError) as e:
    ok += 1
try:
    cj.snappy.decompress_raw(b"\xff\xff\xff\xff\x0f")          # a 5-byte stream announcing 4 GiB
except cj.DecompressionError as e:
    ok += 1
try:
    cj.lz4.decompress_block(struct.pack("<I", 0x7E000000) + b"\x00")   # prefix announcing 2 GiB for a 1-byte block
except cj.DecompressionError as e:
    ok += 1
print("survived", ok)
'''
    r = _run_snippet(code)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "survived 5" in r.stdout, (r.stdout, r.stderr[-2000:])
