"""CPU ORACLE — test infrastructure only (see oracle/cj_oracle.h).

ctypes view of oracle/libcj_oracle.so.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (cramjam_amd) never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcj_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in
            ("lz4_block_oracle.c", "snappy_raw_oracle.c", "snappy_frame_oracle.c", "lz4_frame_oracle.c", "synth_batch_oracle.c", "cj_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcj_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, sz, i64 = C.c_void_p, C.c_size_t, C.c_int64
        for name, res, args in [
            ("cjo_lz4_compress_bound_raw", sz, [sz]),
            ("cjo_lz4_compress_raw", i64, [u8p, sz, u8p, sz]),
            ("cjo_lz4_decompress_raw", i64, [u8p, sz, u8p, sz]),
            ("cjo_lz4_block_compress_bound", sz, [sz, C.c_int]),
            ("cjo_lz4_block_compress", i64, [u8p, sz, u8p, sz, C.c_int]),
            ("cjo_lz4_block_decompress", i64, [u8p, sz, u8p, sz, C.c_int]),
            ("cjo_snappy_max_compress_len", sz, [sz]),
            ("cjo_snappy_decompress_len", i64, [u8p, sz]),
            ("cjo_snappy_compress", i64, [u8p, sz, u8p, sz]),
            ("cjo_snappy_decompress", i64, [u8p, sz, u8p, sz]),
            ("cjo_crc32c", C.c_uint32, [u8p, sz]),
            ("cjo_crc32c_masked", C.c_uint32, [u8p, sz]),
            ("cjo_snappy_frame_max_compress_len", sz, [sz]),
            ("cjo_snappy_frame_compress", i64, [u8p, sz, u8p, sz]),
            ("cjo_snappy_frame_compress_bs", i64, [u8p, sz, u8p, sz, sz]),
            ("cjo_snappy_frame_decompress_len", i64, [u8p, sz]),
            ("cjo_snappy_frame_decompress", i64, [u8p, sz, u8p, sz]),
            ("cjo_xxh32", C.c_uint32, [u8p, sz, C.c_uint32]),
            ("cjo_lz4_frame_compress_bound", sz, [sz, C.c_int]),
            ("cjo_lz4_frame_compress", i64, [u8p, sz, u8p, sz, C.c_int, C.c_int]),
            ("cjo_lz4_frame_decompress_bound", i64, [u8p, sz]),
            ("cjo_lz4_frame_decompress", i64, [u8p, sz, u8p, sz]),
            ("cjo_synth_v1", None, [u8p, sz, C.c_uint64, C.c_uint64]),
            ("cjo_batch_run", C.c_int, [C.c_int, C.c_int, sz, u8p, u8p, u8p, u8p, sz, u8p]),
            ("cjo_batch_run_reps", C.c_int, [C.c_int, C.c_int, C.c_int, sz, u8p, u8p, u8p, u8p, sz, u8p]),
            ("cjo_have_liblz4", C.c_int, []),
            ("cjo_have_libsnappy", C.c_int, []),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _in(b):
    b = bytes(b)
    return C.cast(C.c_char_p(b), C.c_void_p), len(b), b


def _call_out(fn, data, cap, *extra):
    p, n, keep = _in(data)
    out = C.create_string_buffer(max(cap, 1))
    r = fn(p, n, C.cast(out, C.c_void_p), cap, *extra)
    return r, out.raw[:max(r, 0)]


def lz4_compress_raw(data, cap=None):
    cap = lib().cjo_lz4_compress_bound_raw(len(data)) if cap is None else cap
    return _call_out(lib().cjo_lz4_compress_raw, data, cap)


def lz4_decompress_raw(data, cap):
    return _call_out(lib().cjo_lz4_decompress_raw, data, cap)


def lz4_block_compress(data, prepend=True, cap=None):
    cap = lib().cjo_lz4_block_compress_bound(len(data), 1) if cap is None else cap
    return _call_out(lib().cjo_lz4_block_compress, data, cap, int(prepend))


def lz4_block_decompress(data, cap, size_prepended):
    return _call_out(lib().cjo_lz4_block_decompress, data, cap, int(size_prepended))


def snappy_compress(data, cap=None):
    cap = lib().cjo_snappy_max_compress_len(len(data)) if cap is None else cap
    return _call_out(lib().cjo_snappy_compress, data, cap)


def snappy_decompress(data, cap=None):
    if cap is None:
        cap = max(lib().cjo_snappy_decompress_len(_in(data)[0], len(data)), 0) if len(data) else 0
    return _call_out(lib().cjo_snappy_decompress, data, cap)


def snappy_decompress_len(data):
    p, n, keep = _in(data)
    return lib().cjo_snappy_decompress_len(p, n)


def crc32c(data, masked=False):
    p, n, keep = _in(data)
    return (lib().cjo_crc32c_masked if masked else lib().cjo_crc32c)(p, n)


def snappy_frame_compress(data, cap=None, block_size=None):
    cap = lib().cjo_snappy_frame_max_compress_len(len(data)) + (len(data) // (block_size or 65536) + 2) * 8 if cap is None else cap
    if block_size is None:
        return _call_out(lib().cjo_snappy_frame_compress, data, cap)
    return _call_out(lib().cjo_snappy_frame_compress_bs, data, cap, block_size)


def snappy_frame_decompress_len(data):
    p, n, keep = _in(data)
    return lib().cjo_snappy_frame_decompress_len(p, n)


def snappy_frame_decompress(data, cap=None):
    if cap is None:
        cap = max(snappy_frame_decompress_len(data), 0)
    return _call_out(lib().cjo_snappy_frame_decompress, data, cap)


def xxh32(data, seed=0):
    p, n, keep = _in(data)
    return lib().cjo_xxh32(p, n, seed)


LZ4F_LINKED, LZ4F_BLOCK_CHECKSUM, LZ4F_CONTENT_SIZE, LZ4F_NO_CONTENT_CHECKSUM = 1, 2, 4, 8


def lz4_frame_compress(data, bs_code=4, flags=0):
    cap = lib().cjo_lz4_frame_compress_bound(len(data), bs_code)
    return _call_out(lib().cjo_lz4_frame_compress, data, cap, bs_code, flags)


def lz4_frame_decompress_bound(data):
    p, n, keep = _in(data)
    return lib().cjo_lz4_frame_decompress_bound(p, n)


def lz4_frame_decompress(data, cap=None):
    if cap is None:
        cap = max(lz4_frame_decompress_bound(data), 0)
    return _call_out(lib().cjo_lz4_frame_decompress, data, cap)


def synth_v1(chunk_bytes, index, seed=0x5EED):
    out = C.create_string_buffer(max(chunk_bytes, 1))
    lib().cjo_synth_v1(C.cast(out, C.c_void_p), chunk_bytes, index, seed)
    return out.raw[:chunk_bytes]
