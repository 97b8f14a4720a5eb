"""ctypes binding of libcramjam_hip.so (include/cramjam_hip.h).  The library is the product; if it is
missing or no HIP device is usable every compute call fails loudly — there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CJ_HIP_LIB") or os.path.join(_HERE, "libcramjam_hip.so")   # CJ_HIP_LIB: a tuning variant (tools/build_variant.sh)

CODEC_LZ4_BLOCK, CODEC_SNAPPY_RAW = 0, 1
OP_DECOMPRESS, OP_COMPRESS = 0, 1
FLAG_LZ4_SIZE_PREFIX = 1
FLAG_FORCE_WAVE_PER_CHUNK = 0x100
FLAG_FORCE_LANE_PER_CHUNK = 0x200
FLAG_FORCE_LDS_PER_CHUNK = 0x400
FLAG_FORCE_FUSED_PARSE = 0x10     # the workgroup decoder's parse stage inside the decoder kernel / as its own kernel, at any batch size
FLAG_FORCE_PARSE_KERNEL = 0x20
FLAG_CHUNKS_LE_32K = 0x40          # decompress: a promise that no chunk is larger (windows of that size: more workgroups per CU); host batches set them themselves
FLAG_CHUNKS_LE_16K = 0x80
FLAG_BIG_CHUNKS = 0x800            # decompress: reserve record areas for chunks of 64 KiB .. 256 KiB (cramjam_hip.h)
E_NO_DEVICE = -100

_vp, _sz, _i64, _u32, _int = C.c_void_p, C.c_size_t, C.c_int64, C.c_uint32, C.c_int

# every symbol include/cramjam_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "cj_strerror": (C.c_char_p, [_i64]),
    "cj_last_hip_error": (C.c_char_p, []),
    "cj_abi_version": (_int, []),
    "cj_device_count": (_int, []),
    "cj_lz4_block_compress_bound": (_sz, [_sz, _int]),
    "cj_lz4_block_compress": (_i64, [_vp, _sz, _vp, _sz, _int, _int, _int]),
    "cj_lz4_block_decompress": (_i64, [_vp, _sz, _vp, _sz, _int]),
    "cj_lz4_block_prefixed_len": (_i64, [_vp, _sz]),
    "cj_snappy_raw_max_compress_len": (_sz, [_sz]),
    "cj_snappy_raw_decompress_len": (_i64, [_vp, _sz]),
    "cj_snappy_raw_compress": (_i64, [_vp, _sz, _vp, _sz]),
    "cj_snappy_raw_decompress": (_i64, [_vp, _sz, _vp, _sz]),
    "cj_snappy_frame_max_compress_len": (_sz, [_sz]),
    "cj_snappy_frame_compress": (_i64, [_vp, _sz, _vp, _sz]),
    "cj_snappy_frame_decompress_len": (_i64, [_vp, _sz]),
    "cj_snappy_frame_decompress": (_i64, [_vp, _sz, _vp, _sz]),
    "cj_lz4_frame_compress_bound": (_sz, [_sz]),
    "cj_lz4_frame_compress": (_i64, [_vp, _sz, _vp, _sz, C.c_int]),
    "cj_lz4_frame_compress_blocks": (_i64, [_vp, _sz, _vp, _sz]),
    "cj_lz4_frame_decompress_bound": (_i64, [_vp, _sz]),
    "cj_lz4_frame_decompress": (_i64, [_vp, _sz, _vp, _sz]),
    "cj_engine_create": (_int, [_int, C.POINTER(_vp)]),
    "cj_engine_destroy": (None, [_vp]),
    "cj_engine_device": (_int, [_vp]),
    "cj_batch_device": (_int, [_vp, _int, _int, _u32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cj_engine_sync": (_int, [_vp]),
    "cj_stream_sync": (_int, [_vp, _vp]),
    "cj_batch_host": (_int, [_vp, _int, _int, _u32, _sz, _vp, _vp, _vp, _vp, _vp]),
    "cj_batch_device_timed": (C.c_double, [_vp, _int, _int, _u32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "cj_device_alloc": (_vp, [_vp, _sz]),
    "cj_device_free": (None, [_vp, _vp]),
    "cj_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "cj_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "cj_memcpy_d2d": (_int, [_vp, _vp, _vp, _sz]),
    "cj_memset_dev": (_int, [_vp, _vp, _int, _sz]),
}
# benchmark/test utilities exported next to the engine (include/cramjam_hip_debug.h); not part of the drop-in ABI
BENCH_SYMBOLS = {
    "cj_bench_synth_v1": (_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "cj_debug_lds_phase_cycles": (_int, [_vp, _int]),
    "cj_debug_linked_lds_frames": (C.c_ulonglong, []),
    "cj_debug_forwarded_chunks": (C.c_longlong, [_int]),
    "cj_debug_fused_parse_paths": (_int, [_vp, _int]),
    "cj_debug_big_parse": (C.c_int64, [_int, _u32, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "cj_bench_compare": (_int, [_vp, _vp, _vp, C.c_uint64, _u32, C.c_uint64, _u32, _vp, _vp]),
    "cj_debug_big_scratch_bytes": (C.c_uint64, [_vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cramjam_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(BENCH_SYMBOLS.items()):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def strerror(code):
    return lib().cj_strerror(code).decode()


class EngineError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise EngineError("%s (%s)" % (strerror(rc), lib().cj_last_hip_error().decode()))


def _with_hip_error(ex):
    """the host module's message; the HIP error text of THIS library's thread-local slot when the call went through its entry point"""
    msg = str(ex)
    return msg if msg.endswith(")") else "%s (%s)" % (msg, lib().cj_last_hip_error().decode())


def _batch_host_addr():
    """address of cj_batch_host in the library the engines of this process come from (CJ_HIP_LIB may name a tuning variant; the
    CPython module links the product library)"""
    return C.cast(lib().cj_batch_host, C.c_void_p).value


class Engine:
    """One engine per GPU (cj_engine).  Thin: device memory + batch submission."""

    def __init__(self, device=0):
        h = _vp()
        check(lib().cj_engine_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().cj_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, nbytes):
        p = lib().cj_device_alloc(self.h, nbytes)
        if not p:
            raise EngineError("device alloc of %d bytes failed: %s" % (nbytes, lib().cj_last_hip_error().decode()))
        return p

    def free(self, p):
        lib().cj_device_free(self.h, p)

    def h2d(self, dptr, data):
        import numpy as np
        a = np.ascontiguousarray(data)
        check(lib().cj_memcpy_h2d(self.h, dptr, a.ctypes.data, a.nbytes))

    def d2h(self, dptr, nbytes, dtype="uint8"):
        import numpy as np
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        check(lib().cj_memcpy_d2h(self.h, out.ctypes.data, dptr, out.nbytes))
        return out

    def sync(self):
        check(lib().cj_engine_sync(self.h))

    def batch_device(self, codec, op, flags, n, in_base, in_off, in_len, out_base, out_off, out_cap, result, stream=None):
        check(lib().cj_batch_device(self.h, codec, op, flags, n, in_base, in_off, in_len, out_base, out_off,
                                    out_cap, result, stream))

    def batch_device_timed(self, codec, op, flags, n, in_base, in_off, in_len, out_base, out_off, out_cap, result, reps):
        ms = lib().cj_batch_device_timed(self.h, codec, op, flags, n, in_base, in_off, in_len, out_base, out_off,
                                         out_cap, result, reps)
        if ms < 0:
            raise EngineError("timed batch failed: %s" % lib().cj_last_hip_error().decode())
        return ms

    def batch_host(self, codec, op, flags, inputs, out_caps):
        """inputs: list of bytes-like (anything with the buffer protocol: borrowed, not copied); out_caps: list of capacities.
        Returns (results, outputs): results[i] = bytes produced or a negative CJ_E_* code, outputs[i] = bytes."""
        from . import _cramjam                      # the CPython host layer: the engine scatters straight into the bytes objects
        try:
            return _cramjam.batch_host(self.h.value or 0, int(codec), int(op), int(flags), inputs, out_caps, _batch_host_addr())
        except RuntimeError as ex:                  # (a CJ_E_* return code of the call itself, not of a chunk)
            raise EngineError(_with_hip_error(ex)) from None

    def batch_host_into(self, codec, op, flags, inputs, out_caps, out, offsets=None):
        """the same batch into ONE writable buffer (bytearray, numpy array, ...): chunk i at out[offsets[i] : offsets[i] + out_caps[i]],
        back to back when offsets is None.  Returns results."""
        from . import _cramjam
        try:
            return _cramjam.batch_host_into(self.h.value or 0, int(codec), int(op), int(flags), inputs, out_caps, out, offsets, _batch_host_addr())
        except RuntimeError as ex:
            raise EngineError(_with_hip_error(ex)) from None

