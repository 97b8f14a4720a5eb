"""Batch extension: many independent chunks per call (what the GPU is for).  Host buffers in, host
buffers out; chunks shard round-robin over the given engines (one per GPU), no collective."""
from concurrent.futures import ThreadPoolExecutor

from . import _native as N

_engines = {}


def _engine(device):
    if device not in _engines:
        _engines[device] = N.Engine(device)
    return _engines[device]


def _run_into(codec, op, flags, inputs, out_caps, devices, out):
    """results + memoryviews into `out` (one writable buffer, chunk i behind chunk i - 1's capacity): no object per output byte"""
    devices = list(devices) if devices is not None else [0]
    n = len(inputs)
    offsets, run = [], 0
    for c in out_caps:
        offsets.append(run); run += int(c)
    mv = memoryview(out).cast("B")
    if len(devices) == 1:
        res = _engine(devices[0]).batch_host_into(codec, op, flags, inputs, out_caps, out, offsets)
    else:
        shards = [list(range(g, n, len(devices))) for g in range(len(devices))]

        def work(g):
            idx = shards[g]
            return _engine(devices[g]).batch_host_into(codec, op, flags, [inputs[i] for i in idx], [out_caps[i] for i in idx], out, [offsets[i] for i in idx])
        with ThreadPoolExecutor(len(devices)) as ex:
            parts = list(ex.map(work, range(len(devices))))
        res = [None] * n
        for g, r in enumerate(parts):
            for k, i in enumerate(shards[g]):
                res[i] = r[k]
    return res, [mv[offsets[i]:offsets[i] + max(res[i], 0)] for i in range(n)]


def _run(codec, op, flags, inputs, out_caps, devices, out=None):
    if out is not None:
        return _run_into(codec, op, flags, inputs, out_caps, devices, out)
    devices = list(devices) if devices is not None else [0]
    n = len(inputs)
    if len(devices) == 1:
        return _engine(devices[0]).batch_host(codec, op, flags, inputs, out_caps)
    shards = [list(range(g, n, len(devices))) for g in range(len(devices))]   # chunk i -> gpu i mod G

    def work(g):
        idx = shards[g]
        return _engine(devices[g]).batch_host(codec, op, flags, [inputs[i] for i in idx], [out_caps[i] for i in idx])
    with ThreadPoolExecutor(len(devices)) as ex:
        parts = list(ex.map(work, range(len(devices))))
    res, outs = [None] * n, [None] * n
    for g, (r, o) in enumerate(parts):
        for k, i in enumerate(shards[g]):
            res[i], outs[i] = r[k], o[k]
    return res, outs


def lz4_decompress_blocks(blocks, output_lens, store_size=False, devices=None, out=None):
    """decode many LZ4 blocks; returns (results, outputs) with results[i] = length or a negative CJ_E_* code.
    out: ONE writable buffer (bytearray, numpy array) of at least sum(output_lens) bytes — the outputs are then memoryviews into it
    (chunk i behind chunk i - 1's capacity) instead of new `bytes` objects: the C-ABI's host rate without an allocation per chunk."""
    return _run(N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, N.FLAG_LZ4_SIZE_PREFIX if store_size else 0, blocks, output_lens, devices, out)


def lz4_compress_blocks(chunks, store_size=True, devices=None, out=None):
    """out: as in lz4_decompress_blocks; it has to hold sum(compress_block_bound(len(chunk))) bytes"""
    L = N.lib()
    caps = [L.cj_lz4_block_compress_bound(len(c), 1 if store_size else 0) for c in chunks]
    return _run(N.CODEC_LZ4_BLOCK, N.OP_COMPRESS, N.FLAG_LZ4_SIZE_PREFIX if store_size else 0, chunks, caps, devices, out)


def snappy_decompress_raw_many(blocks, devices=None, out=None):
    L = N.lib()
    import ctypes as C
    caps = []
    for b in blocks:
        b = bytes(b)
        caps.append(max(L.cj_snappy_raw_decompress_len(C.cast(C.c_char_p(b), C.c_void_p), len(b)), 0))
    return _run(N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS, 0, blocks, caps, devices, out)


def snappy_compress_raw_many(chunks, devices=None, out=None):
    L = N.lib()
    caps = [L.cj_snappy_raw_max_compress_len(len(c)) for c in chunks]
    return _run(N.CODEC_SNAPPY_RAW, N.OP_COMPRESS, 0, chunks, caps, devices, out)


# ---- device-resident batches: no host copy, no ctypes at the call site ---------------------------------------------------------
# Every buffer argument is any object that exposes `__cuda_array_interface__` (torch tensors on ROCm, cupy arrays, numba device
# arrays) or `__dlpack__` (anything else that lives in HBM); torch is never imported here.  The reference's API is one Python call
# per buffer (/root/reference/src/lz4.rs:78-131, src/snappy.rs:52-78); this is the same call for a batch that already sits in
# HBM: chunk i is inp[in_off[i] : in_off[i] + in_len[i]] and decodes / encodes into out[out_off[i] : out_off[i] + out_cap[i]].
import ctypes as _C


class _DLDevice(_C.Structure):
    _fields_ = [("device_type", _C.c_int32), ("device_id", _C.c_int32)]


class _DLDataType(_C.Structure):
    _fields_ = [("code", _C.c_uint8), ("bits", _C.c_uint8), ("lanes", _C.c_uint16)]


class _DLTensor(_C.Structure):
    _fields_ = [("data", _C.c_void_p), ("device", _DLDevice), ("ndim", _C.c_int32), ("dtype", _DLDataType),
                ("shape", _C.POINTER(_C.c_int64)), ("strides", _C.POINTER(_C.c_int64)), ("byte_offset", _C.c_uint64)]


class _DLManagedTensor(_C.Structure):
    pass


_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", _C.c_void_p),
                             ("deleter", _C.CFUNCTYPE(None, _C.POINTER(_DLManagedTensor)))]
_kDLCPU, _kDLCUDA, _kDLCUDAHost, _kDLROCM, _kDLROCMHost, _kDLCUDAManaged = 1, 2, 3, 10, 11, 13


class _DevView:
    """pointer + byte size of a contiguous device buffer, keeping its owner alive"""
    __slots__ = ("ptr", "nbytes", "itemsize", "count", "device", "_owner", "_capsule")

    def __init__(self, obj):
        self._owner, self._capsule, self.device = obj, None, None
        cai = getattr(obj, "__cuda_array_interface__", None)
        if cai is not None:
            shape, strides, typestr = tuple(cai["shape"]), cai.get("strides"), cai["typestr"]
            self.itemsize = int(typestr[2:])
            self.count = 1
            for s in shape:
                self.count *= int(s)
            if strides is not None and self.count:
                want, run = [], self.itemsize
                for s in reversed(shape):
                    want.append(run); run *= int(s)
                if tuple(strides) != tuple(reversed(want)):
                    raise ValueError("cramjam_amd.batch: device buffers must be contiguous")
            self.ptr = int(cai["data"][0]) if self.count else 0
            dev = getattr(obj, "device", None)
            # torch: .device.index; cupy: .device.id (round-5 advisor: a cupy buffer on GPU N fell back to engine 0)
            self.device = (getattr(dev, "index", None) if getattr(dev, "index", None) is not None else getattr(dev, "id", None)) if dev is not None else None
        elif hasattr(obj, "__dlpack__"):
            cap = obj.__dlpack__()
            api = _C.pythonapi
            api.PyCapsule_GetPointer.restype, api.PyCapsule_GetPointer.argtypes = _C.c_void_p, [_C.py_object, _C.c_char_p]
            p = api.PyCapsule_GetPointer(cap, b"dltensor")
            mt = _C.cast(p, _C.POINTER(_DLManagedTensor)).contents
            t = mt.dl_tensor
            if t.device.device_type not in (_kDLCUDA, _kDLROCM, _kDLCUDAManaged):
                raise ValueError("cramjam_amd.batch: the buffer is not in device memory (DLPack device type %d)" % t.device.device_type)
            if t.dtype.lanes != 1:
                raise ValueError("cramjam_amd.batch: vector dtypes are not supported")
            self.itemsize = t.dtype.bits // 8
            self.count = 1
            for k in range(t.ndim):
                self.count *= int(t.shape[k])
            if t.strides and self.count:
                run = 1
                for k in reversed(range(t.ndim)):
                    if int(t.shape[k]) != 1 and int(t.strides[k]) != run:
                        raise ValueError("cramjam_amd.batch: device buffers must be contiguous")
                    run *= int(t.shape[k])
            self.ptr = (int(t.data or 0) + int(t.byte_offset)) if self.count else 0
            self.device = int(t.device.device_id)
            self._capsule = (cap, mt)          # consumed when this view is released
        else:
            raise TypeError("cramjam_amd.batch: expected a device buffer (an object with __cuda_array_interface__ or __dlpack__), got %s"
                            % type(obj).__name__)
        self.nbytes = self.count * self.itemsize

    def release(self):
        if self._capsule is not None:
            cap, mt = self._capsule
            self._capsule = None
            if mt.deleter:
                mt.deleter(_C.pointer(mt))      # we are the consumer of the capsule: its deleter is ours to call, once
            _C.pythonapi.PyCapsule_SetName.argtypes = [_C.py_object, _C.c_char_p]
            _C.pythonapi.PyCapsule_SetName(cap, b"used_dltensor")


def _is_device_obj(x):
    return hasattr(x, "__cuda_array_interface__") or (hasattr(x, "__dlpack__") and not hasattr(x, "__array_interface__") and not isinstance(x, (list, tuple)))


def _device_batch(codec, op, flags, inp, in_off, in_len, out, out_off, out_cap, result, device, stream, sync):
    import numpy as np
    if stream is not None and int(stream) == 0:
        raise ValueError("cramjam_amd.batch: the NULL stream (torch's default stream has handle 0) cannot be named through the C-ABI, where NULL means "
                         "the engine's own stream — run the producer on a torch.cuda.Stream() and pass its .cuda_stream, or leave stream=None and synchronize")
    views, temps = [], []
    try:
        vin, vout = _DevView(inp), _DevView(out)
        views += [vin, vout]
        if device is None:
            device = vin.device if vin.device is not None else (vout.device if vout.device is not None else 0)
        eng = _engine(device)
        n = None

        def meta(x, name):
            nonlocal n, sync
            if _is_device_obj(x):
                v = _DevView(x)
                views.append(v)
                if v.itemsize != 8:
                    raise TypeError("cramjam_amd.batch: %s must hold 64-bit integers" % name)
                cnt, ptr = v.count, v.ptr
            else:                                   # a host sequence: uploaded for this call
                a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
                cnt = a.size
                ptr = eng.alloc(max(a.nbytes, 8))
                temps.append(ptr)
                eng.h2d(ptr, a)
                sync = True
            if n is None:
                n = cnt
            elif cnt != n:
                raise ValueError("cramjam_amd.batch: %s has %d entries, expected %d" % (name, cnt, n))
            return ptr
        p_in_off, p_in_len = meta(in_off, "in_off"), meta(in_len, "in_len")
        p_out_off, p_out_cap = meta(out_off, "out_off"), meta(out_cap, "out_cap")
        own_result = result is None
        if own_result:
            p_res = eng.alloc(max(8 * n, 8))
            temps.append(p_res)
            sync = True
        else:
            vr = _DevView(result)
            views.append(vr)
            if vr.itemsize != 8 or vr.count != n:
                raise ValueError("cramjam_amd.batch: result must hold %d 64-bit integers" % n)
            p_res = vr.ptr
        eng.batch_device(codec, op, flags, n, vin.ptr, p_in_off, p_in_len, vout.ptr, p_out_off, p_out_cap, p_res, stream)
        if sync:
            if stream is not None:
                N.check(N.lib().cj_stream_sync(eng.h, stream))
            else:
                eng.sync()
        if own_result:
            return eng.d2h(p_res, 8 * n, "int64")
        return result
    finally:
        for p in temps:
            _engine(device if device is not None else 0).free(p)
        for v in views:
            v.release()


def lz4_decompress_blocks_device(inp, in_off, in_len, out, out_off, out_cap, store_size=False, result=None, device=None, stream=None, sync=True):
    """Decode a batch of LZ4 blocks that already sits in HBM (reference call per buffer: src/lz4.rs:78-95).

    inp / out: device byte buffers (torch tensor, cupy array, anything with __cuda_array_interface__ or __dlpack__);
    in_off, in_len, out_off, out_cap: 64-bit integer arrays of one entry per chunk — device arrays are used in place, host
    sequences / numpy arrays are uploaded; result: optional device int64 array that receives the decoded length of every chunk
    (or a negative CJ_E_* code).  Returns `result`, or — when it was None — a numpy int64 array with the same content.
    stream: a hipStream_t handle as an int (a torch.cuda.Stream().cuda_stream; not the default stream, whose handle 0 means "the engine's own" here) to order the
    batch behind the producer of the buffers; without it the batch runs on the engine's own stream and the caller makes sure the
    buffers are ready (torch.cuda.synchronize()).  sync=False returns right after submission (device-resident metadata and result
    only).  In a process that also uses torch, import torch FIRST: both link libamdhip64.so.7, torch loads its own copy by path, and
    two HIP runtimes in one process do not share a device."""
    return _device_batch(N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, N.FLAG_LZ4_SIZE_PREFIX if store_size else 0,
                         inp, in_off, in_len, out, out_off, out_cap, result, device, stream, sync)


def lz4_compress_blocks_device(inp, in_off, in_len, out, out_off, out_cap, store_size=True, result=None, device=None, stream=None, sync=True):
    """Compress a device-resident batch into LZ4 blocks (src/lz4.rs:113-131); out_cap[i] >= cramjam.lz4.compress_block_bound(in_len[i])."""
    return _device_batch(N.CODEC_LZ4_BLOCK, N.OP_COMPRESS, N.FLAG_LZ4_SIZE_PREFIX if store_size else 0,
                         inp, in_off, in_len, out, out_off, out_cap, result, device, stream, sync)


def snappy_decompress_raw_many_device(inp, in_off, in_len, out, out_off, out_cap, result=None, device=None, stream=None, sync=True):
    """Decode a device-resident batch of Snappy raw blocks (src/snappy.rs:52-59)."""
    return _device_batch(N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS, 0, inp, in_off, in_len, out, out_off, out_cap, result, device, stream, sync)


def snappy_compress_raw_many_device(inp, in_off, in_len, out, out_off, out_cap, result=None, device=None, stream=None, sync=True):
    """Compress a device-resident batch into Snappy raw blocks (src/snappy.rs:70-78); out_cap[i] >= cramjam.snappy.compress_raw_max_len(in_len[i])."""
    return _device_batch(N.CODEC_SNAPPY_RAW, N.OP_COMPRESS, 0, inp, in_off, in_len, out, out_off, out_cap, result, device, stream, sync)
