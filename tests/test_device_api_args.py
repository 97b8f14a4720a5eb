"""cramjam_amd.batch.*_device: what it makes of its arguments, without a GPU — device buffers are recognised by
`__cuda_array_interface__` / `__dlpack__`, must be contiguous, metadata must be 64-bit, host bytes and CPU DLPack tensors are refused,
the NULL stream cannot be named.  (The compute side is tests/test_device_api_gpu.py.)"""
import ctypes as C

import numpy as np
import pytest

from cramjam_amd import batch


class FakeDev:
    """an object that claims to live in device memory (nothing here ever dereferences the pointer)"""
    def __init__(self, shape, typestr="|u1", strides=None, ptr=0x7000_0000_0000):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2, "strides": strides}


def test_views_of_cuda_array_interface_objects():
    v = batch._DevView(FakeDev((10, 4), "<u4"))
    assert (v.ptr, v.nbytes, v.itemsize, v.count) == (0x7000_0000_0000, 160, 4, 40)
    v = batch._DevView(FakeDev((10, 4), "<u4", strides=(16, 4)))                 # contiguous strides spelled out
    assert v.nbytes == 160
    with pytest.raises(ValueError):
        batch._DevView(FakeDev((10, 4), "<u4", strides=(32, 4)))                 # a sliced view
    assert batch._DevView(FakeDev((0,), "|u1", ptr=0)).nbytes == 0


def test_host_objects_are_not_device_buffers():
    for x in (b"abc", bytearray(4), [1, 2, 3]):
        with pytest.raises(TypeError):
            batch._DevView(x)
    with pytest.raises(ValueError):
        batch._DevView(np.zeros(4, np.uint8))           # numpy offers __dlpack__, and its capsule says "CPU"
    assert not batch._is_device_obj(np.zeros(4, np.uint64)) and not batch._is_device_obj([0]) and batch._is_device_obj(FakeDev((4,)))


def test_cpu_dlpack_tensor_is_refused():
    class CpuDlpack:                                    # numpy's own capsule says kDLCPU
        def __init__(self): self.a = np.zeros(16, np.uint8)
        def __dlpack__(self, stream=None): return self.a.__dlpack__()
    with pytest.raises(ValueError):
        batch._DevView(CpuDlpack())


def test_the_null_stream_and_ragged_metadata_are_refused_before_anything_runs():
    buf = FakeDev((1 << 16,))
    with pytest.raises(ValueError):
        batch.lz4_decompress_blocks_device(buf, [0], [10], buf, [0], [100], stream=0)
    # (metadata counts are checked once an engine exists: tests/test_device_api_gpu.py)
