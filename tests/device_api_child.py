"""Child process of tests/test_device_api_gpu.py: torch is imported BEFORE cramjam_amd, as a user of both has to (torch ships its own
libamdhip64.so.7 and loads it by path; the library loaded first serves both — two HIP runtimes in one process do not share a device)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from cramjam_amd import batch  # noqa: E402


def _pack(blobs, pad):
    ln = np.array([len(b) for b in blobs], np.uint64)
    off = np.concatenate([[0], np.cumsum((ln + pad + 15) & ~np.uint64(15))[:-1]]).astype(np.uint64)
    buf = np.zeros(int(off[-1] + ln[-1]) + 64, np.uint8)
    for k, b in enumerate(blobs):
        buf[int(off[k]):int(off[k]) + len(b)] = np.frombuffer(b, np.uint8)
    return buf, off, ln


def test_24k_chunks_from_torch_tensors_decode_to_the_oracles_bytes(codec):
    U, n = 96, 24576
    raws = [oracle.synth_v1(65536 if i % 7 else 40000 + 13 * i, 900 + i) for i in range(U)]
    comp = [(oracle.lz4_compress_raw(r)[1] if codec == "lz4" else oracle.snappy_compress(r)[1]) for r in raws]
    blobs = [comp[i % U] for i in range(n)]
    buf, off, ln = _pack(blobs, 3)
    dev = torch.device("cuda:0")
    t_in = torch.from_numpy(buf).to(dev)
    cap = np.array([len(raws[i % U]) for i in range(n)], np.uint64)
    out_off = np.concatenate([[0], np.cumsum(cap + 16)[:-1]]).astype(np.uint64)
    t_out = torch.full((int(out_off[-1] + cap[-1]) + 64,), 0xAB, dtype=torch.uint8, device=dev)
    # metadata as device tensors (int64 views of the uint64 arrays), result as a device tensor: nothing but tensors at the call
    t_off, t_len = torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(ln.view(np.int64)).to(dev)
    t_ooff, t_cap = torch.from_numpy(out_off.view(np.int64)).to(dev), torch.from_numpy(cap.view(np.int64)).to(dev)
    t_res = torch.empty(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    fn = batch.lz4_decompress_blocks_device if codec == "lz4" else batch.snappy_decompress_raw_many_device
    side = torch.cuda.Stream()                              # the batch is ordered on a stream of the caller's
    with torch.cuda.stream(side):
        t_out.fill_(0xAB)                                   # produced on that stream right before the batch
        r = fn(t_in, t_off, t_len, t_out, t_ooff, t_cap, result=t_res, stream=side.cuda_stream)
    assert r is t_res
    try:
        fn(t_in, t_off, t_len, t_out, t_ooff, t_cap, result=t_res, stream=0)
        raise AssertionError("the NULL stream was accepted")
    except ValueError:
        pass
    res = t_res.cpu().numpy()
    out = t_out.cpu().numpy()
    for i in range(n):
        raw = raws[i % U]
        assert res[i] == len(raw), (i, int(res[i]))
        assert out[int(out_off[i]):int(out_off[i]) + len(raw)].tobytes() == raw, i
    # host metadata and no result tensor: the call uploads the arrays and returns the results as numpy
    res2 = fn(t_in, off, ln, t_out, out_off, cap)
    assert isinstance(res2, np.ndarray) and (res2 == res).all()


def test_compress_from_torch_tensors_round_trips_through_the_oracle(codec):
    from cramjam_amd import _native as N
    raws = [oracle.synth_v1(65536, 50 + i) for i in range(40)] + [b"", b"abc", bytes(70000)]
    n = len(raws)
    buf, off, ln = _pack(raws, 0)
    L = N.lib()
    cap = np.array([L.cj_lz4_block_compress_bound(len(r), 1) if codec == "lz4" else L.cj_snappy_raw_max_compress_len(len(r)) for r in raws], np.uint64)
    out_off = np.concatenate([[0], np.cumsum(cap + 8)[:-1]]).astype(np.uint64)
    dev = torch.device("cuda:0")
    t_in = torch.from_numpy(buf).to(dev)
    t_out = torch.zeros(int(out_off[-1] + cap[-1]) + 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    fn = batch.lz4_compress_blocks_device if codec == "lz4" else batch.snappy_compress_raw_many_device
    res = fn(t_in, off, ln, t_out, out_off, cap)
    out = t_out.cpu().numpy()
    for i, raw in enumerate(raws):
        blk = out[int(out_off[i]):int(out_off[i]) + int(res[i])].tobytes()
        if codec == "lz4":
            assert int.from_bytes(blk[:4], "little") == len(raw)          # store_size=True, the reference's default (src/lz4.rs:113)
            assert oracle.lz4_decompress_raw(blk[4:], len(raw)) == (len(raw), raw), i
        else:
            assert oracle.snappy_decompress(blk) == (len(raw), raw), i


def test_dlpack_objects_and_refusals():
    class OnlyDlpack:                                   # an object that offers nothing but __dlpack__ (no __cuda_array_interface__)
        def __init__(self, t): self.t = t
        def __dlpack__(self, stream=None): return self.t.__dlpack__()
        def __dlpack_device__(self): return self.t.__dlpack_device__()
    raw = oracle.synth_v1(65536, 5)
    blk = oracle.lz4_compress_raw(raw)[1]
    dev = torch.device("cuda:0")
    t_in = torch.frombuffer(bytearray(blk) + bytearray(32), dtype=torch.uint8).to(dev)
    t_out = torch.zeros(65536 + 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    res = batch.lz4_decompress_blocks_device(OnlyDlpack(t_in), [0], [len(blk)], OnlyDlpack(t_out), [0], [65536])
    assert list(res) == [65536] and t_out[:65536].cpu().numpy().tobytes() == raw
    for exc, call in ((TypeError, lambda: batch.lz4_decompress_blocks_device(blk, [0], [len(blk)], t_out, [0], [65536])),               # host bytes are not a device buffer
                      (ValueError, lambda: batch.lz4_decompress_blocks_device(t_in, [0, 0], [len(blk)], t_out, [0], [65536])),            # ragged metadata
                      (ValueError, lambda: batch.lz4_decompress_blocks_device(torch.zeros(64, dtype=torch.uint8), [0], [8], t_out, [0], [64]))):   # a CPU tensor
        try:
            call()
        except exc:
            continue
        raise AssertionError("no %s" % exc.__name__)


if __name__ == "__main__":
    for codec in ("lz4", "snappy"):
        test_24k_chunks_from_torch_tensors_decode_to_the_oracles_bytes(codec)
        test_compress_from_torch_tensors_round_trips_through_the_oracle(codec)
    test_dlpack_objects_and_refusals()
    print("device api: ok")
