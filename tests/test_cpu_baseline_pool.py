"""bench.py's cpu_baseline helpers (CPU only): the oracle's persistent thread pool gives the same results as single calls for any
thread / pass count, the liblz4 leg (when the host has the library) agrees with the port, and bench.usable_cores reads a quota."""
import importlib.util
import os

import numpy as np

import oracle
from conftest import ROOT


def _pack(blocks):
    off = np.zeros(len(blocks), dtype=np.uint64)
    pos = 0
    for i, b in enumerate(blocks):
        off[i] = pos
        pos += (len(b) + 15) & ~15
    buf = np.zeros(pos, dtype=np.uint8)
    for i, b in enumerate(blocks):
        buf[int(off[i]):int(off[i]) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    return buf, off, np.array([len(b) for b in blocks], dtype=np.uint64)


def test_pool_passes_and_liblz4_leg():
    L = oracle.lib()
    S = 4096
    raws = [oracle.synth_v1(S, i) for i in range(37)]
    blocks = [oracle.lz4_compress_raw(r)[1] for r in raws]
    buf, off, ln = _pack(blocks)
    want = np.frombuffer(b"".join(raws), dtype=np.uint8)
    for op in ([0, 4] if L.cjo_have_liblz4() else [0]):
        for threads, reps in ((1, 1), (3, 4), (8, 2)):
            out = np.zeros(len(raws) * S, dtype=np.uint8)
            res = np.zeros(len(raws), dtype=np.int64)
            rc = L.cjo_batch_run_reps(op, threads, reps, len(raws), buf.ctypes.data, off.ctypes.data, ln.ctypes.data, out.ctypes.data, S, res.ctypes.data)
            assert rc == 0 and (res == S).all() and (out == want).all(), (op, threads, reps)
    sblocks = [oracle.snappy_compress(r)[1] for r in raws]
    buf, off, ln = _pack(sblocks)
    out = np.zeros(len(raws) * S, dtype=np.uint8)
    res = np.zeros(len(raws), dtype=np.int64)
    assert L.cjo_batch_run_reps(2, 4, 3, len(raws), buf.ctypes.data, off.ctypes.data, ln.ctypes.data, out.ctypes.data, S, res.ctypes.data) == 0
    assert (res == S).all() and (out == want).all()


def test_usable_cores_is_sane():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, note = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and "logical CPUs" in note
