#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch path: times the cj_batch_host C call itself
(pack into pinned staging -> H2D -> kernels -> D2H -> scatter), marshalling excluded."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from cramjam_amd import _native as N
L = N.lib(); eng = N.Engine(0)
n = 16384
raws = [oracle.synth_v1(65536, i % 256) for i in range(256)]
comp = [np.frombuffer(oracle.lz4_compress_raw(r)[1], np.uint8).copy() for r in raws]
ins = [comp[i % 256] for i in range(n)]
outs = [np.zeros(65536, np.uint8) for _ in range(n)]
in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins]); in_lens = (C.c_size_t * n)(*[a.size for a in ins])
out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs]); caps = (C.c_size_t * n)(*[65536] * n)
res = (C.c_int64 * n)()
for rep in range(4):
    t0 = time.perf_counter()
    N.check(L.cj_batch_host(eng.h, N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, 0, n, in_ptrs, in_lens, out_ptrs, caps, res))
    dt = time.perf_counter() - t0
    assert all(r == 65536 for r in res) and outs[5].tobytes() == raws[5]
    print("cj_batch_host LZ4 decode, %d x 64 KiB (%.0f MB in, %.0f MB out): %.1f ms -> %.2f GB/s uncompressed" %
          (n, sum(a.size for a in ins) / 1e6, n * 65536 / 1e6, dt * 1e3, n * 65536 / dt / 1e9), flush=True)
