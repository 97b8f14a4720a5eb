#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch path, compress: times the cj_batch_host C call itself."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from cramjam_amd import _native as N
L = N.lib(); eng = N.Engine(0)
n = 16384
raws = [np.frombuffer(oracle.synth_v1(65536, i), np.uint8).copy() for i in range(256)]
ins = [raws[i % 256] for i in range(n)]
bound = L.cj_lz4_block_compress_bound(65536, 0)
outs = [np.zeros(bound, np.uint8) for _ in range(n)]
in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins]); in_lens = (C.c_size_t * n)(*[a.size for a in ins])
out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs]); caps = (C.c_size_t * n)(*[bound] * n)
res = (C.c_int64 * n)()
for rep in range(4):
    t0 = time.perf_counter()
    N.check(L.cj_batch_host(eng.h, N.CODEC_LZ4_BLOCK, N.OP_COMPRESS, 0, n, in_ptrs, in_lens, out_ptrs, caps, res))
    dt = time.perf_counter() - t0
    assert all(r > 0 for r in res) and oracle.lz4_decompress_raw(outs[5][:res[5]].tobytes(), 65536)[1] == raws[5].tobytes()
    print("cj_batch_host LZ4 compress, %d x 64 KiB (%.0f MB in, %.0f MB out): %.1f ms -> %.2f GB/s uncompressed" %
          (n, n * 65536 / 1e6, sum(res) / 1e6, dt * 1e3, n * 65536 / dt / 1e9), flush=True)
