#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch path (cj_batch_host): pack -> H2D -> kernels -> D2H -> scatter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from cramjam_amd import _native as N
eng = N.Engine(0)
n = 16384
raws = [oracle.synth_v1(65536, i % 256) for i in range(256)]
comp = [oracle.lz4_compress_raw(r)[1] for r in raws]
blobs = [comp[i % 256] for i in range(n)]
caps = [65536] * n
for rep in range(3):
    t0 = time.perf_counter()
    res, outs = eng.batch_host(N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, 0, blobs, caps)
    dt = time.perf_counter() - t0
    assert all(r == 65536 for r in res) and outs[5] == raws[5]
    print("cj_batch_host LZ4 decode, %d x 64 KiB: %.1f ms -> %.2f GB/s uncompressed (includes Python list marshalling)" % (n, dt * 1e3, n * 65536 / dt / 1e9), flush=True)
