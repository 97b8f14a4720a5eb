#!/usr/bin/env python3
"""The host batch through the PYTHON entry (cramjam_amd.batch.lz4_decompress_blocks / lz4_compress_blocks): marshalling included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from cramjam_amd import batch

n = int(os.environ.get("CHUNKS", "16384"))
raws = [oracle.synth_v1(65536, i) for i in range(256)]
comp = [oracle.lz4_compress_raw(r)[1] for r in raws]
ins = [comp[i % 256] for i in range(n)]
lens = [65536] * n
for rep in range(3):
    t0 = time.perf_counter()
    res, outs = batch.lz4_decompress_blocks(ins, lens)
    dt = time.perf_counter() - t0
    assert res[7] == 65536 and bytes(outs[7]) == raws[7]
    print("batch.lz4_decompress_blocks, %d x 64 KiB: %.1f ms -> %.2f GB/s uncompressed" % (n, dt * 1e3, n * 65536 / dt / 1e9), flush=True)
chunks = [raws[i % 256] for i in range(n)]
for rep in range(3):
    t0 = time.perf_counter()
    res, outs = batch.lz4_compress_blocks(chunks, store_size=False)
    dt = time.perf_counter() - t0
    assert oracle.lz4_decompress_raw(bytes(outs[7]), 65536)[1] == raws[7]
    print("batch.lz4_compress_blocks, %d x 64 KiB: %.1f ms -> %.2f GB/s uncompressed" % (n, dt * 1e3, n * 65536 / dt / 1e9), flush=True)
import numpy as np
out = np.empty(n * 65536, np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    res, outs = batch.lz4_decompress_blocks(ins, lens, out=out)
    dt = time.perf_counter() - t0
    assert res[7] == 65536 and bytes(outs[7]) == raws[7]
    print("batch.lz4_decompress_blocks(out=one buffer), %d x 64 KiB: %.1f ms -> %.2f GB/s uncompressed" % (n, dt * 1e3, n * 65536 / dt / 1e9), flush=True)
