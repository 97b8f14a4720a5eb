#!/usr/bin/env python3
"""Host batch of chunks ABOVE 64 KiB (cj_batch_host -> large_decompress_many: piece-parallel parse + slab decoder).
Run under rocprofv3 --kernel-trace --stats to see what the parse kernels and the slab decoder take on this shape
(BASELINE configs[4]: 256 KiB chunks)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from cramjam_amd import _native as N
L = N.lib(); eng = N.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = 262144
for codec, name in ((N.CODEC_LZ4_BLOCK, "lz4"), (N.CODEC_SNAPPY_RAW, "snappy")):
    raws = [oracle.synth_v1(S, 1000 + i) for i in range(64)]
    comp = [np.frombuffer((oracle.lz4_compress_raw(r) if codec == N.CODEC_LZ4_BLOCK else oracle.snappy_compress(r))[1], np.uint8).copy() for r in raws]
    ins = [comp[i % 64] for i in range(n)]
    outs = [np.zeros(S, np.uint8) for _ in range(n)]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins]); in_lens = (C.c_size_t * n)(*[a.size for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs]); caps = (C.c_size_t * n)(*[S] * n)
    res = (C.c_int64 * n)()
    for rep in range(3):
        t0 = time.perf_counter()
        N.check(L.cj_batch_host(eng.h, codec, N.OP_DECOMPRESS, 0, n, in_ptrs, in_lens, out_ptrs, caps, res))
        dt = time.perf_counter() - t0
        assert all(r == S for r in res) and outs[5].tobytes() == raws[5] and outs[n - 1].tobytes() == raws[(n - 1) % 64]
        print("cj_batch_host %s decode, %d x 256 KiB: %.1f ms -> %.2f GB/s uncompressed (PCIe inclusive)" % (name, n, dt * 1e3, n * S / dt / 1e9), flush=True)
