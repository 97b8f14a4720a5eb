#!/usr/bin/env python3
"""Latency / throughput of LZ4 decode over batch sizes for the default pipeline and the forced mappings (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from cramjam_amd import _native as N
L = N.lib(); e = N.Engine(0)
S = 65536
raws = [oracle.synth_v1(S, i) for i in range(64)]
blobs = [oracle.lz4_compress_raw(r)[1] for r in raws]
for n in (1, 8, 64, 512, 4096, 16384, 32768, 65536):
    packed = b"".join(blobs[i % 64].ljust((len(blobs[i % 64]) + 15) & ~15, b"\0") for i in range(min(n, 64)))
    unit = np.frombuffer(packed, dtype=np.uint8)
    reps = (n + 63) // 64
    cin = torch.from_numpy(np.tile(unit, reps).copy()).cuda()
    offs = []; pos = 0
    for i in range(min(n, 64)):
        offs.append(pos); pos += (len(blobs[i % 64]) + 15) & ~15
    in_off = np.array([(i // 64) * len(unit) + offs[i % 64] for i in range(n)], dtype=np.uint64)
    in_len = np.array([len(blobs[i % 64]) for i in range(n)], dtype=np.uint64)
    meta = torch.from_numpy(np.concatenate([in_off, in_len, np.arange(n, dtype=np.uint64) * S, np.full(n, S, np.uint64), np.zeros(n, np.uint64)]).view(np.int64)).cuda()
    out = torch.empty(n * S, dtype=torch.uint8, device="cuda"); mp = meta.data_ptr()
    line = "n=%6d" % n
    for name, flag in (("default", 0), ("wave", N.FLAG_FORCE_WAVE_PER_CHUNK), ("lds", N.FLAG_FORCE_LDS_PER_CHUNK)):
        a = (N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, flag, n, cin.data_ptr(), mp, mp + 8 * n, out.data_ptr(), mp + 16 * n, mp + 24 * n, mp + 32 * n)
        e.batch_device_timed(*a, 2)
        ms = e.batch_device_timed(*a, 5)
        res = meta[4 * n:].cpu().numpy()
        assert (res == S).all(), (name, n, res[:4])
        line += " | %-7s %7.3f ms %7.1f GB/s" % (name, ms, n * S / ms / 1e6)
    print(line, flush=True)
