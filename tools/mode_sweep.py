#!/usr/bin/env python3
"""Sweep LZ4-decode mapping (wave / lane / parse+LDS) over batch sizes; prints ms per launch. GPU only."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cramjam_amd import _native as N

L = N.lib(); eng = N.Engine(0); dev = torch.device("cuda", 0)
S = 65536
lz4 = C.CDLL("liblz4.so.1")
def build(n, kind):
    U = min(n, 2048)
    raw = torch.empty(U * S, dtype=torch.uint8, device=dev)
    if kind == "synth":
        N.check(L.cj_bench_synth_v1(raw.data_ptr(), S, S, 0, U, 0x5EED, None)); torch.cuda.synchronize()
    elif kind == "zeros":
        raw.zero_()
    else:
        raw.copy_(torch.randint(0, 256, (U * S,), dtype=torch.uint8, device=dev))
    rh = raw.cpu().numpy(); bound = S + S // 255 + 16
    comp = np.zeros(U * bound, np.uint8); clen = np.zeros(U, np.uint64)
    for i in range(U):
        clen[i] = lz4.LZ4_compress_default(C.c_void_p(rh.ctypes.data + i * S), C.c_void_p(comp.ctypes.data + i * bound), S, bound)
    ids = np.arange(n) % U
    in_off = (ids * bound).astype(np.uint64); in_len = clen[ids]
    out_off = (np.arange(n) * S).astype(np.uint64); out_cap = np.full(n, S, np.uint64)
    meta = torch.from_numpy(np.concatenate([in_off, in_len, out_off, out_cap, np.zeros(n, np.uint64)]).view(np.int64)).to(dev)
    cin = torch.from_numpy(comp).to(dev); out = torch.empty(n * S, dtype=torch.uint8, device=dev)
    return raw, cin, out, meta, float(clen.mean())
for kind in ("synth", "zeros", "random"):
    for n in (1, 8, 64, 512, 4096, 32768):
        raw, cin, out, meta, cl = build(n, kind); mp = meta.data_ptr(); torch.cuda.synchronize()
        row = []
        for name, fl in (("wave", N.FLAG_FORCE_WAVE_PER_CHUNK), ("lane", N.FLAG_FORCE_LANE_PER_CHUNK), ("lds", N.FLAG_FORCE_LDS_PER_CHUNK)):
            a = (N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, fl, n, cin.data_ptr(), mp, mp + 8 * n, out.data_ptr(), mp + 16 * n, mp + 24 * n, mp + 32 * n)
            eng.batch_device_timed(*a, 1)
            ms = eng.batch_device_timed(*a, 3)
            res = meta[4 * n:].cpu().numpy(); assert (res == S).all(), (kind, n, name, res[:4])
            row.append("%s %8.3f ms (%7.1f GB/s)" % (name, ms, n * S / ms / 1e6))
        print("%-6s n=%6d clen=%7.0f | %s" % (kind, n, cl, " | ".join(row)), flush=True)
