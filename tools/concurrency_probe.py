#!/usr/bin/env python3
"""Probe: do the LDS-path decoder (issue/LDS bound) and the lane decoder (HBM-latency bound, no LDS) overlap when
run concurrently on two streams?  Splits a 100k x 64 KiB synth batch f/(1-f) between them.  GPU only."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cramjam_amd import _native as N

L = N.lib(); dev = torch.device("cuda", 0)
e1 = N.Engine(0); e2 = N.Engine(0)
S = 65536; U = 4096; NCH = 100_000
raw = torch.empty(U * S, dtype=torch.uint8, device=dev)
N.check(L.cj_bench_synth_v1(raw.data_ptr(), S, S, 0, U, 0x5EED, None)); torch.cuda.synchronize()
lz4 = C.CDLL("liblz4.so.1"); rh = raw.cpu().numpy(); bound = S + S // 255 + 16
comp = np.zeros(U * bound, np.uint8); clen = np.zeros(U, np.uint64)
for i in range(U):
    clen[i] = lz4.LZ4_compress_default(C.c_void_p(rh.ctypes.data + i * S), C.c_void_p(comp.ctypes.data + i * bound), S, bound)
cin = torch.from_numpy(comp).to(dev); out = torch.empty(NCH * S, dtype=torch.uint8, device=dev)

def meta_for(idx):
    n = len(idx); ids = idx % U
    m = np.concatenate([(ids * bound).astype(np.uint64), clen[ids], (idx.astype(np.uint64) * S), np.full(n, S, np.uint64), np.zeros(n, np.uint64)])
    return torch.from_numpy(m.view(np.int64)).to(dev), n

def args(meta, n, flag):
    mp = meta.data_ptr()
    return (N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, flag, n, cin.data_ptr(), mp, mp + 8 * n, out.data_ptr(), mp + 16 * n, mp + 24 * n, mp + 32 * n)

allidx = np.arange(NCH)
for frac_lane in (0.35, 0.45, 0.5, 0.55, 0.65, 0.75):
    lane_idx = allidx[(allidx % 20) < int(round(frac_lane * 20))]
    lds_idx = allidx[(allidx % 20) >= int(round(frac_lane * 20))]
    ml, nl = meta_for(lane_idx) if len(lane_idx) else (None, 0)
    md, nd = meta_for(lds_idx) if len(lds_idx) else (None, 0)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        if nd: e1.batch_device(*args(md, nd, N.FLAG_FORCE_LDS_PER_CHUNK))
        if nl: e2.batch_device(*args(ml, nl, N.FLAG_FORCE_LANE_PER_CHUNK))
        if nd: e1.sync()
        if nl: e2.sync()
        dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    ok = True
    if nd: ok &= bool((md[4 * nd:] == S).all().item())
    if nl: ok &= bool((ml[4 * nl:] == S).all().item())
    print("lane fraction %.2f: %.2f ms -> %.1f GB/s  ok=%s" % (frac_lane, best * 1e3, NCH * S / best / 1e9, ok), flush=True)
