#!/usr/bin/env python3
"""Per-phase cycle counters of the slab decoder (CJ_SLAB_PROFILE=1) on a host batch of 256 KiB chunks (configs[4]'s shape) and on one
large stream, next to the same counters of the independent-chunk decoder: where a slab's ~100 us go."""
import ctypes as C, os, sys
os.environ["CJ_SLAB_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle, cramjam_amd as cj
from cramjam_amd import _native as N
L = N.lib(); eng = N.Engine(0)
def phases(tag):
    ph = (C.c_ulonglong * 16)()
    L.cj_debug_lds_phase_cycles(ph, 1)
    nb = max(int(ph[5]), 1)
    print("%-44s S0 %6d  D1 %6d  D2 %6d  D3 %6d  D4 %6d  sum %7d  (%d slabs/chunks, %d through the forwarding phase)" % (tag, ph[0] // nb, ph[1] // nb, ph[2] // nb, ph[3] // nb, ph[4] // nb, sum(ph[:5]) // nb, nb, L.cj_debug_forwarded_chunks(1)), flush=True)
n, S = 512, 262144
for codec, name in ((N.CODEC_LZ4_BLOCK, "lz4"), (N.CODEC_SNAPPY_RAW, "snappy")):
    raws = [oracle.synth_v1(S, 1000 + i) for i in range(64)]
    comp = [np.frombuffer((oracle.lz4_compress_raw(r) if codec == N.CODEC_LZ4_BLOCK else oracle.snappy_compress(r))[1], np.uint8).copy() for r in raws]
    ins = [comp[i % 64] for i in range(n)]
    outs = [np.zeros(S, np.uint8) for _ in range(n)]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins]); in_lens = (C.c_size_t * n)(*[a.size for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs]); caps = (C.c_size_t * n)(*[S] * n)
    res = (C.c_int64 * n)()
    L.cj_debug_lds_phase_cycles((C.c_ulonglong * 16)(), 1)
    for rep in range(3):
        N.check(L.cj_batch_host(eng.h, codec, N.OP_DECOMPRESS, 0, n, in_ptrs, in_lens, out_ptrs, caps, res))
    assert all(r == S for r in res) and outs[7].tobytes() == raws[7]
    phases("%s slabs, 512 x 256 KiB host batch" % name)
    # independent 64 KiB chunks of the same data, same count of 64 KiB units (profile flag 0x1000 of a device batch)
    small = [oracle.synth_v1(65536, 3000 + i) for i in range(64)]
    sc = [(oracle.lz4_compress_raw(r) if codec == N.CODEC_LZ4_BLOCK else oracle.snappy_compress(r))[1] for r in small]
    res2, outs2 = eng.batch_host(codec, N.OP_DECOMPRESS, 0x1000 | N.FLAG_FORCE_LDS_PER_CHUNK, [sc[i % 64] for i in range(2048)], [65536] * 2048)
    assert all(int(r) == 65536 for r in res2)
    phases("%s independent chunks, 2048 x 64 KiB" % name)
big = b"".join(oracle.synth_v1(65536, i) for i in range(512))[:-777]
blob = oracle.lz4_compress_raw(big)[1]
L.cj_debug_lds_phase_cycles((C.c_ulonglong * 16)(), 1)
for _ in range(3): assert bytes(cj.lz4.decompress_block(blob, output_len=len(big))) == big
phases("lz4 slabs, one 32 MiB stream")
