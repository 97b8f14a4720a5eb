#!/usr/bin/env python3
"""One decompress_block call on a stream a little above 64 KiB (the corpus files of configs[0] are 100-700 KB): where the
0.4-0.6 ms go.  Run under rocprofv3 --kernel-trace --stats; prints the wall time per call next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cj
n = int(sys.argv[1]) if len(sys.argv) > 1 else 152089
text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer; take one down\n" % (i * 7919 % 977, i * 104729 % 1013) for i in range(n // 40))[:n]
for name, data in (("text", text), ("synth", b"".join(oracle.synth_v1(65536, i) for i in range(n // 65536 + 1))[:n])):
    blob = oracle.lz4_compress_raw(data)[1]
    sn = oracle.snappy_compress(data)[1]
    for _ in range(3): cj.lz4.decompress_block(blob, output_len=n); cj.snappy.decompress_raw(sn)
    t0 = time.perf_counter()
    for _ in range(50): out = cj.lz4.decompress_block(blob, output_len=n)
    t1 = time.perf_counter()
    for _ in range(50): out2 = cj.snappy.decompress_raw(sn)
    t2 = time.perf_counter()
    assert bytes(out) == data and bytes(out2) == data
    print("%s %d B: lz4 decompress_block %.0f us per call, snappy decompress_raw %.0f us" % (name, n, (t1 - t0) / 50 * 1e6, (t2 - t1) / 50 * 1e6), flush=True)
