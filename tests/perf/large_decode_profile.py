"""One large LZ4 block whose matches cross every 64 KiB slab boundary (synth chunks shifted by 1000 bytes), decompressed
through the single-buffer API.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
import cramjam_amd as cj
mb = int(os.environ.get("MB", "64"))
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = bytes(1000) + b"".join(parts[i % 64] for i in range(mb * 16))
n = len(data)
blob = oracle.lz4_compress_raw(data)[1]
sn = oracle.snappy_compress(data)[1]
for name, fn in (("lz4", lambda: cj.lz4.decompress_block(blob, output_len=n)), ("snappy", lambda: cj.snappy.decompress_raw(sn))):
    assert bytes(fn()) == data
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    print("%s: one %d MiB stream, slab-crossing matches: %.3f GB/s (%.2f ms)" % (name, mb, n / best / 1e9, best * 1e3))
