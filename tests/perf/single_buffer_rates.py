import os, sys, time
sys.path.insert(0, os.getcwd())
import oracle
import cramjam_amd as cj
import numpy as np
def rate(fn, nbytes, reps=2):
    fn(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    return nbytes / best / 1e9
mb = int(os.environ.get("MB", "16"))
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = b"".join(parts[i % 64] for i in range(mb * 16))
n = len(data)
blk = bytes(cj.lz4.compress_block(data, store_size=False))
print("lz4 compress_block   one %d MiB block: %.3f GB/s ratio %.3f" % (mb, rate(lambda: cj.lz4.compress_block(data, store_size=False), n), n / len(blk)))
print("lz4 decompress_block one %d MiB block: %.3f GB/s" % (mb, rate(lambda: cj.lz4.decompress_block(blk, output_len=n), n)))
raw = bytes(cj.snappy.compress_raw(data))
print("snappy compress_raw  one %d MiB buffer: %.3f GB/s ratio %.3f" % (mb, rate(lambda: cj.snappy.compress_raw(data), n), n / len(raw)))
print("snappy decompress_raw one %d MiB buffer: %.3f GB/s" % (mb, rate(lambda: cj.snappy.decompress_raw(raw), n)))
t = time.perf_counter(); r, ob = oracle.lz4_compress_raw(data); t1 = time.perf_counter() - t
print("lz4 decompress_block of the ORACLE's (liblz4-style, matches cross every 64 KiB boundary) %d MiB block: %.3f GB/s" % (mb, rate(lambda: cj.lz4.decompress_block(ob, output_len=n), n)))
osn = oracle.snappy_compress(data)[1]
print("snappy decompress_raw of the oracle's stream: %.3f GB/s" % rate(lambda: cj.snappy.decompress_raw(osn), n))
t = time.perf_counter(); oracle.lz4_decompress_raw(ob, n); t2 = time.perf_counter() - t
print("oracle (1 core) lz4 compress %.3f GB/s decompress %.3f GB/s" % (n / t1 / 1e9, n / t2 / 1e9))
