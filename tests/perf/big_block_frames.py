import os, sys, time
sys.path.insert(0, os.getcwd())
import oracle, cramjam_amd as cj
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = b"".join(parts[i % 64] for i in range(40 * 16))
for code, name in ((5, "256 KiB"), (6, "1 MiB"), (7, "4 MiB")):
    r, fr = oracle.lz4_frame_compress(data, code, 0)
    out = cj.lz4.decompress(fr); assert bytes(out) == data
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); cj.lz4.decompress(fr); best = min(best, time.perf_counter() - t)
    print("40 MiB frame, independent %s blocks: decompress %.2f GB/s (%.1f ms)" % (name, len(data) / best / 1e9, best * 1e3))
small = data[:10 << 20]
r, fr = oracle.lz4_frame_compress(small, 7, 0)
best = 1e9
for _ in range(3):
    t = time.perf_counter(); o = cj.lz4.decompress(fr); best = min(best, time.perf_counter() - t)
assert bytes(o) == small
print("10 MiB frame, 4 MiB blocks: %.2f GB/s (%.1f ms)" % (len(small) / best / 1e9, best * 1e3))
