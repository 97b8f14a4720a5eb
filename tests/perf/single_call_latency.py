"""Latency of ONE call of the four raw entry points by buffer size (host bytes in, Buffer out).  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cj
def lat(fn, reps=20):
    fn(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    return best * 1e3
text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer\n" % (i * 7 % 977, i * 13 % 1013) for i in range(90000))
synth = b"".join(oracle.synth_v1(65536, i) for i in range(64))
print("%-8s %9s | %8s %8s %8s %8s   (ms per call; lz4 c/d, snappy c/d)" % ("data", "bytes", "lz4 c", "lz4 d", "sn c", "sn d"))
for name, src in (("synth", synth), ("text", text)):
    for n in (1 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20):
        d = src[:n]
        lz = bytes(cj.lz4.compress_block(d)); sn = bytes(cj.snappy.compress_raw(d))
        print("%-8s %9d | %8.3f %8.3f %8.3f %8.3f   ratio %.2f / %.2f" % (name, n, lat(lambda: cj.lz4.compress_block(d)), lat(lambda: cj.lz4.decompress_block(lz)),
              lat(lambda: cj.snappy.compress_raw(d)), lat(lambda: cj.snappy.decompress_raw(sn)), n / len(lz), n / len(sn)))
