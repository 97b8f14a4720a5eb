"""Latency of ONE framed compress / decompress call by size (host bytes in, Buffer out).  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cj
def lat(fn, reps=15):
    fn(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    return best * 1e3
text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer\n" % (i * 7 % 977, i * 13 % 1013) for i in range(400000))
synth = b"".join(oracle.synth_v1(65536, i) for i in range(256))
print("%-8s %9s | %8s %8s %8s %8s   (ms per call)" % ("data", "bytes", "lz4 c", "lz4 d", "sn c", "sn d"))
for name, src in (("synth", synth), ("text", text)):
    for n in (64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20):
        d = src[:n]
        lz = bytes(cj.lz4.compress(d)); sn = bytes(cj.snappy.compress(d))
        print("%-8s %9d | %8.3f %8.3f %8.3f %8.3f   ratio %.2f / %.2f" % (name, n, lat(lambda: cj.lz4.compress(d)), lat(lambda: cj.lz4.decompress(lz)),
              lat(lambda: cj.snappy.compress(d)), lat(lambda: cj.snappy.decompress(sn)), n / len(lz), n / len(sn)))
