"""Round-trip time of ONE compress_block + decompress_block call per corpus file (BASELINE.md's table is this measurement on the
reference's CPU path): best of N, host bytes in, Buffer out.  GPU only."""
import bz2, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cramjam_amd as cj
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "corpus")
M = json.load(open(os.path.join(D, "manifest.json")))["files"]
def best(fn, reps=20):
    fn(); b = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t)
    return b * 1e6
print("%-28s %9s | %9s %9s %9s | %9s   (us; lz4 block compress, decompress, round trip | snappy raw round trip)" % ("file", "bytes", "c", "d", "c+d", "snappy"))
for name in sorted(M):
    raw = bz2.decompress(open(os.path.join(D, name + ".bz2"), "rb").read())
    c = bytes(cj.lz4.compress_block(raw)); s = bytes(cj.snappy.compress_raw(raw))
    tc = best(lambda: cj.lz4.compress_block(raw)); td = best(lambda: cj.lz4.decompress_block(c))
    trt = best(lambda: cj.lz4.decompress_block(cj.lz4.compress_block(raw)))
    ts = best(lambda: cj.snappy.decompress_raw(cj.snappy.compress_raw(raw)))
    print("%-28s %9d | %9.0f %9.0f %9.0f | %9.0f   ratio lz4 %.2f snappy %.2f" % (name, len(raw), tc, td, trt, ts, len(raw) / len(c), len(raw) / len(s)))
