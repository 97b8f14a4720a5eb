#!/usr/bin/env python3
"""Host-boundary (PCIe-inclusive) rates of the framed single-buffer API on one MI355X.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
import cramjam_amd as cj

def rate(fn, nbytes, reps=3):
    fn(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    return nbytes / best / 1e9

mb = int(os.environ.get("MB", "64"))
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = b"".join(parts[i % 64] for i in range(mb * 16))
for name, mod in (("snappy", cj.snappy), ("lz4", cj.lz4)):
    comp = bytes(mod.compress(data))
    print("%-7s framed  compress %6.2f GB/s   decompress %6.2f GB/s   (ratio %.2f, %d MiB)" % (
        name, rate(lambda: mod.compress(data), len(data)), rate(lambda: mod.decompress(comp), len(data)), len(data) / len(comp), mb))
r, linked = oracle.lz4_frame_compress(data, 4, 1)
print("lz4 frame with LINKED 64 KiB blocks (what the reference's encoder emits): decompress %.3f GB/s" % rate(lambda: cj.lz4.decompress(linked), len(data)))
