#!/usr/bin/env python3
"""Host-boundary (PCIe-inclusive) rates of the framed single-buffer API on one MI355X.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

# where does the host-boundary time go?  _into with a pre-faulted output skips the result allocation + zero fill
import numpy as np
out = np.zeros(len(data), dtype=np.uint8)
for name, mod in (("snappy", cj.snappy), ("lz4", cj.lz4)):
    comp = bytes(mod.compress(data))
    cout = np.zeros(len(comp) + 1024, dtype=np.uint8)
    print("%-7s framed  compress_into %6.2f GB/s   decompress_into %6.2f GB/s   (pre-faulted outputs)" % (
        name, rate(lambda: mod.compress_into(data, cout), len(data)), rate(lambda: mod.decompress_into(comp, out), len(data))))
blk = bytes(cj.lz4.compress_block(data[:4 << 20], store_size=False))
o4 = np.zeros(4 << 20, dtype=np.uint8)
print("lz4 decompress_block_into of ONE 4 MiB block (single serial stream): %.3f GB/s" % rate(lambda: cj.lz4.decompress_block_into(blk, o4, output_len=4 << 20), 4 << 20))
