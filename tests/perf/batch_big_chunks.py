"""A host batch of chunks above 64 KiB through Engine.batch_host (decompress): they take the large-stream path together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from cramjam_amd import _native as N
parts = [oracle.synth_v1(65536, i) for i in range(64)]
e = N.Engine(0)
for size, count in ((256 << 10, 200), (1 << 20, 64), (8 << 20, 12)):
    chunks = [b"".join(parts[(i + k) % 64] for k in range(size >> 16)) for i in range(count)]
    for codec, comp in ((N.CODEC_LZ4_BLOCK, lambda c: oracle.lz4_compress_raw(c)[1]), (N.CODEC_SNAPPY_RAW, lambda c: oracle.snappy_compress(c)[1])):
        blobs = [comp(c) for c in chunks]
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); res, outs = e.batch_host(codec, N.OP_DECOMPRESS, 0, blobs, [size] * count); best = min(best, time.perf_counter() - t)
        assert all(int(r) == size for r in res) and all(bytes(o) == c for o, c in zip(outs, chunks))
        old = 1e9
        for _ in range(2):
            t = time.perf_counter(); e.batch_host(codec, N.OP_DECOMPRESS, N.FLAG_FORCE_WAVE_PER_CHUNK, blobs, [size] * count); old = min(old, time.perf_counter() - t)
        print("codec %d: %3d chunks of %5d KiB: %.2f GB/s (%.1f ms, host to host); one wavefront per chunk: %.2f GB/s (%.1f ms)" % (
            codec, count, size >> 10, size * count / best / 1e9, best * 1e3, size * count / old / 1e9, old * 1e3))
e.close()
