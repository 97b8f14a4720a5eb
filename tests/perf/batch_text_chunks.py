"""A batch of 64 KiB chunks of text-like data (log lines: deep match chains) through the batch decoder; run under
rocprofv3 --kernel-trace --stats to read the decoder kernel's time (match forwarding)."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from cramjam_amd import _native as N
rnd = random.Random(1)
def log_chunk():
    out = bytearray()
    while len(out) < 65536:
        out += b"2026-09-28T12:%02d:%02d.%03d INFO worker-%d request id=%08x path=/api/v1/items/%d status=%d latency_ms=%d\n" % (
            rnd.randrange(60), rnd.randrange(60), rnd.randrange(1000), rnd.randrange(16), rnd.getrandbits(32), rnd.randrange(5000),
            rnd.choice([200, 200, 200, 404, 500]), rnd.randrange(900))
    return bytes(out[:65536])
uniq = [log_chunk() for _ in range(64)]
n = int(os.environ.get("CHUNKS", "8192"))
for codec, comp in ((N.CODEC_LZ4_BLOCK, lambda c: oracle.lz4_compress_raw(c)[1]), (N.CODEC_SNAPPY_RAW, lambda c: oracle.snappy_compress(c)[1])):
    blobs = [comp(c) for c in uniq]
    ins = [blobs[i % 64] for i in range(n)]
    e = N.Engine(0)
    for _ in range(3):
        t = time.perf_counter(); res, outs = e.batch_host(codec, N.OP_DECOMPRESS, 0, ins, [65536] * n); dt = time.perf_counter() - t
    assert all(r == 65536 for r in res) and all(bytes(outs[i]) == uniq[i % 64] for i in range(0, n, 97))
    print("codec", codec, "ratio %.2f" % (65536 / (sum(map(len, blobs)) / 64)), "host-to-host %.1f ms for %d chunks" % (dt * 1e3, n))
    e.close()
