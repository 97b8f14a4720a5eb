"""Mutation fuzz of the 64 - 256 KiB chunk path of a device batch (CJ_FLAG_BIG_CHUNKS: 32-lane walk + slab decoder) against the oracle:
hand-made Snappy streams of every element form (tests/test_big_chunks_gpu.py: _sn_stream), the oracle encoders' streams of synth / text /
run data for both codecs, each intact and mutated; every verdict and every accepted byte must equal the oracle's.  GPU only.
  CASES=4000 SEED=1 python tests/perf/fuzz_big_chunks.py"""
import os, sys, random, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import importlib.util
import numpy as np
import oracle
from cramjam_amd import _native as N
spec = importlib.util.spec_from_file_location("tb", os.path.join(R, "tests", "test_big_chunks_gpu.py")); tb = importlib.util.module_from_spec(spec); spec.loader.exec_module(tb)

cases = int(os.environ.get("CASES", "2000")); seed = int(os.environ.get("SEED", "1"))
rnd = random.Random(seed)
eng = N.Engine(0)
t0 = time.time(); bad = []; done = 0; accepted = 0
def raw(n):
    k = rnd.randrange(4)
    if k == 0: return oracle.synth_v1(n, rnd.randrange(1000))
    if k == 1: return tb._text(n, rnd.randrange(1000))
    if k == 2: return tb._mixed(n, rnd.randrange(1000))
    return b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 3000) for _ in range(400))[:n]
def mutate(b):
    b = bytearray(b); k = rnd.randrange(6); i = rnd.randrange(len(b))
    if k == 0: b[i] ^= 1 << rnd.randrange(8)
    elif k == 1: b[i] = rnd.choice((0, 0xff, 0x03, 0xf0, 0xfc))
    elif k == 2: b = b[:rnd.randrange(1, len(b))]
    elif k == 3: b[i:i] = rnd.randbytes(rnd.randrange(1, 6))
    elif k == 4: del b[i:i + rnd.randrange(1, 6)]
    else:
        for _ in range(rnd.randrange(2, 5)): b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    return bytes(b)
while done < cases:
    for codec in (tb.LZ4, tb.SN):
        blobs = []
        for _ in range(24):
            n = rnd.choice((70000, 100000, 131072, 200000, 262144))
            if codec == tb.SN and rnd.randrange(2): b = tb._sn_stream(rnd, n, rnd.choice(((1, 3, 3, 3), (3, 1, 1, 3), (1, 0, 1, 6), (2, 4, 4, 0))))
            else: b = (oracle.lz4_compress_raw(raw(n))[1] if codec == tb.LZ4 else oracle.snappy_compress(raw(n))[1])
            blobs.append(b)
            for _ in range(3): blobs.append(mutate(b))
        caps = [min(max(oracle.snappy_decompress_len(d), 0), 1 << 19) for d in blobs] if codec == tb.SN else [rnd.choice((262144, 262144, 131072, 300000)) for _ in blobs]
        res, out, off = tb._run(eng, codec, blobs, caps, N.FLAG_BIG_CHUNKS)
        for k, (d, cap) in enumerate(zip(blobs, caps)):
            er, eo = oracle.lz4_decompress_raw(d, cap) if codec == tb.LZ4 else oracle.snappy_decompress(d, cap)
            if codec == tb.LZ4 and er < 0:
                if res[k] >= 0: bad.append((codec, done + k, "accepted", int(res[k]), er))
            elif res[k] != er: bad.append((codec, done + k, int(res[k]), er))
            elif er >= 0:
                accepted += 1
                if out[int(off[k]):int(off[k]) + er].tobytes() != eo: bad.append((codec, done + k, "bytes"))
        done += len(blobs)
print("big chunks: cases %d, accepted by both %d, %d s" % (done, accepted, time.time() - t0))
print("mismatches:", bad[:10])
