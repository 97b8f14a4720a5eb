"""Mutation fuzz of the large-stream decoders (> 64 KiB: parallel parse + slab decode) against the oracle:
every mutated LZ4 block / Snappy raw stream must get the oracle's verdict, and on success the oracle's bytes.  GPU only.
  CASES=600 SEED=1 python tests/perf/fuzz_large.py"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cj

rnd = random.Random(1)


def shapes():
    parts = [oracle.synth_v1(65536, i) for i in range(8)]
    yield "synth", bytes(rnd.randrange(1, 999)) + b"".join(parts[i % 8] for i in range(rnd.randrange(2, 40)))
    yield "text", b"".join(b"%d bottles of beer on the wall, %d bottles\n" % (rnd.randrange(977), rnd.randrange(1013)) for _ in range(rnd.randrange(3000, 60000)))
    yield "runs", b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 70000) for _ in range(rnd.randrange(4, 40)))
    yield "mixed", b"".join(rnd.choice((rnd.randbytes(rnd.randrange(1, 50000)), bytes(rnd.randrange(1, 300000)), parts[rnd.randrange(8)][:rnd.randrange(1, 65536)])) for _ in range(rnd.randrange(3, 30)))


def mutate(b):
    b = bytearray(b)
    kind = rnd.randrange(9)
    i = rnd.randrange(len(b))
    if kind == 0: b[i] ^= 1 << rnd.randrange(8)
    elif kind == 1: b[i] = 0xFF
    elif kind == 2: b[i] = 0
    elif kind == 3: b = b[:rnd.randrange(65537, len(b))] if len(b) > 65538 else b
    elif kind == 4: b[i:i] = rnd.randbytes(rnd.randrange(1, 9))
    elif kind == 5: del b[i:i + rnd.randrange(1, 9)]
    elif kind == 6:                                         # a run of length bytes
        k = rnd.randrange(2, 400); b[i:i + k] = b"\xff" * min(k, len(b) - i)
    elif kind == 7:                                         # several flips far apart
        for _ in range(rnd.randrange(2, 6)): b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    else:                                                   # swap two 4-byte words (offsets / lengths move)
        j = rnd.randrange(len(b) - 4); i = min(i, len(b) - 4)
        b[i:i + 4], b[j:j + 4] = b[j:j + 4], b[i:i + 4]
    return bytes(b), kind


def run(cases, seed, verbose=False):
    """-> list of mismatches (empty = every verdict and every accepted byte equals the oracle's)"""
    rnd.seed(seed)
    t0 = time.time(); done = 0; ok_count = 0; bad = []
    while done < cases and not bad:
        for name, data in shapes():
            n = len(data)
            lz = oracle.lz4_compress_raw(data)[1]
            sn = oracle.snappy_compress(data)[1]
            for _ in range(6):
                m, kind = mutate(lz)
                cap = n if rnd.randrange(4) else n + rnd.randrange(1, 70000)
                er, eo = oracle.lz4_decompress_raw(m, cap)
                try:
                    got = bytes(cj.lz4.decompress_block(m, output_len=cap))
                    if not (er >= 0 and got[:er] == eo[:er] and got[er:] == bytes(cap - er)): bad.append(("lz4", name, kind, n, er, "accepted"))
                    ok_count += 1
                except cj.DecompressionError:
                    if er >= 0: bad.append(("lz4", name, kind, n, er, "rejected"))
                m, kind = mutate(sn)
                el = oracle.snappy_decompress_len(m)
                er, eo = oracle.snappy_decompress(m) if 0 <= el <= (1 << 28) else (-1, b"")
                try:
                    got = bytes(cj.snappy.decompress_raw(m))
                    if not (er >= 0 and got == eo[:er]): bad.append(("snappy", name, kind, n, er, "accepted"))
                    ok_count += 1
                except cj.DecompressionError:
                    if er >= 0: bad.append(("snappy", name, kind, n, er, "rejected"))
                done += 2
        if verbose: print("cases %d, accepted by both %d, %.0f s" % (done, ok_count, time.time() - t0), flush=True)
    return bad


def run_frames(cases, seed, verbose=False):
    """the same for the framed formats: LZ4 frames (linked / independent blocks of 64 KiB .. 4 MiB, block checksums, content
    size) and Snappy framing; only the verdict and the accepted bytes are compared (an error leaves no output behind)"""
    rnd.seed(seed)
    t0 = time.time(); done = 0; ok_count = 0; bad = []
    while done < cases and not bad:
        for name, data in shapes():
            n = len(data)
            bs, fl = rnd.randrange(4, 8), rnd.randrange(16)
            lz = oracle.lz4_frame_compress(data, bs, fl)[1]
            sn = oracle.snappy_frame_compress(data)[1]
            for _ in range(6):
                m, kind = mutate(lz)
                eb = oracle.lz4_frame_decompress_bound(m)
                er, eo = oracle.lz4_frame_decompress(m) if 0 <= eb <= (1 << 28) else (-1, b"")
                try:
                    got = bytes(cj.lz4.decompress(m))
                    if not (er >= 0 and got == eo[:er]): bad.append(("lz4f", name, kind, n, bs, fl, er, "accepted"))
                    ok_count += 1
                except cj.DecompressionError:
                    if er >= 0: bad.append(("lz4f", name, kind, n, bs, fl, er, "rejected"))
                m, kind = mutate(sn)
                el = oracle.snappy_frame_decompress_len(m)
                er, eo = oracle.snappy_frame_decompress(m) if 0 <= el <= (1 << 28) else (-1, b"")
                try:
                    got = bytes(cj.snappy.decompress(m))
                    if not (er >= 0 and got == eo[:er]): bad.append(("snappyf", name, kind, n, er, "accepted"))
                    ok_count += 1
                except cj.DecompressionError:
                    if er >= 0: bad.append(("snappyf", name, kind, n, er, "rejected"))
                done += 2
        if verbose: print("frames: cases %d, accepted by both %d, %.0f s" % (done, ok_count, time.time() - t0), flush=True)
    return bad


def run_compress(cases, seed, verbose=False):
    """the compressors on random sizes (clustered around the 4 KiB / 16 KiB sub-piece and 64 KiB piece boundaries, their multiples
    and the 8 KiB / 16 MiB switches of the sub-piece size): the oracle's decoders must give the input back from the block, raw
    and both framed outputs"""
    rnd.seed(seed)
    t0 = time.time(); done = 0; bad = []
    pool = b"".join(d for _, d in shapes())
    while done < cases and not bad:
        base = rnd.choice((0, 4096, 8192, 12288, 16384, 32768, 49152, 61440, 65536, 65536 + 4096, 131072, 196608, 262144, 1 << 20,
                           4096 * rnd.randrange(1, 300), rnd.randrange(1 << 22), (16 << 20) if rnd.randrange(8) == 0 else 65536 * rnd.randrange(1, 40)))
        n = max(0, base + rnd.randrange(-40, 41)) if rnd.randrange(3) else rnd.randrange(1 << rnd.randrange(1, 23))
        o = rnd.randrange(max(1, len(pool) - n)) if n < len(pool) else 0
        data = (pool[o:o + n] if rnd.randrange(5) else rnd.randbytes(n))[:n]
        n = len(data)
        b = bytes(cj.lz4.compress_block(data, store_size=False))
        if oracle.lz4_decompress_raw(b, n) != (n, data) and n: bad.append(("lz4 block", n))
        b = bytes(cj.snappy.compress_raw(data))
        if oracle.snappy_decompress(b) != (n, data): bad.append(("snappy raw", n))
        b = bytes(cj.lz4.compress(data))
        if oracle.lz4_frame_decompress(b) != (n, data): bad.append(("lz4 frame", n))
        b = bytes(cj.snappy.compress(data))
        if oracle.snappy_frame_decompress(b) != (n, data): bad.append(("snappy framed", n))
        done += 4
        if verbose and done % 400 == 0: print("compress: cases %d, %.0f s" % (done, time.time() - t0), flush=True)
    return bad


def run_batch(cases, seed, verbose=False):
    """batches of mutated chunks of 1 B .. 64 KiB through Engine.batch_host, default pipeline and each forced mapping: the
    oracle's verdict per chunk and its bytes where it accepts"""
    from cramjam_amd import _native as N
    rnd.seed(seed)
    t0 = time.time(); done = 0; bad = []
    pool = b"".join(d for _, d in shapes())
    eng = N.Engine(0)
    while done < cases and not bad:
        codec = rnd.choice((N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW))
        lz = codec == N.CODEC_LZ4_BLOCK
        streams, caps, want = [], [], []
        maxn = int(os.environ.get("MAXCHUNK", "65536"))          # 32768 / 16384: the batches take the small-window decoders (the host batch sets the flags itself)
        for _ in range(rnd.choice((1, 7, 300, 2500) if maxn == 65536 else (1, 300, 2500, 7000))):
            n = rnd.randrange(1, maxn + 1) if rnd.randrange(3) else maxn
            o = rnd.randrange(len(pool) - n)
            data = pool[o:o + n] if rnd.randrange(6) else rnd.randbytes(n)
            blob = (oracle.lz4_compress_raw if lz else oracle.snappy_compress)(data)[1]
            m = mutate(blob)[0] if rnd.randrange(8) and len(blob) > 8 else blob
            cap = n if rnd.randrange(5) else rnd.randrange(1, maxn + 1)
            streams.append(m); caps.append(cap)
            want.append(oracle.lz4_decompress_raw(m, cap) if lz else oracle.snappy_decompress(m, cap))
        flag = rnd.choice((0, 0, N.FLAG_FORCE_WAVE_PER_CHUNK, N.FLAG_FORCE_LANE_PER_CHUNK if lz else 0, N.FLAG_FORCE_LDS_PER_CHUNK, N.FLAG_FORCE_PARSE_KERNEL, N.FLAG_FORCE_FUSED_PARSE))
        res, outs = eng.batch_host(codec, N.OP_DECOMPRESS, flag, streams, caps)
        for i, ((er, eo), r, o) in enumerate(zip(want, res, outs)):
            if (er < 0) != (r < 0) or (er >= 0 and (r != er or o != eo)): bad.append(("lz4" if lz else "snappy", flag, i, len(streams), er, r)); break
        done += len(streams)
        if verbose: print("batch: cases %d, %.0f s" % (done, time.time() - t0), flush=True)
    eng.close()
    return bad


if __name__ == "__main__":
    if os.environ.get("BATCH"):
        bad = run_batch(int(os.environ.get("CASES", "600")), int(os.environ.get("SEED", "1")), verbose=True)
        print("mismatches:", bad)
        sys.exit(1 if bad else 0)
    if os.environ.get("COMPRESS"):
        bad = run_compress(int(os.environ.get("CASES", "600")), int(os.environ.get("SEED", "1")), verbose=True)
        print("mismatches:", bad)
        sys.exit(1 if bad else 0)
    if os.environ.get("FRAMES"):
        bad = run_frames(int(os.environ.get("CASES", "600")), int(os.environ.get("SEED", "1")), verbose=True)
        print("mismatches:", bad)
        sys.exit(1 if bad else 0)
    bad = run(int(os.environ.get("CASES", "600")), int(os.environ.get("SEED", "1")), verbose=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
