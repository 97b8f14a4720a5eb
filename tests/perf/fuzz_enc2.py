"""Randomised campaign for the encoders: inputs of random structure (literal noise, copies from random distances and lengths, byte runs,
text, periodic patterns, tails that end inside a match) and random sizes 0 .. 150 000 through the batch encoders of both codecs; every
stream must equal the scalar model's bytes (tests/hostsim/enc2_model.c) and decode with the oracle.  N=20000 python tests/perf/fuzz_enc2.py"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from cramjam_amd import _native as N
from test_enc2_model import model_lib, model_lz4, model_snappy

TEXT = (b"It was the best of times, it was the worst of times, it was the age of wisdom, it was the age of foolishness, it was the epoch of belief, "
        b"it was the epoch of incredulity, it was the season of Light, it was the season of Darkness, it was the spring of hope, it was the winter of despair. ")


def make(rnd):
    kind = rnd.randrange(8)
    n = rnd.choice((rnd.randrange(0, 64), rnd.randrange(0, 2000), rnd.randrange(0, 70000), rnd.randrange(60000, 70000), rnd.randrange(0, 150000)))
    out = bytearray()
    alpha = rnd.choice((2, 4, 16, 64, 256))
    while len(out) < n:
        r = rnd.random()
        if r < 0.35 or len(out) < 8:
            out += bytes(rnd.randrange(alpha) for _ in range(rnd.randrange(1, rnd.choice((4, 30, 300, 3000)))))
        elif r < 0.8:
            d = rnd.randrange(1, min(len(out), rnd.choice((8, 300, 70000))) + 1)
            m = rnd.randrange(4, rnd.choice((8, 40, 300, 5000)))
            for _ in range(m): out.append(out[-d])
        elif r < 0.88:
            out += bytes([rnd.randrange(256)]) * rnd.randrange(1, rnd.choice((10, 400, 9000)))
        else:
            k = rnd.randrange(len(TEXT)); out += TEXT[k:k + rnd.randrange(1, 200)]
        if kind == 7 and rnd.random() < 0.05: out += bytes(rnd.randrange(256) for _ in range(rnd.randrange(200, 70000)))
    return bytes(out[:n])


def main():
    total = int(os.environ.get("N", "4000"))
    rnd = random.Random(int(os.environ.get("SEED", "1")))
    M, L, e = model_lib(), N.lib(), N.Engine(0)
    bad = done = 0
    while done < total:
        raws = [make(rnd) for _ in range(min(500, total - done))]
        for codec in (N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW):
            caps = [L.cj_lz4_block_compress_bound(len(r), 0) if codec == 0 else L.cj_snappy_raw_max_compress_len(len(r)) for r in raws]
            res, outs = e.batch_host(codec, N.OP_COMPRESS, 0, raws, caps)
            for r, rr, o in zip(raws, res, outs):
                w = model_lz4(M, r) if codec == 0 else model_snappy(M, r)
                dr, d = oracle.lz4_decompress_raw(bytes(o), len(r)) if codec == 0 else oracle.snappy_decompress(bytes(o))
                if rr != len(w) or bytes(o) != w or dr != len(r) or d != r:
                    bad += 1
                    if bad <= 5: print("MISMATCH codec %d len %d result %d model %d decodes %s" % (codec, len(r), rr, len(w), dr == len(r) and d == r))
        done += len(raws)
    print("encoder fuzz: %d inputs x 2 codecs, %d mismatches" % (done, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
