#!/usr/bin/env python3
"""Fingerprint of the GPU encoders' output on a fixed corpus (sha256 over all compressed chunks) + throughput.
Used to check that a restructured matcher still emits byte-identical streams.  GPU only."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from cramjam_amd import _native as N

e = N.Engine(0)
corpus = [oracle.synth_v1(65536, i) for i in range(48)]
corpus += [bytes(65536), os.urandom(0) + bytes(range(256)) * 256, b"abc" * 21845, corpus[0][:1000], corpus[1][:13], b"x" * 12, b""]
import random
random.seed(4)
corpus += [bytes(random.randrange(4) for _ in range(30000)), bytes(random.randrange(256) for _ in range(50000))]
for codec, name in ((N.CODEC_LZ4_BLOCK, "lz4"), (N.CODEC_SNAPPY_RAW, "snappy")):
    caps = [len(c) + len(c) // 6 + 64 for c in corpus]
    res, outs = e.batch_host(codec, N.OP_COMPRESS, 0, corpus, caps)
    h = hashlib.sha256()
    for r, o in zip(res, outs):
        assert r > 0 or len(o) == 0, r
        h.update(bytes(o))
    print(name, "sha256", h.hexdigest()[:16], "total", sum(res))
