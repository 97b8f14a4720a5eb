"""The segmented parse kernel (csrc/seg_parse.hip: k lanes per chunk, records written by the parse) + the workgroup decoder's
record mode, against the oracle: valid chunks of every shape decode bit-exactly, damaged chunks get the oracle's verdict and
bytes, for every number of lanes per chunk.  Reference behaviour: /root/reference/src/lz4.rs:78-95 (decompress_block),
src/snappy.rs:52-60 (decompress_raw): one call = one chunk; a batch is many of them at once."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_CHECK = r"""
import random, sys
import oracle
from cramjam_amd import _native as N
eng = N.Engine(0)
rnd = random.Random(41)
def text(n, seed):
    r = random.Random(seed); out = bytearray()
    while len(out) < n: out += b"%d bottles of beer on the wall, %d bottles of beer\n" % (r.randrange(977), r.randrange(1013))
    return bytes(out[:n])
def mixed(n, seed):            # short sequences with a few long literal runs (a run longer than the parse's lead-in crosses segment boundaries)
    r = random.Random(seed); out = bytearray()
    while len(out) < n:
        out += oracle.synth_v1(r.randrange(2000, 9000), r.randrange(1 << 20))
        out += r.randbytes(r.choice((40, 700, 1500, 3000, 9000)))
    return bytes(out[:n])
chunks = [oracle.synth_v1(65536, i) for i in range(24)] + [oracle.synth_v1(n, 100 + n) for n in (65535, 40000, 20011, 9000, 3000, 700)]
chunks += [text(65536, 1), text(30000, 2), text(65536, 3), bytes(50000), rnd.randbytes(65536), rnd.randbytes(5000)]
chunks += [mixed(65536, s) for s in range(8)] + [mixed(33333, 99)]
chunks += [b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 300) for _ in range(600))[:65536], (rnd.randbytes(3000) * 22)[:65536],
           b"".join((b"ab" * rnd.randrange(2, 40) + b"xyz" * rnd.randrange(2, 30) + rnd.randbytes(rnd.randrange(1, 9))) for _ in range(900))[:61000],
           b"", b"a", b"howdy neighbor"]
codecs = ((N.CODEC_LZ4_BLOCK, lambda c: oracle.lz4_compress_raw(c)[1], lambda b, cap: oracle.lz4_decompress_raw(b, cap)),
          (N.CODEC_SNAPPY_RAW, lambda c: oracle.snappy_compress(c)[1], lambda b, cap: oracle.snappy_decompress(b, cap)))
reps = int(sys.argv[1])
for codec, comp, dec in codecs:
    blobs = [comp(c) for c in chunks]
    # valid chunks, every workgroup takes several
    res, outs = eng.batch_host(codec, N.OP_DECOMPRESS, 0, blobs * reps, [len(c) for c in chunks] * reps)
    want = chunks * reps
    assert [int(r) for r in res] == [len(c) for c in want], (codec, [(i, int(r), len(c)) for i, (r, c) in enumerate(zip(res, want)) if int(r) != len(c)][:5])
    bad = [i for i, (o, c) in enumerate(zip(outs, want)) if bytes(o) != c]
    assert not bad, (codec, bad[:10])
    # damaged chunks: the oracle's verdict, and its bytes where it accepts
    dam, caps = [], []
    for t in range(400):
        i = rnd.randrange(len(chunks) - 3)
        b = bytearray(blobs[i])
        if not b: continue
        kind = rnd.randrange(4)
        if kind == 0: b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 1: b = b[:rnd.randrange(len(b))]
        elif kind == 2:
            p = rnd.randrange(len(b)); b[p:p + 2] = rnd.randbytes(2)
        else: b += rnd.randbytes(rnd.randrange(1, 5))
        dam.append(bytes(b)); caps.append(len(chunks[i]) if rnd.randrange(3) else max(len(chunks[i]) - rnd.randrange(1, 50), 0))
    if codec == N.CODEC_SNAPPY_RAW:
        caps = [min(max(oracle.snappy_decompress_len(d), 0), 1 << 17) if d else 0 for d in dam]
    res, outs = eng.batch_host(codec, N.OP_DECOMPRESS, 0, dam, caps)
    for d, cap, r, o in zip(dam, caps, res, outs):
        er, eo = dec(d, cap)
        if codec == N.CODEC_LZ4_BLOCK and er < 0: assert int(r) < 0, (len(d), cap, int(r), er)
        else:
            assert int(r) == er, (codec, len(d), cap, int(r), er)
            if er >= 0: assert bytes(o) == eo, (codec, len(d), cap)
print("seg parse ok")
"""


@pytest.mark.parametrize("klog", ["0", "1", "2", "3"])
def test_segmented_parse_and_record_mode_against_the_oracle(klog):
    env = dict(os.environ, CJ_FUSED="0", CJ_SEG_KLOG=klog, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", _CHECK, "60"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "seg parse ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
