"""Batches of small chunks on windows of their own size (CJ_FLAG_CHUNKS_LE_32K / _16K: the workgroup decoder with four workgroups of four
wavefronts / eight of two per CU instead of two of eight, profiles/r06/experiments h01-h04) against the oracle and against the 64 KiB
window: the same results for every chunk — valid, damaged, and chunks that break the promise (they take one wavefront) — with the parse
as its own kernel in front of the decoder and inside it (the one-kernel path on the window's 256 / 128 lanes, f05).
Reference behaviour: one call of /root/reference/src/lz4.rs:78-95 (decompress_block) / src/snappy.rs:52-60 (decompress_raw) per chunk."""
import random

import numpy as np
import pytest

import oracle
from cramjam_amd import _native as N

pytestmark = pytest.mark.gpu
LZ4, SN, DEC = N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS


def _text(n, seed):
    r = random.Random(seed); out = bytearray()
    while len(out) < n: out += b"%d bottles of beer on the wall, %d bottles of beer\n" % (r.randrange(977), r.randrange(1013))
    return bytes(out[:n])


def _run(eng, codec, blobs, caps, flags):
    n = len(blobs)
    in_len = np.array([len(b) for b in blobs], np.uint64)
    in_off = np.concatenate([[0], np.cumsum(in_len + 3)[:-1]]).astype(np.uint64)           # chunks at odd alignments
    packed = np.zeros(int(in_off[-1] + in_len[-1]) + 64, np.uint8)
    for k, b in enumerate(blobs): packed[int(in_off[k]):int(in_off[k]) + len(b)] = np.frombuffer(b, np.uint8)
    out_cap = np.array(caps, np.uint64); out_off = np.concatenate([[0], np.cumsum(out_cap + 5)[:-1]]).astype(np.uint64)
    total = int(out_off[-1] + out_cap[-1]) + 64
    d_in = eng.alloc(packed.nbytes); d_out = eng.alloc(total); d_meta = eng.alloc(5 * n * 8)
    eng.h2d(d_in, packed); eng.h2d(d_meta, np.concatenate([in_off, in_len, out_off, out_cap])); eng.h2d(d_out, np.full(total, 0xAB, np.uint8))
    eng.batch_device(codec, DEC, flags, n, d_in, d_meta, d_meta + 8 * n, d_out, d_meta + 16 * n, d_meta + 24 * n, d_meta + 32 * n)
    eng.sync()
    res = eng.d2h(d_meta + 32 * n, 8 * n, "int64"); out = eng.d2h(d_out, total)
    for p in (d_in, d_out, d_meta): eng.free(p)
    return res, out, out_off


@pytest.fixture(scope="module")
def eng():
    e = N.Engine(0)
    yield e
    e.close()


def _raws(win):
    rnd = random.Random(win)
    r = [oracle.synth_v1(win, 100 + i) for i in range(40)] + [oracle.synth_v1(n, 7 + n) for n in (win - 1, win // 2, win // 3, 4096, 900, 13, 1)]
    r += [_text(win, 1), _text(win - 7, 2), _text(5000, 3), bytes(win), rnd.randbytes(win), rnd.randbytes(win // 2) + bytes(win // 2), b"ab" * (win // 2),
          (rnd.randbytes(300) * 200)[:win], b"", b"x"]
    # chunks that break the promise: they must decode all the same (one wavefront each)
    r += [oracle.synth_v1(win + 1, 5), oracle.synth_v1(65536, 6), _text(40000 if win < 32768 else 50000, 7), oracle.synth_v1(100000, 8)]
    return r


PLACE = [0, N.FLAG_FORCE_PARSE_KERNEL, N.FLAG_FORCE_FUSED_PARSE]      # the engine's choice (one kernel up to 16 384 chunks), parse kernel, one kernel


@pytest.mark.parametrize("place", PLACE)
@pytest.mark.parametrize("win,flag", [(32768, N.FLAG_CHUNKS_LE_32K), (16384, N.FLAG_CHUNKS_LE_16K)])
@pytest.mark.parametrize("codec", [LZ4, SN])
def test_small_windows_against_the_oracle(eng, codec, win, flag, place):
    comp = (lambda r: oracle.lz4_compress_raw(r)[1]) if codec == LZ4 else (lambda r: oracle.snappy_compress(r)[1])
    uniq = _raws(win)
    blobs_u = [comp(r) for r in uniq]
    n = 7000
    idx = [i % len(uniq) for i in range(n)]
    blobs = [blobs_u[i] for i in idx]; want = [uniq[i] for i in idx]
    res, out, off = _run(eng, codec, blobs, [len(r) for r in want], flag | place)
    for i, r in enumerate(want):
        if codec == LZ4 and len(r) == 0:
            assert res[i] == 0
            continue
        assert res[i] == len(r), (codec, win, i, int(res[i]), len(r))
        assert out[int(off[i]):int(off[i]) + len(r)].tobytes() == r, (codec, win, i)
        assert (out[int(off[i]) + len(r):int(off[i]) + len(r) + 5] == 0xAB).all(), (codec, win, i)      # nothing past the capacity


@pytest.mark.parametrize("place", PLACE[1:])
@pytest.mark.parametrize("win,flag", [(32768, N.FLAG_CHUNKS_LE_32K), (16384, N.FLAG_CHUNKS_LE_16K)])
@pytest.mark.parametrize("codec", [LZ4, SN])
def test_damaged_streams_get_the_same_verdict_on_every_window(eng, codec, win, flag, place):
    comp = (lambda r: oracle.lz4_compress_raw(r)[1]) if codec == LZ4 else (lambda r: oracle.snappy_compress(r)[1])
    rnd = random.Random(99 + win)
    base = [oracle.synth_v1(win, 300 + i) for i in range(8)] + [_text(win, 9), _text(win // 2, 10)]
    blobs, caps = [], []
    for k in range(7000):
        raw = base[k % len(base)]
        b = bytearray(comp(raw))
        kind = k % 5
        if kind == 1: b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 2: b = b[:rnd.randrange(1, len(b))]
        elif kind == 3: b += bytes([rnd.randrange(256)])
        elif kind == 4:
            p = rnd.randrange(len(b)); b[p:p + 2] = bytes([rnd.randrange(256), rnd.randrange(256)])
        blobs.append(bytes(b)); caps.append(len(raw))
    ra, oa, off = _run(eng, codec, blobs, caps, N.FLAG_FORCE_PARSE_KERNEL)          # the 64 KiB window
    rb, ob, _ = _run(eng, codec, blobs, caps, flag | place)
    assert (ra == rb).all(), [(i, int(ra[i]), int(rb[i])) for i in np.nonzero(ra != rb)[0][:8]]
    for i in np.nonzero(ra > 0)[0]:
        assert (oa[int(off[i]):int(off[i]) + int(ra[i])] == ob[int(off[i]):int(off[i]) + int(ra[i])]).all(), i
    # and the oracle's verdict for a sample
    for i in range(0, len(blobs), 97):
        r, d = (oracle.lz4_decompress_raw(blobs[i], caps[i]) if codec == LZ4 else oracle.snappy_decompress(blobs[i], caps[i]))
        assert (r > 0) == (rb[i] > 0), (i, r, int(rb[i]))
        if r > 0: assert ob[int(off[i]):int(off[i]) + r].tobytes() == d, i


def test_host_batches_pick_the_window_themselves(eng):
    raws = [oracle.synth_v1(16384, 900 + i) for i in range(64)] + [_text(9000, 4), b"", b"q" * 777]
    blobs = [oracle.lz4_compress_raw(r)[1] for r in raws] * 50
    want = raws * 50
    res, outs = eng.batch_host(LZ4, DEC, 0, blobs, [len(r) for r in want])
    for r, o, w in zip(res, outs, want):
        assert int(r) == len(w) and bytes(o)[:len(w)] == w
