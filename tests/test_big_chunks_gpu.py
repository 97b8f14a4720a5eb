"""Chunks of 64 KiB .. 256 KiB in a device batch (BASELINE configs[4]: 256 KiB chunks, both codecs) through the big-chunk path —
CJ_FLAG_BIG_CHUNKS: 32-lane segmented parse into records (csrc/big_chunks.hip) + the slab mode of the workgroup decoder fed
with them — against the oracle: valid chunks of every shape bit-exactly, damaged chunks with the oracle's verdict and bytes,
small and big chunks mixed in one batch.  Reference behaviour: one call of /root/reference/src/lz4.rs:78-95 (decompress_block) /
src/snappy.rs:52-60 (decompress_raw) per chunk; a batch is many of them at once."""
import random

import numpy as np
import pytest

import oracle
from cramjam_amd import _native as N

pytestmark = pytest.mark.gpu
LZ4, SN, DEC = N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS
S = 262144


def _text(n, seed):
    r = random.Random(seed); out = bytearray()
    while len(out) < n: out += b"%d bottles of beer on the wall, %d bottles of beer\n" % (r.randrange(977), r.randrange(1013))
    return bytes(out[:n])


def _mixed(n, seed):        # short sequences with long literal runs in between (runs longer than a parse segment included)
    r = random.Random(seed); out = bytearray()
    while len(out) < n:
        out += oracle.synth_v1(r.randrange(2000, 30000), r.randrange(1 << 20)); out += r.randbytes(r.choice((40, 700, 1500, 3000, 9000, 70000)))
    return bytes(out[:n])


def _run(eng, codec, blobs, caps, flags):
    n = len(blobs)
    in_len = np.array([len(b) for b in blobs], np.uint64)
    in_off = np.concatenate([[0], np.cumsum(in_len + 7)[:-1]]).astype(np.uint64)           # chunks at odd alignments
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


@pytest.fixture(scope="module")
def raws():
    rnd = random.Random(5)
    r = [oracle.synth_v1(S, 100 + i) for i in range(10)] + [oracle.synth_v1(n, 7 + n) for n in (262143, 200000, 131072, 131073, 70000, 65537, 65536, 30000, 900)]
    r += [bytes(S), rnd.randbytes(S), _text(S, 1), _text(150000, 2), _mixed(S, 3), _mixed(S, 4), _mixed(180000, 5), (rnd.randbytes(3000) * 90)[:S],
          b"ab" * 100000, bytes(100000) + rnd.randbytes(100000), b"", b"x"]
    return r


@pytest.mark.parametrize("codec", [LZ4, SN])
def test_big_and_small_chunks_in_one_batch_against_the_oracle(eng, raws, codec):
    comp = (lambda r: oracle.lz4_compress_raw(r)[1]) if codec == LZ4 else (lambda r: oracle.snappy_compress(r)[1])
    blobs = [comp(r) for r in raws] * 5                       # every workgroup takes several slabs
    want = raws * 5
    for flags in (N.FLAG_BIG_CHUNKS, 0):                      # with the big-chunk path, and without it (one wavefront per big chunk)
        res, out, off = _run(eng, codec, blobs, [len(r) for r in want], flags)
        for i, r in enumerate(want):
            if codec == LZ4 and len(r) == 0:
                assert res[i] == 0
                continue
            assert res[i] == len(r), (codec, flags, i, int(res[i]), len(r))
            assert out[int(off[i]):int(off[i]) + len(r)].tobytes() == r, (codec, flags, i)
            assert (out[int(off[i]) + len(r):int(off[i]) + len(r) + 5] == 0xAB).all(), (codec, flags, i)      # nothing past the capacity


def test_lz4_capacity_above_the_decoded_size_and_size_prefix(eng, raws):
    """output_len is a CAPACITY for LZ4 raw blocks (the reference passes it to LZ4_decompress_safe, src/lz4.rs:88): a big capacity over a
    smaller block, and the u32 size prefix of store_size=True in front of big blocks"""
    pick = [r for r in raws if len(r) > 0][:14]
    blobs = [oracle.lz4_compress_raw(r)[1] for r in pick]
    res, out, off = _run(eng, LZ4, blobs, [S] * len(pick), N.FLAG_BIG_CHUNKS)
    for i, r in enumerate(pick):
        assert res[i] == len(r) and out[int(off[i]):int(off[i]) + len(r)].tobytes() == r, (i, int(res[i]), len(r))
    pref = [len(r).to_bytes(4, "little") + b for r, b in zip(pick, blobs)]
    res, out, off = _run(eng, LZ4, pref, [S + 9] * len(pick), N.FLAG_BIG_CHUNKS | N.FLAG_LZ4_SIZE_PREFIX)
    for i, r in enumerate(pick):
        assert res[i] == len(r) and out[int(off[i]):int(off[i]) + len(r)].tobytes() == r, (i, int(res[i]), len(r))


@pytest.mark.parametrize("codec", [LZ4, SN])
def test_damaged_big_chunks_get_the_oracles_verdict(eng, raws, codec):
    rnd = random.Random(17 + codec)
    big = [r for r in raws if len(r) > 65536]
    comp = (lambda r: oracle.lz4_compress_raw(r)[1]) if codec == LZ4 else (lambda r: oracle.snappy_compress(r)[1])
    blobs = [comp(r) for r in big]
    dam, caps = [], []
    for t in range(160):
        i = rnd.randrange(len(big))
        b = bytearray(blobs[i])
        kind = rnd.randrange(4)
        if kind == 0: b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 1: b = b[:rnd.randrange(1, len(b))]
        elif kind == 2:
            p = rnd.randrange(len(b)); b[p:p + 2] = rnd.randbytes(2)
        else: b += rnd.randbytes(rnd.randrange(1, 5))
        dam.append(bytes(b)); caps.append(len(big[i]) if rnd.randrange(3) else len(big[i]) - rnd.randrange(1, 50))
    if codec == SN:
        caps = [min(max(oracle.snappy_decompress_len(d), 0), 1 << 19) for d in dam]
    res, out, off = _run(eng, codec, dam, caps, N.FLAG_BIG_CHUNKS)
    for k, (d, cap) in enumerate(zip(dam, caps)):
        er, eo = oracle.lz4_decompress_raw(d, cap) if codec == LZ4 else oracle.snappy_decompress(d, cap)
        if codec == LZ4 and er < 0:
            assert res[k] < 0, (k, len(d), cap, int(res[k]), er)
        else:
            assert res[k] == er, (codec, k, len(d), cap, int(res[k]), er)
            if er >= 0:
                assert out[int(off[k]):int(off[k]) + er].tobytes() == eo, (codec, k)


def test_snappy_copy_that_reaches_beyond_the_previous_slab(eng):
    """a Snappy copy-4 element may reach further back than 65 535 bytes (the encoders never emit one, decoders must accept it): such
    a chunk is left to the one-wavefront kernel and decodes like any other"""
    rnd = random.Random(3)
    head = rnd.randbytes(200000)
    raw = head + head[1000:1064] + rnd.randbytes(500)                    # 64 bytes copied from 199 000 bytes back

    def lit(b):
        out = bytearray()
        for i in range(0, len(b), 60):
            p = b[i:i + 60]; out += bytes([(len(p) - 1) << 2]) + p
        return bytes(out)

    def varint(n):
        out = bytearray()
        while n >= 0x80: out.append((n & 0x7f) | 0x80); n >>= 7
        out.append(n)
        return bytes(out)

    stream = varint(len(raw)) + lit(head) + bytes([((64 - 1) << 2) | 3]) + (199000).to_bytes(4, "little") + lit(raw[200064:])
    assert oracle.snappy_decompress(stream) == (len(raw), raw)
    res, out, off = _run(eng, SN, [stream] * 3, [len(raw)] * 3, N.FLAG_BIG_CHUNKS)
    for i in range(3):
        assert res[i] == len(raw) and out[int(off[i]):int(off[i]) + len(raw)].tobytes() == raw


def _sn_varint(n):
    out = bytearray()
    while n >= 0x80: out.append((n & 0x7f) | 0x80); n >>= 7
    out.append(n)
    return bytes(out)


def test_snappy_slab_with_more_records_than_the_slab_tables_hold(eng):
    """Valid Snappy streams of tiny elements: a 64 KiB slab of OUTPUT may hold far more than the 16 384 records an LZ4 slab can
    (min match 4) — the slab decoder's tables are sized for those, so such a chunk has to stay with the one-wavefront kernel
    (round-4 advisor: before the check its cross list ran into the next workgroup's).  Shapes: copies of 3 bytes throughout (21 845
    records per slab, every parse lane inside its region limit), copies of 1 byte behind a long literal, literals of 1 byte."""
    rnd = random.Random(11)
    streams, raws = [], []
    seed = rnd.randbytes(16)
    for k, (clen, off) in enumerate(((3, 3), (3, 16), (2, 5), (1, 7))):
        cnt = 64000 if clen == 3 else 52000
        raw = bytearray(seed)
        body = bytearray([(len(seed) - 1) << 2]) + seed
        for _ in range(cnt):
            body += bytes([((clen - 1) << 2) | 2]) + off.to_bytes(2, "little")
            for _ in range(clen): raw.append(raw[-off])
        streams.append(_sn_varint(len(raw)) + bytes(body)); raws.append(bytes(raw))
    # one-byte copies with offset 65 535 behind 70 000 literal bytes (the advisor's example), and one-byte literals
    head = rnd.randbytes(70000)
    raw = bytearray(head); body = bytearray([62 << 2]) + (len(head) - 1).to_bytes(3, "little") + head
    for _ in range(52000):
        body += bytes([(0 << 2) | 2]) + (65535).to_bytes(2, "little"); raw.append(raw[-65535])
    streams.append(_sn_varint(len(raw)) + bytes(body)); raws.append(bytes(raw))
    tiny = rnd.randbytes(90000)
    streams.append(_sn_varint(len(tiny)) + b"".join(bytes([0, b]) for b in tiny)); raws.append(tiny)
    for s_, r in zip(streams, raws):
        assert oracle.snappy_decompress(s_) == (len(r), r)
    normal = oracle.synth_v1(S, 321)
    blobs = (streams + [oracle.snappy_compress(normal)[1]]) * 6          # between ordinary big chunks: a corrupted neighbour would show
    want = (raws + [normal]) * 6
    res, out, off = _run(eng, SN, blobs, [len(r) for r in want], N.FLAG_BIG_CHUNKS)
    for i, r in enumerate(want):
        assert res[i] == len(r), (i, int(res[i]), len(r))
        assert out[int(off[i]):int(off[i]) + len(r)].tobytes() == r, i


@pytest.mark.parametrize("codec", [LZ4, SN])
def test_more_big_chunks_in_one_call_than_one_group_holds(eng, codec):
    """The engine gives the big chunks of a call record areas in groups of 8 192 (engine.hip: kBigCap); a call with more of them
    (round 4: the rest silently fell to one wavefront each) runs group after group — every chunk against the oracle's bytes."""
    uniq = [oracle.synth_v1(n, 40 + n) for n in (70000, 66000, 90000, 131072, 262144, 65537)] + [_text(80000, 9)]
    comp = (lambda r: oracle.lz4_compress_raw(r)[1]) if codec == LZ4 else (lambda r: oracle.snappy_compress(r)[1])
    ub = [comp(r) for r in uniq]
    small = oracle.synth_v1(40000, 77)
    n = 8192 + 8192 + 700                                                 # three groups, the last one ragged; small chunks in between
    blobs, want = [], []
    for i in range(n):
        if i % 9 == 4: blobs.append(comp(small)); want.append(small)
        else: blobs.append(ub[i % len(ub)]); want.append(uniq[i % len(uniq)])
    res, out, off = _run(eng, codec, blobs, [len(r) for r in want], N.FLAG_BIG_CHUNKS)
    bad = [i for i, r in enumerate(want) if res[i] != len(r) or out[int(off[i]):int(off[i]) + len(r)].tobytes() != r]
    assert not bad, (len(bad), bad[:8])


def _device_batch(eng, codec, blobs, caps):
    """the arrays of a device batch, uploaded; returns (submit, fetch, free)"""
    n = len(blobs)
    in_len = np.array([len(b) for b in blobs], np.uint64)
    in_off = np.concatenate([[0], np.cumsum(in_len)[:-1]]).astype(np.uint64)
    packed = np.frombuffer(b"".join(blobs) + bytes(64), np.uint8)
    out_cap = np.array(caps, np.uint64); out_off = np.concatenate([[0], np.cumsum(out_cap)[:-1]]).astype(np.uint64)
    total = int(out_off[-1] + out_cap[-1]) + 64
    d_in = eng.alloc(packed.nbytes); d_out = eng.alloc(total); d_meta = eng.alloc(5 * n * 8)
    eng.h2d(d_in, packed); eng.h2d(d_meta, np.concatenate([in_off, in_len, out_off, out_cap]))
    submit = lambda flags: eng.batch_device(codec, DEC, flags, n, d_in, d_meta, d_meta + 8 * n, d_out, d_meta + 16 * n, d_meta + 24 * n, d_meta + 32 * n)
    fetch = lambda: (eng.d2h(d_meta + 32 * n, 8 * n, "int64"), eng.d2h(d_out, total), out_off)
    free = lambda: [eng.free(p) for p in (d_in, d_out, d_meta)]
    return submit, fetch, free


def test_reservation_follows_the_counts_the_engine_has_seen():
    """round-5 verdict item 5 / advisor r4: a flagged batch of 8 192 chunks with ONE big chunk must not reserve record areas for 8 192 (8 GiB);
    the engine plans from the counts of big chunks it has seen (engine.hip plan_big), and what exceeds the plan still decodes (one wavefront)"""
    L = N.lib()
    e = N.Engine(0)
    try:
        small = oracle.synth_v1(4096, 1); big = oracle.synth_v1(S, 2)
        raws = [small] * 8191 + [big]
        blobs = [oracle.lz4_compress_raw(r)[1] for r in (small, big)]
        submit, fetch, free = _device_batch(e, LZ4, [blobs[0]] * 8191 + [blobs[1]], [len(r) for r in raws])
        submit(N.FLAG_BIG_CHUNKS); e.sync()
        res, out, off = fetch()
        assert [int(x) for x in res] == [len(r) for r in raws]
        assert out[int(off[8191]):int(off[8191]) + S].tobytes() == big
        assert 0 < L.cj_debug_big_scratch_bytes(e.h) < 64 << 20, L.cj_debug_big_scratch_bytes(e.h)
        free()
        # the same engine now meets 300 big chunks: the first such batch is planned for one (the rest take the wavefront kernel), the ones behind it for 300
        raws = [big, small] * 300
        submit, fetch, free = _device_batch(e, LZ4, [blobs[1], blobs[0]] * 300, [len(r) for r in raws])
        for _ in range(3):
            submit(N.FLAG_BIG_CHUNKS); e.sync()
            res, out, off = fetch()
            assert [int(x) for x in res] == [len(r) for r in raws]
            assert all(out[int(off[i]):int(off[i]) + len(r)].tobytes() == r for i, r in enumerate(raws))
        assert 300 << 20 <= L.cj_debug_big_scratch_bytes(e.h) < 1000 << 20, L.cj_debug_big_scratch_bytes(e.h)
        free()
    finally:
        e.close()


def test_a_flagged_call_only_enqueues_once_the_engine_has_seen_a_count():
    """cj_batch_device with CJ_FLAG_BIG_CHUNKS must not wait for its stream (r5 verdict item 7: above 8 192 chunks it read the list's count
    back): with several batches queued, submitting the next one returns long before the queue drains"""
    import time
    e = N.Engine(0)
    try:
        big = oracle.synth_v1(S, 3)
        blob = oracle.lz4_compress_raw(big)[1]
        n = 9000                                                  # more than one group's worth of chunks: the case that used to synchronise
        submit, fetch, free = _device_batch(e, LZ4, [blob] * n, [S] * n)
        submit(N.FLAG_BIG_CHUNKS); e.sync()                       # the engine sees its first count (and allocates): this call may wait
        submit(N.FLAG_BIG_CHUNKS); e.sync()                       # planned for 9 000 now
        t0 = time.perf_counter(); submit(N.FLAG_BIG_CHUNKS); e.sync(); one = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(6): submit(N.FLAG_BIG_CHUNKS)
        queued = time.perf_counter() - t0
        e.sync()
        drained = time.perf_counter() - t0
        res, out, off = fetch()
        assert (res == S).all() and out[int(off[n - 1]):int(off[n - 1]) + S].tobytes() == big
        assert drained > 3 * one and queued < 0.5 * drained, (one, queued, drained)      # six batches were still running when the sixth submit returned
        free()
    finally:
        e.close()


def _sn_stream(rnd, n_out, weights):
    """a hand-made Snappy raw stream of n_out bytes from every element form the format has: literals with headers of 1 .. 5 bytes (also
    non-minimal ones), copies with 1-, 2- and 4-byte offsets, a literal behind a literal.  weights: literal, copy-1, copy-2, copy-4"""
    body = bytearray(); made = 0
    def lit(n, form):
        nonlocal made
        if form == 0 and n <= 60: body.append((n - 1) << 2)
        else:
            nb = max(form, 1 if n <= 256 else 2 if n <= 65536 else 3)
            body.append((59 + nb) << 2); body.extend((n - 1).to_bytes(nb, "little"))
        body.extend(rnd.randbytes(n)); made += n
    lit(rnd.randrange(1, 40), 0)
    while made < n_out:
        room = n_out - made
        k = rnd.choices((0, 1, 2, 3), weights)[0]
        if k == 0:
            n = min(room, rnd.choice((3000, 70000)) if rnd.randrange(60) == 0 else rnd.choice((1, 2, 5, 17, 60, 61, 62, 200, 257))); lit(n, rnd.choice((0, 0, 1, 2, 3, 4)))
        elif k == 1:
            n = min(room, rnd.randrange(4, 12)); off = rnd.randrange(1, min(made, 2047) + 1)
            if n < 4: lit(n, 0); continue
            body.append(1 | ((n - 4) << 2) | ((off >> 8) << 5)); body.append(off & 0xff); made += n
        else:
            n = min(room, rnd.randrange(1, 65)); off = rnd.randrange(1, min(made, 65535) + 1)
            body.append((2 if k == 2 else 3) | ((n - 1) << 2)); body.extend(off.to_bytes(2 if k == 2 else 4, "little")); made += n
    return bytes(_sn_varint(n_out)) + bytes(body)


def test_every_snappy_element_form_on_the_segmented_walk(eng):
    """round 6 (f06): the walk's straight-line step takes literal headers of up to 4 bytes, copy-4 elements and a literal behind a literal
    (until then: the grammar's general function, through global memory); 5-byte headers still take that one.  Streams of 70 .. 256 KiB
    made of every form, against the oracle — and with a flipped byte each, for the verdicts."""
    rnd = random.Random(606)
    blobs = [_sn_stream(rnd, rnd.choice((70000, 131072, 200000, 262144)), w) for w in ((1, 3, 3, 3), (3, 1, 1, 3), (1, 0, 1, 6), (2, 4, 4, 0)) for _ in range(6)]
    dam = []
    for b in blobs[:12]:
        d = bytearray(b); d[rnd.randrange(5, len(d))] ^= 1 << rnd.randrange(8); dam.append(bytes(d))
    allb = blobs + dam
    caps = [min(max(oracle.snappy_decompress_len(d), 0), 1 << 19) for d in allb]
    res, out, off = _run(eng, SN, allb, caps, N.FLAG_BIG_CHUNKS)
    n_ok = 0
    for k, (d, cap) in enumerate(zip(allb, caps)):
        er, eo = oracle.snappy_decompress(d, cap)
        assert res[k] == er, (k, len(d), cap, int(res[k]), er)
        if er >= 0:
            assert out[int(off[k]):int(off[k]) + er].tobytes() == eo, k
            n_ok += 1
    assert n_ok >= len(blobs)


@pytest.mark.parametrize("place", [0, N.FLAG_FORCE_PARSE_KERNEL, N.FLAG_FORCE_FUSED_PARSE])
def test_every_snappy_element_form_on_the_small_chunk_paths(eng, place):
    """the same streams at 3 .. 64 KiB through the parse kernel and through the parse inside the decoder (whose walks from guessed starts
    take the straight-line step for all these forms since f06), intact and with a flipped byte"""
    rnd = random.Random(607 + place)
    blobs = [_sn_stream(rnd, rnd.choice((3000, 20000, 50000, 65536)), w) for w in ((1, 3, 3, 3), (3, 1, 1, 3), (1, 0, 1, 6), (2, 4, 4, 0), (1, 6, 1, 1)) for _ in range(8)]
    dam = []
    for b in blobs[:20]:
        d = bytearray(b); d[rnd.randrange(3, len(d))] ^= 1 << rnd.randrange(8); dam.append(bytes(d))
    allb = (blobs + dam) * 8                                      # (a few hundred chunks: the workgroup decoder's paths, not only the wavefront kernel)
    caps = [min(max(oracle.snappy_decompress_len(d), 0), 1 << 17) for d in allb]
    res, out, off = _run(eng, SN, allb, caps, place)
    for k, (d, cap) in enumerate(zip(allb, caps)):
        if k >= len(blobs) + len(dam): 
            assert res[k] == res[k - len(blobs) - len(dam)]
            continue
        er, eo = oracle.snappy_decompress(d, cap)
        assert res[k] == er, (place, k, len(d), cap, int(res[k]), er)
        if er >= 0: assert out[int(off[k]):int(off[k]) + er].tobytes() == eo, (place, k)
