"""GPU parity tests: HIP path (through the C-ABI of libcramjam_hip.so) vs the CPU oracle and the golden
fixtures.  Bit-exact for every decoder output; encoder output must decode losslessly with the oracle's
decoder (the reference pins compressed bytes only for all-literal inputs)."""
import hashlib

import numpy as np
import pytest

import oracle
from conftest import b64d, sha

pytestmark = pytest.mark.gpu

from cramjam_amd import _native as N  # noqa: E402

LZ4, SNAPPY, DEC, ENC, PREFIX = N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS, N.OP_COMPRESS, N.FLAG_LZ4_SIZE_PREFIX


@pytest.fixture(scope="module")
def eng():
    e = N.Engine(0)
    yield e
    e.close()


# both chunk->hardware mappings of the LZ4 decoder must give identical results
MAPPINGS = [pytest.param(N.FLAG_FORCE_WAVE_PER_CHUNK, id="wave-per-chunk"), pytest.param(N.FLAG_FORCE_LANE_PER_CHUNK, id="lane-per-chunk"),
            pytest.param(N.FLAG_FORCE_LDS_PER_CHUNK, id="parse+lds-workgroup")]


def test_device_present():
    assert N.lib().cj_device_count() >= 1


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_decode_golden_vectors(eng, golden, mapping):
    vs = golden["vectors"]
    for codec, key in ((LZ4, "lz4"), (SNAPPY, "snappy")):
        for extra in (0, 77):
            res, outs = eng.batch_host(codec, DEC, mapping, [b64d(v[key]) for v in vs], [v["n"] + extra for v in vs])
            for v, r, o in zip(vs, res, outs):
                assert r == v["n"], (key, v["name"], r)
                assert sha(o) == v["sha256"], (key, v["name"])


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_decode_reference_fixture_blocks(eng, golden, plaintext, mapping):
    import os
    from conftest import GOLDEN_DIR
    fb = golden["reference_fixture_blocks"]
    a, b = fb["lz4_frame_block"]
    blk = open(os.path.join(GOLDEN_DIR, "plaintext.txt.lz4"), "rb").read()[a:b]
    res, outs = eng.batch_host(LZ4, DEC, mapping, [blk], [len(plaintext)])
    assert res == [857] and outs[0] == plaintext
    a, b = fb["snappy_framed_raw"]
    blk = open(os.path.join(GOLDEN_DIR, "plaintext.txt.snappy"), "rb").read()[a:b]
    res, outs = eng.batch_host(SNAPPY, DEC, 0, [blk], [len(plaintext)])
    assert res == [857] and outs[0] == plaintext


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_decode_malformed_matches_oracle(eng, golden, mapping):
    ms = golden["malformed_lz4"]
    res, outs = eng.batch_host(LZ4, DEC, mapping, [b64d(m["data"]) for m in ms], [m["cap"] for m in ms])
    for m, r, o in zip(ms, res, outs):
        er, eo = oracle.lz4_decompress_raw(b64d(m["data"]), m["cap"])
        if er < 0:
            assert r == -7, (m["src"], m["kind"], m["k"], m["cap"], r)
        else:
            assert r == er and o == eo, (m["src"], m["kind"], m["k"], m["cap"], r, er)
    ms = [m for m in golden["malformed_snappy"]]
    caps = [max(oracle.snappy_decompress_len(b64d(m["data"])), 0) for m in ms]
    caps = [min(c, 1 << 20) for c in caps]
    res, outs = eng.batch_host(SNAPPY, DEC, mapping, [b64d(m["data"]) for m in ms], caps)
    for m, c, r, o in zip(ms, caps, res, outs):
        er, eo = oracle.snappy_decompress(b64d(m["data"]), c)
        assert r == er, (m["src"], m["kind"], m["k"], r, er)
        if er >= 0:
            assert o == eo


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_lz4_prefix_flag(eng, golden_raw, mapping):
    raw = golden_raw["plaintext"]
    _, blk = oracle.lz4_block_compress(raw, prepend=True)
    res, outs = eng.batch_host(LZ4, DEC, PREFIX | mapping, [blk, blk, blk[:3], b"\xff\xff\xff\xff\x00", blk],
                               [len(raw), len(raw) + 9, 10, 10, len(raw) - 1])
    assert res == [len(raw), len(raw), -3, -4, -6]
    assert outs[0] == raw and outs[1] == raw


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_encode_roundtrips_through_oracle(eng, golden, golden_raw, codec):
    names = [v["name"] for v in golden["vectors"]]
    raws = [golden_raw[n] for n in names]
    L = N.lib()
    if codec == LZ4:
        caps = [L.cj_lz4_block_compress_bound(len(r), 0) for r in raws]
    else:
        caps = [L.cj_snappy_raw_max_compress_len(len(r)) for r in raws]
    res, outs = eng.batch_host(codec, ENC, 0, raws, caps)
    tot_gpu = tot_cpu = 0
    for name, raw, r, o in zip(names, raws, res, outs):
        assert r > 0 and r == len(o), (name, r)
        if codec == LZ4:
            dr, d = oracle.lz4_decompress_raw(o, len(raw))          # exact capacity: end-of-block rules bite
            cr, _ = oracle.lz4_compress_raw(raw)
        else:
            dr, d = oracle.snappy_decompress(o)
            cr, _ = oracle.snappy_compress(raw)
        assert dr == len(raw) and d == raw, (name, dr)
        tot_gpu += r
        tot_cpu += cr
    # "ratio held": the wave-parallel matcher must stay close to the CPU encoders on the same data
    assert tot_gpu <= tot_cpu * 1.10, (tot_gpu, tot_cpu)


def test_lz4_encode_known_answers(eng):
    # /root/reference/tests/test_variants.py:329-334 (all-literal block: bytes are pinned)
    res, outs = eng.batch_host(LZ4, ENC, PREFIX, [b"howdy neighbor"], [64])
    assert outs[0] == b"\x0e\x00\x00\x00\xe0howdy neighbor"
    res, outs = eng.batch_host(LZ4, ENC, 0, [b"howdy neighbor", b""], [64, 16])
    assert outs[0] == b"\xe0howdy neighbor" and outs[1] == b"\x00"
    res, outs = eng.batch_host(SNAPPY, ENC, 0, [b"howdy neighbor", b""], [64, 32])
    assert outs[0] == b"\x0e4howdy neighbor" and outs[1] == b"\x00"


def test_single_buffer_c_abi(plaintext):
    import ctypes as C
    L = N.lib()
    out = C.create_string_buffer(2048)
    n = L.cj_lz4_block_compress(plaintext, len(plaintext), C.cast(out, C.c_void_p), 2048, -1, -1, -1)
    assert n > 4 and int.from_bytes(out.raw[:4], "little") == 857
    assert oracle.lz4_block_decompress(out.raw[:n], 857, True) == (857, plaintext)
    back = C.create_string_buffer(1000)
    m = L.cj_lz4_block_decompress(out.raw[:n], n, C.cast(back, C.c_void_p), 1000, 1)
    assert m == 857 and back.raw[:857] == plaintext
    n = L.cj_snappy_raw_compress(plaintext, len(plaintext), C.cast(out, C.c_void_p), 2048)
    assert n > 0 and oracle.snappy_decompress(out.raw[:n]) == (857, plaintext)
    m = L.cj_snappy_raw_decompress(out.raw[:n], n, C.cast(back, C.c_void_p), 1000)
    assert m == 857 and back.raw[:857] == plaintext
    assert L.cj_snappy_raw_compress(plaintext, len(plaintext), C.cast(out, C.c_void_p), 100) == -11


def _device_batch(eng, codec, op, flags, blobs, caps):
    """pack -> device-resident batch -> results, out bytes (device path, no host staging by the engine)"""
    n = len(blobs)
    in_off = np.zeros(n, np.uint64); in_len = np.array([len(b) for b in blobs], np.uint64)
    pos = 0
    for i, b in enumerate(blobs):
        in_off[i] = pos
        pos += len(b) + 3          # deliberately unaligned packing
    packed = np.zeros(pos + 16, np.uint8)
    for i, b in enumerate(blobs):
        packed[int(in_off[i]):int(in_off[i]) + len(b)] = np.frombuffer(b, np.uint8)
    out_cap = np.array(caps, np.uint64)
    out_off = np.concatenate([[0], np.cumsum(out_cap + 5)[:-1]]).astype(np.uint64)
    total_out = int(out_off[-1] + out_cap[-1]) + 16
    d_in = eng.alloc(packed.nbytes); d_out = eng.alloc(total_out)
    d_meta = eng.alloc(5 * n * 8)
    eng.h2d(d_in, packed)
    eng.h2d(d_meta, np.concatenate([in_off, in_len, out_off, out_cap]))
    N.check(N.lib().cj_memset_dev(eng.h, d_out, 0xAB, total_out))
    eng.batch_device(codec, op, flags, n, d_in, d_meta, d_meta + 8 * n, d_out, d_meta + 16 * n, d_meta + 24 * n, d_meta + 32 * n)
    eng.sync()
    res = eng.d2h(d_meta + 32 * n, 8 * n, "int64")
    out = eng.d2h(d_out, total_out)
    for p in (d_in, d_out, d_meta):
        eng.free(p)
    return res, out, out_off


@pytest.mark.parametrize("mapping", MAPPINGS)
@pytest.mark.parametrize("chunk", [65536, 262144])
def test_device_batch_synth_roundtrip(eng, chunk, mapping):
    n = 96 if chunk == 65536 else 24
    raws = [oracle.synth_v1(chunk, i) for i in range(n)]
    raws[3] = bytes(chunk); raws[5] = hashlib.shake_256(b"x").digest(chunk); raws[7] = raws[7][:chunk - 1]; raws[9] = b""
    for codec in (LZ4, SNAPPY):
        comp = [(oracle.lz4_compress_raw(r) if codec == LZ4 else oracle.snappy_compress(r))[1] for r in raws]
        big = N.FLAG_BIG_CHUNKS if chunk > 65536 else 0          # (ignored by the forced one-wavefront / one-lane mappings)
        res, out, off = _device_batch(eng, codec, DEC, mapping | big, comp, [len(r) for r in raws])
        for i, r in enumerate(raws):
            if codec == LZ4 and len(r) == 0:
                assert res[i] == 0
                continue
            assert res[i] == len(r), (codec, i, res[i])
            assert out[int(off[i]):int(off[i]) + len(r)].tobytes() == r, (codec, i)
            # nothing written past the chunk's capacity
            assert (out[int(off[i]) + len(r):int(off[i]) + len(r) + 5] == 0xAB).all()
        L = N.lib()
        caps = [(L.cj_lz4_block_compress_bound(len(r), 0) if codec == LZ4 else L.cj_snappy_raw_max_compress_len(len(r))) for r in raws]
        res, out, off = _device_batch(eng, codec, ENC, 0, raws, caps)
        for i, r in enumerate(raws):
            blob = out[int(off[i]):int(off[i]) + int(res[i])].tobytes()
            d = oracle.lz4_decompress_raw(blob, len(r)) if codec == LZ4 else oracle.snappy_decompress(blob)
            assert d == (len(r), r), (codec, i, res[i])


# both sides of CJ_FUSED_MAX_CHUNKS (16 384): parse + decode in one kernel below it, parse pass + workgroup decoder above
PIPELINE_BATCHES = [8192 + 37, 24576 + 41]


@pytest.mark.parametrize("n", PIPELINE_BATCHES)
def test_default_pipeline_mixed_large_batch(eng, n):
    """A batch through the DEFAULT path (up to CJ_FUSED_MAX_CHUNKS chunks: parse + decode in one kernel; more: the parse
    pass, then the workgroup decoder): every chunk goes to the LDS workgroup decoder (many short sequences) or the wave
    decoder (few long runs, oversize, tiny); malformed chunks must fail alone.  Every result and byte is compared with
    the oracle."""
    import random
    rnd = random.Random(11)
    kinds = []
    uniq = []
    for i in range(48):
        k = i % 8
        if k == 0: raw = oracle.synth_v1(65536, i)
        elif k == 1: raw = bytes(65536)
        elif k == 2: raw = hashlib.shake_256(b"r%d" % i).digest(65536)
        elif k == 3: raw = oracle.synth_v1(rnd.randrange(1, 65536), i)
        elif k == 4: raw = (b"abcdefgh" * 9000)[:rnd.randrange(20, 65536)]
        elif k == 5: raw = oracle.synth_v1(70000, i)                       # larger than the LDS window
        elif k == 6: raw = oracle.synth_v1(4096, i) + bytes(3000) + hashlib.shake_256(b"x").digest(5000)
        else: raw = b"tiny%d" % i
        blk = oracle.lz4_compress_raw(raw)[1]
        if i % 16 == 7:                                                   # corrupt a few
            b = bytearray(blk); b[len(b) // 2] ^= 0x5A; blk = bytes(b[:max(3, len(b) - 9)])
        uniq.append((raw, blk))
    blobs = [uniq[i % 48][1] for i in range(n)]
    caps = [len(uniq[i % 48][0]) for i in range(n)]
    res, out, off = _device_batch(eng, LZ4, DEC, 0, blobs, caps)
    exp = [oracle.lz4_decompress_raw(b, len(r)) for r, b in uniq]
    n_bad = 0
    for i in range(n):
        er, eo = exp[i % 48]
        if er < 0:
            assert res[i] == -7, (i, res[i])
            n_bad += 1
        else:
            assert res[i] == er, (i, i % 48, res[i], er)
            assert out[int(off[i]):int(off[i]) + er].tobytes() == eo, (i, i % 48)
            assert (out[int(off[i]) + caps[i]:int(off[i]) + caps[i] + 5] == 0xAB).all()
    assert n_bad > 100


@pytest.mark.parametrize("n", PIPELINE_BATCHES)
def test_default_pipeline_mixed_large_batch_snappy(eng, n):
    """Snappy twin of the test above."""
    import random
    rnd = random.Random(12)
    uniq = []
    for i in range(40):
        k = i % 8
        if k == 0: raw = oracle.synth_v1(65536, i)
        elif k == 1: raw = bytes(65536)
        elif k == 2: raw = hashlib.shake_256(b"s%d" % i).digest(65536)
        elif k == 3: raw = oracle.synth_v1(rnd.randrange(1, 65536), i)
        elif k == 4: raw = (b"abcdefgh" * 9000)[:rnd.randrange(20, 65536)]
        elif k == 5: raw = oracle.synth_v1(70000, i)
        elif k == 6: raw = oracle.synth_v1(4096, i) + bytes(3000) + hashlib.shake_256(b"y").digest(5000)
        else: raw = b"tiny%d" % i
        blk = oracle.snappy_compress(raw)[1]
        if i % 16 == 7:
            b = bytearray(blk); b[len(b) // 2] ^= 0x5A; blk = bytes(b[:max(3, len(b) - 9)])
        uniq.append((raw, blk))
    blobs = [uniq[i % 40][1] for i in range(n)]
    caps = [len(uniq[i % 40][0]) for i in range(n)]
    res, out, off = _device_batch(eng, SNAPPY, DEC, 0, blobs, caps)
    exp = [oracle.snappy_decompress(b, len(r)) for r, b in uniq]
    n_bad = 0
    for i in range(n):
        er, eo = exp[i % 40]
        assert res[i] == er, (i, i % 40, res[i], er)
        if er < 0:
            n_bad += 1
        else:
            assert out[int(off[i]):int(off[i]) + er].tobytes() == eo, (i, i % 40)
            assert (out[int(off[i]) + caps[i]:int(off[i]) + caps[i] + 5] == 0xAB).all()
    assert n_bad > 100


def test_mixed_codecs_256k_concurrent_streams():
    """BASELINE configs[4] at reduced N: interleaved LZ4-block / Snappy-raw 256 KiB chunks, one engine (= one HIP
    stream) per codec, both batches in flight at once through the big-chunk path (CJ_FLAG_BIG_CHUNKS: segmented parse + slab
    decoder); every chunk checked against the oracle."""
    e_lz4, e_sn = N.Engine(0), N.Engine(0)
    S = 262144
    raws = [oracle.synth_v1(S, 100 + i) for i in range(24)]
    raws[2] = bytes(S); raws[3] = hashlib.shake_256(b"m").digest(S)
    lz = [(i, oracle.lz4_compress_raw(r)[1]) for i, r in enumerate(raws) if i % 2 == 0]
    sn = [(i, oracle.snappy_compress(r)[1]) for i, r in enumerate(raws) if i % 2 == 1]

    def submit(eng, codec, items):
        n = len(items)
        blobs = [b for _, b in items]
        in_len = np.array([len(b) for b in blobs], np.uint64)
        in_off = np.concatenate([[0], np.cumsum(in_len + 7)[:-1]]).astype(np.uint64)
        packed = np.zeros(int(in_off[-1] + in_len[-1]) + 16, np.uint8)
        for k, b in enumerate(blobs):
            packed[int(in_off[k]):int(in_off[k]) + len(b)] = np.frombuffer(b, np.uint8)
        out_off = (np.arange(n) * S).astype(np.uint64); out_cap = np.full(n, S, np.uint64)
        d_in = eng.alloc(packed.nbytes); d_out = eng.alloc(n * S); d_meta = eng.alloc(5 * n * 8)
        eng.h2d(d_in, packed); eng.h2d(d_meta, np.concatenate([in_off, in_len, out_off, out_cap]))
        eng.batch_device(codec, DEC, N.FLAG_BIG_CHUNKS, n, d_in, d_meta, d_meta + 8 * n, d_out, d_meta + 16 * n, d_meta + 24 * n, d_meta + 32 * n)
        return d_in, d_out, d_meta, n

    h1 = submit(e_lz4, LZ4, lz)            # asynchronous: returns after enqueue
    h2 = submit(e_sn, SNAPPY, sn)
    for eng, (d_in, d_out, d_meta, n), items in ((e_lz4, h1, lz), (e_sn, h2, sn)):
        eng.sync()
        res = eng.d2h(d_meta + 32 * n, 8 * n, "int64")
        out = eng.d2h(d_out, n * S)
        for k, (i, _) in enumerate(items):
            assert res[k] == S and out[k * S:(k + 1) * S].tobytes() == raws[i], (i, res[k])
        for p in (d_in, d_out, d_meta):
            eng.free(p)
    e_lz4.close(); e_sn.close()


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_large_batch_compress_then_decompress_property(eng, codec):
    """BASELINE configs[2]-style at 12 288 x 64 KiB: GPU compress -> GPU decompress (default large-batch pipelines)
    must reproduce the input exactly for every chunk (device-side compare), and a sample of the GPU-compressed
    blocks must decode with the CPU oracle (the stand-in for the reference's CPU decoder)."""
    L = N.lib()
    S, U, n = 65536, 512, 12288
    d_raw = eng.alloc(U * S)
    N.check(L.cj_bench_synth_v1(d_raw, S, S, 7000, U, 0x5EED, None))
    N.check(L.cj_engine_sync(eng.h))
    import ctypes as C
    bound = L.cj_lz4_block_compress_bound(S, 0) if codec == LZ4 else L.cj_snappy_raw_max_compress_len(S)
    stride = (bound + 15) & ~15
    ids = np.arange(n, dtype=np.uint64)
    d_comp = eng.alloc(n * stride)
    d_m1 = eng.alloc(5 * n * 8)
    eng.h2d(d_m1, np.concatenate([(ids % U) * S, np.full(n, S, np.uint64), ids * stride, np.full(n, stride, np.uint64), np.zeros(n, np.uint64)]))
    # the generator ran on the NULL stream, the engine stream is non-blocking: make the data visible first
    eng.d2h(d_raw, 16)
    eng.batch_device(codec, ENC, 0, n, d_raw, d_m1, d_m1 + 8 * n, d_comp, d_m1 + 16 * n, d_m1 + 24 * n, d_m1 + 32 * n)
    eng.sync()
    clen = eng.d2h(d_m1 + 32 * n, 8 * n, "int64")
    assert (clen > 0).all() and (clen <= bound).all()
    d_out = eng.alloc(n * S)
    N.check(L.cj_memset_dev(eng.h, d_out, 0xCD, n * S))
    d_m2 = eng.alloc(5 * n * 8)
    eng.h2d(d_m2, np.concatenate([ids * stride, clen.astype(np.uint64), ids * S, np.full(n, S, np.uint64), np.zeros(n, np.uint64)]))
    eng.batch_device(codec, DEC, 0, n, d_comp, d_m2, d_m2 + 8 * n, d_out, d_m2 + 16 * n, d_m2 + 24 * n, d_m2 + 32 * n)
    eng.sync()
    assert (eng.d2h(d_m2 + 32 * n, 8 * n, "int64") == S).all()
    d_mism = eng.alloc(8)
    N.check(L.cj_memset_dev(eng.h, d_mism, 0, 8))
    N.check(L.cj_bench_compare(d_out, d_m2 + 16 * n, d_raw, S, U, S, n, d_mism, None))
    assert int(eng.d2h(d_mism, 8, "int64")[0]) == 0
    comp_h = eng.d2h(d_comp, n * stride); raw_h = eng.d2h(d_raw, U * S)
    for p_ in (d_raw, d_comp, d_m1, d_out, d_m2, d_mism):
        eng.free(p_)
    for i in range(0, n, 251):
        blob = comp_h[i * stride:i * stride + int(clen[i])].tobytes()
        want = raw_h[(i % U) * S:(i % U + 1) * S].tobytes()
        got = oracle.lz4_decompress_raw(blob, S) if codec == LZ4 else oracle.snappy_decompress(blob)
        assert got == (S, want), i
    # batches of this size run as persistent blocks, some with their hash table in LDS and some with it in global memory
    # (engine.hip launch_encode): which kind compresses a chunk is a race, the bytes must not depend on it
    assert (clen.reshape(-1, U) == clen[:U]).all()
    rows = comp_h.reshape(n, stride)
    for i in range(U):
        assert (rows[i::U, :int(clen[i])] == rows[i, :int(clen[i])]).all(), i


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_encoder_block_kinds_agree_with_the_plain_kernel(codec):
    """A large batch runs as persistent blocks of two kinds (hash table in LDS / in global memory, engine.hip launch_encode);
    a small one as one block per chunk.  Every copy of a chunk in the large batch must be byte-identical to what the plain
    kernel emits for it — the global-memory table has to reproduce the LDS table's semantics exactly (stores acknowledged
    before the lookups that follow, the highest position winning a contested slot: cj_match.hpp insert_round)."""
    e = N.Engine(0)
    U, n = 64, 8192
    raws = [oracle.synth_v1(65536, 500 + i) for i in range(U)]
    raws[5] = raws[5][:40000]; raws[9] = bytes(65536); raws[11] = (b"abcdefgh" * 9000)[:65536]; raws[13] = hashlib.shake_256(b"k").digest(30000) + raws[13][:30000]
    res0, outs0 = e.batch_host(codec, ENC, 0, raws, [80000] * U)
    ref = [bytes(o) for o in outs0]
    res, outs = e.batch_host(codec, ENC, 0, [raws[i % U] for i in range(n)], [80000] * n)
    bad = [i for i in range(n) if bytes(outs[i]) != ref[i % U]]
    assert not bad, (len(bad), bad[:8])
    e.close()


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_encoder_persistent_blocks_on_mixed_chunks(eng, codec):
    """The large-batch encoder path (persistent LDS-table + global-table blocks) on chunks of every shape: empty, tiny, zeros,
    random, short periods, ragged sizes, above 64 KiB.  Every copy of a chunk must compress to the same bytes, and those
    bytes must decode with the oracle."""
    import random
    rnd = random.Random(21)
    uniq = [b"", b"a", b"abcd" * 3, bytes(65536), hashlib.shake_256(b"e").digest(65536), (b"xyz" * 30000)[:65536],
            oracle.synth_v1(70000, 3), oracle.synth_v1(4096, 4) + bytes(3000) + hashlib.shake_256(b"f").digest(5000)]
    uniq += [oracle.synth_v1(rnd.randrange(1, 65537), 10 + i) for i in range(16)]
    uniq += [(bytes([i]) * rnd.randrange(1, 300) + oracle.synth_v1(3000, i))[:rnd.randrange(13, 3300)] for i in range(8)]
    U = len(uniq)
    n = 8192 + 3 * U + 5
    L = N.lib()
    bound = [(L.cj_lz4_block_compress_bound(len(r), 0) if codec == LZ4 else L.cj_snappy_raw_max_compress_len(len(r))) for r in uniq]
    res, out, off = _device_batch(eng, codec, ENC, 0, [uniq[i % U] for i in range(n)], [bound[i % U] for i in range(n)])
    first = []
    for i in range(U):
        assert 0 < res[i] <= bound[i], (i, res[i])
        blob = out[int(off[i]):int(off[i]) + int(res[i])].tobytes()
        got = oracle.lz4_decompress_raw(blob, len(uniq[i])) if codec == LZ4 else oracle.snappy_decompress(blob)
        assert got == (len(uniq[i]), uniq[i]), i
        first.append(blob)
    for i in range(U, n):
        assert res[i] == res[i % U], i
        assert out[int(off[i]):int(off[i]) + int(res[i])].tobytes() == first[i % U], i


def _shaped_chunks(seed, count, size=65536):
    """chunks with very different sequence shapes: short/long literals, short/long matches, self-overlapping matches with
    small periods, near and far offsets, long runs — all with enough sequences to take the LDS workgroup decoder"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        kind = i % 6
        buf = bytearray()
        while len(buf) < size:
            r = rng.random()
            if kind == 0:      # text-like: small alphabet words
                buf += bytes(rng.integers(97, 101, int(rng.integers(1, 12)), dtype=np.uint8))
            elif kind == 1:    # periodic runs with tiny periods (offsets 1..8, overlapping matches) between short literals
                p = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
                buf += p * int(rng.integers(2, 40)) + bytes(rng.integers(0, 256, int(rng.integers(0, 6)), dtype=np.uint8))
            elif kind == 2:    # far copies of earlier content, lengths 4..300
                if len(buf) > 64 and r < 0.7:
                    s = int(rng.integers(0, len(buf) - 4)); ln = int(rng.integers(4, 300))
                    buf += buf[s:s + ln]
                else:
                    buf += bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
            elif kind == 3:    # long literals (random) alternating with long matches
                buf += bytes(rng.integers(0, 256, int(rng.integers(30, 700)), dtype=np.uint8))
                if len(buf) > 1000:
                    s = int(rng.integers(0, len(buf) - 600)); buf += buf[s:s + int(rng.integers(40, 600))]
            elif kind == 4:    # mostly short matches from the last 256 bytes (dense dependency chains)
                if len(buf) > 16 and r < 0.85:
                    s = len(buf) - int(rng.integers(4, min(len(buf), 256))); ln = int(rng.integers(4, 24))
                    buf += (bytes(buf[s:]) * (ln // max(1, len(buf) - s) + 1))[:ln]
                else:
                    buf += bytes(rng.integers(0, 256, int(rng.integers(1, 8)), dtype=np.uint8))
            else:              # zero runs and counters
                buf += bytes(int(rng.integers(1, 400))) if r < 0.3 else bytes((np.arange(int(rng.integers(4, 60))) % 7).astype(np.uint8))
        out.append(bytes(buf[:size]))
    return out


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_lds_decoder_on_shaped_streams(eng, codec):
    """the parse + LDS workgroup pipeline (forced) on 240 structurally diverse 64 KiB chunks, compressed by the CPU oracle
    (= liblz4 / libsnappy bytes) and by the GPU encoder: bit-exact, at exact and at loose capacity"""
    raws = _shaped_chunks(11, 240)
    comp = oracle.lz4_compress_raw if codec == LZ4 else oracle.snappy_compress
    blobs = [comp(r)[1] for r in raws]
    caps = [len(r) + 64 + len(r) // 5 for r in raws]
    res, gpu_blobs = eng.batch_host(codec, ENC, 0, raws, caps)
    assert all(r > 0 for r in res)
    for streams in (blobs, [bytes(b) for b in gpu_blobs]):
        for extra in (0, 33):
            res, outs = eng.batch_host(codec, DEC, N.FLAG_FORCE_LDS_PER_CHUNK, streams, [len(r) + extra for r in raws])
            bad = [i for i, (r, o, raw) in enumerate(zip(res, outs, raws)) if r != len(raw) or o != raw]
            assert not bad, (codec, extra, bad[:10], [res[i] for i in bad[:10]])
    # the same streams through the wave kernel give the same bytes (cross-mapping check)
    res, outs = eng.batch_host(codec, DEC, N.FLAG_FORCE_WAVE_PER_CHUNK, blobs, [len(r) for r in raws])
    assert all(r == len(raw) and o == raw for r, o, raw in zip(res, outs, raws))


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_lds_pipeline_on_damaged_large_chunks(eng, codec):
    """1 500 full-size chunks with one random bit flipped somewhere in the compressed stream, through the parse + LDS
    pipeline: same verdict as the CPU oracle for every chunk, same bytes where the oracle still accepts the stream"""
    rng = np.random.default_rng(17)
    raws = _shaped_chunks(23, 60) + [oracle.synth_v1(65536, i) for i in range(40)]
    comp = oracle.lz4_compress_raw if codec == LZ4 else oracle.snappy_compress
    dec = (lambda b, cap: oracle.lz4_decompress_raw(b, cap)) if codec == LZ4 else (lambda b, cap: oracle.snappy_decompress(b, cap))
    blobs = [comp(r)[1] for r in raws]
    streams, caps = [], []
    for t in range(1500):
        b = bytearray(blobs[t % len(blobs)])
        pos = int(rng.integers(0, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        streams.append(bytes(b)); caps.append(65536)
    res, outs = eng.batch_host(codec, DEC, N.FLAG_FORCE_LDS_PER_CHUNK, streams, caps)
    n_ok = 0
    for i, (s, r, o) in enumerate(zip(streams, res, outs)):
        er, eo = dec(s, 65536)
        if er < 0:
            assert r < 0, (i, r, er)
            if codec == SNAPPY:
                assert r == er, (i, r, er)
        else:
            assert r == er and o == eo, (i, r, er)
            n_ok += 1
    assert 0 < n_ok < len(streams)              # both outcomes occur (a flipped literal byte still decodes)


def test_deep_chain_chunks_take_match_forwarding(eng):
    """text-like chunks (every phrase copied from its previous occurrence: dependency chains thousands of levels deep) go
    through the decoder's match-forwarding phase (lz4_decode_lds.hip, D1f) and must decode bit-exactly"""
    import json, random
    rnd = random.Random(12)
    def lines(fn, n=65536):
        out = bytearray()
        while len(out) < n: out += fn()
        return bytes(out[:n])
    kinds = [
        lambda: b"%d bottles of beer on the wall, %d bottles of beer\n" % (rnd.randrange(977), rnd.randrange(1013)),
        lambda: b"2026-09-28T12:%02d:%02d.%03d INFO worker-%d request id=%08x path=/api/v1/items/%d status=%d\n" % (
            rnd.randrange(60), rnd.randrange(60), rnd.randrange(1000), rnd.randrange(16), rnd.getrandbits(32), rnd.randrange(5000), rnd.choice([200, 404, 500])),
        lambda: json.dumps({"id": rnd.randrange(10 ** 6), "name": "user%d" % rnd.randrange(1000), "tags": ["a", "b", rnd.choice("xyz")]}).encode() + b"\n",
        lambda: b"abcabcabc" * rnd.randrange(1, 9) + bytes([rnd.randrange(97, 123)]),
    ]
    chunks = [lines(k, 65536 - 1000 * (i % 3)) for i, k in enumerate(kinds * 6)]
    L = N.lib()
    for codec, comp in ((LZ4, lambda c: oracle.lz4_compress_raw(c)[1]), (SNAPPY, lambda c: oracle.snappy_compress(c)[1])):
        blobs = [comp(c) for c in chunks]
        L.cj_debug_forwarded_chunks(1)
        res, outs = eng.batch_host(codec, DEC, N.FLAG_FORCE_LDS_PER_CHUNK | 0x1000, blobs, [len(c) for c in chunks])      # (0x1000 = CJ_FLAG_DEBUG_PROFILE: the counter counts)
        assert [int(r) for r in res] == [len(c) for c in chunks]
        assert all(bytes(o) == c for o, c in zip(outs, chunks))
        assert L.cj_debug_forwarded_chunks(0) >= len(chunks) // 2, codec


def test_one_kernel_parse_lists_its_walks_and_walks_again_where_they_outgrow_the_lists(eng):
    """the parse inside the decoder kernel (lds_shared.hpp: fused_parse) takes its phases 3 / 4 from the lists its walks wrote, and
    walks them where the walks outgrow the lists' rows: a stream of 4-byte sequences on which no guessed start ever meets the true
    path (every lane of the workgroup walks to the end of the chunk).  Both ways decode bit-exactly; the debug counter says which
    way the chunks went."""
    import ctypes as C
    L = N.lib()
    paths = (C.c_ulonglong * 3)()
    def never_meets(nseq):
        b = bytearray(bytes([0x40]) + b"wxyz" + bytes([4, 0]))
        for i in range(nseq): b += bytes([0x10, 0x61 + (i % 7), 3, 0])
        return bytes(b + bytes([0xC0]) + b"abcdefghijkl")
    def dense(nseq):                                    # 3-byte sequences: ~30 per segment, the guesses meet the path
        return bytes(bytearray([0x10, 0x61, 1, 0]) + bytes([0x00, 1, 0]) * nseq + bytes([0xC0]) + b"abcdefghijkl")
    ordinary = [oracle.lz4_compress_raw(oracle.synth_v1(65536, i))[1] for i in range(6)]
    hard = [never_meets(n) for n in (3000, 9000, 12000)]
    for blobs, slot in ((ordinary + [dense(15000), dense(16300)], 0), (hard, 2)):
        want = [oracle.lz4_decompress_raw(b, 65536) for b in blobs]
        assert all(r > 0 for r, _ in want)
        assert L.cj_debug_fused_parse_paths(paths, 1) == 0
        res, outs = eng.batch_host(LZ4, DEC, N.FLAG_FORCE_LDS_PER_CHUNK | 0x1000, blobs, [65536] * len(blobs))      # (0x1000 = CJ_FLAG_DEBUG_PROFILE: the counter counts)
        assert [int(r) for r in res] == [r for r, _ in want]
        assert all(bytes(o) == w for o, (_, w) in zip(outs, want))
        assert L.cj_debug_fused_parse_paths(paths, 0) == 0
        assert paths[slot] == len(blobs) and sum(paths) == len(blobs), (slot, list(paths))
    # Snappy: 2-byte copy elements behind one literal (66 elements per segment at most: the lists' rows hold them)
    raw = (b"ab" * 40000)[:65536]
    body = bytearray(bytes([0x04]) + b"ab")            # literal "ab", then copies of 4 bytes at offset 2: tag 0b000_000_01, offset byte 2
    while len(body) < 30000: body += bytes([0x01, 0x02])
    n_out = 2 + 4 * ((len(body) - 3) // 2)
    hdr = bytearray(); v = n_out
    while v >= 0x80: hdr.append((v & 0x7f) | 0x80); v >>= 7
    hdr.append(v)
    blob = bytes(hdr + body)
    er, eo = oracle.snappy_decompress(blob, 1 << 17)
    assert er == n_out and eo == raw[:n_out]
    assert L.cj_debug_fused_parse_paths(paths, 1) == 0
    res, outs = eng.batch_host(SNAPPY, DEC, N.FLAG_FORCE_LDS_PER_CHUNK | 0x1000, [blob] * 3, [n_out] * 3)
    assert [int(r) for r in res] == [n_out] * 3 and all(bytes(o) == eo for o in outs)
    assert L.cj_debug_fused_parse_paths(paths, 0) == 0 and sum(paths) == 3, list(paths)


def test_internal_flag_bits_are_refused_at_the_c_abi(eng):
    """piece splitting / tail report / linked-frame bits (cj_common.hpp) belong to the library's own large-buffer and frame paths; a
    C-ABI caller that sets one gets CJ_E_BAD_ARG instead of a kernel that reads descriptors which are not there"""
    blob = oracle.lz4_compress_raw(oracle.synth_v1(4096, 1))[1]
    for bad in (0x8000, 0x8000 | (4 << 16), 0x2000, 0x4000, 1 << 20):
        with pytest.raises(N.EngineError):
            eng.batch_host(LZ4, DEC, bad, [blob], [4096])
        with pytest.raises(N.EngineError):
            eng.batch_host(LZ4, N.OP_COMPRESS, bad, [b"x" * 100], [200])
    res, outs = eng.batch_host(LZ4, DEC, N.FLAG_FORCE_LDS_PER_CHUNK, [blob], [4096])          # public bits still pass
    assert int(res[0]) == 4096


def test_chunks_of_up_to_16384_sequences_take_the_workgroup_decoder(eng):
    """a 64 KiB LZ4 block holds at most 16 384 sequences; round 3 raised the parse kernels' sync-point capacity from 8 192 to that, so
    chunks of 10-16 k four-byte matches (and real text, tests/test_corpus_gpu.py) decode on the workgroup path instead of one wavefront"""
    import random
    rnd = random.Random(7)
    words = [rnd.randbytes(4) for _ in range(64)]
    def mk(n, sep):
        out = bytearray(b"".join(words))
        while len(out) < n:
            out += rnd.choice(words) + rnd.randbytes(sep)
        return bytes(out[:n])
    chunks = [mk(65536, 1), mk(65536, 2), mk(65536, 0), mk(65000, 0), mk(40000, 1)] * 13
    for codec, comp in ((LZ4, lambda c: oracle.lz4_compress_raw(c)[1]), (SNAPPY, lambda c: oracle.snappy_compress(c)[1])):
        blobs = [comp(c) for c in chunks[:5]] * 13
        for flags in (0, N.FLAG_FORCE_LDS_PER_CHUNK):
            res, outs = eng.batch_host(codec, DEC, flags, blobs, [len(c) for c in chunks])
            assert [int(r) for r in res] == [len(c) for c in chunks], (codec, flags)
            assert all(bytes(o) == c for o, c in zip(outs, chunks)), (codec, flags)


def _lz4_block(seqs, tail):
    """an LZ4 block from (literal bytes, offset, match length) triples + the final literals — written by hand so that the
    test chooses every length and alignment itself"""
    out = bytearray()
    def ext(v):
        while v >= 255: out.append(255); v -= 255
        out.append(v)
    for lit, off, m in seqs:
        out.append((min(len(lit), 15) << 4) | min(m - 4, 15))
        if len(lit) >= 15: ext(len(lit) - 15)
        out += lit
        out += bytes((off & 255, off >> 8))
        if m - 4 >= 15: ext(m - 4 - 15)
    out.append(min(len(tail), 15) << 4)
    if len(tail) >= 15: ext(len(tail) - 15)
    out += tail
    return bytes(out)


def _snappy_raw(n, seqs, tail):
    out = bytearray()
    v = n
    while v >= 128: out.append((v & 127) | 128); v >>= 7
    out.append(v)
    def lit(b):
        if not b: return
        k = len(b) - 1
        if k < 60: out.append(k << 2)
        else: out.append(61 << 2); out.extend((k & 255, k >> 8))
        out.extend(b)
    for l, off, m in seqs:
        lit(l)
        while m > 0:                                  # copies of at most 64 bytes, 2-byte offsets
            k = min(m, 64) if m - min(m, 64) == 0 or m - min(m, 64) >= 4 else m - 4
            out.append(((k - 1) << 2) | 2); out.extend((off & 255, off >> 8))
            m -= k
    lit(tail)
    return bytes(out)


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_every_alignment_and_length_of_literals_and_matches(eng, codec):
    """hand-written streams in which every literal length 0..40 meets every destination alignment, and every match length
    4..40 every combination of source and destination alignment modulo 8 — far matches (ready at the resolver's first look:
    the aligned dense copy) and near ones (the exact-address copy), through the workgroup decoder and the default pipeline"""
    rng = np.random.default_rng(5)
    chunks = []
    for variant in range(6):
        data = bytearray(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes())
        seqs = []
        first = bytes(data)
        pending_lit = first
        combos = [(m, sa, ll) for m in range(4, 41) for sa in range(8) for ll in (0, 1, 2, 3, 5, 7, 11, 17, 29, 40)]
        rng.shuffle(combos)
        for m, sa, ll in combos:
            if len(data) + ll + m > 65000: break
            lit = bytes(rng.integers(0, 256, ll, dtype=np.uint8).tobytes()) if seqs else pending_lit
            if seqs: data += lit
            dst = len(data)
            if variant % 2 == 0:                              # far: somewhere in the first half of what exists, aligned as asked
                src = (int(rng.integers(0, max(8, dst // 2 - 64))) & ~7) + sa
            else:                                             # near: within the last 200 bytes, not overlapping
                src = max(0, ((dst - m - int(rng.integers(0, 160))) & ~7) + sa - 8)
            if src + m > dst: src = ((dst - m) & ~7)
            if src < 0 or dst - src > 65535 or dst - src < m: continue
            data += data[src:src + m]
            seqs.append((lit, dst - src, m))
        tail = bytes(rng.integers(0, 256, 12 + variant, dtype=np.uint8).tobytes())
        data += tail
        raw = bytes(data)
        blob = _lz4_block(seqs, tail) if codec == LZ4 else _snappy_raw(len(raw), seqs, tail)
        want = oracle.lz4_decompress_raw(blob, len(raw)) if codec == LZ4 else oracle.snappy_decompress(blob)
        assert want == (len(raw), raw), (variant, want[0], len(raw))      # the hand-written stream is valid and means what was meant
        assert len(seqs) >= 600
        chunks.append((blob, raw))
    for flag in (N.FLAG_FORCE_LDS_PER_CHUNK, 0, N.FLAG_FORCE_WAVE_PER_CHUNK):
        for extra in (0, 19):
            streams = [c[0] for c in chunks] * 40                 # (a batch large enough for the parse kernel + decoder pipeline as well)
            raws = [c[1] for c in chunks] * 40
            res, outs = eng.batch_host(codec, DEC, flag, streams, [len(r) + extra for r in raws])
            bad = [i for i, (r, o, raw) in enumerate(zip(res, outs, raws)) if r != len(raw) or o != raw]
            assert not bad, (codec, flag, extra, bad[:8])
