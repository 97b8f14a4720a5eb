"""ONE large buffer through the single-buffer API (csrc/large.hip): compress_block / compress_raw cut the input into
64 KiB pieces, compress them as a batch and join the pieces' streams into one valid block.  The judge is the oracle's
decoder (the reference's bar for compressed output: it must decode losslessly — reference tests/test_variants.py
round trips); sizes and ratios are checked against a batch of independent 64 KiB chunks."""
import random

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

import cramjam_amd as cramjam  # noqa: E402

PIECE = 65536


def payloads():
    rnd = random.Random(5)
    rb = lambda n: rnd.randbytes(n)
    text = b"".join(b"line %d: the quick brown fox jumps over the lazy dog\n" % (i * 7919 % 10007) for i in range(40000))
    synth = b"".join(oracle.synth_v1(PIECE, i) for i in range(40))
    yield "just-over-one-piece", text[:PIECE + 1]
    yield "two-pieces-exact", text[:2 * PIECE]
    yield "synth-2.5MiB", synth
    yield "zeros", bytes(5 * PIECE + 17)
    yield "random-no-match", rb(3 * PIECE + 100)
    yield "random-then-text", rb(PIECE + 500) + text[:PIECE]
    yield "text-random-text", text[:PIECE - 7] + rb(2 * PIECE + 7) + text[:30000]
    yield "tiny-last-piece", text[:2 * PIECE + 3]
    yield "random-tiny-last", rb(PIECE) + b"ab"
    yield "long-run-of-15s", (b"\xff" * 300 + rb(40)) * 700


CASES = list(payloads())
IDS = [c[0] for c in CASES]


@pytest.mark.parametrize("name,data", CASES, ids=IDS)
@pytest.mark.parametrize("store_size", [True, False])
def test_lz4_compress_block_large(name, data, store_size):
    blob = bytes(cramjam.lz4.compress_block(data, store_size=store_size))
    body = blob[4:] if store_size else blob
    if store_size:
        assert int.from_bytes(blob[:4], "little") == len(data)
    r, out = oracle.lz4_decompress_raw(body, len(data))
    assert r == len(data) and out == data
    assert bytes(cramjam.lz4.decompress_block(blob, output_len=None if store_size else len(data))) == data
    bound = cramjam.lz4.compress_block_bound(data)
    assert len(blob) <= bound
    # same pieces as a framed / batched 64 KiB split would produce: the joined block must not be larger than their sum
    pieces = [data[i:i + PIECE] for i in range(0, len(data), PIECE)]
    assert len(body) <= sum(len(bytes(cramjam.lz4.compress_block(p, store_size=False))) for p in pieces)


@pytest.mark.parametrize("name,data", CASES, ids=IDS)
def test_snappy_compress_raw_large(name, data):
    blob = bytes(cramjam.snappy.compress_raw(data))
    r, out = oracle.snappy_decompress(blob, len(data))
    assert r == len(data) and out == data
    assert bytes(cramjam.snappy.decompress_raw(blob)) == data
    assert len(blob) <= cramjam.snappy.compress_raw_max_len(data)


def test_into_variants_and_small_outputs():
    data = b"".join(oracle.synth_v1(PIECE, i) for i in range(6))[:-999]
    out = np.zeros(cramjam.lz4.compress_block_bound(data), dtype=np.uint8)
    n = cramjam.lz4.compress_block_into(data, out)
    assert bytes(cramjam.lz4.decompress_block(out[:n].tobytes())) == data
    out = np.zeros(cramjam.snappy.compress_raw_max_len(data), dtype=np.uint8)
    n = cramjam.snappy.compress_raw_into(data, out)
    assert bytes(cramjam.snappy.decompress_raw(out[:n].tobytes())) == data
    small = np.zeros(1000, dtype=np.uint8)
    with pytest.raises(cramjam.CompressionError):
        cramjam.lz4.compress_block_into(data, small)
    with pytest.raises(cramjam.CompressionError):
        cramjam.snappy.compress_raw_into(data, small)


# ---------------------------------------------------------------------------------------------------------------------
# decompress: ONE large stream (big_parse.hip + the slab mode of the LDS decoder).  Streams come from the oracle's
# encoders (liblz4-style: matches reach back across every 64 KiB boundary) and from the GPU's own piece-wise encoder.

def lz4_streams(data):
    yield "oracle", oracle.lz4_compress_raw(data)[1]
    yield "gpu", bytes(cramjam.lz4.compress_block(data, store_size=False))


@pytest.mark.parametrize("name,data", CASES, ids=IDS)
def test_lz4_decompress_block_large(name, data):
    for src, blob in lz4_streams(data):
        if len(blob) <= PIECE:
            continue                                      # small stream: the ordinary single-chunk path
        assert bytes(cramjam.lz4.decompress_block(blob, output_len=len(data))) == data, src
        pre = len(data).to_bytes(4, "little") + blob
        assert bytes(cramjam.lz4.decompress_block(pre)) == data, src
        out = np.zeros(len(data) + 100, dtype=np.uint8)
        n = cramjam.lz4.decompress_block_into(pre, out)
        assert n == len(data) and out[:n].tobytes() == data, src


@pytest.mark.parametrize("name,data", CASES, ids=IDS)
def test_snappy_decompress_raw_large(name, data):
    for src, blob in (("oracle", oracle.snappy_compress(data)[1]), ("gpu", bytes(cramjam.snappy.compress_raw(data)))):
        if len(blob) <= PIECE:
            continue
        assert bytes(cramjam.snappy.decompress_raw(blob)) == data, src
        out = np.zeros(len(data), dtype=np.uint8)
        assert cramjam.snappy.decompress_raw_into(blob, out) == len(data) and out.tobytes() == data, src


def test_large_decompress_verdicts_match_the_oracle():
    rnd = random.Random(77)
    data = b"".join(oracle.synth_v1(PIECE, i) for i in range(5)) + bytes(70000) + rnd.randbytes(90000)
    n = len(data)
    lz = oracle.lz4_compress_raw(data)[1]
    sn = oracle.snappy_compress(data)[1]
    for t in range(40):
        b = bytearray(lz); i = rnd.randrange(len(b)); b[i] ^= 1 << rnd.randrange(8)
        if t % 5 == 0: b = b[:rnd.randrange(PIECE + 1, len(b))]
        er, eo = oracle.lz4_decompress_raw(bytes(b), n)
        try:
            got = bytes(cramjam.lz4.decompress_block(bytes(b), output_len=n))      # Some(n): length n, zero tail (src/lz4.rs:78-95)
            assert er >= 0 and got[:er] == eo[:er] and got[er:] == bytes(n - er), ("lz4", t, er)
        except cramjam.DecompressionError:
            assert er < 0, ("lz4", t, er)
    for t in range(40):
        b = bytearray(sn); i = rnd.randrange(len(b)); b[i] ^= 1 << rnd.randrange(8)
        if t % 5 == 0: b = b[:rnd.randrange(PIECE + 1, len(b))]
        er, eo = oracle.snappy_decompress(bytes(b), n + 64)
        try:
            got = bytes(cramjam.snappy.decompress_raw(bytes(b)))
            assert er >= 0 and got == eo[:er], ("snappy", t, er)
        except cramjam.DecompressionError:
            assert er < 0, ("snappy", t, er)
    # capacity rules
    assert bytes(cramjam.lz4.decompress_block(lz, output_len=n + 1000)) == data + bytes(1000)
    with pytest.raises(cramjam.DecompressionError):
        cramjam.lz4.decompress_block(lz, output_len=n - 1)
    with pytest.raises(cramjam.DecompressionError):
        cramjam.snappy.decompress_raw_into(sn, np.zeros(n - 1, dtype=np.uint8))
