"""The reference's hot-path tests, restated against cramjam_amd on the GPU:
  /root/reference/tests/test_variants.py:247-266  test_variant_snappy_raw_into
  /root/reference/tests/test_variants.py:269-289  test_variant_lz4_block_into
  /root/reference/tests/test_variants.py:314-341  test_lz4_block (byte-exact known answers)
  /root/reference/tests/test_integration.py:70-102 test_lz4_decompress_block_into_non_prepended_size
  /root/reference/benchmarks/test_bench.py:245-266,97-121 round trips (on reduced synthetic inputs)
Decoded bytes are compared bit-exactly; blocks produced by the GPU encoders are additionally decoded
by the CPU oracle (the stand-in for the reference's CPU decoder)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle
import cramjam_amd as cramjam

pytestmark = pytest.mark.gpu
FAST = settings(max_examples=25, deadline=None)


def same_same(a, b):
    return bytes(a) == bytes(b)


@pytest.mark.parametrize(
    "compress_kwargs",
    (
        dict(mode="default", acceleration=1, compression=1, store_size=True),
        dict(mode="fast", acceleration=2, compression=2, store_size=False),
        dict(mode="high_compression", acceleration=3, compression=3, store_size=True),
        dict(mode="default", acceleration=5, compression=4, store_size=False),
    ),
)
def test_lz4_block(compress_kwargs):
    lz4 = cramjam.lz4
    data = b"howdy neighbor"
    assert bytes(lz4.compress_block(data)) == b"\x0e\x00\x00\x00\xe0howdy neighbor"
    assert bytes(lz4.compress_block(data, store_size=False)) == b"\xe0howdy neighbor"
    out = lz4.decompress_block(
        lz4.compress_block(data, **compress_kwargs),
        output_len=len(data) if not compress_kwargs["store_size"] else None,
    )
    assert isinstance(out, cramjam.Buffer) and same_same(out, data)


def test_decompress_block_output_len_is_capacity_not_truncated():
    # src/lz4.rs:86-89: `.map(|_| buf)` keeps the n-byte buffer
    blk = cramjam.lz4.compress_block(b"abcd" * 10, store_size=False)
    out = cramjam.lz4.decompress_block(blk, output_len=100)
    assert len(out) == 100 and bytes(out)[:40] == b"abcd" * 10 and bytes(out)[40:] == bytes(60)
    with pytest.raises(cramjam.DecompressionError):
        cramjam.lz4.decompress_block(blk, output_len=10)
    with pytest.raises(cramjam.DecompressionError):
        cramjam.lz4.decompress_block(b"\x01\x02")            # shorter than the size prefix
    with pytest.raises(cramjam.DecompressionError):
        cramjam.lz4.decompress_block(b"\x10\x00\x00\x00\xff\xff\xff")


@FAST
@given(data=st.binary())
def test_variant_lz4_block_into(data):
    compressed = cramjam.lz4.compress_block(data)
    compressed_size = cramjam.lz4.compress_block_bound(data)
    compressed_buffer = np.zeros(compressed_size, dtype=np.uint8)
    n_bytes = cramjam.lz4.compress_block_into(data, compressed_buffer)
    assert n_bytes == len(compressed)
    assert same_same(compressed, compressed_buffer[:n_bytes])
    assert oracle.lz4_block_decompress(bytes(compressed), len(data), True) == (len(data), data)

    decompressed_buffer = np.zeros(len(data), dtype=np.uint8)
    n_bytes = cramjam.lz4.decompress_block_into(compressed_buffer[:n_bytes].tobytes(), decompressed_buffer)
    assert n_bytes == len(data)
    assert same_same(decompressed_buffer[:n_bytes], data)


@FAST
@given(data=st.binary(min_size=1, max_size=int(1e5)))
@pytest.mark.parametrize("set_output_len", (True, False))
def test_lz4_decompress_block_into_non_prepended_size(data, set_output_len):
    compressed = cramjam.lz4.compress_block(data, store_size=False)
    output_len = len(data) if set_output_len else None
    with pytest.raises(cramjam.DecompressionError):
        cramjam.lz4.decompress_block_into(compressed, bytearray(0), output_len=output_len)
    match = f"output_len set to {len(data)}, but output is less"
    with pytest.raises(cramjam.DecompressionError, match=match):
        cramjam.lz4.decompress_block_into(compressed, bytearray(0), output_len=len(data))
    out = bytearray(len(data))
    n = cramjam.lz4.decompress_block_into(compressed, out, output_len=output_len)
    assert same_same(out, data)
    out = bytearray(len(compressed) * 2)
    n = cramjam.lz4.decompress_block_into(compressed, out, output_len=output_len)
    assert same_same(out[:n], data)


@FAST
@given(data=st.binary())
def test_variant_snappy_raw_into(data):
    compressed = cramjam.snappy.compress_raw(data)
    compressed_size = cramjam.snappy.compress_raw_max_len(data)
    compressed_buffer = np.zeros(compressed_size, dtype=np.uint8)
    n_bytes = cramjam.snappy.compress_raw_into(data, compressed_buffer)
    assert n_bytes == len(compressed)
    assert oracle.snappy_decompress(bytes(compressed)) == (len(data), data)
    decompressed_buffer = np.zeros(len(data), dtype=np.uint8)
    n_bytes = cramjam.snappy.decompress_raw_into(compressed_buffer[:n_bytes].tobytes(), decompressed_buffer)
    assert n_bytes == len(data)
    assert same_same(decompressed_buffer[:n_bytes], data)


def test_snappy_raw_errors():
    with pytest.raises(cramjam.DecompressionError):
        cramjam.snappy.decompress_raw(b"")
    with pytest.raises(cramjam.DecompressionError):
        cramjam.snappy.decompress_raw(b"\x05\x0cab")
    with pytest.raises(cramjam.CompressionError):
        cramjam.snappy.compress_raw_into(b"abcdef", bytearray(8))      # < max_compress_len
    with pytest.raises(cramjam.DecompressionError):
        cramjam.snappy.decompress_raw_into(bytes(cramjam.snappy.compress_raw(b"x" * 100)), bytearray(99))
    assert bytes(cramjam.snappy.compress_raw(b"")) == b"\x00"
    assert bytes(cramjam.snappy.decompress_raw(b"\x00")) == b""


@pytest.mark.parametrize("kind", ["repeating", "random", "text"])
@pytest.mark.parametrize("codec", ["lz4_block", "snappy_raw"])
def test_bench_round_trips(kind, codec, plaintext):
    # benchmarks/test_bench.py:38-60 uses 54 MB inputs; same constructions at 5.4 MB (seeded)
    if kind == "repeating":
        data = b"oh what a beautiful morning, oh what a beautiful day!!" * 100_000
    elif kind == "random":
        data = np.random.default_rng(7).integers(0, 255, size=5_400_000, dtype=np.uint8).tobytes()
    else:
        data = plaintext * 300
    if codec == "lz4_block":
        comp = cramjam.lz4.compress_block(data)
        assert oracle.lz4_block_decompress(bytes(comp), len(data), True) == (len(data), data)
        back = cramjam.lz4.decompress_block(comp)
        back2 = cramjam.lz4.decompress_block(bytes(comp)[4:], output_len=len(data))
        assert same_same(back2, data)
    else:
        comp = cramjam.snappy.compress_raw(data)
        assert oracle.snappy_decompress(bytes(comp)) == (len(data), data)
        back = cramjam.snappy.decompress_raw(comp)
    assert same_same(back, data)
    if kind != "random":
        assert len(comp) < len(data) // 3


def test_inputs_accept_all_bytes_like(plaintext):
    # BytesType: bytes, bytearray, numpy, memoryview, Buffer (reference src/lib.rs:104-148)
    ref = bytes(cramjam.lz4.compress_block(plaintext))
    for variant in (bytearray(plaintext), np.frombuffer(plaintext, dtype=np.uint8), memoryview(plaintext), cramjam.Buffer(plaintext)):
        assert bytes(cramjam.lz4.compress_block(variant)) == ref
    out = cramjam.Buffer()
    out.set_len(len(plaintext))
    assert cramjam.lz4.decompress_block_into(ref, out) == len(plaintext) and bytes(out) == plaintext


def test_batch_extension(plaintext):
    chunks = [oracle.synth_v1(65536, i) for i in range(40)] + [plaintext, b"", b"a"]
    res, blocks = cramjam.batch.lz4_compress_blocks(chunks, store_size=False)
    assert all(r > 0 for r in res)
    res2, outs = cramjam.batch.lz4_decompress_blocks(blocks, [len(c) for c in chunks])
    assert res2 == [len(c) for c in chunks] and outs == chunks
    res, blocks = cramjam.batch.snappy_compress_raw_many(chunks)
    res2, outs = cramjam.batch.snappy_decompress_raw_many(blocks)
    assert res2 == [len(c) for c in chunks] and outs == chunks
    # one corrupt block does not poison its neighbours
    bad = list(blocks)
    bad[1] = bad[1][:50]
    res3, outs3 = cramjam.batch.snappy_decompress_raw_many(bad)
    assert res3[1] < 0 and res3[0] == 65536 and outs3[2] == chunks[2]
