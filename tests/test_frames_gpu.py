"""GPU parity for the framed formats: Snappy framing format through the C-ABI
(cj_snappy_frame_*) and through the reference's Python surface (cramjam.snappy.compress / decompress /
compress_into / decompress_into, /root/reference/src/snappy.rs:22-42,80-91), against the CPU oracle, the reference's
fixture and hand-built streams.  Restates /root/reference/tests/test_variants.py:48-245 for the snappy variant."""
import ctypes as C
import os
import random

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as st_np

import oracle
from conftest import GOLDEN_DIR
from framing import SNAPPY_IDENT, crc32c_masked, snappy_chunk, snappy_compressed, snappy_stored

pytestmark = pytest.mark.gpu

import cramjam_amd as cramjam  # noqa: E402
from cramjam_amd import _native as N  # noqa: E402

FAST = settings(max_examples=20, deadline=None)
E_EOF, E_WRITE, E_HDR, E_TYPE, E_LEN, E_SUM = -13, -14, -15, -16, -17, -18


def _fx(name):
    with open(os.path.join(GOLDEN_DIR, name), "rb") as f:
        return f.read()


def abi_decompress(framed, cap=None):
    L = N.lib()
    framed = bytes(framed)
    if cap is None:
        cap = max(L.cj_snappy_frame_decompress_len(framed, len(framed)), 0)
    out = C.create_string_buffer(max(cap, 1))
    r = L.cj_snappy_frame_decompress(framed, len(framed), out, cap)
    return r, out.raw[:max(r, 0)]


def abi_compress(data, cap=None):
    L = N.lib()
    data = bytes(data)
    cap = L.cj_snappy_frame_max_compress_len(len(data)) if cap is None else cap
    out = C.create_string_buffer(max(cap, 1))
    r = L.cj_snappy_frame_compress(data, len(data), out, cap)
    return r, out.raw[:max(r, 0)]


def mixed_data(seed, n_text=90, n_rand=200000, n_zero=50000):
    random.seed(seed)
    return _fx("plaintext.txt") * n_text + bytes(random.randrange(256) for _ in range(n_rand)) + bytes(n_zero)


def test_reference_fixture():
    framed, plain = _fx("plaintext.txt.snappy"), _fx("plaintext.txt")
    assert N.lib().cj_snappy_frame_decompress_len(framed, len(framed)) == len(plain)
    assert abi_decompress(framed) == (len(plain), plain)
    assert bytes(cramjam.snappy.decompress(framed)) == plain


@pytest.mark.parametrize("block_size", [None, 4096, 1000, 333, 7])
def test_decode_oracle_streams(block_size):
    """streams minted by the oracle with ragged piece sizes: unaligned payloads and piece offsets, stored + compressed"""
    data = mixed_data(5) if (block_size or 65536) >= 1000 else mixed_data(5, 3, 3000, 1000)
    r, framed = oracle.snappy_frame_compress(data, block_size=block_size)
    assert r > 0
    assert N.lib().cj_snappy_frame_decompress_len(framed, len(framed)) == len(data)
    assert abi_decompress(framed) == (len(data), data)


def test_encode_layout_and_round_trip():
    data = mixed_data(6)
    r, framed = abi_compress(data)
    assert r == len(framed) <= N.lib().cj_snappy_frame_max_compress_len(len(data))
    assert framed[:10] == SNAPPY_IDENT
    pos, off, types = 10, 0, set()
    while pos < len(framed):                    # one chunk per 64 KiB piece; stored iff the block did not shrink by 1/8
        ty, ln = framed[pos], int.from_bytes(framed[pos + 1:pos + 4], "little")
        piece = data[off:off + 65536]
        assert int.from_bytes(framed[pos + 4:pos + 8], "little") == crc32c_masked(piece), off
        body = framed[pos + 8:pos + 4 + ln]
        if ty == 1:
            assert body == piece
        else:
            assert ty == 0 and len(body) < len(piece) - len(piece) // 8
            assert oracle.snappy_decompress(body) == (len(piece), piece)
        types.add(ty); pos += 4 + ln; off += len(piece)
    assert off == len(data) and types == {0, 1}
    assert oracle.snappy_frame_decompress(framed) == (len(data), data)      # the CPU decoder accepts the GPU's stream
    assert abi_decompress(framed) == (len(data), data)
    assert abi_compress(data, cap=len(framed) - 1)[0] == E_WRITE
    assert abi_compress(data, cap=len(framed))[0] == len(framed)


def test_empty_and_hand_built_streams():
    assert abi_compress(b"") == (0, b"") and abi_decompress(b"") == (0, b"")
    a, b = b"hello hello hello hello", bytes(range(256)) * 3
    s = (SNAPPY_IDENT + snappy_stored(a) + snappy_chunk(0xfe, b"\0" * 13) + snappy_chunk(0x80, b"skip me")
         + SNAPPY_IDENT + snappy_compressed(b, oracle.snappy_compress(b)[1]) + snappy_chunk(0xfd, b""))
    assert abi_decompress(s) == (len(a) + len(b), a + b)
    assert abi_decompress(SNAPPY_IDENT) == (0, b"")


def test_malformed_streams_match_oracle():
    a = b"abcdefgh" * 40
    blk = oracle.snappy_compress(a)[1]
    good = SNAPPY_IDENT + snappy_compressed(a, blk)
    bad = bytearray(blk); bad[-1] ^= 0xff; bad[3] ^= 0x40
    big = oracle.snappy_compress(bytes(65537))[1]
    cases = [b"sknow", good[10:], b"\xff\x06\x00\x00sNaPpX" + good[10:], b"\xff\x05\x00\x00sNaPp" + good[10:], good[:10]]
    cases += [good[:cut] for cut in (1, 3, 11, 13, 15, 17, len(good) - 1)]
    cases += [SNAPPY_IDENT + snappy_chunk(0x02, b"xx"), SNAPPY_IDENT + snappy_chunk(0x7f, b""),
              SNAPPY_IDENT + snappy_chunk(0x00, b"abc"), SNAPPY_IDENT + b"\x01" + (76491).to_bytes(3, "little") + bytes(76491),
              SNAPPY_IDENT + snappy_stored(bytes(65537)), SNAPPY_IDENT + snappy_stored(bytes(65536)),
              SNAPPY_IDENT + snappy_stored(a, crc=1), SNAPPY_IDENT + snappy_compressed(a, blk, crc=crc32c_masked(a) ^ 1),
              SNAPPY_IDENT + snappy_chunk(0x00, b"\0\0\0\0"), SNAPPY_IDENT + snappy_compressed(bytes(65537), big),
              SNAPPY_IDENT + snappy_compressed(a, bytes(bad)),
              SNAPPY_IDENT + snappy_stored(a, crc=1) + snappy_chunk(0x02, b""),
              SNAPPY_IDENT + snappy_chunk(0x02, b"") + snappy_stored(a, crc=1),
              good + snappy_stored(a) + snappy_compressed(a, bytes(bad)) + snappy_stored(a, crc=3)]
    random.seed(11)
    for _ in range(60):                           # random single-byte damage anywhere in a 3-chunk stream
        s = bytearray(good + snappy_stored(a[:100]) + snappy_compressed(a, blk))
        s[random.randrange(len(s))] ^= 1 << random.randrange(8)
        cases.append(bytes(s))
    for s in cases:
        for cap in (70000, 100):
            want = oracle.snappy_frame_decompress(s, cap)
            got = abi_decompress(s, cap)
            assert got[0] == want[0], (s[:40], cap, got[0], want[0])
            if want[0] >= 0:
                assert got[1] == want[1]
    assert abi_decompress(good, cap=len(a) - 1)[0] == E_WRITE
    # validate-only mode names the first error in stream order without an output buffer
    L = N.lib()
    s = SNAPPY_IDENT + snappy_stored(a, crc=1) + snappy_chunk(0x02, b"")
    assert L.cj_snappy_frame_decompress(s, len(s), None, 0) == E_SUM
    assert L.cj_snappy_frame_decompress(good, len(good), None, 0) == len(a)


def test_large_stream_round_trip():
    """64 MiB of benchmark-like data: 1024 pieces in one batch (the large-batch decode pipeline), every piece checksummed"""
    n = 1024
    rng = np.random.default_rng(3)
    pieces = [oracle.synth_v1(65536, i) for i in range(64)]
    data = b"".join(pieces[i % 64] for i in range(n - 8)) + rng.integers(0, 256, 8 * 65536 - 12345, dtype=np.uint8).tobytes()
    r, framed = abi_compress(data)
    assert 0 < r < len(data)
    assert abi_decompress(framed) == (len(data), data)
    head = 10 + 8 + int.from_bytes(framed[11:14], "little") - 4
    assert oracle.snappy_frame_decompress(framed[:head]) == (65536, data[:65536])   # spot check with the CPU decoder
    dmg = bytearray(framed); dmg[len(framed) // 2] ^= 0x10                         # damage somewhere in the middle
    assert abi_decompress(bytes(dmg), cap=len(data))[0] == oracle.snappy_frame_decompress(bytes(dmg), len(data))[0] < 0


# ---- the reference's generic variant tests, restated for cramjam.snappy and cramjam.lz4 (tests/test_variants.py:48-245) ----
VARIANTS = ("snappy", "lz4")


def _oracle_decode(variant_str, framed):
    return (oracle.snappy_frame_decompress if variant_str == "snappy" else oracle.lz4_frame_decompress)(framed)[1]


def same_same(a, b):
    return bytes(a) == bytes(b)


@pytest.mark.parametrize("variant_str", VARIANTS)
@FAST
@given(arr=st_np.arrays(st_np.scalar_dtypes(), shape=st.integers(0, int(1e4))))
def test_variants_different_dtypes(variant_str, arr):
    variant = getattr(cramjam, variant_str)
    compressed = variant.compress(arr)
    assert same_same(variant.decompress(compressed), arr.tobytes())
    if arr.shape[0] % 2 == 0:
        arr = arr.reshape((2, -1))
        assert same_same(variant.decompress(variant.compress(arr)), arr.tobytes())


@pytest.mark.parametrize("is_bytearray", (True, False))
@pytest.mark.parametrize("variant_str", VARIANTS)
@FAST
@given(uncompressed=st.binary(min_size=1))
def test_variants_simple(variant_str, is_bytearray, uncompressed):
    variant = getattr(cramjam, variant_str)
    if is_bytearray:
        uncompressed = bytearray(uncompressed)
    compressed = variant.compress(uncompressed)
    assert compressed.read() != uncompressed
    compressed.seek(0)
    assert isinstance(compressed, cramjam.Buffer)
    assert _oracle_decode(variant_str, bytes(compressed)) == bytes(uncompressed)     # the CPU decoder reads the GPU's container
    decompressed = variant.decompress(compressed, output_len=len(uncompressed))
    assert same_same(decompressed.read(), uncompressed)
    assert isinstance(decompressed, cramjam.Buffer)


@pytest.mark.parametrize("variant_str", VARIANTS)
def test_variants_raise_exception(variant_str):
    with pytest.raises(cramjam.DecompressionError):
        getattr(cramjam, variant_str).decompress(b"sknow")


@pytest.mark.parametrize("variant_str", VARIANTS)
def test_file_in_file_out(variant_str, tmp_path):
    """compress a File into a File and back (reference generic! macro, src/lib.rs:237-265): both are streamed at their positions"""
    variant = getattr(cramjam, variant_str)
    data = mixed_data(31, 40, 30000, 5000)
    src = cramjam.File(str(tmp_path / "plain.bin")); src.write(data); src.seek(0)
    dst = cramjam.File(str(tmp_path / "packed.bin"))
    n = variant.compress_into(src, dst)
    assert n == len(dst) and dst.seek(0) == 0
    assert bytes(variant.decompress(dst)) == data          # File as `data`: read from its position to the end
    dst.seek(0)
    back = cramjam.File(str(tmp_path / "back.bin"))
    assert variant.decompress_into(dst, back) == len(data) and back.seek(0) == 0 and back.read() == data


@pytest.mark.parametrize("variant_str", VARIANTS)
def test_output_len_is_a_floor(variant_str):
    # generic!: vec![0; output_len] under a Cursor -> never shorter than output_len (src/lib.rs:216-219)
    variant = getattr(cramjam, variant_str)
    out = variant.decompress(variant.compress(b"abc" * 10), output_len=100)
    assert len(out) == 100 and bytes(out)[:30] == b"abc" * 10 and bytes(out)[30:] == bytes(70)
    out = variant.decompress(variant.compress(b"abc" * 10), output_len=5)
    assert bytes(out) == b"abc" * 10


def _make(kind, payload, tmp=None):
    # NB: the reference's tests write THROUGH immutable `bytes` / `memoryview(bytes)` outputs (CPython only).  CPython shares
    # one object for b"" and for every 1-byte bytes value, so doing that with a 0/1-byte output would corrupt an
    # interpreter-wide singleton (it made this suite fail once in ~20 runs); such outputs are made of 2+ bytes here.
    if kind is cramjam.File:
        f = cramjam.File(str(tmp.mktemp("f").joinpath("data.bin")))
        f.write(payload); f.seek(0)
        return f
    if kind == "numpy":
        return np.frombuffer(payload, dtype=np.uint8).copy()
    if kind is cramjam.Buffer:
        b = cramjam.Buffer(); b.write(payload); b.seek(0)
        return b
    return kind(payload)


def _collect(obj):
    if isinstance(obj, (cramjam.Buffer, cramjam.File)):
        obj.seek(0)
        return obj.read()
    return obj.tobytes() if hasattr(obj, "tobytes") else bytes(obj)


TYPES = (bytes, bytearray, "numpy", cramjam.Buffer, cramjam.File, memoryview)


@pytest.mark.parametrize("input_type", TYPES)
@pytest.mark.parametrize("output_type", TYPES)
@pytest.mark.parametrize("variant_str", VARIANTS)
@settings(max_examples=6, deadline=None)
@given(raw_data=st.binary())
def test_variants_compress_into(variant_str, input_type, output_type, raw_data, tmp_path_factory):
    variant = getattr(cramjam, variant_str)
    inp = _make(input_type, raw_data, tmp_path_factory)
    compressed_len = len(variant.compress(raw_data))
    output = (cramjam.Buffer() if output_type is cramjam.Buffer else _make(cramjam.File, b"", tmp_path_factory) if output_type is cramjam.File
              else _make(output_type, b"0" * compressed_len))
    n_bytes = variant.compress_into(inp, output)
    assert n_bytes == compressed_len
    assert same_same(raw_data, variant.decompress(_collect(output)[:n_bytes]))


@pytest.mark.parametrize("input_type", TYPES)
@pytest.mark.parametrize("output_type", TYPES)
@pytest.mark.parametrize("variant_str", VARIANTS)
@settings(max_examples=6, deadline=None)
@given(raw_data=st.binary())
def test_variants_decompress_into(variant_str, input_type, output_type, raw_data, tmp_path_factory):
    variant = getattr(cramjam, variant_str)
    compressed = bytes(variant.compress(raw_data))
    inp = _make(input_type, compressed, tmp_path_factory)
    pad = 2 if output_type in (bytes, memoryview) and len(raw_data) < 2 else 0        # never write into a shared 0/1-byte object
    output = (cramjam.Buffer() if output_type is cramjam.Buffer else _make(cramjam.File, b"", tmp_path_factory) if output_type is cramjam.File
              else _make(output_type, b"0" * (len(raw_data) + pad)))
    n_bytes = variant.decompress_into(inp, output)
    assert n_bytes == len(raw_data)
    assert same_same(_collect(output)[:n_bytes], raw_data)


@pytest.mark.parametrize("variant_str", VARIANTS)
def test_into_output_too_small(variant_str):
    variant = getattr(cramjam, variant_str)
    data = b"some bytes here" * 100
    with pytest.raises(cramjam.CompressionError):
        variant.compress_into(data, bytearray(10))
    with pytest.raises(cramjam.DecompressionError):
        variant.decompress_into(variant.compress(data), bytearray(len(data) - 1))


# ---- LZ4 frame through the C-ABI ----------------------------------------------------------------------------------

def lz4f_decompress(framed, cap=None):
    L = N.lib()
    framed = bytes(framed)
    if cap is None:
        cap = max(L.cj_lz4_frame_decompress_bound(framed, len(framed)), 0)
    out = C.create_string_buffer(max(cap, 1))
    r = L.cj_lz4_frame_decompress(framed, len(framed), out, cap)
    return r, out.raw[:max(r, 0)]


def lz4f_compress(data, cap=None, level=-1):
    L = N.lib()
    data = bytes(data)
    cap = L.cj_lz4_frame_compress_bound(len(data)) if cap is None else cap
    out = C.create_string_buffer(max(cap, 1))
    r = L.cj_lz4_frame_compress(data, len(data), out, cap, level)
    return r, out.raw[:max(r, 0)]


def test_lz4_frame_fixture_and_golden_frames():
    """reference fixture + 50 frames minted by liblz4's LZ4F: linked and independent blocks, all block sizes, checksums"""
    import base64, hashlib, importlib.util, json
    framed, plain = _fx("plaintext.txt.lz4"), _fx("plaintext.txt")
    assert lz4f_decompress(framed) == (len(plain), plain)
    assert bytes(cramjam.lz4.decompress(framed)) == plain
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(GOLDEN_DIR, "make_golden_frames.py"))
    mgf = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgf)
    g = json.load(open(os.path.join(GOLDEN_DIR, "golden_frames.json")))
    kinds = set()
    lds0 = N.lib().cj_debug_linked_lds_frames()
    for v in g["vectors"]:
        data = mgf.content(v["kind"], v["n"])
        frame = base64.b64decode(v["frame"])
        assert hashlib.sha256(data).hexdigest() == v["sha256"]
        assert N.lib().cj_lz4_frame_decompress_bound(frame, len(frame)) == oracle.lz4_frame_decompress_bound(frame)
        r, out = lz4f_decompress(frame)
        assert r == len(data) and out == data, {k: v[k] for k in v if k != "frame"}
        kinds.add((frame[4] >> 5) & 1)
    assert kinds == {0, 1}                      # both linked and independent (batch) frames were exercised
    # linked frames with 64 KiB blocks took the two-window LDS decoder, the other linked ones the chain kernel
    assert N.lib().cj_debug_linked_lds_frames() - lds0 >= 6


def test_lz4_frame_linked_blocks_at_scale():
    """8 MiB linked-block frame (128 blocks) with matches reaching across block boundaries, long runs straddling block
    starts, stored blocks in the middle: the two-window LDS decoder against the CPU oracle's bytes"""
    rng = np.random.default_rng(9)
    parts = [oracle.synth_v1(65536, i) for i in range(24)]
    body = bytearray()
    for i in range(128):
        kind = i % 8
        if kind == 5:
            body += bytes(65536)                               # zero run: every block starts with a match that straddles the boundary
        elif kind == 6:
            body += rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()      # incompressible: stored block
        elif kind == 7:
            body += bytes(body[-70000:-70000 + 65536]) if len(body) > 70000 else parts[0]    # far copies of the previous block
        else:
            body += parts[int(rng.integers(0, 24))]
    data = bytes(body[:128 * 65536 - 4321])
    r, frame = oracle.lz4_frame_compress(data, 4, 1)          # linked, 64 KiB blocks
    assert r > 0 and (frame[4] >> 5) & 1 == 0
    lds0 = N.lib().cj_debug_linked_lds_frames()
    got = lz4f_decompress(frame)
    assert got[0] == len(data) and got[1] == data
    assert N.lib().cj_debug_linked_lds_frames() == lds0 + 1
    assert bytes(cramjam.lz4.decompress(frame)) == data
    dmg = bytearray(frame); dmg[len(frame) // 2] ^= 0x08     # damage -> same verdict as the oracle (falls back to the chain kernel)
    assert lz4f_decompress(bytes(dmg), cap=len(data))[0] == oracle.lz4_frame_decompress(bytes(dmg), len(data))[0]


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 7, 9])
@pytest.mark.parametrize("bs", [4, 5, 7])
def test_lz4_frame_decode_oracle_frames(flags, bs):
    data = mixed_data(7 + flags, 200, 70000, 100000)
    r, frame = oracle.lz4_frame_compress(data, bs, flags)
    assert lz4f_decompress(frame) == (len(data), data)


def test_lz4_frame_encode_layout_and_round_trip():
    data = mixed_data(8)
    r, frame = lz4f_compress(data)
    assert r == len(frame) <= N.lib().cj_lz4_frame_compress_bound(len(data))
    assert frame[:6] == b"\x04\x22\x4d\x18\x64\x40" and frame[6] == (oracle.xxh32(frame[4:6]) >> 8) & 0xff
    pos, off, kinds = 7, 0, set()
    while True:                                   # 64 KiB independent blocks; stored iff the block did not shrink
        w = int.from_bytes(frame[pos:pos + 4], "little"); pos += 4
        if w == 0:
            break
        sz, stored = w & 0x7FFFFFFF, w >> 31
        piece = data[off:off + 65536]
        if stored:
            assert frame[pos:pos + sz] == piece
        else:
            assert sz < len(piece) and oracle.lz4_decompress_raw(frame[pos:pos + sz], len(piece)) == (len(piece), piece)
        kinds.add(stored); pos += sz; off += len(piece)
    assert off == len(data) and kinds == {0, 1}
    assert int.from_bytes(frame[pos:pos + 4], "little") == oracle.xxh32(data) and pos + 4 == len(frame)
    assert oracle.lz4_frame_decompress(frame) == (len(data), data)           # the CPU decoder accepts the GPU's frame
    assert lz4f_decompress(frame) == (len(data), data)
    assert lz4f_compress(data, cap=len(frame) - 1)[0] == E_WRITE and lz4f_compress(data, cap=len(frame))[0] == len(frame)
    assert lz4f_compress(b"") == (15, bytes.fromhex("04224d186440a700000000055dcc02"))       # = liblz4's empty frame
    assert lz4f_decompress(lz4f_compress(b"")[1]) == (0, b"")
    for level in (0, 4, 12):
        assert lz4f_decompress(lz4f_compress(data[:100000], level=level)[1])[1] == data[:100000]


def test_lz4_frame_malformed_matches_oracle():
    data = _fx("plaintext.txt") * 100
    cases = []
    for flags in (2, 8, 4 | 8, 1 | 2):
        r, fr = oracle.lz4_frame_compress(data, 4, flags)
        cases += [fr, fr[:3], fr[:6], fr[:8], fr[:12], fr[:len(fr) - 5], fr[:len(fr) - 1], fr + b"trailing"]
        for pos, mask in ((4, 0x80), (4, 0x02), (5, 0x70), (6, 1), (12, 0xFF), (13, 0xFF), (20, 1), (len(fr) - 1, 1), (len(fr) - 6, 0x40), (7, 0x10), (9, 0x01)):
            b = bytearray(fr); b[pos] ^= mask
            cases.append(bytes(b))
    cases += [b"sknow", b"sknowsknow", b"", b"\x50\x2a\x4d\x18\x03\x00\x00\x00abc" + cases[0], b"\x50\x2a\x4d\x18\x09\x00\x00\x00abc"]
    random.seed(13)
    r, fr = oracle.lz4_frame_compress(data + bytes(random.randrange(256) for _ in range(3000)), 4, 1)      # linked, 2 blocks
    for _ in range(40):
        b = bytearray(fr); b[random.randrange(len(b))] ^= 1 << random.randrange(8)
        cases.append(bytes(b))
    for s in cases:
        for cap in (200000, 100):
            want = oracle.lz4_frame_decompress(s, cap)
            got = lz4f_decompress(s, cap)
            assert got[0] == want[0], (s[:24].hex(), len(s), cap, got[0], want[0])
            if want[0] >= 0:
                assert got[1] == want[1]


def test_lz4_frame_large_round_trip():
    """64 MiB: 1024 independent 64 KiB blocks through the batch engine, host XXH32 overlapped"""
    rng = np.random.default_rng(5)
    pieces = [oracle.synth_v1(65536, i) for i in range(64)]
    data = b"".join(pieces[i % 64] for i in range(1016)) + rng.integers(0, 256, 8 * 65536 - 777, dtype=np.uint8).tobytes()
    r, frame = lz4f_compress(data)
    assert 0 < r < len(data)
    assert lz4f_decompress(frame) == (len(data), data)
    dmg = bytearray(frame); dmg[len(frame) // 3] ^= 0x04
    assert lz4f_decompress(bytes(dmg), cap=len(data))[0] == oracle.lz4_frame_decompress(bytes(dmg), len(data))[0] < 0


# ---- streaming objects (reference tests/test_variants.py:361-417, restated for snappy + lz4) ------------------------

@pytest.mark.parametrize("variant_str", VARIANTS)
@FAST
@given(first=st.binary(), second=st.binary())
def test_streams_compressor(variant_str, first, second):
    mod = getattr(cramjam, variant_str)
    compressor = mod.Compressor()
    compressor.compress(first)
    out = bytes(compressor.flush())
    compressor.compress(second)
    out += bytes(compressor.flush())
    out += bytes(compressor.finish())
    assert same_same(bytes(mod.decompress(out)), first + second)
    assert _oracle_decode(variant_str, out) == first + second             # the CPU decoder reads the concatenated flushes
    assert bytes(compressor.finish()) == b""                               # just empty bytes after the first .finish()
    with pytest.raises(cramjam.CompressionError):
        compressor.compress(b"data")


@pytest.mark.parametrize("variant_str", VARIANTS)
def test_variants_stream_decompressors(variant_str):
    variant = getattr(cramjam, variant_str)
    decompressor = variant.Decompressor()
    compressed = variant.compress(b"bytes")
    for _ in range(2):
        assert decompressor.decompress(bytes(compressed)) == 5
    assert len(decompressor) == 10 and decompressor.len() == 10 and b"sby" in decompressor and bool(decompressor)
    assert repr(decompressor) == "Decompressor<len=10>"
    assert bytes(decompressor.flush()) == b"bytesbytes"
    assert bytes(decompressor.flush()) == b""
    decompressor.decompress(bytes(compressed))
    assert bytes(decompressor.finish()) == b"bytes"
    with pytest.raises(cramjam.DecompressionError):
        decompressor.finish()
    with pytest.raises(cramjam.DecompressionError):
        decompressor.decompress(bytes(compressed))


def test_stream_compressor_large_and_options():
    data = mixed_data(21)
    for mod, kw in ((cramjam.snappy, {}), (cramjam.lz4, {}), (cramjam.lz4, dict(level=9, content_checksum=False, block_linked=False))):
        c = mod.Compressor(**kw)
        out = b""
        for i in range(0, len(data), 100_000):                   # pieces that do not line up with the 64 KiB blocks
            assert c.compress(data[i:i + 100_000]) == len(data[i:i + 100_000])
            if (i // 100_000) % 2:
                out += bytes(c.flush())
        out += bytes(c.finish())
        assert bytes(mod.decompress(out)) == data
        assert _oracle_decode("snappy" if mod is cramjam.snappy else "lz4", out) == data
    empty = cramjam.lz4.Compressor()
    assert bytes(empty.finish()) == bytes.fromhex("04224d186440a700000000055dcc02")      # liblz4's empty frame
    assert bytes(cramjam.snappy.Compressor().finish()) == b""


def test_concurrent_calls_from_python_threads():
    """every hot call releases the GIL (src/lib.rs allow_threads), so Python may enter the exports from many threads at once
    (SURVEY.md §8b "Threading"): results must be the same as when called alone"""
    import threading
    datas = [mixed_data(40 + i, 20 + i, 40000 + 1000 * i, 3000) for i in range(6)]
    errors = []

    def worker(i):
        try:
            d = datas[i % len(datas)]
            for _ in range(4):
                assert bytes(cramjam.snappy.decompress(cramjam.snappy.compress(d))) == d
                assert bytes(cramjam.lz4.decompress(cramjam.lz4.compress(d))) == d
                blk = cramjam.lz4.compress_block(d[:200000])
                assert bytes(cramjam.lz4.decompress_block(blk)) == d[:200000]
                raw = cramjam.snappy.compress_raw(d[:150000])
                assert bytes(cramjam.snappy.decompress_raw(raw)) == d[:150000]
                assert bytes(cramjam.lz4.decompress(oracle.lz4_frame_compress(d, 4, 1)[1])) == d      # linked path (own scratch lock)
        except Exception as exc:                      # noqa: BLE001
            errors.append((i, repr(exc)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors
