"""CPU: the framed-format oracle (oracle/snappy_frame_oracle.c) against the reference's fixture, the CRC-32C check
value and hand-built streams; and the lane-parallel CRC-32C of the HIP kernel (crc32c_lanes.hpp) built for the host."""
import ctypes as C
import os
import random
import subprocess

import pytest

import oracle
from conftest import GOLDEN_DIR, ROOT
from framing import SNAPPY_IDENT, crc32c, crc32c_masked, snappy_chunk, snappy_compressed, snappy_stored

E_EOF, E_WRITE, E_HDR, E_TYPE, E_LEN, E_SUM = -13, -14, -15, -16, -17, -18


def _fx(name):
    with open(os.path.join(GOLDEN_DIR, name), "rb") as f:
        return f.read()


def test_crc32c_known_answers():
    assert oracle.crc32c(b"123456789") == 0xE3069283          # the standard CRC-32C check value
    assert oracle.crc32c(b"") == 0
    assert oracle.crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4 test patterns
    assert oracle.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert oracle.crc32c(bytes(range(32))) == 0x46DD794E
    random.seed(1)
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 1000):
        d = bytes(random.randrange(256) for _ in range(n))
        assert oracle.crc32c(d) == crc32c(d)
        assert oracle.crc32c(d, masked=True) == crc32c_masked(d)


def test_reference_fixture_decodes():
    """reference tests/data/integration/plaintext.txt.snappy (third-party framed stream) -> plaintext.txt"""
    framed, plain = _fx("plaintext.txt.snappy"), _fx("plaintext.txt")
    assert framed[:10] == SNAPPY_IDENT and framed[10] == 0x00
    assert int.from_bytes(framed[14:18], "little") == oracle.crc32c(plain, masked=True)
    assert oracle.snappy_frame_decompress_len(framed) == len(plain)
    r, out = oracle.snappy_frame_decompress(framed)
    assert r == len(plain) and out == plain


@pytest.mark.parametrize("block_size", [None, 65536, 4096, 1000, 7])
def test_round_trip_and_layout(block_size):
    random.seed(5)
    data = _fx("plaintext.txt") * 90 + bytes(random.randrange(256) for _ in range(200000)) + bytes(50000)
    r, framed = oracle.snappy_frame_compress(data, block_size=block_size)
    assert r == len(framed) and framed[:10] == SNAPPY_IDENT
    bs = block_size or 65536
    pos, off, types = 10, 0, set()
    while pos < len(framed):                                  # layout: one chunk per piece, stored iff it does not shrink by 1/8
        ty, ln = framed[pos], int.from_bytes(framed[pos + 1:pos + 4], "little")
        piece = data[off:off + bs]
        assert int.from_bytes(framed[pos + 4:pos + 8], "little") == crc32c_masked(piece)
        body = framed[pos + 8:pos + 4 + ln]
        if ty == 1:
            assert body == piece
            assert len(oracle.snappy_compress(piece)[1]) >= len(piece) - len(piece) // 8
        else:
            assert ty == 0 and oracle.snappy_decompress(body)[1] == piece and len(body) < len(piece) - len(piece) // 8
        types.add(ty); pos += 4 + ln; off += len(piece)
    assert off == len(data) and (block_size == 7 or types == {0, 1})
    assert oracle.snappy_frame_decompress(framed) == (len(data), data)
    assert len(framed) <= oracle.lib().cjo_snappy_frame_max_compress_len(len(data)) or block_size not in (None, 65536)


def test_empty_and_hand_built_streams():
    assert oracle.snappy_frame_compress(b"") == (0, b"")        # snap emits the identifier with the first chunk only
    assert oracle.snappy_frame_decompress(b"") == (0, b"")
    a, b = b"hello hello hello hello", bytes(range(256)) * 3
    s = (SNAPPY_IDENT + snappy_stored(a) + snappy_chunk(0xfe, b"\0" * 13) + snappy_chunk(0x80, b"skip me")
         + SNAPPY_IDENT + snappy_compressed(b, oracle.snappy_compress(b)[1]) + snappy_chunk(0xfd, b""))
    assert oracle.snappy_frame_decompress(s) == (len(a) + len(b), a + b)
    assert oracle.snappy_frame_decompress(SNAPPY_IDENT) == (0, b"")


def test_malformed_streams():
    a = b"abcdefgh" * 40
    good = SNAPPY_IDENT + snappy_compressed(a, oracle.snappy_compress(a)[1])
    dec = lambda s, cap=None: oracle.snappy_frame_decompress(s, cap if cap is not None else 4096)[0]
    assert dec(good) == len(a)
    assert dec(b"sknow") == E_HDR                                   # reference tests/test_variants.py:92-96
    assert dec(good[10:]) == E_HDR                                  # no stream identifier
    assert dec(b"\xff\x06\x00\x00sNaPpX" + good[10:]) == E_HDR
    assert dec(b"\xff\x05\x00\x00sNaPp" + good[10:]) == E_LEN
    for cut in (1, 3, 11, 13, 15, 17, len(good) - 1):
        assert dec(good[:cut]) == E_EOF, cut
    assert dec(good[:10]) == 0
    assert dec(SNAPPY_IDENT + snappy_chunk(0x02, b"xx")) == E_TYPE
    assert dec(SNAPPY_IDENT + snappy_chunk(0x7f, b"")) == E_TYPE
    assert dec(SNAPPY_IDENT + snappy_chunk(0x00, b"abc")) == E_LEN       # shorter than its checksum
    assert dec(SNAPPY_IDENT + b"\x01" + (76491).to_bytes(3, "little") + bytes(76491)) == E_LEN
    assert dec(SNAPPY_IDENT + snappy_stored(bytes(65537)), 70000) == E_LEN
    assert dec(SNAPPY_IDENT + snappy_stored(bytes(65536)), 70000) == 65536
    assert dec(SNAPPY_IDENT + snappy_stored(a, crc=1)) == E_SUM
    assert dec(SNAPPY_IDENT + snappy_compressed(a, oracle.snappy_compress(a)[1], crc=crc32c_masked(a) ^ 1)) == E_SUM
    assert dec(SNAPPY_IDENT + snappy_chunk(0x00, b"\0\0\0\0")) == -8      # empty raw block: snap Error::Empty
    big = oracle.snappy_compress(bytes(65537))[1]
    assert dec(SNAPPY_IDENT + snappy_compressed(bytes(65537), big), 70000) == E_LEN   # piece decodes to > 64 KiB
    bad = bytearray(oracle.snappy_compress(a)[1]); bad[-1] ^= 0xff; bad[3] ^= 0x40
    assert dec(SNAPPY_IDENT + snappy_compressed(a, bytes(bad))) in (-12, E_SUM)
    assert dec(good, cap=len(a) - 1) == E_WRITE
    # stream order: a corrupt piece before a header error wins; a header error before it hides it
    assert dec(SNAPPY_IDENT + snappy_stored(a, crc=1) + snappy_chunk(0x02, b"")) == E_SUM
    assert dec(SNAPPY_IDENT + snappy_chunk(0x02, b"") + snappy_stored(a, crc=1)) == E_TYPE


SIM_SO = os.path.join(ROOT, "tests", "hostsim", "libsim_crc32c.so")


def test_hostsim_lane_parallel_crc32c():
    """The kernel's 64-lane strided CRC (GF(2) advance tables + tail multipliers) equals the serial CRC for every
    length / alignment class, incl. the 64 KiB piece size."""
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "cramjam_amd", "csrc"),
                           os.path.join(ROOT, "tests", "hostsim", "sim_crc32c.cpp"), "-o", SIM_SO])
    S = C.CDLL(SIM_SO)
    S.sim_crc32c.restype = C.c_uint32
    S.sim_crc32c.argtypes = [C.c_void_p, C.c_uint32]
    S.sim_crc32c_mask.restype = C.c_uint32
    random.seed(9)
    buf = bytes(random.randrange(256) for _ in range(70100))
    base = C.create_string_buffer(buf, len(buf))
    addr = C.addressof(base)
    cases = [(o, n) for n in range(0, 1300) for o in (0, 1, 3)]
    cases += [(random.randrange(8), random.randrange(70000)) for _ in range(200)] + [(0, 65536), (3, 65536), (1, 65535)]
    for o, n in cases:
        assert S.sim_crc32c(addr + o, n) == oracle.crc32c(buf[o:o + n]), (o, n)
    assert S.sim_crc32c_mask(oracle.crc32c(b"abc")) == oracle.crc32c(b"abc", masked=True)


def test_hostsim_snappy_records(golden):
    """The record grammar shared by the GPU Snappy parse kernel and the LDS decoder (snappy_records.hpp), built for the
    host: same verdicts and bytes as the oracle on the golden vectors, the malformed set and fuzzed blocks."""
    from conftest import b64d
    so = os.path.join(ROOT, "tests", "hostsim", "libsim_snappy_records.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC",
                           "-I", os.path.join(ROOT, "cramjam_amd", "csrc"), os.path.join(ROOT, "tests", "hostsim", "sim_snappy_records.cpp"), "-o", so])
    child = r'''
import ctypes as C, json, random, sys
sys.path.insert(0, %r)
import oracle
from base64 import b64decode as b64d
S = C.CDLL(%r); S.sim_snappy_decode.restype = C.c_int64
S.sim_snappy_decode.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32)]
g = json.load(open(%r))
bad = 0
def chk(blob, tag):
    global bad
    cap = min(max(oracle.snappy_decompress_len(blob), 0), 65536)
    out = C.create_string_buffer(max(cap, 1)); nr = C.c_uint32(0)
    r = S.sim_snappy_decode(blob, len(blob), out, cap, C.byref(nr))
    er, eo = oracle.snappy_decompress(blob, cap)
    if r != er or (er >= 0 and out.raw[:r] != eo):
        bad += 1; print("MISMATCH", tag, r, er)
for v in g["vectors"]:
    if v["n"] <= 65536: chk(b64d(v["snappy"]), v["name"])
for m in g["malformed_snappy"]: chk(b64d(m["data"]), (m["src"], m["kind"], m["k"]))
random.seed(5)
for t in range(200):
    n = random.choice([1, 5, 13, 40, 100, 1000, 5000, 65536]); alpha = random.choice([2, 4, 16, 256])
    raw = bytes(random.randrange(alpha) for _ in range(n))
    if random.random() < 0.5 and n > 10: raw = (raw[:random.randrange(1, 20)] * n)[:n]
    _, blob = oracle.snappy_compress(raw)
    chk(blob, ("fuzz", t))
    b = bytearray(blob); i = random.randrange(len(b)); b[i] ^= 1 << random.randrange(8); chk(bytes(b), ("fuzzbad", t))
for i in range(4):
    _, blob = oracle.snappy_compress(oracle.synth_v1(65536, i)); chk(blob, ("synth", i))
print("HOSTSIM bad=%%d" %% bad)
''' % (ROOT, so, os.path.join(GOLDEN_DIR, "golden_vectors.json"))
    import sys
    env = dict(os.environ, LD_PRELOAD=subprocess.check_output(["g++", "-print-file-name=libasan.so"]).decode().strip(), ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env)
    assert "HOSTSIM bad=0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- LZ4 frame ------------------------------------------------------------------------------------------------
import base64
import hashlib
import json


def _frame_vectors():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(GOLDEN_DIR, "make_golden_frames.py"))
    mgf = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgf)
    g = json.load(open(os.path.join(GOLDEN_DIR, "golden_frames.json")))
    for v in g["vectors"]:
        data = mgf.content(v["kind"], v["n"])
        assert hashlib.sha256(data).hexdigest() == v["sha256"]
        yield v, base64.b64decode(v["frame"]), data


def test_xxh32_known_answers():
    assert oracle.xxh32(b"") == 0x02CC5D05 and oracle.xxh32(b"", 1) == 0x0B2CB792
    assert oracle.xxh32(b"a") == 0x550D7456 and oracle.xxh32(b"abc") == 0x32D153FF
    assert oracle.xxh32(b"Nobody inspects the spammish repetition") == 0xE2293B2F


def test_lz4_frame_fixture_and_golden_frames():
    """the reference's fixture + 48 frames minted by liblz4's LZ4F (linked/independent, all block sizes, checksums, HC)"""
    framed, plain = _fx("plaintext.txt.lz4"), _fx("plaintext.txt")
    assert oracle.lz4_frame_decompress(framed) == (len(plain), plain)
    n = 0
    for v, frame, data in _frame_vectors():
        b = oracle.lz4_frame_decompress_bound(frame)
        assert b >= len(data) and (not v["content_size"] or b == len(data)), v
        assert oracle.lz4_frame_decompress(frame) == (len(data), data), {k: v[k] for k in v if k != "frame"}
        n += 1
    assert n >= 40


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 4, 7, 8, 9])
@pytest.mark.parametrize("bs", [4, 5, 7])
def test_lz4_frame_round_trip(flags, bs):
    random.seed(flags * 8 + bs)
    data = _fx("plaintext.txt") * 200 + bytes(random.randrange(256) for _ in range(70000)) + bytes(100000)
    r, fr = oracle.lz4_frame_compress(data, bs, flags)
    assert r == len(fr) and fr[:4] == b"\x04\x22\x4d\x18"
    assert fr[4] == 0x40 | (0 if flags & 1 else 0x20) | (0x10 if flags & 2 else 0) | (0x08 if flags & 4 else 0) | (0 if flags & 8 else 0x04)
    assert oracle.lz4_frame_decompress(fr) == (len(data), data)
    assert oracle.lz4_frame_compress(b"", bs, flags & ~4)[1][4:6] != b"" and oracle.lz4_frame_decompress(oracle.lz4_frame_compress(b"", bs, flags)[1]) == (0, b"")


def test_lz4_frame_malformed():
    data = _fx("plaintext.txt") * 100
    r, fr = oracle.lz4_frame_compress(data, 4, 2)            # independent blocks, block + content checksums
    dec = lambda s, cap=200000: oracle.lz4_frame_decompress(s, cap)[0]
    assert dec(fr) == len(data)
    assert dec(b"sknow") == -26 and dec(b"sknowsknow") == -20          # reference tests/test_variants.py:92-96: DecompressionError
    assert dec(b"") == -26
    for cut in (3, 6, 8, 12, len(fr) - 5, len(fr) - 1):
        assert dec(fr[:cut]) == -26, cut
    assert dec(fr + b"trailing bytes are ignored by the lz4 crate's Decoder") == len(data)
    b = bytearray(fr); b[4] ^= 0x80; assert dec(bytes(b)) == -21       # version
    b = bytearray(fr); b[4] |= 0x02; assert dec(bytes(b)) == -21       # reserved flag
    b = bytearray(fr); b[5] = 0x30; assert dec(bytes(b)) in (-21, -22) # block size code 3
    b = bytearray(fr); b[6] ^= 1; assert dec(bytes(b)) == -21          # header checksum
    b = bytearray(fr); b[20] ^= 1; assert dec(bytes(b)) == -23         # block data vs block checksum
    b = bytearray(fr); b[-1] ^= 1; assert dec(bytes(b)) == -24         # content checksum
    r2, fr2 = oracle.lz4_frame_compress(data, 4, 8)                    # no checksums at all: damage reaches the block decoder
    b = bytearray(fr2); b[12] = 0xFF; b[13] = 0xFF; assert dec(bytes(b)) in (-27, len(data))
    r3, fr3 = oracle.lz4_frame_compress(data, 4, 4 | 8)                # content size stored
    b = bytearray(fr3); b[6] ^= 1; b[14] = (oracle.xxh32(bytes(b[4:14])) >> 8) & 0xff
    assert dec(bytes(b)) == -25
    big = bytearray(fr2); big[7:11] = (70000).to_bytes(4, "little"); assert dec(bytes(big)) in (-22, -26)
    assert dec(fr, cap=len(data) - 1) == -14
    assert dec(b"\x50\x2a\x4d\x18\x03\x00\x00\x00abc" + fr) == 0       # skippable frame first: the crate's Decoder stops there
