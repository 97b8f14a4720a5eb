#!/usr/bin/env python3
"""Mint tests/golden/golden_vectors.json.  Run ONLY in the build container (needs /root/reference's
benchmark corpus and the stand-in C libraries); the output is committed, this script is the recipe.

Sources of truth recorded in the fixture:
  * liblz4 (system, 1.9.3 — same C code family as the lz4-sys 1.11.1+lz4-1.10.0 the reference links):
      LZ4_compress_default output bytes, LZ4_decompress_safe verdicts on malformed streams
  * libsnappy 1.1.8 (same format/algorithm family as the Rust `snap` 1.1.1 the reference links):
      snappy_compress output bytes, snappy_uncompress verdicts on malformed streams
  * pyarrow codecs lz4_raw / snappy as a third independent decoder on the valid streams
Data files plaintext.txt{,.lz4,.snappy} next to this script are copies of the reference's own test
fixtures (tests/data/integration/, used by tests/test_integration.py:32-50): the .lz4 frame holds one raw
LZ4 block at bytes [11,657), the .snappy framed file holds one raw Snappy stream at bytes [18,660).
"""
import base64, bz2, ctypes as C, glob, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402  (only for synth_v1 so the generator itself is pinned by sha256)
import pyarrow as pa

L4 = C.CDLL("/lib/x86_64-linux-gnu/liblz4.so.1")
SN = C.CDLL("/opt/conda/lib/libsnappy.so.1")
SN.snappy_max_compressed_length.restype = C.c_size_t
SN.snappy_max_compressed_length.argtypes = [C.c_size_t]


def lz4c(d):
    cap = L4.LZ4_compressBound(len(d)); o = C.create_string_buffer(cap)
    r = L4.LZ4_compress_default(d, o, len(d), cap); assert r > 0
    return o.raw[:r]


def lz4d(d, cap):
    o = C.create_string_buffer(max(cap, 1))
    r = L4.LZ4_decompress_safe(d, o, len(d), cap)
    return r, o.raw[:max(r, 0)]


def snc(d):
    cap = C.c_size_t(SN.snappy_max_compressed_length(len(d))); o = C.create_string_buffer(cap.value)
    assert SN.snappy_compress(d, C.c_size_t(len(d)), o, C.byref(cap)) == 0
    return o.raw[:cap.value]


def snd(d):
    n = C.c_size_t(0)
    if SN.snappy_uncompressed_length(d, C.c_size_t(len(d)), C.byref(n)) != 0:
        return -1, b""
    if n.value > (1 << 26):
        return -1, b""
    o = C.create_string_buffer(max(n.value, 1)); m = C.c_size_t(n.value)
    if SN.snappy_uncompress(d, C.c_size_t(len(d)), o, C.byref(m)) != 0:
        return -1, b""
    return m.value, o.raw[:m.value]


def rnd(seed, n):
    return hashlib.shake_256(("cj-golden-%d" % seed).encode()).digest(n)


def b64(b):
    return base64.b64encode(b).decode()


def sha(b):
    return hashlib.sha256(b).hexdigest()


vectors = []


def add(name, raw, keep_raw):
    cl, cs = lz4c(raw), snc(raw)
    assert lz4d(cl, len(raw)) == (len(raw), raw) and snd(cs) == (len(raw), raw)
    if raw:
        assert pa.decompress(cl, len(raw), codec="lz4_raw").to_pybytes() == raw
        assert pa.decompress(cs, len(raw), codec="snappy").to_pybytes() == raw
    v = dict(name=name, n=len(raw), sha256=sha(raw), lz4=b64(cl), snappy=b64(cs))
    if keep_raw:
        v["raw"] = b64(raw)
    vectors.append(v)


plain = open(os.path.join(HERE, "plaintext.txt"), "rb").read()
add("empty", b"", True)
add("one", b"a", True)
add("howdy", b"howdy neighbor", True)
for n in (4, 5, 11, 12, 13, 14, 16, 17, 18, 63, 64, 65, 255, 256, 270, 271):
    add("abc%d" % n, (b"abcabcabd" * 40)[:n], True)
add("plaintext", plain, True)
add("plaintext_x20", plain * 20, False)
add("zeros_64k", bytes(65536), False)
add("zeros_65535", bytes(65535), False)
add("ff_300", b"\xff" * 300, True)
add("rand_64k", rnd(1, 65536), False)
add("rand_1000", rnd(2, 1000), True)
add("period3_5000", (b"xyz" * 2000)[:5000], False)
add("period70_9000", (rnd(3, 70) * 200)[:9000], False)
add("longlit_then_rep", rnd(4, 700) + rnd(4, 700)[:650] + bytes(3000), False)
for size, idx in ((13, 0), (64, 1), (1000, 2), (4096, 3), (65535, 4), (65536, 5), (65536, 6), (65547, 7), (70000, 8),
                  (131072, 9), (262144, 10)):
    add("synth_v1_%d_%d" % (size, idx), oracle.synth_v1(size, idx), False)
for f in ("alice29.txt", "html", "geo.protodata", "urls.10K", "kppkn.gtb", "fireworks.jpeg", "xml", "x-ray"):
    raw = bz2.decompress(open("/root/reference/benchmarks/data/%s.bz2" % f, "rb").read())
    add("corpus_%s_64k" % f, raw[:65536], False)
    add("corpus_%s_tail" % f, raw[65536:65536 + 30011], False)

# ---- malformed streams: verdicts of the stand-in decoders ----
bad_lz4, bad_sn = [], []
seedn = [100]


def mutate(blob, kind, k):
    seedn[0] += 1
    r = rnd(seedn[0], 8)
    if kind == "trunc":
        return blob[:max(0, len(blob) * k // 7)]
    if kind == "flip":
        i = int.from_bytes(r[:4], "little") % max(1, len(blob))
        return blob[:i] + bytes([blob[i] ^ (1 << (r[4] & 7))]) + blob[i + 1:]
    if kind == "extend":
        return blob + r[:1 + k]
    raise ValueError


base_names = ("howdy", "abc64", "abc271", "plaintext", "rand_1000", "ff_300", "synth_v1_1000_2", "synth_v1_4096_3")
for v in vectors:
    if v["name"] not in base_names:
        continue
    cl, cs, n = base64.b64decode(v["lz4"]), base64.b64decode(v["snappy"]), v["n"]
    for kind in ("trunc", "flip", "extend"):
        for k in range(1, 7):
            m = mutate(cl, kind, k)
            for cap in (n, n + 64):
                r, out = lz4d(m, cap)
                # offset-0 matches are accepted by liblz4 but read unwritten output: exclude (policy = reject)
                bad_lz4.append(dict(src=v["name"], kind=kind, k=k, cap=cap, data=b64(m), ret=r,
                                    sha256=sha(out) if r >= 0 else None))
            m = mutate(cs, kind, k)
            r, out = snd(m)
            bad_sn.append(dict(src=v["name"], kind=kind, k=k, data=b64(m), ret=r,
                               sha256=sha(out) if r >= 0 else None))

hand_lz4 = [  # (name, stream, cap)
    ("offset_gt_pos", bytes([0x10, 0x41, 0x05, 0x00, 0x50]) + b"abcde", 64),
    ("match_into_last5", bytes([0x10, 0x41, 0x01, 0x00, 0x10, 0x42]), 7),
    ("no_final_literals", bytes([0x14, 0x41, 0x01, 0x00]), 64),
    ("lit_overrun", bytes([0x50, 0x41, 0x42]), 64),
    ("only_token_f0", bytes([0xF0]), 64),
    ("ext_255_run", bytes([0xF0, 0xFF, 0xFF]), 1000),
    ("single_zero", bytes([0x00]), 0),
    ("single_zero_cap5", bytes([0x00]), 5),
    ("two_zero", bytes([0x00, 0x00]), 5),
    ("valid_rle", bytes([0x1F, 0x61, 0x01, 0x00, 0x05, 0x50]) + b"bcdef", 64),
    ("valid_rle_exact", bytes([0x1F, 0x61, 0x01, 0x00, 0x05, 0x50]) + b"bcdef", 30),
    ("valid_rle_small", bytes([0x1F, 0x61, 0x01, 0x00, 0x05, 0x50]) + b"bcdef", 29),
]
for name, m, cap in hand_lz4:
    r, out = lz4d(m, cap)
    bad_lz4.append(dict(src=name, kind="hand", k=0, cap=cap, data=b64(m), ret=r, sha256=sha(out) if r >= 0 else None))
hand_sn = [
    ("empty", b""), ("zero_len", b"\x00"), ("zero_len_extra", b"\x00\x00"),
    ("hdr_only", b"\x05"), ("bad_varint", b"\xff\xff\xff\xff\xff\xff"),
    ("offset0", b"\x08\x0cabcd\x01\x00"), ("offset_gt", b"\x08\x0cabcd\x01\x09"),
    ("copy4", b"\x08\x0cabcd\x0f\x04\x00\x00\x00"), ("copy2_overlap", b"\x0a\x00a\x22\x01\x00"),
    ("short", b"\x08\x0cabcd"), ("long", b"\x03\x0cabcd"), ("lit61", b"\x02\xf4\x01\x00ab"),
]
for name, m in hand_sn:
    r, out = snd(m)
    bad_sn.append(dict(src=name, kind="hand", k=0, data=b64(m), ret=r, sha256=sha(out) if r >= 0 else None))

gold = dict(
    about="golden vectors for LZ4-block / Snappy-raw; see make_golden.py",
    lz4_lib="liblz4 %d" % L4.LZ4_versionNumber(), snappy_lib="libsnappy 1.1.8", pyarrow=pa.__version__,
    reference_known_answers=dict(  # /root/reference/tests/test_variants.py:329-334
        data=b64(b"howdy neighbor"),
        lz4_block_store_size=b64(b"\x0e\x00\x00\x00\xe0howdy neighbor"),
        lz4_block_no_size=b64(b"\xe0howdy neighbor")),
    reference_fixture_blocks=dict(lz4_frame_block=[11, 657], snappy_framed_raw=[18, 660], sha256=sha(plain)),
    vectors=vectors, malformed_lz4=bad_lz4, malformed_snappy=bad_sn)
out = os.path.join(HERE, "golden_vectors.json")
json.dump(gold, open(out, "w"), indent=0)
print(out, os.path.getsize(out), len(vectors), len(bad_lz4), len(bad_sn))
