"""Inputs shared by the encoder-model tests (CPU) and the kernel-vs-model tests (GPU): every shape the round-based matcher
treats differently — empty table, long runs, capped extensions, literal runs above the queue's limit, windows of more than 64
heads, inputs around the round and end-of-block boundaries."""
import bz2
import os
import random

import oracle
from conftest import GOLDEN_DIR


def synth(i, n=65536):
    return oracle.synth_v1(n, i, 1) if hasattr(oracle, "synth_v1") else _synth(i, n)


def _synth(i, n):
    import ctypes as C
    buf = C.create_string_buffer(n)
    oracle.lib().cjo_synth_v1(C.cast(buf, C.c_void_p), n, i, 1)
    return buf.raw


def corpus(name):
    with open(os.path.join(GOLDEN_DIR, "corpus", name + ".bz2"), "rb") as f:
        return bz2.decompress(f.read())


def cases():
    rnd = random.Random(77)
    out = []
    for i in range(6):
        out.append(("synth%d" % i, synth(i)))
    out.append(("synth_3000", synth(9, 3000)))
    out.append(("synth_65535", synth(10, 65535)))
    out.append(("synth_100k", synth(11, 100000)))               # second 64 KiB lap of the 16-bit table
    out.append(("synth_200k", synth(12, 200000)))
    text = corpus("alice29.txt")
    out.append(("alice_64k", text[:65536]))
    out.append(("alice_all", text))
    html = corpus("html")
    out.append(("html", html))
    out.append(("kppkn_64k", corpus("kppkn.gtb")[:65536]))
    out.append(("geo_64k", corpus("geo.protodata")[:65536]))
    out.append(("jpeg_64k", corpus("fireworks.jpeg")[:65536]))  # incompressible: literal runs far above 256
    out.append(("zeros_64k", bytes(65536)))                      # one match to the end: capped forward extension
    out.append(("zeros_1000", bytes(1000)))
    out.append(("random_64k", bytes(rnd.getrandbits(8) for _ in range(65536))))
    out.append(("period7", (b"abcdefg" * 10000)[:65536]))
    out.append(("period300", bytes(rnd.getrandbits(8) for _ in range(300)) * 200))
    out.append(("two_symbols", bytes(rnd.choice(b"ab") for _ in range(20000))))      # windows of more than 64 heads
    out.append(("four_symbols", bytes(rnd.choice(b"abcd") for _ in range(30000))))
    blk = bytes(rnd.getrandbits(8) for _ in range(700))
    out.append(("far_repeat", blk + bytes(rnd.getrandbits(8) for _ in range(40000)) + blk + b"tail of some literals"))    # literal run > 256 before a long match
    out.append(("back_ext", b"x" * 40 + bytes(rnd.getrandbits(8) for _ in range(5000)) + b"x" * 40 + b"yz" * 30))
    long_lit_then_match = bytes(rnd.getrandbits(8) for _ in range(300))
    out.append(("lit300_match", long_lit_then_match + long_lit_then_match[10:90] + bytes(20)))
    r70k = bytes(rnd.getrandbits(8) for _ in range(70000))
    out.append(("lit70000_match", r70k + r70k[69000:69900] + b"the end of it"))      # a literal run above 64 KiB in front of a match: the serial path
    out.append(("back_ext_far", b"q" * 300 + bytes(rnd.getrandbits(8) for _ in range(3000)) + b"Z" + b"q" * 300 + bytes(50)))   # backward extension far beyond 16 bytes
    base = bytes(rnd.getrandbits(8) for _ in range(2000))
    for t in range(0, 72, 3):                                    # a match that runs into the end-of-block zone, every distance from the end
        out.append(("tail_match_%d" % t, base + base[100:140 + t]))
        out.append(("tail_match_lit_%d" % t, base + base[100:170] + bytes(rnd.getrandbits(8) for _ in range(t))))
    for k in range(24):                                          # ragged sizes: the last round's positions end anywhere in a lane's four
        out.append(("synth_ragged_%d" % k, synth(300 + k, 30000 + 37 * k + k * k)))
    # long matches finished by groups of sixteen lanes, four heads at a time: windows with fewer / more than four heads that still match
    # after the lanes' first trips, lengths on both sides of every step (32 per lane trip, 256 per group trip), ends in the end-of-block zone
    dic = [bytes(rnd.getrandbits(8) for _ in range(1400)) for _ in range(8)]
    for name, lens in (("slices_short", (40, 45, 52, 60)), ("slices_mixed", (37, 44, 70, 100, 259, 260, 261, 300, 516, 517, 600, 1100)), ("slices_long", (300, 520, 800, 1290))):
        body = b"".join(dic)
        for _ in range(400):
            L = rnd.choice(lens)
            a = rnd.randrange(0, 1400 - L)
            body += rnd.choice(dic)[a:a + L]
        out.append((name, body[:65536]))
    rec = bytes(rnd.getrandbits(8) for _ in range(700))
    recs = bytearray()
    for k in range(90):                                           # repeated records with a few bytes changed each (what geo.protodata / xml look like)
        r = bytearray(rec)
        for _ in range(rnd.randrange(1, 6)): r[rnd.randrange(700)] = rnd.getrandbits(8)
        if k % 7 == 0: rec = bytes(r)
        recs += r
    out.append(("records_mutated", bytes(recs)))
    for t in range(0, 40, 3):                                    # long matches (one, then two side by side) that run into the end of the input
        out.append(("tail_long_%d" % t, base + base[100:700 + t]))
        out.append(("tail_long2_%d" % t, base + base[100:400] + base[900:1500 + t]))
    for n in (0, 1, 4, 7, 8, 9, 12, 13, 14, 15, 16, 17, 20, 31, 63, 64, 65, 67, 68, 255, 256, 257, 260, 511, 512, 513, 1023, 1024):
        out.append(("abab_%d" % n, (b"abcab" * 300)[:n]))
        out.append(("text_%d" % n, text[1000:1000 + n]))
    return out
