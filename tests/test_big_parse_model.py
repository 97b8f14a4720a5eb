"""CPU model of the large-stream parse (cramjam_amd/csrc/big_parse.hip): K1 marks per piece + per-lane merge / exit + the
next-entry tables for the first 64 bytes, K2 threading of the entries through the tables (with the walk through the stream
as fallback), K3 chain + counts, K4 scan, K5 validation + absolute sync points.  Small pieces (4 KiB, 64-byte sub-segments)
so that streams of some 10 KiB exercise many pieces.  Checked against the oracle decoders and a serial walk."""
import random

import pytest

import oracle
from conftest import b64d
from test_spec_parse_model import END, ERR, lz4_check, seq_at, snappy_at, snappy_check

P, SUB = 4096, 64


def big_parse(b, start, cap, at, check):
    """returns (ok, total, nseq, sync points)"""
    iend = len(b)
    if iend <= start: return False, 0, 0, []
    np_ = (iend - start + P - 1) // P
    nxt = lambda q: (lambda r: r[4] if r[0] else ERR)(at(b, q, iend))
    K1 = []
    for p in range(np_):
        B = start + p * P; E = min(B + P, iend)
        marks = set(); pos = []
        for l in range(64):                                               # 1a
            q = B + l * SUB if B + l * SUB < E else END
            s1 = min(B + (l + 1) * SUB, E)
            while 0 <= q < s1:
                marks.add(q); q = nxt(q)
            pos.append(q)
        merge = []
        for l in range(64):                                               # 1b
            q = pos[l]
            while 0 <= q < E and q not in marks: q = nxt(q)
            merge.append(q if B + l * SUB < E else ERR)
        ex = list(merge)
        for l in range(63, -1, -1):
            if 0 <= ex[l] < E: ex[l] = ex[(ex[l] - B) // SUB]
        ntab, fetab = [], []
        for j in range(64):                                               # 1c
            q = B + j if B + j < E else ERR
            while 0 <= q < E and q not in marks: q = nxt(q)
            if 0 <= q < E:
                o = (q - B) // SUB
                fetab.append(merge[0] if o == 0 else q); ntab.append(ex[o])
            else:
                fetab.append(q); ntab.append(q)
        K1.append((marks, merge, ex, ntab, fetab))
    entry = []; e = start                                                 # K2
    for p in range(np_):
        B = start + p * P; E = min(B + P, iend)
        marks, merge, ex, ntab, fetab = K1[p]
        if not (0 <= e < E): entry.append((END, END)); continue
        if e - B < 64:
            entry.append((e, fetab[e - B])); e = ntab[e - B]
        else:
            lx = (e - B) // SUB; q = e
            while 0 <= q < E and q not in marks: q = nxt(q)
            if 0 <= q < E:
                o = (q - B) // SUB
                entry.append((e, merge[lx] if o == lx else q)); e = ex[o]
            else:
                entry.append((e, q)); e = q
    if e != END: return False, 0, 0, []
    pieces = []                                                           # K3: chain + counts
    for p in range(np_):
        B = start + p * P; E = min(B + P, iend)
        ent, fe = entry[p]
        lanes = []
        if ent != END:
            merge = K1[p][1]
            cur, s, f = (ent - B) // SUB, ent, fe
            for _ in range(64):
                lanes.append([s, f, 0, 0])
                if not (0 <= f < E): break
                cur = (f - B) // SUB; s = f; f = merge[cur]
            for ln in lanes:
                q = ln[0]
                while 0 <= q < E and q != ln[1]:
                    ok, lit, ml, _, n2, _ = at(b, q, iend)
                    if not ok: q = ERR; break
                    ln[2] += 1; ln[3] += lit + ml; q = n2
        pieces.append(lanes)
    sync = {}; idx = op = 0; final = None; bad = False                    # K4 + K5
    for p in range(np_):
        B = start + p * P; E = min(B + P, iend)
        for s, f, cnt, outb in pieces[p]:
            q, i, o = s, idx, op
            while 0 <= q < E and q != f and not bad:
                if i % 8 == 0: sync[i // 8] = (q, o)
                ok, lit, ml, off, n2, last = at(b, q, iend)
                if not ok: bad = True; break
                ok, o, fin = check(lit, ml, off, last, o, cap)
                if not ok: bad = True
                elif fin: final = o; q = END
                else: q = n2; i += 1
            if f == ERR: bad = True
            idx += cnt; op += outb
    if bad or final is None: return False, 0, 0, []
    return True, final, idx, [sync[k] for k in sorted(sync)]


def serial(b, start, at):
    ip, op, n, pts = start, 0, 0, []
    while True:
        if n % 8 == 0: pts.append((ip, op))
        n += 1
        ok, lit, ml, off, n2, last = at(b, ip, len(b))
        assert ok
        op += lit + ml
        if last or n2 == END: return n, op, pts
        ip = n2


def payloads():
    rnd = random.Random(3)
    text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer\n" % (rnd.randrange(977), rnd.randrange(1013)) for _ in range(4000))
    yield "synth", oracle.synth_v1(65536, 1) + oracle.synth_v1(65536, 2)[:30000]
    yield "text", text
    yield "random-then-text", rnd.randbytes(9000) + text[:40000] + rnd.randbytes(300)      # a literal run over two pieces
    yield "zeros", bytes(150000)
    yield "mixed", text[:20000] + bytes(30000) + rnd.randbytes(6000) + oracle.synth_v1(65536, 5)[:40000]


@pytest.mark.parametrize("name,data", list(payloads()), ids=[c[0] for c in payloads()])
def test_piecewise_parse_equals_the_serial_walk(name, data):
    rnd = random.Random(len(data))
    lz = oracle.lz4_compress_raw(data)[1]
    ok, total, nseq, pts = big_parse(lz, 0, len(data), seq_at, lz4_check)
    n2, op2, p2 = serial(lz, 0, seq_at)
    assert ok and total == len(data) and nseq == n2 and pts == p2
    ok, total, _, _ = big_parse(lz, 0, len(data) + 77, seq_at, lz4_check)
    assert ok and total == len(data)
    assert not big_parse(lz, 0, len(data) - 1, seq_at, lz4_check)[0]
    sn = oracle.snappy_compress(data)[1]
    hdr = 0
    while sn[hdr] & 0x80: hdr += 1
    hdr += 1
    ok, total, nrec, pts = big_parse(sn, hdr, len(data), snappy_at, snappy_check)
    n2, op2, p2 = serial(sn, hdr, snappy_at)
    assert ok and total == len(data) and nrec == n2 and pts == p2
    for t in range(12):                                                   # damaged streams: the oracle's verdict
        for blob, start, cap, at, check, dec in ((lz, 0, len(data), seq_at, lz4_check, lambda x: oracle.lz4_decompress_raw(x, len(data))[0]),
                                                (sn, hdr, len(data), snappy_at, snappy_check, lambda x: oracle.snappy_decompress(x, len(data))[0])):
            bb = bytearray(blob); i = rnd.randrange(start, len(bb)); bb[i] ^= 1 << rnd.randrange(8)
            if t % 4 == 0: bb = bb[:rnd.randrange(start + 1, len(bb))]
            er = dec(bytes(bb))
            ok, total, _, _ = big_parse(bytes(bb), start, cap, at, check)
            if at is snappy_at and ok and total != cap: ok = False
            assert ok == (er >= 0) and (not ok or total == er), (name, t, er, ok, total)
