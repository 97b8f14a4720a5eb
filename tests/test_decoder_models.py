"""CPU models of two pieces of decoder logic that otherwise only run on the GPU (cramjam_amd/csrc/lz4_decode_lds.hip):
 * D1f, match forwarding: matches are redirected to the source of the match (or the literal run) their source lies in;
   the model applies the kernel's rules round by round and then decodes FROM THE FORWARDED RECORDS — the bytes must be
   the oracle's, and the dependency depth must collapse on chain-like data.
 * the slab mode's D1: one large stream cut into 64 KiB slabs of output — clipping of the sequences that straddle a slab,
   the cross list (bytes taken from earlier slabs), remainder matches, extra literal records.  Every slab is decoded from
   its own records + the finished earlier output only."""
import bisect
import json
import random

import pytest

import oracle

LIT = 1 << 31


def lz4_records(blk):
    """[(lit_src, lit, dst, off, m)] — the decoder's record per sequence; dst = output position after the literals"""
    out = []; ip = op = 0; n = len(blk)
    while ip < n:
        tok = blk[ip]; ip += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = blk[ip]; ip += 1; lit += b
                if b != 255: break
        src = ip; ip += lit; op += lit
        if ip >= n:
            out.append((src, lit, op, 0, 0)); break
        off = blk[ip] | (blk[ip + 1] << 8); ip += 2
        m = tok & 15
        if m == 15:
            while True:
                b = blk[ip]; ip += 1; m += b
                if b != 255: break
        m += 4
        out.append((src, lit, op, off, m)); op += m
    return out


def forward(recs, max_rounds=16):
    """the kernel's D1f on one chunk's records; returns (records', extras, rounds)"""
    n = len(recs)
    start = [min(r[2] - r[1], 65535) for r in recs]
    dst = [min(r[2], 65535) for r in recs]
    m = [r[4] for r in recs]
    litsrc = [r[0] & 0xffff for r in recs]
    st = [r[3] for r in recs]
    blk = [0] * 4096
    for i in range(n):
        b0 = 0 if i == 0 else (start[i] + 15) >> 4
        b1 = (start[i + 1] + 15) >> 4 if i + 1 < n else 4096
        for b in range(b0, b1): blk[b] = i
    rounds = 0
    for _ in range(max_rounds):
        changed = False
        new = list(st)                                     # (the kernel updates in place; any interleaving is valid)
        for i in range(n):
            s = st[i]
            if m[i] == 0 or (s & LIT) or s < m[i]: continue
            sp = dst[i] - s
            r = blk[sp >> 4]
            while r + 1 < n and start[r + 1] <= sp: r += 1
            if r > i: continue
            if sp >= start[r] and sp + m[i] <= dst[r]:
                new[i] = LIT | (litsrc[r] + (sp - start[r])); changed = True
            elif m[r] > 0 and sp >= dst[r] and sp + m[i] <= dst[r] + m[r]:
                if st[r] & LIT: new[i] = LIT | ((st[r] & ~LIT) + (sp - dst[r])); changed = True
                elif st[r] >= m[r]: new[i] = s + st[r]; changed = True
        st = new; rounds += 1
        if not changed: break
    out, extras = [], []
    split = set()
    for i in range(n - 2):                                 # straddlers: both halves end in literal runs -> two literal copies
        s = st[i]
        if m[i] == 0 or (s & LIT) or s < m[i] or s > dst[i]: continue
        sp = dst[i] - s
        r = blk[sp >> 4]
        while r + 1 < n and start[r + 1] <= sp: r += 1
        if r + 2 >= n or r + 1 > i: continue
        cut = start[r + 1]
        if sp + m[i] <= cut or sp + m[i] > start[r + 2]: continue
        parts = []
        for q, p0, p1 in ((r, sp, cut), (r + 1, cut, sp + m[i])):
            if p0 >= start[q] and p1 <= dst[q]: parts.append(litsrc[q] + (p0 - start[q]))
            elif p0 >= dst[q] and p1 <= dst[q] + m[q] and (st[q] & LIT) and q not in split: parts.append((st[q] & ~LIT) + (p0 - dst[q]))
        if len(parts) != 2: continue
        d = recs[i][2]
        for pos, ln in zip(parts, (cut - sp, sp + m[i] - cut)):
            extras.append((pos, ln, d + ln, 0, 0)); d += ln
        split.add(i)
    for i, (src, lit, d, off, mm) in enumerate(recs):
        if i in split:
            out.append((src, lit, d, 0, 0))
        elif mm and (st[i] & LIT):
            extras.append((st[i] & ~LIT, mm, d + mm, 0, 0)); out.append((src, lit, d, 0, 0))
        else:
            out.append((src, lit, d, st[i] if mm else 0, mm))
    return out, extras, rounds


def decode(blk, recs, n):
    out = bytearray(n)
    for src, lit, d, off, m in recs:                       # literals (and forwarded literal copies): from the input
        out[d - lit:d] = blk[src:src + lit]
    for src, lit, d, off, m in recs:                       # matches in record order
        for k in range(m): out[d + k] = out[d + k - off]
    return bytes(out)


def depth(recs, n):
    lvl = bytearray(n) if n < 1 else [0] * n               # level of every output byte
    deepest = 0
    for src, lit, d, off, m in recs:
        if not m: continue
        need = min(off, m)
        L = 1 + max(lvl[d - off:d - off + need], default=0)
        for k in range(m): lvl[d + k] = L
        deepest = max(deepest, L)
    return deepest


def chunks():
    rnd = random.Random(4)
    def fill(fn, n=65536):
        out = bytearray()
        while len(out) < n: out += fn()
        return bytes(out[:n])
    yield "bottles", fill(lambda: b"%d bottles of beer on the wall, %d bottles of beer\n" % (rnd.randrange(977), rnd.randrange(1013)))
    yield "log", fill(lambda: b"2026-09-28T12:%02d:%02d INFO worker-%d id=%08x path=/api/v1/items/%d status=%d\n" % (
        rnd.randrange(60), rnd.randrange(60), rnd.randrange(16), rnd.getrandbits(32), rnd.randrange(5000), rnd.choice([200, 404, 500])))
    yield "json", fill(lambda: json.dumps({"id": rnd.randrange(10 ** 6), "name": "user%d" % rnd.randrange(1000), "t": ["a", rnd.choice("xyz")]}).encode() + b"\n")
    yield "synth-v1", oracle.synth_v1(65536, 3)
    yield "periodic", fill(lambda: b"abcabcabc" * rnd.randrange(1, 9) + bytes([rnd.randrange(97, 123)]), 40000)
    yield "zeros", bytes(50000)


@pytest.mark.parametrize("name,data", list(chunks()), ids=[c[0] for c in chunks()])
def test_forwarded_records_decode_to_the_same_bytes(name, data):
    blk = oracle.lz4_compress_raw(data)[1]
    recs = lz4_records(blk)
    assert decode(blk, recs, len(data)) == data            # the model's decoder itself
    fwd, extras, rounds = forward(recs)
    assert decode(blk, fwd + extras, len(data)) == data, name
    d0, d1 = depth(recs, len(data)), depth(fwd + extras, len(data))
    assert d1 <= d0
    if name == "bottles": assert d0 > 500 and d1 < 100, (d0, d1, rounds)
    if name == "log": assert d1 * 3 < d0, (d0, d1)


def slab_records(recs, S, U):
    """the slab mode's D1 for the slab [S, S + U): (table records, extras, cross list) from the stream's absolute records"""
    table, extras, cross = [], [], []
    for src, lit, d, off, m in recs:
        o0 = d - lit - S                                   # literal start, slab coordinates (may be negative)
        if o0 + lit + m <= 0: continue
        if o0 >= U: break
        if o0 < 0: cut = min(lit, -o0); src += cut; lit -= cut; o0 += cut
        d0 = o0 + lit
        if d0 < 0: cut = min(m, -d0); m -= cut; d0 += cut
        if d0 < 0: d0 = o0 = 0
        o0 = max(o0, 0)
        if o0 + lit > U: lit = U - o0; m = 0
        d0 = o0 + lit
        if d0 + m > U: m = U - d0
        rec = (src, lit, d0, 0, 0)
        if m:
            if off > d0:
                n1 = min(m, off - d0)
                cross.append((S + d0 - off, d0, n1))
                if m > n1:
                    if lit: extras.append(rec)
                    rec = (0, 0, d0 + n1, off, m - n1)
            else: rec = (src, lit, d0, off, m)
        table.append(rec)
    return table, extras, cross


def test_slab_records_decode_every_slab_from_its_own_records_and_earlier_output():
    rnd = random.Random(9)
    text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer\n" % (rnd.randrange(977), rnd.randrange(1013)) for _ in range(6000))
    data = bytes(777) + oracle.synth_v1(65536, 1) + text + rnd.randbytes(70000) + bytes(200000) + oracle.synth_v1(65536, 2)[:30000]
    blk = oracle.lz4_compress_raw(data)[1]
    recs = lz4_records(blk)
    out = bytearray()
    for S in range(0, len(data), 65536):
        U = min(65536, len(data) - S)
        table, extras, cross = slab_records(recs, S, U)
        win = bytearray(U)
        for src, lit, d, off, m in table + extras: win[d - lit:d] = blk[src:src + lit]
        for src_abs, d, n1 in cross:
            assert 0 <= src_abs and src_abs + n1 <= S      # finished output of earlier slabs only
            win[d:d + n1] = out[src_abs:src_abs + n1]
        for src, lit, d, off, m in table:                  # record order: a record waits only for records before it
            assert off <= d or m == 0
            for k in range(m): win[d + k] = win[d + k - off]
        assert bytes(win) == data[S:S + U], S
        out += win
