"""CPU model of the segmented parse (cramjam_amd/csrc/parse_spec.hip): 64 lanes walk 64 segments of a block from
guessed start positions, join each other's paths, and the true path is stitched from the pieces.  The model mirrors the
kernel phase by phase, for both grammars (LZ4 sequences, Snappy records), and is checked against the oracle decoders:
same verdict, same decoded size, and sync points that are exactly the (ip, op) of every 8th sequence of a serial walk."""
import random

import pytest

import oracle
from conftest import b64d

END, ERR = -1, -2


def seq_at(b, ip, iend):
    """(ok, lit, mlen, offset, nxt, last) from the input bytes only — kernel seq_at()"""
    def rd(p):
        return b[p] if p < len(b) else 0
    token = rd(ip); ip += 1
    lit = token >> 4
    if lit == 15:
        if ip + 15 >= iend: return (False,) + (0,) * 5
        x = rd(ip); ip += 1; lit += x
        if ip + 15 > iend: return (False,) + (0,) * 5
        while x == 255:
            x = rd(ip); ip += 1; lit += x
            if ip + 15 > iend: return (False,) + (0,) * 5
    rem_in = iend - ip
    if rem_in < lit + 8:
        return (rem_in == lit, lit, 0, 0, END, True)
    ip += lit
    off = rd(ip) | (rd(ip + 1) << 8); ip += 2
    mlen = token & 15
    if mlen == 15:
        x = rd(ip); ip += 1; mlen += x
        if ip + 4 > iend: return (False,) + (0,) * 5
        while x == 255:
            x = rd(ip); ip += 1; mlen += x
            if ip + 4 > iend: return (False,) + (0,) * 5
    return (True, lit, mlen + 4, off, ip, False)


def lz4_check(lit, mlen, off, last, op, cap):
    """(ok, op, final) — kernel Lz4Grammar::check"""
    rem_out = cap - op
    if last or rem_out < lit + 12:
        if not last or rem_out < lit: return False, op, False
        return True, op + lit, True
    op += lit
    if off == 0 or off > op or cap - op < mlen + 5: return False, op, False
    return True, op + mlen, False


def snappy_at(b, ip, iend):
    """kernel SnappyGrammar::at: a record = optional literal element + optional copy element"""
    def rd(p, n=1):
        return sum((b[p + k] if p + k < len(b) else 0) << (8 * k) for k in range(n))
    bad = (False,) + (0,) * 5
    tag = rd(ip); lit = mlen = off = 0
    if tag & 3 == 0:
        ip += 1
        ln = (tag >> 2) + 1
        if ln > 60:
            nb = ln - 60
            if iend - ip < nb: return bad
            ln = rd(ip, nb) + 1; ip += nb
        if ln > iend - ip: return bad
        lit = ln; ip += ln
        if ip >= iend: return (True, lit, 0, 0, END, True)
        tag = rd(ip)
        if tag & 3 == 0: return (True, lit, 0, 0, ip, False)
    kind = tag & 3; ip += 1
    if kind == 1:
        if iend - ip < 1: return bad
        mlen = 4 + ((tag >> 2) & 7); off = ((tag >> 5) << 8) | rd(ip); ip += 1
    elif kind == 2:
        if iend - ip < 2: return bad
        mlen = 1 + (tag >> 2); off = rd(ip, 2); ip += 2
    else:
        if iend - ip < 4: return bad
        mlen = 1 + (tag >> 2); off = rd(ip, 4); ip += 4
    if ip >= iend: return (True, lit, mlen, off, END, True)
    return (True, lit, mlen, off, ip, False)


def snappy_check(lit, mlen, off, last, op, dn):
    if lit > dn - op: return False, op, last
    op += lit
    if mlen:
        if off == 0 or off > op or mlen > dn - op: return False, op, last
        op += mlen
    return True, op, last


def spec_parse(b, cap, seq_at=seq_at, seq_check=lz4_check):
    """returns (result, nseq, sync points) like the kernel's spec_walk; result < 0 = corrupt"""
    iend = len(b)
    nl = min(64, (iend + 255) // 256)
    seg = ((((iend + nl - 1) // nl) + 3) & ~3) | 4
    marks = set()
    pos = []
    for l in range(64):                                   # 1a
        p = l * seg if (l < nl and l * seg < iend) else END
        while p >= 0 and p < (l + 1) * seg and p < iend:
            marks.add(p)
            ok, _, _, _, nxt, _ = seq_at(b, p, iend)
            p = nxt if ok else ERR
        pos.append(p)
    merge = []
    for l in range(64):                                   # 1b
        p = pos[l]
        while p >= 0 and p < iend and p not in marks:
            ok, _, _, _, nxt, _ = seq_at(b, p, iend)
            p = nxt if ok else ERR
        if p >= iend: p = ERR
        merge.append(p)
    entry = [0] * 64; chain = []; cur = 0                 # 2
    for _ in range(64):
        chain.append(cur)
        m = merge[cur]
        if m < 0: break
        nxt = m // seg
        assert nxt > cur
        entry[nxt] = m; cur = nxt
    cnt = [0] * 64; outb = [0] * 64                       # 3
    for l in chain:
        q = entry[l]
        while q >= 0 and q != merge[l]:
            ok, lit, mlen, _, nxt, _ = seq_at(b, q, iend)
            if not ok: q = ERR; break
            cnt[l] += 1; outb[l] += lit + mlen; q = nxt
    sync = {}; bad = False; final = None                  # 4
    for l in chain:
        q, idx, op = entry[l], sum(cnt[:l]), sum(outb[:l])
        while q >= 0 and q != merge[l] and not bad:
            if idx % 8 == 0: sync[idx // 8] = (q, op)
            ok, lit, mlen, off, nxt, last = seq_at(b, q, iend)
            if not ok: bad = True; break
            ok, op, fin = seq_check(lit, mlen, off, last, op, cap)
            if not ok: bad = True
            elif fin: final = op; q = END
            else: q = nxt; idx += 1
        if merge[l] == ERR: bad = True
    if bad or final is None: return -7, 0, []
    return final, sum(cnt), [sync[k] for k in sorted(sync)]


def serial_sync(b, cap, seq_at=seq_at):
    ip = op = n = 0; pts = []
    while True:
        if n % 8 == 0: pts.append((ip, op))
        n += 1
        ok, lit, mlen, off, nxt, last = seq_at(b, ip, len(b))
        assert ok
        op += lit + mlen
        if last or nxt == END: return n, pts
        ip = nxt


def check(blob, cap, tag):
    er, eo = oracle.lz4_decompress_raw(blob, cap)
    if cap == 0 or len(blob) == 0: return          # answered by the kernel's prologue, not the walk
    r, nseq, pts = spec_parse(blob, cap)
    assert (r < 0) == (er < 0) and (er < 0 or r == er), (tag, cap, r, er)
    if er >= 0 and cap > 0 and len(blob) > 0:
        n2, p2 = serial_sync(blob, cap)
        assert nseq == n2 and pts == p2, (tag, nseq, n2)


def test_model_on_golden_malformed_and_fuzz(golden):
    for v in golden["vectors"]:
        if v["n"] <= 65536:
            for extra in (0, 5, 33):
                check(b64d(v["lz4"]), v["n"] + extra, v["name"])
    for m in golden["malformed_lz4"]:
        check(b64d(m["data"]), m["cap"], (m["src"], m["kind"], m["k"]))
    random.seed(8)
    for t in range(120):
        n = random.choice([1, 13, 40, 300, 3000, 20000, 65536]); alpha = random.choice([2, 4, 16, 256])
        raw = bytes(random.randrange(alpha) for _ in range(n))
        if random.random() < 0.5 and n > 10: raw = (raw[:random.randrange(1, 20)] * n)[:n]
        _, blob = oracle.lz4_compress_raw(raw)
        check(blob, n, ("fuzz", t)); check(blob, n + random.randrange(1, 40), ("fuzz+", t))
        if n > 20: check(blob, n - random.randrange(1, 12), ("fuzz-", t))
        bb = bytearray(blob); i = random.randrange(len(bb)); bb[i] ^= 1 << random.randrange(8)
        check(bytes(bb), n + 8, ("fuzzbad", t))
    for i in range(4):
        _, blob = oracle.lz4_compress_raw(oracle.synth_v1(65536, i))
        check(blob, 65536, ("synth", i))


def check_snappy(blob, tag):
    """the kernel's prologue (varint length) is restated here; the walk sees the element stream only"""
    er, eo = oracle.snappy_decompress(blob, 1 << 17)
    dn = shift = hdr = 0
    while hdr < len(blob):
        x = blob[hdr]; hdr += 1
        dn |= (x & 0x7f) << shift; shift += 7
        if x < 0x80: break
    else:
        return
    if dn == 0 or dn > 65536 or hdr == len(blob): return
    body = blob[hdr:]
    r, nrec, pts = spec_parse(body, dn, snappy_at, snappy_check)
    if r >= 0 and r != dn: r = -1
    assert (r < 0) == (er < 0) and (er < 0 or r == er), (tag, r, er)
    if er >= 0:
        n2, p2 = serial_sync(body, dn, snappy_at)
        assert nrec == n2 and pts == p2, (tag, nrec, n2)


def test_snappy_model_on_golden_malformed_and_fuzz(golden):
    for v in golden["vectors"]:
        if v["n"] <= 65536: check_snappy(b64d(v["snappy"]), v["name"])
    for m in golden["malformed_snappy"]:
        check_snappy(b64d(m["data"]), (m["src"], m["kind"], m["k"]))
    random.seed(9)
    for t in range(120):
        n = random.choice([1, 13, 40, 300, 3000, 20000, 65536]); alpha = random.choice([2, 4, 16, 256])
        raw = bytes(random.randrange(alpha) for _ in range(n))
        if random.random() < 0.5 and n > 10: raw = (raw[:random.randrange(1, 20)] * n)[:n]
        _, blob = oracle.snappy_compress(raw)
        check_snappy(blob, ("fuzz", t))
        for _ in range(3):
            bb = bytearray(blob); i = random.randrange(len(bb)); bb[i] ^= 1 << random.randrange(8)
            check_snappy(bytes(bb), ("fuzzbad", t))
    for i in range(4):
        _, blob = oracle.snappy_compress(oracle.synth_v1(65536, i))
        check_snappy(blob, ("synth", i))
