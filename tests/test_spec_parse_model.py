"""CPU model of the segmented parse inside the workgroup decoder (cramjam_amd/csrc/lz4_decode_lds.hip: fused_parse; round 1's
parse_spec.hip was the same walk on one wavefront): LANES lanes walk LANES segments of a block from guessed start positions, join
each other's paths, the true path is marked from lane 0 by pointer doubling and stitched from the pieces.  The model mirrors the
kernel phase by phase, for both grammars (LZ4 sequences, Snappy records), and is checked against the oracle decoders: same
verdict, same decoded size, and positions that are exactly the (ip, op) of every 8th sequence of a serial walk.  Every step
also runs the kernels' STRAIGHT-LINE step (parse_grammar.hpp: walk_step) next to the general grammar function and requires
the same answer wherever the straight-line step says it applies."""
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


def lz4_fast_step(b, ip, iend):
    """parse_grammar.hpp walk_step<Lz4Grammar>: None where the straight-line step does not apply"""
    def rd32(p):
        return [b[p + k] if p + k < len(b) else 0xA5 for k in range(4)]      # bytes past the end may be anything
    t = rd32(ip)
    token, e1 = t[0], t[1]
    x1 = (token >> 4) == 15
    lit = (token >> 4) + (e1 if x1 else 0)
    ip1 = ip + 1 + (1 if x1 else 0); ip2 = ip1 + lit
    o = rd32(ip2 if ip2 < iend else ip)
    mc, e2 = token & 15, o[2]
    x2 = mc == 15
    fast = not (x1 and e1 == 255) and not (x2 and e2 == 255) and ip1 + 16 <= iend and iend - ip1 >= lit + 8
    if not fast: return None
    return (True, lit, mc + (e2 if x2 else 0) + 4, o[0] | (o[1] << 8), ip2 + 2 + (1 if x2 else 0), False)


def snappy_fast_step(b, ip, iend):
    """parse_grammar.hpp walk_step<SnappyGrammar> (round 6: literal headers of up to 4 bytes and copy-4 elements are straight-line too)"""
    def rd32(p):
        return [b[p + k] if p + k < len(b) else 0xA5 for k in range(4)]
    t = rd32(ip)
    tag = t[0]; l6 = tag >> 2
    is_lit = (tag & 3) == 0
    lhdr = (1 if l6 < 60 else l6 - 58) if is_lit else 0
    v24 = t[1] | (t[2] << 8) | (t[3] << 16)
    lit = (l6 + 1 if l6 < 60 else (v24 & (0xffffff >> (8 * (62 - min(l6, 62))))) + 1) if is_lit else 0
    ip2 = ip + lhdr + lit
    in2 = ip2 + 4 <= iend
    c = rd32(ip2 if in2 else ip) if is_lit else t
    ctag = c[0]; kind = ctag & 3
    clen = 4 + ((ctag >> 2) & 7) if kind == 1 else 1 + (ctag >> 2)
    off = (((ctag >> 5) << 8) | c[1]) if kind == 1 else (c[1] | (c[2] << 8))
    ip3 = ip2 if kind == 0 else ip2 + (2 if kind == 1 else 3 if kind == 2 else 5)
    fast = not (is_lit and l6 > 62) and in2 and ip3 < iend and (kind != 0 or is_lit)
    if not fast: return None
    if kind == 3:
        o = rd32(ip2 + 1); off = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24)
    return (True, lit, 0 if kind == 0 else clen, 0 if kind == 0 else off, ip3, False)


def with_fast(general, fast):
    def at(b, ip, iend):
        g = general(b, ip, iend)
        f = fast(b, ip, iend)
        if f is not None:
            assert g == f, ("straight-line step disagrees with the grammar function", ip, g, f)
        return g
    return at


LANES = 256


ROWS = 144                         # lds_shared.hpp: kFlRows (P1b's rows follow the longest P1a of the workgroup)
GROUP, MAX_GROUPS = 4, 4096         # lds_shared.hpp: kFlGroup / kFlMaxGroups (more groups: the kernel walks its phase 4)
STATS = {"listed": 0, "walked": 0, "max_b": 0, "groups": 0}


def spec_parse(b, cap, seq_at=seq_at, seq_check=lz4_check, lanes=LANES):
    """returns (result, nseq, sync points) like the kernel's fused_parse; result < 0 = corrupt (the kernel hands those to the
    wave kernel).  Round 6: the kernel LISTS what its walks of 1a / 1b read (one row per loop iteration, one column per lane) and
    takes phase 3 / 4 from the lists — a piece is the suffix of its lane's list that begins at the step standing on the piece's
    entry, and that step's number is the count of the lane's own marks below the entry; the walking phases 3 / 4 remain for chunks
    whose walks outgrow the rows.  The model runs both and requires the same verdict, counts and records."""
    seq_at = with_fast(seq_at, lz4_fast_step if seq_at is globals()["seq_at"] else snappy_fast_step)
    iend = len(b)
    nl = min(lanes, (iend + 63) // 64)
    seg = ((((iend + nl - 1) // nl) + 3) & ~3) | 4
    marks = set()
    pos = []
    lists = [[] for _ in range(lanes)]                       # (lit, mlen, off, output bytes of the walk before the step) or None = not representable in 16-bit fields
    n_a = [0] * lanes
    over = False; row_b = 0
    def put(l, lit, mlen, off):
        ob = lists[l][-1][3] + lists[l][-1][0] + lists[l][-1][1] if lists[l] else 0
        rep = lit <= 0xffff and mlen <= 0xffff and off <= 0xffff
        lists[l].append((lit if rep else 0, mlen if rep else 0, off, ob & 0xffffffff, rep))
    for l in range(lanes):                                   # 1a
        p = l * seg if (l < nl and l * seg < iend) else END
        it = 0
        while p >= 0 and p < (l + 1) * seg and p < iend:
            marks.add(p)
            ok, lit, mlen, off, nxt, _ = seq_at(b, p, iend)
            if ok and it < ROWS: put(l, lit, mlen, off); n_a[l] += 1
            it += 1
            p = nxt if ok else ERR
        row_b = max(row_b, it)
        pos.append(p)
    own_marks = [sorted(m for m in marks if m // seg == l) for l in range(lanes)]
    merge = []
    for l in range(lanes):                                # 1b
        p = pos[l]
        it = 0
        while p >= 0 and p < iend and p not in marks:
            ok, lit, mlen, off, nxt, _ = seq_at(b, p, iend)
            if ok and row_b + it < ROWS: put(l, lit, mlen, off)
            it += 1
            p = nxt if ok else ERR
        over |= row_b + it > ROWS
        STATS["max_b"] = max(STATS["max_b"], it)
        if p >= iend: p = ERR
        merge.append(p)
    # 2: the chain from lane 0, marked by pointer doubling (a piece that ends the stream points at itself)
    active = [l < nl and l * seg < iend for l in range(lanes)]
    nxt = [(merge[l] // seg if (active[l] and 0 <= merge[l] < iend) else l) for l in range(lanes)]
    for l in range(lanes):
        assert nxt[l] >= l
    mark = [l == 0 for l in range(lanes)]
    rounds = 0
    while (1 << rounds) < lanes: rounds += 1
    for _ in range(rounds):
        new = list(mark)
        for l in range(lanes):
            if mark[l]: new[nxt[l]] = True
        nxt = [nxt[nxt[l]] for l in range(lanes)]
        mark = new
    entry = [0] * lanes
    chain = [l for l in range(lanes) if mark[l] and active[l]]
    for l in chain:
        if 0 <= merge[l] < iend: entry[merge[l] // seg] = merge[l]
    # the serial chain of round 1's kernel visits the same lanes
    ser = []; cur = 0
    while True:
        ser.append(cur)
        if not (0 <= merge[cur] < iend): break
        cur = merge[cur] // seg
    assert ser == chain, (ser[:8], chain[:8])
    cnt = [0] * lanes; outb = [0] * lanes                 # 3
    for l in chain:
        q = entry[l]
        while q >= 0 and q != merge[l]:
            ok, lit, mlen, _, nxt, _ = seq_at(b, q, iend)
            if not ok: q = ERR; break
            cnt[l] += 1; outb[l] += lit + mlen; q = nxt
    sync = {}; bad = False; final = None; recs = []       # 4
    for l in chain:
        q, idx, op = entry[l], sum(cnt[:l]), sum(outb[:l])
        while q >= 0 and q != merge[l] and not bad:
            if idx % 8 == 0: sync[idx // 8] = (q, op)
            ok, lit, mlen, off, nxt, last = seq_at(b, q, iend)
            if not ok: bad = True; break
            op0 = op
            ok, op, fin = seq_check(lit, mlen, off, last, op, cap)
            if not ok: bad = True
            else:
                recs.append((idx, lit, mlen, off, op0))
                if fin: final = op; q = END
                else: q = nxt; idx += 1
        if merge[l] == ERR: bad = True
    walked = (-7, 0, []) if (bad or final is None) else (final, sum(cnt), [sync[k] for k in sorted(sync)])
    if over or iend > 65504:                             # (the kernel stages at most 65 504 bytes: longer inputs never reach its parse)
        STATS["walked"] += 1
        return walked
    # ---- phases 3 / 4 from the lists ----
    STATS["listed"] += 1
    k = [0] * lanes; cnt2 = [0] * lanes; outb2 = [0] * lanes
    for l in chain:
        k[l] = sum(1 for m in own_marks[l] if m < entry[l])          # the kernel: popcount of the bitmap words of [l * seg, entry)
        assert k[l] <= len(lists[l])
        cnt2[l] = len(lists[l]) - k[l]
        if cnt2[l]:
            last_e = lists[l][-1]
            outb2[l] = (last_e[3] + last_e[0] + last_e[1] - lists[l][k[l]][3]) & 0xffffffff
    STATS["groups"] = sum((c + GROUP - 1) // GROUP for c in cnt2)
    bad2 = False; final2 = None; recs2 = []
    total = sum(cnt2)
    for l in chain:
        base_idx, base_op = sum(cnt2[:l]), sum(outb2[:l]) & 0xffffffff
        ends = merge[l] == END                               # the lane's last step consumed the input exactly
        for j in range(k[l], len(lists[l])):                 # every cell on its own: its output position is a difference of two list words
            lit, mlen, off, ob, rep = lists[l][j]
            idx, op0 = base_idx + j - k[l], (base_op + ob - lists[l][k[l]][3]) & 0xffffffff
            last = ends and j + 1 == len(lists[l])
            if not rep or idx >= total or op0 > cap: bad2 = True; continue
            ok, op, fin = seq_check(lit, mlen, off, last, op0, cap)
            if not ok: bad2 = True; continue
            recs2.append((idx, lit, mlen, off, op0))
            if fin:
                if final2 is not None: bad2 = True
                final2 = op
        if merge[l] == ERR: bad2 = True
    listed = (-7, 0, []) if (bad2 or final2 is None) else (final2, total, None)
    assert (listed[0] < 0) == (walked[0] < 0), ("listed / walked verdicts differ", listed[:2], walked[:2])
    if walked[0] >= 0:
        assert listed[:2] == walked[:2] and sorted(recs2) == recs, ("listed / walked records differ", listed[:2], walked[:2])
    return walked


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


def test_listed_and_walked_phases_agree_where_the_walks_outgrow_the_rows():
    """hand-made streams for both ends of the round-6 lists: ~30 three-byte sequences per segment (listed; the groups of four cells stay
    below the work list's 4 096 even at the decoder's 16 384 sequences) and four-byte sequences on which no guessed start ever meets
    the true path (every lane walks to the end of the chunk: the rows overflow, phases 3 / 4 walk).  spec_parse itself compares the
    two ways whenever the lists are used."""
    def dense(nseq):
        return bytes(bytearray([0x10, 0x61, 1, 0]) + bytes([0x00, 1, 0]) * nseq + bytes([0xC0]) + b"abcdefghijkl")
    def never_meets(nseq):
        b = bytearray(bytes([0x40]) + b"wxyz" + bytes([4, 0]))
        for i in range(nseq): b += bytes([0x10, 0x61 + (i % 7), 3, 0])
        return bytes(b + bytes([0xC0]) + b"abcdefghijkl")
    for blob, listed in ((dense(16300), True), (dense(4000), True), (never_meets(1200), False)):
        er, _ = oracle.lz4_decompress_raw(blob, 65536)
        STATS.update(listed=0, walked=0, max_b=0, groups=0)
        r, nseq, _ = spec_parse(blob, er, lanes=512)
        assert r == er and (STATS["listed"] == 1) == listed, (len(blob), STATS)
        if listed: assert STATS["groups"] <= MAX_GROUPS
