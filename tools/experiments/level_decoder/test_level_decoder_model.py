"""CPU model of the LEVEL-ORDERED workgroup decoder (cramjam_amd/csrc/lz4_decode_lvl.hip), phase by phase with the
kernel's own arithmetic, checked against the oracle without a GPU:
 * the position index: per 16 bytes of output one word {record-start bits, match-start bits} + {records starting before
   the granule, "the granule's first byte lies in a match"} -> O(1) "which record holds byte p, and is p in its match part";
 * the dependency range of a match = the records [qa, qb] its source bytes touch, and level = 1 + max(level[qa..qb]),
   computed in ANY order with "unknown" markers (the kernel's eight wavefronts race through the batches);
 * the counting sort by level and the level-by-level copy (every match of a level reads only bytes that lower levels and
   the literals produced), self-overlapping matches as a chain of doubling non-overlapping copies.
The model's decode must give the oracle's bytes, its levels must equal the byte-exact longest-path levels."""
import random

import pytest

import oracle
from test_decoder_models import chunks, lz4_records

UNKNOWN = 0xFFFF
G = 16                                         # bytes of output per index granule


def build_index(recs, U):
    """A[g] = start bits | match-start bits << 16;  C[g] = records with start < 16 g | (byte 16 g inside a match) << 15"""
    ng = (U + G - 1) // G + 1
    A = [0] * ng
    C = [0] * ng
    n = len(recs)
    for r, (src, lit, dst, off, m) in enumerate(recs):
        start = dst - lit
        A[start // G] |= 1 << (start % G)
        if m: A[dst // G] |= 0x10000 << (dst % G)
        # granules whose first byte lies in (start of r-1, start of r]: the first record starting at or after them is r
        gp = -1 if r == 0 else (recs[r - 1][2] - recs[r - 1][1]) // G
        pd = 0 if r == 0 else recs[r - 1][2]               # the previous record's match start
        for g in range(gp + 1, start // G + 1):
            C[g] = r | ((1 << 15) if (r > 0 and g * G >= pd and g * G < start) else 0)
    last_start = recs[-1][2] - recs[-1][1]
    for g in range(last_start // G + 1, ng):               # behind the last record's start
        C[g] = n | ((1 << 15) if g * G >= recs[-1][2] and recs[-1][4] else 0)
    return A, C


def lookup(A, C, p):
    """(record holding output byte p, p lies in that record's match part)"""
    g, i = p // G, p % G
    mask = (2 << i) - 1
    s = A[g] & mask
    d = (A[g] >> 16) & mask
    q = (C[g] & 0x7fff) + bin(s).count("1") - 1
    if (s | d) == 0: inm = bool(C[g] >> 15)
    else: inm = d != 0 and d.bit_length() >= s.bit_length()
    return q, inm


def dep_range(recs, A, C, r):
    src, lit, dst, off, m = recs[r]
    s0 = dst - off
    s1 = s0 + min(off, m)
    q0, _ = lookup(A, C, s0)
    q1, inm = lookup(A, C, s1 - 1)
    qb = q1 if inm else q1 - 1
    return q0, min(qb, r - 1)


def levels_any_order(recs, A, C, rnd):
    n = len(recs)
    lvl = [UNKNOWN] * n
    pending = list(range(n))
    sweeps = 0
    while pending:
        rnd.shuffle(pending)                               # the waves' interleaving is arbitrary
        nxt = []
        for r in pending:
            if recs[r][4] == 0: lvl[r] = 0; continue
            qa, qb = dep_range(recs, A, C, r)
            vals = lvl[qa:qb + 1] if qa <= qb else []
            if UNKNOWN in vals: nxt.append(r); continue
            lvl[r] = 1 + max(vals, default=0)
        assert len(nxt) < len(pending)                     # the lowest unresolved record always resolves
        pending = nxt; sweeps += 1
    return lvl


def exact_levels(recs, U):
    """longest path over BYTES (what tools/dag_stats.py computes)"""
    lb = [0] * (U + 1)
    out = []
    for src, lit, dst, off, m in recs:
        if not m: out.append(0); continue
        need = min(off, m)
        L = 1 + max(lb[dst - off:dst - off + need])
        for k in range(dst, dst + m): lb[k] = L
        out.append(L)
    return out


def copy_piecewise(win, dst, off, m, piece=32):
    """a lane's match copy: non-overlapping pieces of <= `piece` bytes; a self-overlapping match reads from a distance that
    doubles while the copied region is still shorter than it (the distance stays a multiple of the period)"""
    d = off
    done = 0
    steps = 0
    while done < m:
        n = min(m - done, d, piece)
        a = dst + done
        assert a - d >= 0
        win[a:a + n] = win[a - d:a - d + n]                # (non-overlapping: n <= d)
        done += n
        if n == d: d *= 2
        steps += 1
    return steps


def decode_by_levels(blk, recs, U, lvl):
    win = bytearray(U)
    for src, lit, dst, off, m in recs:                     # D2: literals (level 0)
        win[dst - lit:dst] = blk[src:src + lit]
    depth = max(lvl)
    hist = [0] * (depth + 2)
    for L in lvl:
        if L: hist[L] += 1
    base = [0] * (depth + 2)
    for L in range(1, depth + 2): base[L] = base[L - 1] + hist[L - 1]
    slot = list(base)
    order = [None] * sum(hist)
    ids = list(range(len(recs)))
    random.Random(1).shuffle(ids)                          # ranks inside a level are arbitrary (atomics)
    for r in ids:
        if lvl[r]:
            order[slot[lvl[r]]] = r; slot[lvl[r]] += 1
    # D3: level by level; inside a level any order, and a copy may only read FINAL bytes: check with a "final" map
    final = bytearray(U)
    for src, lit, dst, off, m in recs:
        for k in range(dst - lit, dst): final[k] = 1
    for L in range(1, depth + 1):
        members = order[base[L]:base[L] + hist[L]]
        for r in members:
            src, lit, dst, off, m = recs[r]
            need = min(off, m)
            assert all(final[dst - off:dst - off + need]), (L, r)
        for r in members:
            src, lit, dst, off, m = recs[r]
            copy_piecewise(win, dst, off, m)
        for r in members:
            src, lit, dst, off, m = recs[r]
            for k in range(dst, dst + m): final[k] = 1
    return bytes(win), hist


def extra_chunks():
    rnd = random.Random(11)
    yield "rle-mix", b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 300) for _ in range(600))[:65536]
    yield "one-literal", rnd.randbytes(65536)
    yield "one-match", bytes(65536)
    yield "tiny", b"howdy neighbor, howdy neighbor!"
    yield "short-periods", b"".join((b"ab" * rnd.randrange(2, 40) + b"xyz" * rnd.randrange(2, 30) + rnd.randbytes(rnd.randrange(1, 9))) for _ in range(700))[:60000]
    yield "long-matches", (rnd.randbytes(3000) * 20)[:65536]
    yield "text", (b"It was the best of times, it was the worst of times, it was the age of wisdom, it was the age of foolishness, " * 700)[:65536]


ALL = list(chunks()) + list(extra_chunks())


@pytest.mark.parametrize("name,data", ALL, ids=[c[0] for c in ALL])
def test_level_ordered_decode_matches_the_oracle(name, data):
    blk = oracle.lz4_compress_raw(data)[1]
    recs = lz4_records(blk)
    U = len(data)
    A, C = build_index(recs, U)
    # the index answers "which record, literal or match part" for every byte
    r = 0
    for p in range(0, U, 7 if U > 4096 else 1):
        while r + 1 < len(recs) and recs[r + 1][2] - recs[r + 1][1] <= p: r += 1
        assert lookup(A, C, p) == (r, p >= recs[r][2]), (name, p)
    lvl = levels_any_order(recs, A, C, random.Random(5))
    assert lvl == exact_levels(recs, U), name
    out, hist = decode_by_levels(blk, recs, U, lvl)
    assert out == data, name


def test_doubling_copy_of_self_overlapping_matches():
    for off in (1, 2, 3, 5, 8, 13, 31, 32, 33, 64):
        for m in (1, 4, 7, 32, 33, 100, 1000):
            win = bytearray(random.Random(off * 1000 + m).randbytes(200)) + bytearray(m)
            ref = bytearray(win)
            for k in range(m): ref[200 + k] = ref[200 + k - off]
            steps = copy_piecewise(win, 200, off, m)
            assert win == ref, (off, m)
            assert steps <= (m + 31) // 32 + 6, (off, m, steps)
