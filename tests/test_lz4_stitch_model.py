"""CPU model of large.hip's LZ4 stitch: the streams of independently compressed 64 KiB pieces are joined into ONE LZ4 block
by dropping every non-final piece's trailing literal-only sequence and prepending its bytes to the literal run of the
next sequence.  The model uses the oracle encoder for the pieces and the oracle decoder as the judge, so it checks the
ALGORITHM (host plan arithmetic + what the stitch kernel writes) without a GPU."""
import random

import pytest

import oracle

PIECE = 65536


def ext(n):
    return 0 if n < 15 else (n - 15) // 255 + 1


def parse(stream):
    """[(token_pos, lit, has_match, end_pos)] of one LZ4 block"""
    out = []; ip = 0; n = len(stream)
    while ip < n:
        t0 = ip; tok = stream[ip]; ip += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = stream[ip]; ip += 1; lit += b
                if b != 255: break
        ip += lit
        if ip >= n:
            out.append((t0, lit, False, ip)); break
        ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = stream[ip]; ip += 1; ml += b
                if b != 255: break
        out.append((t0, lit, True, ip))
    return out


def stitch(data):
    pieces = [data[i:i + PIECE] for i in range(0, len(data), PIECE)]
    out = bytearray(); pending = 0
    for i, p in enumerate(pieces):
        _, s = oracle.lz4_compress_raw(p)
        seqs = parse(s)
        l2 = seqs[0][1]; tail = seqs[-1][1]; r = len(s)
        last = i + 1 == len(pieces); has_match = l2 < len(p)
        assert not seqs[-1][2] and has_match == seqs[0][2]
        if has_match:
            skip = 1 + ext(l2) + l2
            end = r if last else r - (1 + ext(tail) + tail)
            run = pending + l2
            start = i * PIECE - pending
            out.append((min(run, 15) << 4) | (s[0] & 15))
            if run >= 15:
                rem = run - 15; k = ext(run)
                out += bytes([255] * (k - 1) + [rem - 255 * (k - 1)])
            out += data[start:start + run] + s[skip:end]
            pending = 0 if last else tail
        else:
            pending += len(p)
            if last:
                out.append(min(pending, 15) << 4)
                if pending >= 15:
                    rem = pending - 15; k = ext(pending)
                    out += bytes([255] * (k - 1) + [rem - 255 * (k - 1)])
                out += data[len(data) - pending:]
    return bytes(out)


def cases():
    rnd = random.Random(11)
    rb = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
    text = (b"the quick brown fox jumps over the lazy dog, " * 4000)
    yield "synth", b"".join(oracle.synth_v1(PIECE, i) for i in range(3)) + oracle.synth_v1(PIECE, 3)[:12345]
    yield "zeros", bytes(3 * PIECE + 17)
    yield "random", rb(2 * PIECE + 100)                                   # no match anywhere: one literal-only sequence
    yield "random-then-text", rb(PIECE + 500) + text[:PIECE]
    yield "text-random-text", text[:PIECE - 7] + rb(2 * PIECE + 7) + text[:30000]
    yield "tiny-last-piece", text[:2 * PIECE + 3]
    yield "random-tiny-last", rb(PIECE) + b"ab"
    yield "exact", text[:2 * PIECE]


@pytest.mark.parametrize("name,data", list(cases()), ids=[c[0] for c in cases()])
def test_stitched_stream_decodes_with_the_oracle(name, data):
    s = stitch(data)
    r, out = oracle.lz4_decompress_raw(s, len(data))
    assert r == len(data) and out == data
    r2, _ = oracle.lz4_decompress_raw(s, len(data) + 100)                 # and with spare capacity (different end-of-block branch)
    assert r2 == len(data)
    whole = oracle.lz4_compress_raw(data)[1]
    assert len(s) <= len(whole) * 1.02 + 64 * (len(data) // PIECE + 1)
