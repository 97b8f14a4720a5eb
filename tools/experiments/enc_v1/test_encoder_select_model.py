"""CPU model of the encoders' greedy selection (cramjam_amd/csrc/cj_match.hpp: select_walk).

The kernel walks the verified candidates of a round serially but carries only "where did the previous selected match end";
sizes -> output positions, the whole-wave-emission test and the coverage of positions by matches are computed afterwards for all
candidates at once with wave scans (prefix sum, prefix max of the ends, suffix min of the starts).  This model states both
formulations — the plain serial loop that does everything per match, and the walk + scans — and checks that they agree on
random rounds, including the corner cases the scans have to get right (backward extension clamped by the previous match, a
match that starts at its own lane, matches of earlier rounds reaching into this one, empty sub-rounds)."""
import random

KSUB = 5
N = 64 * KSUB


def size_lz4(lit, mcode):
    return 1 + (1 + (lit - 15) // 255 if lit >= 15 else 0) + lit + 2 + (1 + (mcode - 15) // 255 if mcode >= 15 else 0)


def serial(pos, anchor, op, ok, fwd, back):
    """the reference formulation: one loop does everything (what the kernels did before the rewrite)"""
    cur, cur_op = anchor, op
    sel, prev_end, out_pos = [False] * N, [0] * N, [0] * N
    covered = [pos + i < anchor for i in range(N)]
    i = max(0, cur - pos)
    while i < N:
        if not ok[i]:
            i += 1
            continue
        p = pos + i
        room = p - cur
        bk = min(back[i], room)
        lit, mcode, e = room - bk, fwd[i] + bk, p + 4 + fwd[i]
        sel[i], prev_end[i], out_pos[i] = True, cur, cur_op
        cur_op += size_lz4(lit, mcode)
        for k in range(N):
            if p - bk < pos + k < e:
                covered[k] = True
        cur = e
        i = max(i + 1, e - pos)
    return sel, prev_end, out_pos, covered, cur, cur_op


def walk_and_scans(pos, anchor, op, ok, fwd, back):
    """select_walk: minimal serial walk, then everything else per lane with scans over the 320 positions"""
    cur = anchor
    sel, prev_end = [False] * N, [0] * N
    i = max(0, cur - pos)
    while i < N:                                            # the chain: which lane, where the previous match ended
        if ok[i]:
            sel[i], prev_end[i] = True, cur
            cur = pos + i + 4 + fwd[i]
            i = max(i + 1, cur - pos)
        else:
            i += 1
    size, end, start = [0] * N, [0] * N, [1 << 32] * N
    for i in range(N):                                      # per lane, no dependence between lanes
        if sel[i]:
            p = pos + i
            room = p - prev_end[i]
            bk = min(back[i], room)
            size[i] = size_lz4(room - bk, fwd[i] + bk)
            end[i] = p + 4 + fwd[i]
            start[i] = p - bk
    out_pos, run = [0] * N, op
    for i in range(N):                                      # exclusive prefix sum
        out_pos[i] = run
        run += size[i]
    covered, run_end = [False] * N, anchor
    for i in range(N):                                      # exclusive prefix max of the ends (earlier rounds: anchor)
        covered[i] = pos + i < run_end
        run_end = max(run_end, end[i])
    run_start = 1 << 32
    for i in reversed(range(N)):                            # inclusive suffix min of the starts
        run_start = min(run_start, start[i])
        covered[i] = covered[i] or pos + i > run_start
    return sel, prev_end, out_pos, covered, cur, run


def random_round(rnd):
    pos = rnd.randrange(0, 60000)
    anchor = pos + rnd.choice((0, 0, -5, -300, 3, 17, 64, 200, 400))        # a match of an earlier round may reach into this one
    anchor = max(anchor, 0)
    density = rnd.choice((0.0, 0.02, 0.1, 0.3, 0.9))
    ok = [rnd.random() < density for _ in range(N)]
    if rnd.randrange(4) == 0:
        for i in range(64 * rnd.randrange(KSUB), 64 * (rnd.randrange(KSUB) + 1)):
            if i < N: ok[i] = False                                         # empty sub-rounds
    fwd = [rnd.choice((0, 0, 1, 3, 12, 12, 30, 60, 150)) for _ in range(N)]
    back = [rnd.choice((0, 0, 0, 1, 2, 7, 16)) for _ in range(N)]
    return pos, anchor, rnd.randrange(0, 80000), ok, fwd, back


def test_walk_and_scans_equal_the_serial_selection():
    rnd = random.Random(2024)
    for _ in range(3000):
        pos, anchor, op, ok, fwd, back = random_round(rnd)
        a = serial(pos, anchor, op, ok, fwd, back)
        b = walk_and_scans(pos, anchor, op, ok, fwd, back)
        assert a[0] == b[0]                                 # the same lanes are selected
        for i in range(N):
            if a[0][i]:
                assert a[1][i] == b[1][i] and a[2][i] == b[2][i], i       # previous end, output position
        assert a[3] == b[3]                                 # coverage of every position
        assert a[4] == b[4] and a[5] == b[5]                # anchor and output position after the round
