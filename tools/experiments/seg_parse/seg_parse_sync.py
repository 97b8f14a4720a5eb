"""How often does a speculative LZ4 / Snappy walk that starts W bytes in front of a segment boundary arrive at the boundary ON the
true token path?  (design input for the segmented parse kernel, csrc/seg_parse.hip: a lane whose lead-in did not synchronise is
dead and its predecessor walks on through its segment.)  CPU only; uses the oracle encoders as data source => lives under tools/models
but is run by hand, not by the product."""
import bz2, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle


def lz4_next(b, p, n):
    """position of the token after the sequence at p, or None (malformed / last)"""
    if p >= n: return None
    t = b[p]; p += 1
    lit = t >> 4
    if lit == 15:
        while True:
            if p >= n: return None
            x = b[p]; p += 1; lit += x
            if x != 255: break
    if n - p < lit + 8: return None
    p += lit + 2
    m = t & 15
    if m == 15:
        while True:
            if p >= n: return None
            x = b[p]; p += 1
            if x != 255: break
    return p


def snappy_next(b, p, n):
    if p >= n: return None
    t = b[p]
    k = t & 3
    if k == 0:
        l = (t >> 2) + 1; p += 1
        if l > 60:
            nb = l - 60
            if p + nb > n: return None
            l = int.from_bytes(b[p:p + nb], "little") + 1; p += nb
        p += l
    elif k == 1: p += 2
    elif k == 2: p += 3
    else: p += 5
    return p if p < n else None


def stats(blobs, nxt, skip_hdr):
    res = {}
    for W in (128, 256, 512, 1024, 2048):
        for k in (4, 8, 16):
            fail = tot = 0
            for b in blobs:
                n = len(b)
                p = skip_hdr(b)
                true = set()
                order = []
                while p is not None:
                    true.add(p); order.append(p); p = nxt(b, p, n)
                seg = -(-n // k)
                for j in range(1, k):
                    bj = j * seg
                    if bj >= n: break
                    want = next((t for t in order if t >= bj), None)
                    q = max(bj - W, skip_hdr(b))
                    while q is not None and q < bj: q = nxt(b, q, n)
                    tot += 1
                    if q != want: fail += 1
            res[(W, k)] = (fail, tot)
    return res


def varint_skip(b):
    p = 0
    while b[p] & 0x80: p += 1
    return p + 1


if __name__ == "__main__":
    synth = [oracle.synth_v1(65536, i) for i in range(64)]
    corpus = []
    for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "corpus", "*.bz2"))):
        d = bz2.decompress(open(f, "rb").read())
        corpus += [d[i:i + 65536] for i in range(0, len(d) - 65535, 65536)][:4]
    for name, raws in (("synth-v1", synth), ("corpus", corpus)):
        for codec, enc, nxt, skip in (("lz4", lambda r: oracle.lz4_compress_raw(r)[1], lz4_next, lambda b: 0),
                                      ("snappy", lambda r: oracle.snappy_compress(r)[1], snappy_next, varint_skip)):
            blobs = [enc(r) for r in raws]
            r = stats(blobs, nxt, skip)
            print(name, codec, "chunks", len(blobs), "avg compressed", sum(map(len, blobs)) // len(blobs))
            for W in (128, 256, 512, 1024, 2048):
                print("   W=%4d " % W + "  ".join("k=%d: %d/%d (%.2f%%)" % (k, r[(W, k)][0], r[(W, k)][1], 100.0 * r[(W, k)][0] / max(r[(W, k)][1], 1)) for k in (4, 8, 16)))


def simulate(b, nxt, start, k, W):
    """the kernel's algorithm on one stream: candidates from lead-ins (restart at p + 1 on a malformed garbage step), every lane walks
    from its candidate until it links (its position == the candidate of the segment it is in) or the stream ends.
    Returns (steps of the slowest lane incl. lead-in, true sequence count, number of live lanes, lanes that had to walk on)"""
    n = len(b)
    seg = max(-(-n // k), 1)
    bnd = [j * seg for j in range(k)]
    cand = [None] * k
    lead_steps = [0] * k
    cand[0] = start
    for j in range(1, k):
        if bnd[j] >= n or bnd[j] <= start: continue
        q = max(bnd[j] - W, start)
        while q is not None and q < bnd[j]:
            nq = nxt(b, q, n)
            lead_steps[j] += 1
            q = nq if nq is not None else (q + 1 if q + 1 < n else None)
        cand[j] = q
    steps = [0] * k; link = [None] * k; extended = 0
    for j in range(k):
        p = cand[j]
        if p is None: continue
        nb = j + 1
        first = True
        while p is not None:
            if nb < k and p >= bnd[nb]:
                m = min(p // seg, k - 1)
                if cand[m] == p: link[j] = m; break
                nb = m + 1
                if j == 0 or True: extended += 0
            p = nxt(b, p, n); steps[j] += 1
    live = [0]; ext_live = 0
    while link[live[-1]] is not None: live.append(link[live[-1]])
    nseq = sum(steps[j] for j in live)
    worst = max(lead_steps[j] + steps[j] for j in range(k))
    return worst, nseq, len(live)


if __name__ == "__main__":
    print("\nfull algorithm: lock-step cost = slowest lane's steps / (sequences / k)")
    for name, raws in (("synth-v1", synth[:32]), ("corpus", corpus)):
        for codec, enc, nxt, skip in (("lz4", lambda r: oracle.lz4_compress_raw(r)[1], lz4_next, lambda b: 0),
                                      ("snappy", lambda r: oracle.snappy_compress(r)[1], snappy_next, varint_skip)):
            blobs = [enc(r) for r in raws]
            for W in (512, 1024, 2048):
                for k in (4, 8, 16):
                    tot_w = tot_i = 0; dead = 0; n_big = 0
                    for b in blobs:
                        w, ns, nl = simulate(b, nxt, skip(b), k, W)
                        if ns < 256: continue
                        n_big += 1
                        tot_w += w; tot_i += ns / k; dead += k - nl
                    print("  %-8s %-6s W=%4d k=%2d: chunks>=256seq %3d  lockstep/ideal %.3f  dead lanes %d" % (name, codec, W, k, n_big, tot_w / max(tot_i, 1), dead))
