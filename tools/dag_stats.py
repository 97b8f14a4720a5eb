#!/usr/bin/env python3
"""Dependency structure of LZ4 blocks on the benchmark data (synth-v1, liblz4-style encoder from the oracle):
levels of the byte-exact match DAG, what an in-order window of W matches leaves pending, and a small event model
of the resolver loop (per-level latency vs poll period) used to choose between D3 designs.  Dev tool only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle


def parse(blk):
    ip, op, seqs, n = 0, 0, [], len(blk)
    while ip < n:
        t = blk[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = blk[ip]; ip += 1; lit += b
                if b != 255: break
        lit_src = ip
        ip += lit; op += lit
        if ip >= n:
            seqs.append((op - lit, lit, op, 0, 0)); break
        off = blk[ip] | (blk[ip + 1] << 8); ip += 2
        m = t & 15
        if m == 15:
            while True:
                b = blk[ip]; ip += 1; m += b
                if b != 255: break
        m += 4
        seqs.append((op - lit, lit, op, off, m))
        op += m
    return seqs, op


def levels(seqs, U):
    lvl_byte = np.zeros(U + 64, dtype=np.int32)       # level of the producer of every byte (literals 0)
    out = []
    for (ls, lit, dst, off, m) in seqs:
        if m == 0:
            out.append(0); continue
        src = dst - off
        need = min(off, m)
        l = int(lvl_byte[src:src + need].max()) + 1
        lvl_byte[dst:dst + m] = l
        out.append(l)
    return np.array(out)


def main():
    S = 65536
    for idx in range(3):
        raw = oracle.synth_v1(S, idx)
        _, blk = oracle.lz4_compress_raw(raw)
        seqs, U = parse(blk)
        lv = levels(seqs, U)
        ms = np.array([s[4] for s in seqs]); offs = np.array([s[3] for s in seqs]); lits = np.array([s[1] for s in seqs])
        print("chunk %d: C=%d nseq=%d depth=%d mean level %.1f  m>32: %d  overlap: %d  lit>16: %d lit>32: %d  m<=16: %d" % (
            idx, len(blk), len(seqs), lv.max(), lv.mean(), (ms > 32).sum(), ((offs < ms) & (ms > 0)).sum(), (lits > 16).sum(), (lits > 32).sum(), (ms <= 16).sum()))
        h = np.bincount(lv)
        print("  matches per level:", h.tolist())
        # first index at which each level appears; level of the first 64/256/512 records
        for k in (64, 256, 512, 1024):
            print("  max level among first %d records: %d" % (k, lv[:k].max()))
        # record-index distance to the deepest producer
        prod = np.zeros(U + 64, dtype=np.int32)
        dist = []
        for i, (ls, lit, dst, off, m) in enumerate(seqs):
            prod[ls:ls + lit] = -1
            if m:
                src = dst - off; need = min(off, m)
                p = prod[src:src + need].max()
                dist.append(i - p if p >= 0 else -1)
                prod[dst:dst + m] = i
        dist = np.array(dist)
        dep = dist[dist >= 0]
        print("  matches depending on a match: %d of %d; producer within 64 records: %d, within 512: %d, within 1024: %d" % (
            len(dep), len(dist), (dep <= 64).sum(), (dep <= 512).sum(), (dep <= 1024).sum()))


if __name__ == "__main__":
    main()
