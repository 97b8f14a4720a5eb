/* enc2_model.c — scalar CPU statement of the round-based matcher of cramjam_amd/csrc/cj_enc2.hpp (TEST INFRASTRUCTURE).
 *
 * The GPU encoders are specified by this model: a round covers R consecutive positions in BLOCKS of 128 — a block's positions are probed
 * against the hash table and then enter it (all but the followers of a run of equal candidate distances), before the next block is probed (round 6: the table a position sees is at most
 * 128 positions stale; until round 5 a round probed the table as it was when the round began and inserted at its end, only the positions
 * outside the emitted matches — html 4.15 -> 4.48, kppkn.gtb 2.08 -> 2.17, the whole corpus 1.850 -> 1.898 against liblz4's 1.906);
 * only the HEADS of runs of equal candidate distances are verified and become candidates; every head is extended to its true length
 * forwards and backwards; the heads are walked in position order, greedily (the first head whose interval still has four bytes after the
 * previous match ends wins).  tests/test_enc2_gpu.py asserts that the kernels emit exactly these bytes, tests/test_enc2_model.py that the
 * streams decode with the oracle and keep the CPU encoders' ratio — per file of the reference's corpus.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define HASH_BITS 13
#define HASH_SIZE (1u << HASH_BITS)
#define RMAX 1024
#define BLOCK 128u

static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash_slot(uint32_t v) { return (v * 2654435761u) >> (32 - HASH_BITS); }

typedef struct { uint32_t s, e, off; } sel_t;

/* one round; returns the number of selected matches, updates *cur (end of the last selected match) */
static int model_round(const uint8_t* in, uint32_t n, uint16_t* tab, uint32_t pos, uint32_t span, uint32_t R, uint32_t last_start,
                       uint32_t limit, uint32_t* cur_io, sel_t* sel) {
    static uint32_t hs[RMAX], d[RMAX];
    static uint8_t ok[RMAX], valid[RMAX];
    (void)n;
    const uint32_t anchor = *cur_io;
    /* block by block: candidates (the slot's position, as a distance modulo the 64 KiB lap of the 16-bit table), then the block's own
     * positions into the table — k-major (lane l owns positions 4 l + k of its group of 256, half a wavefront is a block): the kernel issues
     * one store instruction per (block, k), and within an instruction the highest lane wins a contested slot */
    for (uint32_t b0 = 0; b0 < R; b0 += BLOCK) {
        for (uint32_t i = b0; i < b0 + BLOCK; i++) {
            const uint32_t p = pos + i;
            valid[i] = i < span && p <= last_start;
            ok[i] = 0; d[i] = 0;
            if (!valid[i]) continue;
            hs[i] = hash_slot(ld32(in + p));
            const uint32_t dist = (p - tab[hs[i]]) & 0xffffu;
            if (dist != 0u && dist <= p) d[i] = dist;
        }
        /* ... except the FOLLOWERS: a position whose candidate distance equals its left neighbour's lies inside that neighbour's match if
         * it is one — what the coverage rule of rounds 1-5 kept out of the table, decided here before anything is verified (benchmark
         * data 1.616 -> 1.622, xml 4.31 -> 4.39; the corpus 1.898 -> 1.901) */
        for (uint32_t k = 0; k < 4u; k++)
            for (uint32_t i = b0 + k; i < b0 + BLOCK; i += 4u) {
                if (!valid[i]) continue;
                if (d[i] != 0u && (i & 255u) != 0u && d[i - 1] == d[i]) continue;
                tab[hs[i]] = (uint16_t)(pos + i);
            }
    }
    /* Only the FIRST position of a run of equal distances is verified (its left neighbour IN THE SAME GROUP OF 256 has another distance or
     * no candidate): the positions behind it lie inside its match if it is one, and a candidate dword costs a scattered memory access
     * each — on match-heavy data half of all positions are such followers. */
    for (uint32_t i = 0; i < R; i++) {
        if (d[i] == 0u) continue;
        if ((i & 255u) != 0u && d[i - 1] == d[i]) continue;
        if (ld32(in + pos + i - d[i]) == ld32(in + pos + i)) ok[i] = 1;
    }
    uint32_t cur = anchor;
    int ns = 0;
    for (uint32_t i = 0; i < R; i++) {
        if (!ok[i]) continue;            /* a verified head */
        const uint32_t p = pos + i, c = p - d[i];
        uint32_t e = p + 4u;
        while (e < limit && in[e] == in[e - d[i]]) e++;
        uint32_t s;
        if (p >= cur) {
            uint32_t room = p - cur, bk = 0;
            if (room > c) room = c;
            while (bk < room && in[p - 1u - bk] == in[c - 1u - bk]) bk++;
            s = p - bk;
        } else s = cur;
        if (e < s + 4u || s > last_start) continue;
        sel[ns].s = s; sel[ns].e = e; sel[ns].off = d[i]; ns++;
        cur = e;
    }
    *cur_io = cur;
    return ns;
}

static size_t lz4_put_seq(uint8_t* out, size_t op, const uint8_t* lit_src, uint32_t lit, uint32_t off, uint32_t mlen, int last) {
    const uint32_t mcode = last ? 0u : mlen - 4u;
    out[op++] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
    if (lit >= 15u) { uint32_t v = lit - 15u; while (v >= 255u) { out[op++] = 255u; v -= 255u; } out[op++] = (uint8_t)v; }
    memcpy(out + op, lit_src, lit); op += lit;
    if (last) return op;
    out[op++] = (uint8_t)off; out[op++] = (uint8_t)(off >> 8);
    if (mcode >= 15u) { uint32_t v = mcode - 15u; while (v >= 255u) { out[op++] = 255u; v -= 255u; } out[op++] = (uint8_t)v; }
    return op;
}

/* LZ4 block (no size prefix); out must hold LZ4_compressBound(n).  R = positions per round (multiple of 256, <= RMAX) */
int64_t enc2_model_lz4(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t R) {
    size_t op = 0;
    uint32_t anchor = 0;
    if (n >= 13u) {
        uint16_t* tab = (uint16_t*)calloc(HASH_SIZE, 2);
        sel_t* sel = (sel_t*)malloc(sizeof(sel_t) * (RMAX + 2));
        const uint32_t last_start = n - 12u, limit = n - 5u;
        uint32_t pos = 0, span = 64u;
        while (pos <= last_start) {
            uint32_t cur = anchor;
            const int ns = model_round(in, n, tab, pos, span, R, last_start, limit, &cur, sel);
            uint32_t a = anchor;
            for (int q = 0; q < ns; q++) { op = lz4_put_seq(out, op, in + a, sel[q].s - a, sel[q].off, sel[q].e - sel[q].s, 0); a = sel[q].e; }
            anchor = cur;
            const uint32_t round_end = pos + span;
            span = span * 2u < R ? span * 2u : R;
            pos = anchor > round_end ? anchor : round_end;
        }
        free(tab); free(sel);
    }
    op = lz4_put_seq(out, op, in + anchor, n - anchor, 0, 0, 1);
    return (int64_t)op;
}

static size_t sn_put_literal(uint8_t* out, size_t op, const uint8_t* src, uint32_t len) {
    const uint32_t n1 = len - 1u;
    if (n1 < 60u) out[op++] = (uint8_t)(n1 << 2);
    else {
        const uint32_t nb = n1 < 256u ? 1u : n1 < 65536u ? 2u : n1 < 16777216u ? 3u : 4u;
        out[op++] = (uint8_t)((59u + nb) << 2);
        for (uint32_t k = 0; k < nb; k++) out[op++] = (uint8_t)(n1 >> (8u * k));
    }
    memcpy(out + op, src, len);
    return op + len;
}
static size_t sn_put_copy(uint8_t* out, size_t op, uint32_t off, uint32_t len) {
    while (len >= 68u) { out[op++] = (uint8_t)(2u | (63u << 2)); out[op++] = (uint8_t)off; out[op++] = (uint8_t)(off >> 8); len -= 64u; }
    if (len > 64u) { out[op++] = (uint8_t)(2u | (59u << 2)); out[op++] = (uint8_t)off; out[op++] = (uint8_t)(off >> 8); len -= 60u; }
    if (len < 12u && off < 2048u) { out[op++] = (uint8_t)(1u | ((len - 4u) << 2) | ((off >> 8) << 5)); out[op++] = (uint8_t)off; }
    else { out[op++] = (uint8_t)(2u | ((len - 1u) << 2)); out[op++] = (uint8_t)off; out[op++] = (uint8_t)(off >> 8); }
    return op;
}

/* Snappy raw (with the varint preamble); out must hold 32 + n + n / 6 */
int64_t enc2_model_snappy(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t R) {
    size_t op = 0;
    { uint32_t v = n; while (v >= 0x80u) { out[op++] = (uint8_t)(v | 0x80u); v >>= 7; } out[op++] = (uint8_t)v; }
    uint32_t anchor = 0;
    if (n >= 8u) {
        uint16_t* tab = (uint16_t*)calloc(HASH_SIZE, 2);
        sel_t* sel = (sel_t*)malloc(sizeof(sel_t) * (RMAX + 2));
        const uint32_t last_start = n - 8u, limit = n;      /* the kernels' position lanes read 8 bytes at a time */
        uint32_t pos = 0, span = 64u;
        while (pos <= last_start) {
            uint32_t cur = anchor;
            const int ns = model_round(in, n, tab, pos, span, R, last_start, limit, &cur, sel);
            uint32_t a = anchor;
            for (int q = 0; q < ns; q++) {
                if (sel[q].s > a) op = sn_put_literal(out, op, in + a, sel[q].s - a);
                op = sn_put_copy(out, op, sel[q].off, sel[q].e - sel[q].s);
                a = sel[q].e;
            }
            anchor = cur;
            const uint32_t round_end = pos + span;
            span = span * 2u < R ? span * 2u : R;
            pos = anchor > round_end ? anchor : round_end;
        }
        free(tab); free(sel);
    }
    if (anchor < n) op = sn_put_literal(out, op, in + anchor, n - anchor);
    return (int64_t)op;
}
