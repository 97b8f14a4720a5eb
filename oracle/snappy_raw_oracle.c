/*
 * snappy_raw_oracle.c — CPU ORACLE (test infrastructure only; see cj_oracle.h header).
 *
 * Restates the Snappy *raw* codec the reference reaches through
 *   /root/reference/src/snappy.rs:57,75,97,106,114,121
 *   -> libcramjam 0.8.0 snappy::raw::{compress,decompress,compress_vec,decompress_vec}
 *   -> snap 1.1.1 raw::{Encoder::compress, Decoder::decompress, max_compress_len, decompress_len}
 * snap is not under /root/reference; this file is written from the public format description
 * and the published encoder algorithm (64 KiB fragments, <=16384-entry u16 hash table,
 * skip heuristic starting at 32) and is checked bit-for-bit against libsnappy in
 * tests/golden/make_golden.py.
 */
#include "cj_oracle.h"
#include <string.h>

#define MAX_BLOCK 65536u
#define INPUT_MARGIN 15u
#define MIN_NON_LITERAL_BLOCK (1 + 1 + INPUT_MARGIN)
#define MAX_TABLE 16384u

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

size_t cjo_snappy_max_compress_len(size_t n) {
    if ((uint64_t)n > 0xFFFFFFFFull) return 0;
    uint64_t m = 32 + (uint64_t)n + (uint64_t)n / 6;
    return m > 0xFFFFFFFFull ? 0 : (size_t)m;
}

/* snap bytes::read_varu64: returns header length, 0 on malformed/overflow */
static size_t read_varu64(const uint8_t* in, size_t n, uint64_t* out) {
    uint64_t v = 0;
    unsigned shift = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t b = in[i];
        if (b < 0x80) {
            if (i > 9 || (i == 9 && b > 1)) return 0;
            *out = v | ((uint64_t)b << shift);
            return i + 1;
        }
        v |= ((uint64_t)(b & 0x7f)) << shift;
        shift += 7;
        if (shift > 63) return 0;
    }
    return 0;
}

int64_t cjo_snappy_decompress_len(const uint8_t* in, size_t n) {
    if (n == 0) return 0;
    uint64_t len;
    size_t h = read_varu64(in, n, &len);
    if (h == 0) return CJO_E_SNAPPY_HEADER;
    if (len > 0xFFFFFFFFull) return CJO_E_SNAPPY_TOO_BIG;
    return (int64_t)len;
}

int64_t cjo_snappy_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if (n == 0) return CJO_E_SNAPPY_EMPTY;
    uint64_t ulen;
    size_t h = read_varu64(in, n, &ulen);
    if (h == 0) return CJO_E_SNAPPY_HEADER;
    if (ulen > 0xFFFFFFFFull) return CJO_E_SNAPPY_TOO_BIG;
    if (ulen > cap) return CJO_E_SNAPPY_BUF_SMALL;
    const uint8_t* src = in + h;
    size_t sn = n - h, s = 0, d = 0, dn = (size_t)ulen;
    while (s < sn) {
        unsigned tag = src[s++];
        size_t len, offset;
        switch (tag & 3) {
        case 0: {
            len = (tag >> 2) + 1;
            if (len > 60) {
                unsigned nb = (unsigned)len - 60;
                if (sn - s < nb) return CJO_E_SNAPPY_CORRUPT;
                uint32_t v = 0;
                for (unsigned i = 0; i < nb; i++) v |= (uint32_t)src[s + i] << (8 * i);
                s += nb;
                len = (size_t)v + 1;
            }
            if (len > sn - s || len > dn - d) return CJO_E_SNAPPY_CORRUPT;
            memcpy(out + d, src + s, len);
            s += len; d += len;
            continue;
        }
        case 1:
            if (sn - s < 1) return CJO_E_SNAPPY_CORRUPT;
            len = 4 + ((tag >> 2) & 7);
            offset = ((size_t)(tag >> 5) << 8) | src[s];
            s += 1;
            break;
        case 2:
            if (sn - s < 2) return CJO_E_SNAPPY_CORRUPT;
            len = 1 + (tag >> 2);
            offset = src[s] | ((size_t)src[s + 1] << 8);
            s += 2;
            break;
        default:
            if (sn - s < 4) return CJO_E_SNAPPY_CORRUPT;
            len = 1 + (tag >> 2);
            offset = rd32(src + s);
            s += 4;
            break;
        }
        if (offset == 0 || offset > d) return CJO_E_SNAPPY_CORRUPT;
        if (len > dn - d) return CJO_E_SNAPPY_CORRUPT;
        for (size_t i = 0; i < len; i++) out[d + i] = out[d - offset + i];
        d += len;
    }
    if (d != dn) return CJO_E_SNAPPY_CORRUPT;
    return (int64_t)dn;
}

/* ---------------- encoder ---------------- */

static uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, size_t len) {
    size_t n = len - 1;
    if (n < 60) {
        *op++ = (uint8_t)(n << 2);
    } else if (n < 256) {
        *op++ = 60 << 2; *op++ = (uint8_t)n;
    } else if (n < 65536) {
        *op++ = 61 << 2; *op++ = (uint8_t)n; *op++ = (uint8_t)(n >> 8);
    } else if (n < 16777216) {
        *op++ = 62 << 2; *op++ = (uint8_t)n; *op++ = (uint8_t)(n >> 8); *op++ = (uint8_t)(n >> 16);
    } else {
        *op++ = 63 << 2; *op++ = (uint8_t)n; *op++ = (uint8_t)(n >> 8); *op++ = (uint8_t)(n >> 16); *op++ = (uint8_t)(n >> 24);
    }
    memcpy(op, lit, len);
    return op + len;
}

static uint8_t* emit_copy_upto64(uint8_t* op, size_t offset, size_t len) {
    if (len < 12 && offset < 2048) {
        *op++ = (uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5));
        *op++ = (uint8_t)offset;
    } else {
        *op++ = (uint8_t)(2 | ((len - 1) << 2));
        *op++ = (uint8_t)offset; *op++ = (uint8_t)(offset >> 8);
    }
    return op;
}

static uint8_t* emit_copy(uint8_t* op, size_t offset, size_t len) {
    while (len >= 68) { op = emit_copy_upto64(op, offset, 64); len -= 64; }
    if (len > 64) { op = emit_copy_upto64(op, offset, 60); len -= 60; }
    return emit_copy_upto64(op, offset, len);
}

static uint8_t* compress_fragment(const uint8_t* src, size_t n, uint8_t* op, uint16_t* table) {
    /* table size: smallest power of two >= n, clamped to [256, 16384] */
    unsigned tsize = 256, shift = 32 - 8;
    while (tsize < MAX_TABLE && tsize < n) { tsize <<= 1; shift--; }
    memset(table, 0, tsize * sizeof(uint16_t));
#define HASH(v) (((uint32_t)(v) * 0x1e35a7bdu) >> shift)
    size_t s = 0, next_emit = 0;
    if (n >= MIN_NON_LITERAL_BLOCK) {
        const size_t s_limit = n - INPUT_MARGIN;
        s = 1;
        uint32_t next_hash = HASH(rd32(src + s));
        for (;;) {
            uint32_t skip = 32;
            size_t s_next = s, cand;
            for (;;) {
                s = s_next;
                uint32_t between = skip >> 5;
                s_next = s + between;
                skip += between;
                if (s_next > s_limit) goto emit_remainder;
                cand = table[next_hash];
                table[next_hash] = (uint16_t)s;
                next_hash = HASH(rd32(src + s_next));
                if (rd32(src + s) == rd32(src + cand)) break;
            }
            op = emit_literal(op, src + next_emit, s - next_emit);
            for (;;) {
                size_t base = s;
                s += 4;
                size_t c = cand + 4;
                while (s < n && src[s] == src[c]) { s++; c++; }
                op = emit_copy(op, base - cand, s - base);
                next_emit = s;
                if (s >= s_limit) goto emit_remainder;
                uint64_t x = rd64(src + s - 1);
                table[HASH((uint32_t)x)] = (uint16_t)(s - 1);
                uint32_t cur_hash = HASH((uint32_t)(x >> 8));
                cand = table[cur_hash];
                table[cur_hash] = (uint16_t)s;
                if ((uint32_t)(x >> 8) != rd32(src + cand)) {
                    next_hash = HASH((uint32_t)(x >> 16));
                    s += 1;
                    break;
                }
            }
        }
    }
emit_remainder:
    if (next_emit < n) op = emit_literal(op, src + next_emit, n - next_emit);
    return op;
#undef HASH
}

int64_t cjo_snappy_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    size_t need = cjo_snappy_max_compress_len(n);
    if (need == 0) return CJO_E_SNAPPY_TOO_BIG;
    if (cap < need) return CJO_E_SNAPPY_BUF_SMALL;
    if (n == 0) { out[0] = 0; return 1; }
    uint8_t* op = out;
    uint64_t v = n;
    while (v >= 0x80) { *op++ = (uint8_t)(v | 0x80); v >>= 7; }
    *op++ = (uint8_t)v;
    uint16_t table[MAX_TABLE];
    for (size_t pos = 0; pos < n; pos += MAX_BLOCK) {
        size_t len = n - pos < MAX_BLOCK ? n - pos : MAX_BLOCK;
        op = compress_fragment(in + pos, len, op, table);
    }
    return (int64_t)(op - out);
}
