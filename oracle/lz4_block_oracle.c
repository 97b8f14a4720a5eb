/*
 * lz4_block_oracle.c — CPU ORACLE (test infrastructure only; see cj_oracle.h header).
 *
 * Restates the LZ4 *block* codec the reference reaches through
 *   /root/reference/src/lz4.rs:88,90,127,164,168,206,228
 *   -> libcramjam 0.8.0 lz4::block::{compress_into,decompress_into,compress_bound}
 *   -> lz4 1.28.1 block::{compress_to_buffer,decompress_to_buffer}
 *   -> liblz4 1.10.0 LZ4_compress_default / LZ4_decompress_safe / LZ4_compressBound
 * None of those sources are under /root/reference; this file is written from the public
 * block-format description and the published behaviour of the fast compressor (greedy,
 * single-probe hash table, skip acceleration 1) and is checked bit-for-bit against the
 * system liblz4 in tests/golden/make_golden.py.
 */
#include "cj_oracle.h"
#include <string.h>

#define MINMATCH 4
#define MFLIMIT 12
#define LASTLITERALS 5
#define LZ4_MIN_LENGTH (MFLIMIT + 1)
#define LZ4_MAX_INPUT 0x7E000000u
#define LZ4_64K_LIMIT (65536 + (MFLIMIT - 1))
#define SKIP_TRIGGER 6
#define MAX_DISTANCE 65535u

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

size_t cjo_lz4_compress_bound_raw(size_t n) {
    return n > LZ4_MAX_INPUT ? 0 : n + n / 255 + 16;
}

/* hash of the 4 bytes at p into a 13-bit (u16 table) index */
static inline uint32_t hash_u16tab(const uint8_t* p) { return (rd32(p) * 2654435761u) >> (32 - 13); }
/* 64-bit builds hash 5 bytes into a 12-bit index when the table holds u32 positions */
static inline uint32_t hash_u32tab(const uint8_t* p) {
    const uint64_t prime5 = 889523592379ull;
    return (uint32_t)(((rd64(p) << 24) * prime5) >> (64 - 12));
}

static inline unsigned count_match(const uint8_t* a, const uint8_t* b, const uint8_t* alimit) {
    const uint8_t* s = a;
    while (a < alimit && *a == *b) { a++; b++; }
    return (unsigned)(a - s);
}

/* Greedy fast compressor, acceleration 1, no dictionary.  `small` selects the 8192 x u16 table
 * (inputs below 64 KiB + 11) versus the 4096 x u32 table. Returns 0 when dst is too small. */
static int64_t lz4_fast(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int small) {
    uint32_t table[4096];           /* 16 KiB either way */
    uint16_t* t16 = (uint16_t*)table;
    memset(table, 0, sizeof table);
#define HASH(p) (small ? hash_u16tab(p) : hash_u32tab(p))
#define GET(h) (small ? (uint32_t)t16[h] : table[h])
#define PUT(h, v) do { if (small) t16[h] = (uint16_t)(v); else table[h] = (uint32_t)(v); } while (0)
    const uint8_t* ip = src;
    const uint8_t* anchor = src;
    const uint8_t* const iend = src + n;
    const uint8_t* const mflimit_plus1 = iend - MFLIMIT + 1;
    const uint8_t* const matchlimit = iend - LASTLITERALS;
    uint8_t* op = dst;
    uint8_t* const olimit = dst + cap;
    const int limited = cap < cjo_lz4_compress_bound_raw(n);

    if (n < LZ4_MIN_LENGTH) goto last_literals;

    PUT(HASH(ip), 0);
    ip++;
    uint32_t forward_h = HASH(ip);

    for (;;) {
        const uint8_t* match;
        uint8_t* token;
        {   /* find a match: probe, insert, step grows after 64 misses */
            const uint8_t* forward_ip = ip;
            unsigned step = 1, search_nb = 1u << SKIP_TRIGGER;
            for (;;) {
                uint32_t h = forward_h;
                uint32_t cur = (uint32_t)(forward_ip - src);
                uint32_t mi = GET(h);
                ip = forward_ip;
                forward_ip += step;
                step = search_nb++ >> SKIP_TRIGGER;
                if (forward_ip > mflimit_plus1) goto last_literals;
                match = src + mi;
                forward_h = HASH(forward_ip);
                PUT(h, cur);
                if (!small && mi + MAX_DISTANCE < cur) continue;   /* too far */
                if (rd32(match) == rd32(ip)) break;
            }
        }
        /* catch up: extend backwards */
        while (ip > anchor && match > src && ip[-1] == match[-1]) { ip--; match--; }

        {   /* literals */
            unsigned lit = (unsigned)(ip - anchor);
            token = op++;
            if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > olimit) return 0;
            if (lit >= 15) {
                int len = (int)lit - 15;
                *token = 15 << 4;
                for (; len >= 255; len -= 255) *op++ = 255;
                *op++ = (uint8_t)len;
            } else *token = (uint8_t)(lit << 4);
            memcpy(op, anchor, lit);
            op += lit;
        }
    next_match:
        op[0] = (uint8_t)(ip - match); op[1] = (uint8_t)((ip - match) >> 8); op += 2;
        {
            unsigned mc = count_match(ip + MINMATCH, match + MINMATCH, matchlimit);
            ip += (size_t)mc + MINMATCH;
            if (limited && op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit) return 0;
            if (mc >= 15) {
                *token += 15;
                mc -= 15;
                while (mc >= 255) { *op++ = 255; mc -= 255; }
                *op++ = (uint8_t)mc;
            } else *token += (uint8_t)mc;
        }
        anchor = ip;
        if (ip >= mflimit_plus1) break;

        PUT(HASH(ip - 2), (uint32_t)(ip - 2 - src));
        {   /* test next position immediately */
            uint32_t h = HASH(ip);
            uint32_t cur = (uint32_t)(ip - src);
            uint32_t mi = GET(h);
            match = src + mi;
            PUT(h, cur);
            if ((small || mi + MAX_DISTANCE >= cur) && rd32(match) == rd32(ip)) {
                token = op++;
                *token = 0;
                goto next_match;
            }
        }
        forward_h = HASH(++ip);
    }

last_literals:
    {
        size_t last = (size_t)(iend - anchor);
        if (limited && op + last + 1 + ((last + 255 - 15) / 255) > olimit) return 0;
        if (last >= 15) {
            size_t acc = last - 15;
            *op++ = 15 << 4;
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else *op++ = (uint8_t)(last << 4);
        memcpy(op, anchor, last);
        op += last;
    }
    return (int64_t)(op - dst);
#undef HASH
#undef GET
#undef PUT
}

int64_t cjo_lz4_compress_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (n > LZ4_MAX_INPUT) return 0;
    if (cap == 0) return 0;
    return lz4_fast(src, n, dst, cap, n < LZ4_64K_LIMIT);
}

/* variable-length field reader of the 1.10.0 safe decoder: returns -1 on error */
static inline int64_t read_vlen(const uint8_t** ipp, const uint8_t* ilimit, int initial_check) {
    const uint8_t* ip = *ipp;
    int64_t len = 0;
    unsigned s;
    if (initial_check && ip >= ilimit) return -1;
    do {
        s = *ip++;
        len += s;
        if (ip > ilimit) return -1;
    } while (s == 255);
    *ipp = ip;
    return len;
}

/* hist: bytes of valid history directly before dst (LZ4_decompress_safe_usingDict with a contiguous prefix) */
static int64_t lz4_dec(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t hist) {
    const uint8_t* ip = src;
    const uint8_t* const iend = src + n;
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    if (cap == 0) return (n == 1 && src[0] == 0) ? 0 : -1;
    if (n == 0) return -1;
    for (;;) {
        unsigned token = *ip++;
        size_t len = token >> 4;
        if (len == 15) {
            int64_t a = read_vlen(&ip, iend - 15, 1);
            if (a < 0) return -1;
            len += (size_t)a;
        }
        /* literals: a run that ends within 12 bytes of the output end or 8 bytes of the
         * input end must be the final sequence and must consume the input exactly */
        if ((size_t)(oend - op) < len + MFLIMIT || (size_t)(iend - ip) < len + (2 + 1 + LASTLITERALS)) {
            if ((size_t)(iend - ip) != len || (size_t)(oend - op) < len) return -1;
            memmove(op, ip, len);
            op += len;
            break;
        }
        memcpy(op, ip, len);
        ip += len;
        op += len;
        unsigned offset = ip[0] | ((unsigned)ip[1] << 8);
        ip += 2;
        size_t mlen = token & 15;
        if (mlen == 15) {
            int64_t a = read_vlen(&ip, iend - LASTLITERALS + 1, 0);
            if (a < 0) return -1;
            mlen += (size_t)a;
        }
        mlen += MINMATCH;
        if (offset > (size_t)(op - dst) + hist) return -1;
        /* offset 0 is spec-invalid; liblz4 leaves the output bytes untouched (reads garbage).
         * The build rejects it (documented deviation). */
        if (offset == 0) return -1;
        if ((size_t)(oend - op) < mlen + LASTLITERALS) return -1;   /* last 5 bytes must be literals */
        {
            const uint8_t* m = op - offset;
            for (size_t i = 0; i < mlen; i++) op[i] = m[i];
            op += mlen;
        }
    }
    return (int64_t)(op - dst);
}

int64_t cjo_lz4_decompress_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    return lz4_dec(src, n, dst, cap, 0);
}

/* linked LZ4-frame blocks: buf[0..hist) is the previous output, the block decodes to buf + hist (capacity cap) */
int64_t cjo_lz4_decompress_with_prefix(const uint8_t* src, size_t n, uint8_t* buf, size_t hist, size_t cap) {
    return lz4_dec(src, n, buf + hist, cap, hist);
}

/* Compress base[hist .. hist+len) as ONE block whose matches may reach into base[0..hist) (hist <= 65536): a plain
 * greedy single-probe matcher with the history pre-indexed.  Used only to mint linked-block test frames — the bytes are
 * not meant to equal liblz4's.  Returns 0 when the result does not fit cap; hist == 0 uses the liblz4-identical encoder. */
int64_t cjo_lz4_compress_with_prefix(const uint8_t* base, size_t hist, size_t len, uint8_t* dst, size_t cap) {
    if (hist == 0) return cap == 0 ? 0 : cjo_lz4_compress_raw(base, len, dst, cap);
    enum { HB = 15 };
    static uint32_t tab[1 << HB];
    const size_t end = hist + len;
    memset(tab, 0xff, sizeof tab);
    for (size_t i = 0; i + 4 <= hist + (len >= 4 ? 0 : 0) && i + 4 <= end; i++) {
        if (i >= hist) break;
        tab[(rd32(base + i) * 2654435761u) >> (32 - HB)] = (uint32_t)i;
    }
    uint8_t* op = dst; uint8_t* const oend = dst + cap;
    size_t anchor = hist, ip = hist;
    const size_t mflimit = end >= MFLIMIT ? end - MFLIMIT : 0, matchlimit = end >= LASTLITERALS ? end - LASTLITERALS : 0;
    if (len >= LZ4_MIN_LENGTH) {
        while (ip <= mflimit && ip + 4 <= end) {
            uint32_t h = (rd32(base + ip) * 2654435761u) >> (32 - HB);
            uint32_t c = tab[h];
            tab[h] = (uint32_t)ip;
            if (c != 0xffffffffu && ip - c <= MAX_DISTANCE && rd32(base + c) == rd32(base + ip)) {
                size_t ml = 4;
                while (ip + ml < matchlimit && base[c + ml] == base[ip + ml]) ml++;
                size_t lit = ip - anchor, mc = ml - 4;
                if ((size_t)(oend - op) < 1 + lit + lit / 255 + 1 + 2 + mc / 255 + 1) return 0;
                uint8_t* tok = op++;
                if (lit >= 15) { size_t v = lit - 15; *tok = 0xF0; while (v >= 255) { *op++ = 255; v -= 255; } *op++ = (uint8_t)v; }
                else *tok = (uint8_t)(lit << 4);
                memcpy(op, base + anchor, lit); op += lit;
                size_t off = ip - c;
                *op++ = (uint8_t)off; *op++ = (uint8_t)(off >> 8);
                if (mc >= 15) { size_t v = mc - 15; *tok |= 15; while (v >= 255) { *op++ = 255; v -= 255; } *op++ = (uint8_t)v; }
                else *tok |= (uint8_t)mc;
                ip += ml; anchor = ip;
            } else ip++;
        }
    }
    size_t lit = end - anchor;
    if ((size_t)(oend - op) < 1 + lit + lit / 255 + 1) return 0;
    uint8_t* tok = op++;
    if (lit >= 15) { size_t v = lit - 15; *tok = 0xF0; while (v >= 255) { *op++ = 255; v -= 255; } *op++ = (uint8_t)v; }
    else *tok = (uint8_t)(lit << 4);
    memcpy(op, base + anchor, lit); op += lit;
    return (int64_t)(op - dst);
}

/* ---------------- libcramjam::lz4::block wrapper semantics ---------------- */

size_t cjo_lz4_block_compress_bound(size_t n, int prepend) {
    size_t b = cjo_lz4_compress_bound_raw(n);
    /* libcramjam: bound + 4 when the size prefix is stored (reference tests/test_variants.py:277-279) */
    return prepend ? b + 4 : b;
}

int64_t cjo_lz4_block_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int prepend) {
    if (n > 0x7FFFFFFFu || cjo_lz4_compress_bound_raw(n) == 0) return CJO_E_INPUT_TOO_LARGE;
    uint8_t* dst = out;
    size_t dcap = cap;
    if (prepend) {
        if (cap < 4) return CJO_E_COMPRESS_FAILED;
        out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 24);
        dst += 4; dcap -= 4;
    }
    int64_t r = cjo_lz4_compress_raw(in, n, dst, dcap);
    if (r <= 0) return CJO_E_COMPRESS_FAILED;
    return prepend ? r + 4 : r;
}

int64_t cjo_lz4_block_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int size_prepended) {
    int64_t size;
    if (size_prepended) {
        if (n < 4) return CJO_E_NO_PREFIX;
        size = (int32_t)((uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24));
        in += 4; n -= 4;
    } else {
        size = (int32_t)(uint32_t)cap;   /* lz4 crate takes the capacity as an i32 */
    }
    if (size < 0) return CJO_E_NEG_PREFIX;
    if (cjo_lz4_compress_bound_raw((size_t)size) == 0) return CJO_E_PREFIX_TOO_BIG;
    if ((size_t)size > cap) return CJO_E_OUT_TOO_SMALL;
    int64_t r = cjo_lz4_decompress_raw(in, n, out, (size_t)size);
    return r < 0 ? CJO_E_CORRUPT : r;
}
