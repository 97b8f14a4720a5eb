/*
 * snappy_frame_oracle.c — CPU ORACLE (test infrastructure) for the Snappy FRAMING format behind
 * cramjam.snappy.compress / decompress / compress_into / decompress_into
 * (/root/reference/src/snappy.rs:24,38,82,88 -> libcramjam::snappy::{compress,decompress} ->
 * snap 1.1.1 read::FrameEncoder / read::FrameDecoder; the crates are not under /root/reference,
 * Cargo.lock:407-413,744-747).
 *
 * Restated from the published framing_format.txt and snap 1.1.1's frame.rs / read.rs behaviour:
 *   stream  = chunk*            chunk = type(1) len(3, LE) body(len)
 *   0xff    stream identifier, body "sNaPpY" (may repeat inside a stream)
 *   0x00    compressed data:   masked-CRC32C(uncompressed) (4, LE) + snappy raw block
 *   0x01    uncompressed data: masked-CRC32C (4, LE) + bytes
 *   0x02-0x7f reserved unskippable (error), 0x80-0xfd reserved skippable, 0xfe padding (skipped)
 *   masked(c) = ((c >> 15) | (c << 17)) + 0xa282ead8
 * Encoder: nothing at all for empty input (snap writes the identifier with the first chunk); input is cut
 * into 65536-byte pieces; a piece is stored uncompressed when compressed_len >= len - len/8.
 * Decoder limits: chunk length <= max_compress_len(65536) = 76490, decoded piece <= 65536.
 *
 * Pinning: the reference's fixture tests/data/integration/plaintext.txt.snappy (a third-party framed stream;
 * copied as data under tests/golden/) must decode to plaintext.txt, and its stored checksum must equal this
 * file's masked CRC32C; CRC32C itself is pinned by the standard check value crc32c("123456789") = 0xE3069283.
 */
#include "cj_oracle.h"
#include <stdlib.h>
#include <string.h>

#define FRAME_BLOCK 65536u
#define FRAME_MAX_CHUNK 76490u   /* max_compress_len(65536) */

static uint32_t crc_tab[256];
static int crc_ready;

static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
        crc_tab[i] = c;
    }
    crc_ready = 1;
}

uint32_t cjo_crc32c(const uint8_t* p, size_t n) {
    if (!crc_ready) crc_init();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = crc_tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

uint32_t cjo_crc32c_masked(const uint8_t* p, size_t n) {
    uint32_t c = cjo_crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

static const uint8_t kIdent[10] = { 0xff, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y' };

size_t cjo_snappy_frame_max_compress_len(size_t n) {
    if (n == 0) return 0;
    size_t chunks = (n + FRAME_BLOCK - 1) / FRAME_BLOCK;
    return 10 + chunks * 8 + n;      /* a piece never grows: it is stored raw unless it shrinks by 1/8 */
}

/* block_size: 65536 for the reference's behaviour; tests pass smaller values to mint ragged foreign streams */
int64_t cjo_snappy_frame_compress_bs(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t block_size) {
    if (block_size == 0 || block_size > FRAME_BLOCK) return CJO_E_SNAPPY_TOO_BIG;
    if (n == 0) return 0;
    uint8_t* tmp = (uint8_t*)malloc(cjo_snappy_max_compress_len(FRAME_BLOCK));
    if (!tmp) return CJO_E_FRAME_WRITE;
    size_t op = 0;
    int64_t rc = 0;
    if (cap < 10) { rc = CJO_E_FRAME_WRITE; goto done; }
    memcpy(out, kIdent, 10); op = 10;
    for (size_t pos = 0; pos < n; pos += block_size) {
        size_t len = n - pos < block_size ? n - pos : block_size;
        int64_t cl = cjo_snappy_compress(in + pos, len, tmp, cjo_snappy_max_compress_len(FRAME_BLOCK));
        if (cl < 0) { rc = cl; goto done; }
        int stored = (size_t)cl >= len - len / 8;
        size_t body = stored ? len : (size_t)cl;
        if (cap - op < 8 + body) { rc = CJO_E_FRAME_WRITE; goto done; }
        uint32_t clen = (uint32_t)(4 + body), crc = cjo_crc32c_masked(in + pos, len);
        out[op] = stored ? 0x01 : 0x00;
        out[op + 1] = (uint8_t)clen; out[op + 2] = (uint8_t)(clen >> 8); out[op + 3] = (uint8_t)(clen >> 16);
        out[op + 4] = (uint8_t)crc; out[op + 5] = (uint8_t)(crc >> 8); out[op + 6] = (uint8_t)(crc >> 16); out[op + 7] = (uint8_t)(crc >> 24);
        memcpy(out + op + 8, stored ? in + pos : tmp, body);
        op += 8 + body;
    }
    rc = (int64_t)op;
done:
    free(tmp);
    return rc;
}

int64_t cjo_snappy_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    return cjo_snappy_frame_compress_bs(in, n, out, cap, FRAME_BLOCK);
}

/* out == NULL: only validate the grammar / sizes that need no decoding and return the decoded length */
static int64_t frame_walk(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    size_t pos = 0, op = 0;
    int ident = 0;
    while (pos < n) {
        if (n - pos < 4) return CJO_E_FRAME_EOF;
        uint8_t ty = in[pos];
        if (!ident) {
            if (ty != 0xff) return CJO_E_SNAPPY_STREAM_HEADER;
            ident = 1;
        }
        size_t len = (size_t)in[pos + 1] | ((size_t)in[pos + 2] << 8) | ((size_t)in[pos + 3] << 16);
        if (len > FRAME_MAX_CHUNK) return CJO_E_SNAPPY_CHUNK_LEN;
        pos += 4;
        if (ty >= 0x02 && ty <= 0x7f) return CJO_E_SNAPPY_CHUNK_TYPE;
        if (ty >= 0x80 && ty <= 0xfe) {             /* skippable + padding */
            if (n - pos < len) return CJO_E_FRAME_EOF;
            pos += len;
            continue;
        }
        if (ty == 0xff) {
            if (len != 6) return CJO_E_SNAPPY_CHUNK_LEN;
            if (n - pos < 6) return CJO_E_FRAME_EOF;
            if (memcmp(in + pos, kIdent + 4, 6) != 0) return CJO_E_SNAPPY_STREAM_HEADER;
            pos += 6;
            continue;
        }
        if (len < 4) return CJO_E_SNAPPY_CHUNK_LEN;
        if (n - pos < 4) return CJO_E_FRAME_EOF;
        uint32_t want = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
        size_t sn = len - 4;
        pos += 4;
        size_t dn;
        const uint8_t* piece_src;
        uint8_t piece[FRAME_BLOCK];
        if (ty == 0x01 && sn > FRAME_BLOCK) return CJO_E_SNAPPY_CHUNK_LEN;   /* checked before the body is read */
        if (n - pos < sn) return CJO_E_FRAME_EOF;
        if (ty == 0x01) {
            dn = sn;
            piece_src = in + pos;
        } else {
            int64_t d = cjo_snappy_decompress_len(in + pos, sn);
            if (d < 0) return d;
            if ((uint64_t)d > FRAME_BLOCK) return CJO_E_SNAPPY_CHUNK_LEN;
            dn = (size_t)d;
            piece_src = piece;
            if (out) {
                int64_t r = cjo_snappy_decompress(in + pos, sn, piece, dn);   /* snap decodes into dst[..dn] */
                if (r < 0) return r;
            }
        }
        if (out) {                                   /* snap: checksum first, then the writer sees the piece */
            if (cjo_crc32c_masked(piece_src, dn) != want) return CJO_E_SNAPPY_CHECKSUM;
            if (cap - op < dn) return CJO_E_FRAME_WRITE;
            memcpy(out + op, piece_src, dn);
        }
        pos += sn;
        op += dn;
    }
    return (int64_t)op;
}

int64_t cjo_snappy_frame_decompress_len(const uint8_t* in, size_t n) { return frame_walk(in, n, NULL, 0); }

int64_t cjo_snappy_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    static uint8_t dummy;
    return frame_walk(in, n, out ? out : &dummy, cap);
}
