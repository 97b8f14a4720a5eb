/*
 * lz4_frame_oracle.c — CPU ORACLE (test infrastructure) for the LZ4 FRAME format behind
 * cramjam.lz4.compress / decompress / compress_into / decompress_into
 * (/root/reference/src/lz4.rs:28-66 -> libcramjam::lz4::{compress,decompress} -> lz4 crate 1.28.1
 * Encoder/Decoder -> LZ4F_* of liblz4 1.10.0; the crates are not under /root/reference, Cargo.lock:407-473).
 *
 * Restated from the published lz4_Frame_format.md (v1.6.x) and the observable behaviour of LZ4F_decompress / the lz4
 * crate's Decoder:
 *   frame   = magic 0x184D2204 | FLG BD [content size u64] [dict id u32] HC | block* | EndMark 0 | [content xxh32]
 *   FLG     = version 01 | B.Indep | B.Checksum | C.Size | C.Checksum | reserved 0 | DictID
 *   BD      = block max size code 4..7 (64 KiB, 256 KiB, 1 MiB, 4 MiB), other bits 0
 *   HC      = (xxh32(FLG .. before HC) >> 8) & 0xff
 *   block   = u32 size (bit 31: stored uncompressed) | data | [xxh32(data)]        size <= block max size
 *   linked blocks (B.Indep = 0): matches may reach into the previous 64 KiB of output
 *   skippable frames (magic 0x184D2A5x) are skipped; the lz4 crate's Decoder stops after the FIRST frame it completes
 *   (trailing bytes are ignored) and fails with "Finish runned before read end of compressed stream" on truncation.
 * Encoder defaults of the reference (lz4 crate EncoderBuilder via libcramjam): 64 KiB blocks, content checksum on,
 * no content size, blocks LINKED, level 4 (LZ4HC).  Compressed BYTES are not pinned by the reference (round trips
 * only), so this oracle's encoder emits the same container with the greedy LZ4_compress_default matcher of
 * lz4_block_oracle.c and — selectable — independent or linked blocks.
 *
 * Pinning: the reference's fixture tests/data/integration/plaintext.txt.lz4 (copied as data under tests/golden/);
 * golden frames minted with system liblz4's LZ4F_compressFrame (linked / independent, block + content checksums,
 * content size, all four block sizes) in tests/golden/make_golden.py; XXH32 by its published test values.
 */
#include "cj_oracle.h"
#include <stdlib.h>
#include <string.h>

#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4 668265263u
#define P5 374761393u

static uint32_t rotl(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

uint32_t cjo_xxh32(const uint8_t* p, size_t n, uint32_t seed) {
    const uint8_t* end = p + n;
    uint32_t h;
    if (n >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* lim = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1;
            v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1;
            v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= lim);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)n;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

static size_t block_max(int code) { return (size_t)1 << (8 + 2 * code); }      /* 4 -> 64 KiB ... 7 -> 4 MiB */

size_t cjo_lz4_frame_compress_bound(size_t n, int bs_code) {
    if (bs_code < 4 || bs_code > 7) bs_code = 4;
    size_t b = block_max(bs_code), nb = (n + b - 1) / b;
    return 7 + 8 + 4 + nb * 8 + n + 4 + 4;       /* header (+size) + per-block size/checksum words + stored data + end + checksum */
}

/* flags: bit0 linked blocks, bit1 block checksums, bit2 content size, bit3 NO content checksum */
int64_t cjo_lz4_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int bs_code, int flags) {
    if (bs_code < 4 || bs_code > 7) bs_code = 4;
    const int linked = flags & 1, bsum = (flags >> 1) & 1, csize = (flags >> 2) & 1, csum = !((flags >> 3) & 1);
    const size_t B = block_max(bs_code);
    if (cap < cjo_lz4_frame_compress_bound(n, bs_code)) return CJO_E_FRAME_WRITE;
    size_t op = 0;
    wr32(out, 0x184D2204u); op = 4;
    size_t d0 = op;
    out[op++] = (uint8_t)(0x40 | (linked ? 0 : 0x20) | (bsum ? 0x10 : 0) | (csize ? 0x08 : 0) | (csum ? 0x04 : 0));
    out[op++] = (uint8_t)(bs_code << 4);
    if (csize) { wr32(out + op, (uint32_t)n); wr32(out + op + 4, (uint32_t)((uint64_t)n >> 32)); op += 8; }
    out[op] = (uint8_t)(cjo_xxh32(out + d0, op - d0, 0) >> 8); op++;
    /* linked blocks: compress [max(0, pos - 64 KiB), pos + len) as one buffer is NOT how liblz4 does it, but any valid
     * sequence stream whose matches stay inside the 64 KiB history is a valid linked block: we compress the block with
     * the history prepended and keep only the sequences of the block itself by encoding from a primed matcher state.
     * To stay simple the linked variant emits blocks whose matches reference the previous block through a
     * dictionary-primed greedy pass (cjo_lz4_compress_with_prefix). */
    uint8_t* tmp = (uint8_t*)malloc(cjo_lz4_compress_bound_raw(B) + 16);
    if (!tmp) return CJO_E_FRAME_WRITE;
    for (size_t pos = 0; pos < n; pos += B) {
        size_t len = n - pos < B ? n - pos : B;
        size_t hist = linked ? (pos < 65536 ? pos : 65536) : 0;
        int64_t c = cjo_lz4_compress_with_prefix(in + pos - hist, hist, len, tmp, len > 0 ? len - 1 : 0);
        uint32_t word;
        const uint8_t* src;
        size_t sz;
        if (c <= 0) { word = (uint32_t)len | 0x80000000u; src = in + pos; sz = len; }
        else { word = (uint32_t)c; src = tmp; sz = (size_t)c; }
        wr32(out + op, word); op += 4;
        memcpy(out + op, src, sz); op += sz;
        if (bsum) { wr32(out + op, cjo_xxh32(src, sz, 0)); op += 4; }
    }
    free(tmp);
    wr32(out + op, 0); op += 4;
    if (csum) { wr32(out + op, cjo_xxh32(in, n, 0)); op += 4; }
    return (int64_t)op;
}

/* header walk shared by bound + decode.  Returns header length or an error. */
typedef struct { int indep, bsum, csize, csum, dictid, bs_code; uint64_t content_size; size_t hdr_len; } FrameInfo;

static int64_t parse_header(const uint8_t* in, size_t n, FrameInfo* fi) {
    if (n < 7) return CJO_E_LZ4F_INCOMPLETE;
    if (rd32(in) != 0x184D2204u) return CJO_E_LZ4F_FRAME_TYPE;
    const uint8_t flg = in[4], bd = in[5];
    if ((flg >> 6) != 1 || (flg & 0x02)) return CJO_E_LZ4F_HEADER;
    if ((bd & 0x8F) != 0) return CJO_E_LZ4F_HEADER;
    fi->indep = (flg >> 5) & 1; fi->bsum = (flg >> 4) & 1; fi->csize = (flg >> 3) & 1; fi->csum = (flg >> 2) & 1; fi->dictid = flg & 1;
    fi->bs_code = (bd >> 4) & 7;
    if (fi->bs_code < 4) return CJO_E_LZ4F_BLOCK_SIZE;
    size_t hl = 6 + (fi->csize ? 8 : 0) + (fi->dictid ? 4 : 0);
    if (n < hl + 1) return CJO_E_LZ4F_INCOMPLETE;
    fi->content_size = 0;
    if (fi->csize) fi->content_size = (uint64_t)rd32(in + 6) | ((uint64_t)rd32(in + 10) << 32);
    if (in[hl] != (uint8_t)(cjo_xxh32(in + 4, hl - 4, 0) >> 8)) return CJO_E_LZ4F_HEADER;
    fi->hdr_len = hl + 1;
    return (int64_t)fi->hdr_len;
}

/* upper bound of the decoded size from the headers alone: what the blocks can produce at most (a stored block its size, a
 * compressed block of c bytes min(block size, 255 c + 64): an LZ4 length byte stands for at most 255 bytes), capped by the
 * announced content size when there is one — the announcement never raises the bound (it is attacker-controlled) */
int64_t cjo_lz4_frame_decompress_bound(const uint8_t* in, size_t n) {
    if (n >= 8 && (rd32(in) & 0xFFFFFFF0u) == 0x184D2A50u) return 0;       /* skippable frame first: decoder yields nothing */
    FrameInfo fi;
    int64_t h = parse_header(in, n, &fi);
    if (h < 0) return h;
    const size_t B = block_max(fi.bs_code);
    size_t pos = (size_t)h;
    uint64_t total = 0;
    for (;;) {
        if (n - pos < 4) return CJO_E_LZ4F_INCOMPLETE;
        uint32_t w = rd32(in + pos); pos += 4;
        if (w == 0) break;
        size_t sz = w & 0x7FFFFFFFu;
        if (sz > B) return CJO_E_LZ4F_BLOCK_SIZE;
        if (n - pos < sz + (fi.bsum ? 4u : 0u)) return CJO_E_LZ4F_INCOMPLETE;
        uint64_t most = 255ull * sz + 64ull;
        total += (w & 0x80000000u) ? sz : (most < B ? most : B);
        pos += sz + (fi.bsum ? 4 : 0);
    }
    if (fi.csize && fi.content_size < total) return (int64_t)fi.content_size;
    return (int64_t)total;
}

int64_t cjo_lz4_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if (n >= 8 && (rd32(in) & 0xFFFFFFF0u) == 0x184D2A50u) {             /* skippable frame: skipped, then "frame complete" */
        size_t sz = rd32(in + 4);
        return n - 8 < sz ? CJO_E_LZ4F_INCOMPLETE : 0;
    }
    FrameInfo fi;
    int64_t h = parse_header(in, n, &fi);
    if (h < 0) return h;
    const size_t B = block_max(fi.bs_code);
    size_t pos = (size_t)h, op = 0;
    uint8_t* blk = (uint8_t*)malloc(B + 65536 + 16);
    if (!blk) return CJO_E_FRAME_WRITE;
    int64_t rc = 0;
    for (;;) {
        if (n - pos < 4) { rc = CJO_E_LZ4F_INCOMPLETE; break; }
        uint32_t w = rd32(in + pos); pos += 4;
        if (w == 0) break;
        size_t sz = w & 0x7FFFFFFFu;
        if (sz > B) { rc = CJO_E_LZ4F_BLOCK_SIZE; break; }
        if (n - pos < sz + (fi.bsum ? 4u : 0u)) { rc = CJO_E_LZ4F_INCOMPLETE; break; }
        if (fi.bsum && rd32(in + pos + sz) != cjo_xxh32(in + pos, sz, 0)) { rc = CJO_E_LZ4F_BLOCK_CHECKSUM; break; }
        size_t dn;
        if (w & 0x80000000u) {
            dn = sz;
            if (cap - op < dn) { rc = CJO_E_FRAME_WRITE; break; }
            memcpy(out + op, in + pos, dn);
        } else {
            /* decode with the previous output as prefix when blocks are linked; capacity = max block size */
            size_t hist = fi.indep ? 0 : (op < 65536 ? op : 65536);
            memcpy(blk, out + op - hist, hist);
            int64_t r = cjo_lz4_decompress_with_prefix(in + pos, sz, blk, hist, B);
            if (r < 0) { rc = CJO_E_LZ4F_DECOMPRESS; break; }
            dn = (size_t)r;
            if (cap - op < dn) { rc = CJO_E_FRAME_WRITE; break; }
            memcpy(out + op, blk + hist, dn);
        }
        op += dn;
        pos += sz + (fi.bsum ? 4 : 0);
    }
    free(blk);
    if (rc < 0) return rc;
    if (fi.csize && fi.content_size != op) return CJO_E_LZ4F_CONTENT_SIZE;
    if (fi.csum) {
        if (n - pos < 4) return CJO_E_LZ4F_INCOMPLETE;
        if (rd32(in + pos) != cjo_xxh32(out, op, 0)) return CJO_E_LZ4F_CONTENT_CHECKSUM;
    }
    return (int64_t)op;
}
