// frame.hip — framed container formats on top of the batch engine.
//
// Snappy framing format: what the reference reaches at /root/reference/src/snappy.rs:24 (decompress), :38
// (compress), :82 (compress_into), :88 (decompress_into) through libcramjam::snappy::{compress,decompress} ->
// snap 1.1.1 read::FrameEncoder / read::FrameDecoder.  A framed stream is a sequence of independent <= 64 KiB
// pieces, which is exactly the batch the block engine wants: the host walks the 4-byte chunk headers (a serial
// but trivial scan), the GPU decodes/encodes every piece in one batch, checksums every piece (crc32c_pieces) and
// assembles the stream (copy_segments).  No codec or checksum arithmetic runs on the host.
#include "cj_engine.hpp"
#include "xxh32_host.hpp"
#include "lz4_lane_walk.hpp"

#include <atomic>

namespace cj {
void launch_crc32c_pieces(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t* out, uint32_t n, hipStream_t s);
void launch_copy_segments(const uint64_t* src, uint8_t* dst_base, const uint64_t* dst_off, const uint64_t* len,
                          const uint64_t* hdr, uint32_t hdr_len, uint32_t n, hipStream_t s);
void launch_lz4_frame_chain(const uint8_t* in, const uint64_t* blk_off, const uint32_t* word, uint32_t nblk, uint8_t* out,
                            uint64_t out_cap, uint32_t block_max, int64_t* result, hipStream_t s);
}

namespace {
// (tuning builds only: every linked-block frame through the chain kernel; the shipped library reads no environment)
inline bool lz4f_chain_only() {
#ifdef CJ_DEBUG_KNOBS
    static const bool v = std::getenv("CJ_LZ4F_CHAIN_ONLY") != nullptr;
    return v;
#else
    return false;
#endif
}

constexpr size_t kPiece = 65536;          // snap MAX_BLOCK_SIZE
constexpr size_t kMaxChunk = 76490;       // snap MAX_COMPRESS_BLOCK_SIZE = max_compress_len(65536)
constexpr size_t kTmpStride = 76496;      // kMaxChunk rounded up to 16
const uint8_t kIdent[10] = { 0xff, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y' };

struct Piece {
    uint64_t src_off;     // payload offset in the framed stream
    uint64_t dst_off;     // offset of the decoded piece in the output
    uint32_t src_len, dst_len, crc;
    bool stored;
};

// Walk the chunk grammar (snap read::FrameDecoder::read).  Data chunks are appended to `pieces` (may be null);
// returns 0 or the first header-level error, in which case the pieces before it are still listed: snap would
// have decoded those first, so their errors take precedence.
int64_t snappy_frame_walk(const uint8_t* in, size_t n, std::vector<Piece>* pieces, uint64_t* total) {
    size_t pos = 0;
    uint64_t op = 0;
    bool ident = false;
    int64_t err = 0;
    while (pos < n) {
        if (n - pos < 4) { err = CJ_E_FRAME_EOF; break; }
        const uint8_t ty = in[pos];
        if (!ident) {
            if (ty != 0xff) { err = CJ_E_SNAPPY_STREAM_HEADER; break; }
            ident = true;
        }
        const size_t len = (size_t)in[pos + 1] | ((size_t)in[pos + 2] << 8) | ((size_t)in[pos + 3] << 16);
        if (len > kMaxChunk) { err = CJ_E_SNAPPY_CHUNK_LEN; break; }
        pos += 4;
        if (ty >= 0x02 && ty <= 0x7f) { err = CJ_E_SNAPPY_CHUNK_TYPE; break; }
        if (ty >= 0x80 && ty <= 0xfe) {                 // reserved skippable, padding
            if (n - pos < len) { err = CJ_E_FRAME_EOF; break; }
            pos += len;
            continue;
        }
        if (ty == 0xff) {
            if (len != 6) { err = CJ_E_SNAPPY_CHUNK_LEN; break; }
            if (n - pos < 6) { err = CJ_E_FRAME_EOF; break; }
            if (std::memcmp(in + pos, kIdent + 4, 6) != 0) { err = CJ_E_SNAPPY_STREAM_HEADER; break; }
            pos += 6;
            continue;
        }
        if (len < 4) { err = CJ_E_SNAPPY_CHUNK_LEN; break; }
        if (n - pos < 4) { err = CJ_E_FRAME_EOF; break; }
        Piece p;
        p.crc = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
        pos += 4;
        const size_t sn = len - 4;
        p.stored = ty == 0x01;
        if (p.stored && sn > kPiece) { err = CJ_E_SNAPPY_CHUNK_LEN; break; }
        if (n - pos < sn) { err = CJ_E_FRAME_EOF; break; }
        uint64_t dn = sn;
        if (!p.stored) {
            const int64_t d = cj_snappy_raw_decompress_len(in + pos, sn);   // empty block -> 0; the decoder then reports Empty
            if (d < 0) { err = d; break; }
            if ((uint64_t)d > kPiece) { err = CJ_E_SNAPPY_CHUNK_LEN; break; }
            dn = (uint64_t)d;
        }
        p.src_off = pos; p.src_len = (uint32_t)sn; p.dst_off = op; p.dst_len = (uint32_t)dn;
        if (pieces) pieces->push_back(p);
        pos += sn;
        op += dn;
    }
    if (total) *total = op;
    return err;
}

}  // namespace

extern "C" {

size_t cj_snappy_frame_max_compress_len(size_t n) {
    if (n == 0) return 0;
    return 10 + ((n + kPiece - 1) / kPiece) * 8 + n;
}

int64_t cj_snappy_frame_decompress_len(const uint8_t* in, size_t n) {
    if (n && !in) return CJ_E_BAD_ARG;
    uint64_t total = 0;
    const int64_t err = snappy_frame_walk(in, n, nullptr, &total);
    return err ? err : (int64_t)total;
}

int64_t cj_snappy_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if ((n && !in) || (cap && !out)) return CJ_E_BAD_ARG;
    cj_engine* e = cj::default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    std::vector<Piece> pieces;
    uint64_t total = 0;
    const int64_t gerr = snappy_frame_walk(in, n, &pieces, &total);
    const size_t np = pieces.size();
    if (np == 0) return gerr;
    if (np > 0xFFFFFFF0ull) return CJ_E_BAD_ARG;
    size_t nc = 0;
    for (const Piece& p : pieces) nc += p.stored ? 0 : 1;
    const size_t ns = np - nc;

    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    // device meta rows (u64): compressed pieces in_off|in_len|out_off|out_cap|result (nc each), stored pieces
    // src|dst_off|len (ns each), all pieces off|len (np each), then np u32 checksums
    const size_t r_c = 0, r_s = 5 * nc, r_p = r_s + 3 * ns, r_crc = r_p + 2 * np, rows = r_crc + (np + 1) / 2;
    if (!e->d_frame.reserve(n + 16) || !e->d_out.reserve(total + 16) || !e->d_meta.reserve(rows * 8)) return CJ_E_OOM;
    uint8_t* d_in = (uint8_t*)e->d_frame.p;
    uint8_t* d_out = (uint8_t*)e->d_out.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    std::vector<uint64_t>& m = e->h_meta;
    m.assign(rows, 0);
    size_t ci = 0, si = 0;
    for (size_t i = 0; i < np; i++) {
        const Piece& p = pieces[i];
        if (p.stored) {
            m[r_s + si] = (uint64_t)(uintptr_t)(d_in + p.src_off);
            m[r_s + ns + si] = p.dst_off;
            m[r_s + 2 * ns + si] = p.dst_len;
            si++;
        } else {
            m[r_c + ci] = p.src_off;
            m[r_c + nc + ci] = p.src_len;
            m[r_c + 2 * nc + ci] = p.dst_off;
            m[r_c + 3 * nc + ci] = p.dst_len;       // snap decodes into dst[..decompress_len]
            ci++;
        }
        m[r_p + i] = p.dst_off;
        m[r_p + np + i] = p.dst_len;
    }
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), r_crc * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    if (nc) {
        cj::BatchArgs a;
        cj::fill_args(a, 0, nc, d_in, d_meta + r_c, d_meta + r_c + nc, d_out, d_meta + r_c + 2 * nc, d_meta + r_c + 3 * nc,
                      (int64_t*)(d_meta + r_c + 4 * nc));
        const int rc = cj::launch(e, CJ_CODEC_SNAPPY_RAW, CJ_OP_DECOMPRESS, a, s);
        if (rc != 0) return rc;
    }
    cj::launch_copy_segments(d_meta + r_s, d_out, d_meta + r_s + ns, d_meta + r_s + 2 * ns, nullptr, 0, (uint32_t)ns, s);
    cj::launch_crc32c_pieces(d_out, d_meta + r_p, d_meta + r_p + np, (uint32_t*)(d_meta + r_crc), (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<int64_t> res(nc);
    std::vector<uint32_t> crc(np);
    if (nc) HIP_TRY(hipMemcpyAsync(res.data(), d_meta + r_c + 4 * nc, nc * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(crc.data(), d_meta + r_crc, np * 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    // first failure in stream order: block error, checksum, then the writer running out of room
    ci = 0;
    uint64_t written = 0;
    for (size_t i = 0; i < np; i++) {
        const Piece& p = pieces[i];
        if (!p.stored) {
            const int64_t r = res[ci++];
            if (r < 0) return r;
        }
        if (crc[i] != p.crc) return CJ_E_SNAPPY_CHECKSUM;
        if (out != nullptr && written + p.dst_len > cap) return CJ_E_FRAME_WRITE;
        written += p.dst_len;
    }
    if (gerr) return gerr;
    if (out != nullptr && total) HIP_TRY(hipMemcpy(out, d_out, total, hipMemcpyDeviceToHost), CJ_E_NO_DEVICE);
    return (int64_t)total;
}

int64_t cj_snappy_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if ((n && !in) || (cap && !out)) return CJ_E_BAD_ARG;
    cj_engine* e = cj::default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    if (n == 0) return 0;                  // snap emits the stream identifier together with the first chunk only
    const size_t np = (n + kPiece - 1) / kPiece;
    if (np > 0xFFFFFFF0ull) return CJ_E_BAD_ARG;
    if (n > 8192 && n <= cj::large_split_max()) return cj::large_snappy_frame(in, n, out, cap);     // sub-pieces, one wavefront each (large.hip)

    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    // rows: in_off|in_len|tmp_off|tmp_cap|result | src|dst_off|len|hdr (np each) | np u32 checksums
    const size_t r_crc = 9 * np, rows = r_crc + (np + 1) / 2;
    if (!e->d_in.reserve(n + 16) || !e->d_out.reserve(np * kTmpStride + 16) || !e->d_meta.reserve(rows * 8)) return CJ_E_OOM;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint8_t* d_tmp = (uint8_t*)e->d_out.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    std::vector<uint64_t>& m = e->h_meta;
    m.assign(rows, 0);
    for (size_t i = 0; i < np; i++) {
        m[i] = i * kPiece;
        m[np + i] = std::min(kPiece, n - i * kPiece);
        m[2 * np + i] = i * kTmpStride;
        m[3 * np + i] = kTmpStride;
    }
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), 4 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    cj::BatchArgs a;
    cj::fill_args(a, 0, np, d_in, d_meta, d_meta + np, d_tmp, d_meta + 2 * np, d_meta + 3 * np, (int64_t*)(d_meta + 4 * np));
    const int rc = cj::launch(e, CJ_CODEC_SNAPPY_RAW, CJ_OP_COMPRESS, a, s);
    if (rc != 0) return rc;
    cj::launch_crc32c_pieces(d_in, d_meta, d_meta + np, (uint32_t*)(d_meta + r_crc), (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<int64_t> res(np);
    std::vector<uint32_t> crc(np);
    HIP_TRY(hipMemcpyAsync(res.data(), d_meta + 4 * np, np * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(crc.data(), d_meta + r_crc, np * 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    // chunk layout (snap frame.rs compress_frame): stored when compressed_len >= len - len/8
    uint64_t fpos = 10;
    for (size_t i = 0; i < np; i++) {
        if (res[i] < 0) return res[i];
        const uint64_t len = m[np + i], cl = (uint64_t)res[i];
        const bool stored = cl >= len - len / 8;
        const uint64_t body = stored ? len : cl;
        m[5 * np + i] = (uint64_t)(uintptr_t)(stored ? d_in + i * kPiece : d_tmp + i * kTmpStride);
        m[6 * np + i] = fpos + 8;
        m[7 * np + i] = body;
        m[8 * np + i] = (stored ? 1ull : 0ull) | ((body + 4) << 8) | ((uint64_t)crc[i] << 32);
        fpos += 8 + body;
    }
    if (fpos > cap) return CJ_E_FRAME_WRITE;
    if (!e->d_frame.reserve(fpos + 16)) return CJ_E_OOM;
    uint8_t* d_frame = (uint8_t*)e->d_frame.p;
    HIP_TRY(hipMemcpyAsync(d_frame, kIdent, 10, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_meta + 5 * np, m.data() + 5 * np, 4 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    cj::launch_copy_segments(d_meta + 5 * np, d_frame, d_meta + 6 * np, d_meta + 7 * np, d_meta + 8 * np, 8, (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, d_frame, fpos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)fpos;
}

// =====================================================================================================================
// LZ4 FRAME format: what the reference reaches at /root/reference/src/lz4.rs:28 (decompress), :43 (compress), :56
// (compress_into), :63 (decompress_into) through libcramjam::lz4::{compress,decompress} -> lz4 crate Encoder/Decoder ->
// LZ4F_* (liblz4 1.10.0).  Blocks are de/compressed on the GPU: frames with independent blocks (and single-block
// frames) as ONE batch through the block engine, frames with linked blocks by the chain kernel (lz4_decode.hip).
// The frame's XXH32 checksums are the one piece of arithmetic done on the host: XXH32 is a serial recurrence of four
// 32-bit multiply-rotate accumulators per frame — one CPU core sustains ~6 GB/s on it, one GPU wavefront ~1 GB/s
// (quarter-rate 32-bit multiplies on a dependent chain) — so it runs on a host thread concurrently with the device batch.
// =====================================================================================================================
namespace {

using cj::xxh32;
inline uint32_t xrd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline void xwr32(uint8_t* p, uint32_t v) { std::memcpy(p, &v, 4); }

constexpr size_t kLz4fBlock = 65536;                      // the reference encoder's block size (lz4 crate BlockSize::Default)
constexpr size_t kLz4fTmpStride = 65824;                  // LZ4_compressBound(65536) = 65809, rounded up to 16

struct Lz4Block { uint64_t src_off; uint32_t word; };     // word = size | bit 31 (stored)
struct Lz4Frame {
    bool indep = true, bsum = false, csize = false, csum = false;
    uint32_t block_max = 0;
    uint64_t content_size = 0;
    uint32_t content_sum = 0;
    std::vector<Lz4Block> blocks;
    bool skippable = false;
    bool complete = false;        // the EndMark was reached
    int64_t late_err = 0;         // error met while walking the blocks (the blocks listed before it are intact)
};

// header + block walk (LZ4F_decompress's own checks; truncation = the lz4 crate's "Finish runned before read end ...")
int64_t lz4_frame_walk(const uint8_t* in, size_t n, Lz4Frame& f, bool verify_block_sums) {
    if (n >= 8 && (xrd32(in) & 0xFFFFFFF0u) == 0x184D2A50u) {
        f.skippable = true;
        return n - 8 < xrd32(in + 4) ? (int64_t)CJ_E_LZ4F_INCOMPLETE : 0;
    }
    if (n < 7) return CJ_E_LZ4F_INCOMPLETE;
    if (xrd32(in) != 0x184D2204u) return CJ_E_LZ4F_FRAME_TYPE;
    const uint8_t flg = in[4], bd = in[5];
    if ((flg >> 6) != 1 || (flg & 0x02)) return CJ_E_LZ4F_HEADER;
    if ((bd & 0x8F) != 0) return CJ_E_LZ4F_HEADER;
    f.indep = (flg >> 5) & 1; f.bsum = (flg >> 4) & 1; f.csize = (flg >> 3) & 1; f.csum = (flg >> 2) & 1;
    const bool dictid = flg & 1;
    const uint32_t code = (bd >> 4) & 7;
    if (code < 4) return CJ_E_LZ4F_BLOCK_SIZE;
    f.block_max = 1u << (8 + 2 * code);
    const size_t hl = 6 + (f.csize ? 8 : 0) + (dictid ? 4 : 0);
    if (n < hl + 1) return CJ_E_LZ4F_INCOMPLETE;
    if (f.csize) f.content_size = (uint64_t)xrd32(in + 6) | ((uint64_t)xrd32(in + 10) << 32);
    if (in[hl] != (uint8_t)(xxh32(in + 4, hl - 4, 0) >> 8)) return CJ_E_LZ4F_HEADER;
    // block walk: an error here is a LATE error — the streaming decoder has already written the blocks before it
    size_t pos = hl + 1;
    for (;;) {
        if (n - pos < 4) { f.late_err = CJ_E_LZ4F_INCOMPLETE; return 0; }
        const uint32_t w = xrd32(in + pos);
        pos += 4;
        if (w == 0) break;
        const size_t sz = w & 0x7FFFFFFFu;
        if (sz > f.block_max) { f.late_err = CJ_E_LZ4F_BLOCK_SIZE; return 0; }
        if (n - pos < sz + (f.bsum ? 4u : 0u)) { f.late_err = CJ_E_LZ4F_INCOMPLETE; return 0; }
        if (f.bsum && verify_block_sums && xrd32(in + pos + sz) != xxh32(in + pos, sz, 0)) { f.late_err = CJ_E_LZ4F_BLOCK_CHECKSUM; return 0; }
        f.blocks.push_back({pos, w});
        pos += sz + (f.bsum ? 4 : 0);
    }
    f.complete = true;
    if (f.csum) {
        if (n - pos < 4) { f.late_err = CJ_E_LZ4F_INCOMPLETE; return 0; }
        f.content_sum = xrd32(in + pos);
    }
    return 0;
}

}  // namespace

namespace {

std::atomic<unsigned long long> g_linked_lds_frames{0};

// Linked 64 KiB blocks through parse + lz4_decode_lds2_kernel<LZ4, linked>: ~100x the chain kernel on short-sequence data.
// Needs every non-last block to decode to exactly 64 KiB (block k then starts at k * 64 KiB and its history is the whole
// previous block) and every block to fit the LDS decoder (<= 16 384 sequences, <= 65 504 input bytes); the parse kernel
// establishes both.  Returns 0 when the frame was decoded this way, 1 when the caller must fall back to the chain
// kernel, < 0 on a device error.
int lz4_frame_linked_lds(cj_engine* e, const Lz4Frame& f, const uint8_t* d_in, std::vector<int64_t>& res, uint8_t** d_final) {
    const size_t nb = f.blocks.size();
    const uint64_t B = 65536;
    hipStream_t s = e->stream;
    std::lock_guard<std::mutex> lock(e->scratch_mu);
    // rows: in_off | in_len | out_off | out_cap | result | hist(u32) | frames(uint2) | first(uint2 per block) | counter | done(u32 per block)
    const size_t hw = (nb * 4 + 7) / 8, r_first = 5 * nb + hw + 1, r_cnt = r_first + nb, r_done = r_cnt + 1, rows = r_done + hw;
    const size_t list_bytes = 16;
    if (e->scratch_free) HIP_TRY(hipEventSynchronize(e->scratch_free), CJ_E_NO_DEVICE);
    if (!e->d_out.reserve(nb * B + 16) || !e->d_meta.reserve(rows * 8) || !e->d_sync.reserve(cj::lz4_lds_scratch_sync_bytes(nb)) ||
        !e->d_pmeta.reserve(cj::lz4_lds_scratch_meta_bytes(nb)) || !e->d_lanelist.reserve(list_bytes) ||
        !e->d_tab.reserve(cj::lz4_lds2_tab_bytes(1))) return CJ_E_OOM;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    std::vector<uint64_t>& m = e->h_meta;
    m.assign(rows, 0);
    uint32_t* hist = reinterpret_cast<uint32_t*>(m.data() + 5 * nb);
    for (size_t i = 0; i < nb; i++) {
        const Lz4Block& b = f.blocks[i];
        m[i] = b.src_off;
        m[nb + i] = (uint64_t)(b.word & 0x7FFFFFFFu) | ((b.word & 0x80000000u) ? (1ull << 63) : 0ull);
        m[2 * nb + i] = i * B;
        m[3 * nb + i] = B;
        hist[i] = i ? 65536u : 0u;
    }
    uint32_t* fr = reinterpret_cast<uint32_t*>(m.data() + 5 * nb + hw);
    fr[0] = 0u; fr[1] = (uint32_t)nb;
    uint32_t* first = reinterpret_cast<uint32_t*>(m.data() + r_first);
    for (size_t i = 0; i < nb; i++) { first[2 * i] = (uint32_t)(i * cj::kSyncPitch); first[2 * i + 1] = 0u; }
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), rows * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemsetAsync(e->d_pmeta.p, 0, cj::lz4_lds_scratch_meta_bytes(nb), s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemsetAsync(e->d_lanelist.p, 0, 16, s), CJ_E_NO_DEVICE);
    cj::BatchArgs a;
    cj::fill_args(a, cj::kFlagLinkedFrame, nb, d_in, d_meta, d_meta + nb, (uint8_t*)e->d_out.p, d_meta + 2 * nb, d_meta + 3 * nb,
                  (int64_t*)(d_meta + 4 * nb));
    a.hist = reinterpret_cast<const uint32_t*>(d_meta + 5 * nb);
    cj::launch_lz4_parse(a, e->d_sync.p, e->d_pmeta.p, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<uint64_t> pmeta(nb);
    HIP_TRY(hipMemcpyAsync(res.data(), d_meta + 4 * nb, nb * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(pmeta.data(), e->d_pmeta.p, nb * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    for (size_t i = 0; i < nb; i++) {
        const uint32_t in_skip = (uint32_t)(pmeta[i] >> 32);          // ParseMeta {nseq, in_skip}
        if (res[i] < 0 || (in_skip & 0x80000000u)) return 1;          // malformed (the chain kernel names the block) or not LDS-capable
        if (i + 1 < nb && (uint64_t)res[i] != B) return 1;            // a short block in the middle: positions are not k * 64 KiB
        if (res[i] == 0 && !(in_skip & 0x20000000u)) return 1;
    }
#ifdef CJ_DEBUG_KNOBS
    static const bool one_wg = std::getenv("CJ_LZ4F_ONE_WORKGROUP") != nullptr;      // (tuning builds only: the older one-workgroup path, tests/perf/linked_frame_rate.py)
#else
    constexpr bool one_wg = false;
#endif
    if (one_wg) {
        // one workgroup walks the frame's blocks in order, the previous block in a second LDS window
        cj::launch_lz4_decode_lds2_linked(a, e->d_sync.p, e->d_pmeta.p, e->d_tab.p, (uint32_t*)e->d_lanelist.p + 2,
                                          d_meta + 5 * nb + hw, 1u, 1u, s);
    } else {
        // every block is a slab of the large-stream decoder (large.hip / DESIGN 5.6): two workgroups per CU take the blocks in
        // order, a match that reaches into earlier blocks is copied from their finished output once the predecessor's
        // completion flag is up, everything else is resolved meanwhile
        if (e->n_cu == 0) HIP_TRY(hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, e->device), CJ_E_NO_DEVICE);
        const uint32_t grid = (uint32_t)std::min<size_t>(2u * (size_t)e->n_cu, nb);
        const uint32_t cross_stride = 3u * cj::kSyncStride * cj::kSyncEvery, tab_stride = 4u * cj::kSyncStride * cj::kSyncEvery;
        const size_t tab_bytes = (size_t)grid * tab_stride * 16;
        if (!e->d_bigtab.reserve(tab_bytes + (size_t)grid * cross_stride * 16 + (size_t)grid * (tab_stride + 512u) * 4)) return CJ_E_OOM;
        cj::launch_lz4_decode_lds2_slabs(a, e->d_sync.p, e->d_pmeta.p, e->d_bigtab.p, (uint32_t*)(d_meta + r_cnt), d_meta + r_first, 0u,
                                         (uint32_t*)(d_meta + r_done), (uint8_t*)e->d_bigtab.p + tab_bytes, tab_stride, cross_stride,
                                         grid, s, CJ_CODEC_LZ4_BLOCK, true);
    }
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(res.data(), d_meta + 4 * nb, nb * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);     // the decoder's stall guard reports here
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    *d_final = (uint8_t*)e->d_out.p;
    g_linked_lds_frames.fetch_add(1);
    return 0;
}

}  // namespace

// debug aid (tests): number of linked-block frames decoded by the two-window LDS decoder in this process
extern "C" unsigned long long cj_debug_linked_lds_frames(void) { return g_linked_lds_frames.load(); }

size_t cj_lz4_frame_compress_bound(size_t n) {
    return 7 + ((n + kLz4fBlock - 1) / kLz4fBlock) * 4 + n + 4 + 4;
}

int64_t cj_lz4_frame_decompress_bound(const uint8_t* in, size_t n) {
    if (n && !in) return CJ_E_BAD_ARG;
    Lz4Frame f;
    const int64_t err = lz4_frame_walk(in, n, f, false);
    if (err) return err;
    if (f.skippable) return 0;
    if (f.late_err) return f.late_err;
    // what the blocks can produce at most: a stored block its own size, a compressed block of c bytes at most 255 c + 64 (an
    // LZ4 length byte stands for at most 255 bytes) and never more than the frame's block size.  The announced content size
    // is attacker-controlled: it bounds the result from above, it never raises it (a 19-byte frame announcing 2^46 bytes
    // used to make the caller allocate that; values >= 2^63 turned into bogus negative "error codes")
    uint64_t total = 0;
    for (const Lz4Block& b : f.blocks) {
        const uint64_t c = b.word & 0x7FFFFFFFu;
        total += (b.word & 0x80000000u) ? c : std::min<uint64_t>(f.block_max, 255ull * c + 64ull);
    }
    if (f.csize && f.content_size < total) return (int64_t)f.content_size;
    return (int64_t)total;
}

// the block sequence of a frame (u32 size word + data per 64 KiB of input, no header, no EndMark) written to out
int64_t cj_lz4_frame_compress_blocks(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if ((n && !in) || (cap && !out)) return CJ_E_BAD_ARG;
    cj_engine* e = cj::default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    const size_t np = (n + kLz4fBlock - 1) / kLz4fBlock;
    if (np > 0xFFFFFFF0ull) return CJ_E_BAD_ARG;
    if (np == 0) return 0;
    // up to 32 MiB: sixteen (above 16 MiB: four) wavefronts per 64 KiB block (sub-pieces joined into one LZ4 block, large.hip) —
    // one wavefront per block would make every call at least the 1.7 ms it needs for 64 KiB
    if (n > 8192 && n <= cj::large_split_max()) return cj::large_lz4_frame_blocks(in, n, out, cap);
    uint64_t fpos = 0;
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    // rows: in_off|in_len|tmp_off|tmp_cap|result | src|dst_off|len|hdr (np each)
    const size_t rows = 9 * np;
    if (!e->d_in.reserve(n + 16) || !e->d_out.reserve(np * kLz4fTmpStride + 16) || !e->d_meta.reserve(rows * 8)) return CJ_E_OOM;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint8_t* d_tmp = (uint8_t*)e->d_out.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    std::vector<uint64_t>& m = e->h_meta;
    m.assign(rows, 0);
    for (size_t i = 0; i < np; i++) {
        m[i] = i * kLz4fBlock;
        m[np + i] = std::min(kLz4fBlock, n - i * kLz4fBlock);
        m[2 * np + i] = i * kLz4fTmpStride;
        m[3 * np + i] = kLz4fTmpStride;
    }
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), 4 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    cj::BatchArgs a;
    cj::fill_args(a, 0, np, d_in, d_meta, d_meta + np, d_tmp, d_meta + 2 * np, d_meta + 3 * np, (int64_t*)(d_meta + 4 * np));
    const int rc = cj::launch(e, CJ_CODEC_LZ4_BLOCK, CJ_OP_COMPRESS, a, s);
    if (rc != 0) return rc;
    std::vector<int64_t> res(np);
    HIP_TRY(hipMemcpyAsync(res.data(), d_meta + 4 * np, np * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    // LZ4F_makeBlock: a block that does not shrink is stored (bit 31 of the size word)
    for (size_t i = 0; i < np; i++) {
        if (res[i] < 0) return res[i];
        const uint64_t len = m[np + i], cl = (uint64_t)res[i];
        const bool stored = cl >= len;
        const uint64_t body = stored ? len : cl;
        m[5 * np + i] = (uint64_t)(uintptr_t)(stored ? d_in + i * kLz4fBlock : d_tmp + i * kLz4fTmpStride);
        m[6 * np + i] = fpos + 4;
        m[7 * np + i] = body;
        m[8 * np + i] = body | (stored ? 0x80000000ull : 0ull);
        fpos += 4 + body;
    }
    if (fpos > cap) return CJ_E_FRAME_WRITE;
    if (!e->d_frame.reserve(fpos + 16)) return CJ_E_OOM;
    uint8_t* d_frame = (uint8_t*)e->d_frame.p;
    HIP_TRY(hipMemcpyAsync(d_meta + 5 * np, m.data() + 5 * np, 4 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    cj::launch_copy_segments(d_meta + 5 * np, d_frame, d_meta + 6 * np, d_meta + 7 * np, d_meta + 8 * np, 4, (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, d_frame, fpos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)fpos;
}

int64_t cj_lz4_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int level) {
    (void)level;                           // the GPU matcher has one mode; any level yields a valid frame (see header)
    if ((n && !in) || (cap && !out)) return CJ_E_BAD_ARG;
    if (!cj::default_engine()) return CJ_E_NO_DEVICE;
    if (cap < 15) return CJ_E_FRAME_WRITE;
    // frame header: version 01, independent blocks, content checksum, 64 KiB blocks (FLG 0x64, BD 0x40)
    uint8_t hdr[7] = { 0x04, 0x22, 0x4D, 0x18, 0x64, 0x40, 0 };
    hdr[6] = (uint8_t)(xxh32(hdr + 4, 2, 0) >> 8);
    std::memcpy(out, hdr, 7);
    uint32_t content_sum = 0;
    std::thread summer([&] { content_sum = xxh32(in, n, 0); });          // overlaps the device batch
    const int64_t r = cj_lz4_frame_compress_blocks(in, n, out + 7, cap - 15);
    summer.join();
    if (r < 0) return r;
    xwr32(out + 7 + r, 0u);                  // EndMark
    xwr32(out + 11 + r, content_sum);
    return r + 15;
}

int64_t cj_lz4_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if ((n && !in) || (cap && !out)) return CJ_E_BAD_ARG;
    cj_engine* e = cj::default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    Lz4Frame f;
    const int64_t werr = lz4_frame_walk(in, n, f, true);
    if (werr) return werr;
    if (f.skippable) return 0;
    const size_t nb = f.blocks.size();
    if (nb > 0xFFFFFFF0ull) return CJ_E_BAD_ARG;
    uint64_t total = 0;
    // Independent blocks above 64 KiB (the lz4 command line's 4 MiB default, 256 KiB / 1 MiB frames): as chunks of a batch each
    // would be one serial stream on one wavefront (0.035 GB/s per block); instead all blocks go through the large-stream
    // path together (large.hip: one parallel parse over the pieces of all blocks, one slab decoder launch over all slabs).
    bool big_done = false;
    const bool few_big = f.block_max > 65536;
    if (nb > 0 && few_big && out != nullptr && f.indep) {
        uint64_t bound = 0;
        for (const Lz4Block& b : f.blocks) bound += (b.word & 0x80000000u) ? (b.word & 0x7FFFFFFFu) : f.block_max;
        if (bound <= cap) {                                                    // (else: the generic path knows the writer's error order)
            // every compressed block decodes to its slot (where it would lie if every block before it decoded to block_max
            // bytes); the parse runs over all blocks at once; then the output is closed up where a block came out shorter
            // (only the last one does in frames liblz4 writes) and the stored blocks are copied in
            std::vector<const uint8_t*> ins; std::vector<size_t> lens, caps; std::vector<uint8_t*> outs;
            std::vector<uint64_t> slots(nb);
            uint64_t slot = 0;
            for (size_t i = 0; i < nb; i++) {
                const Lz4Block& b = f.blocks[i];
                const uint32_t sz = b.word & 0x7FFFFFFFu;
                slots[i] = slot;
                if (b.word & 0x80000000u) slot += sz;
                else { ins.push_back(in + b.src_off); lens.push_back(sz); outs.push_back(out + slot); caps.push_back(f.block_max); slot += f.block_max; }
            }
            std::vector<int64_t> rj(ins.size());
            const int rc = ins.empty() ? 0 : cj::large_decompress_many(e, CJ_CODEC_LZ4_BLOCK, ins.size(), ins.data(), lens.data(), nullptr, outs.data(), caps.data(), rj.data());
            if (rc == 0) {
                size_t k = 0;
                for (size_t i = 0; i < nb; i++) {
                    const Lz4Block& b = f.blocks[i];
                    uint64_t got;
                    if (b.word & 0x80000000u) { got = b.word & 0x7FFFFFFFu; std::memmove(out + total, in + b.src_off, got); }
                    else {
                        if (rj[k] < 0) return CJ_E_LZ4F_DECOMPRESS;
                        got = (uint64_t)rj[k++];
                        if (slots[i] != total) std::memmove(out + total, out + slots[i], got);
                    }
                    total += got;
                }
                big_done = true;
            } else if (rc != CJ_E_BAD_ARG) return rc;
        }
    }
    if (nb > 0 && !big_done) {
        const uint64_t B = f.block_max;
        std::lock_guard<std::mutex> lock(e->mu);
        HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
        hipStream_t s = e->stream;
        if (!e->d_frame.reserve(n + 16)) return CJ_E_OOM;
        uint8_t* d_in = (uint8_t*)e->d_frame.p;
        HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
        std::vector<int64_t> res(nb);
        uint8_t* d_final = nullptr;
        if (f.indep || nb == 1) {
            // one batch: compressed blocks decode into slots of the maximal block size, then everything is compacted
            size_t nc = 0;
            for (const Lz4Block& b : f.blocks) nc += (b.word & 0x80000000u) ? 0 : 1;
            // rows: in_off|in_len|out_off|out_cap|result (nc each) | src|dst_off|len (nb each)
            const size_t r_g = 5 * nc, rows = r_g + 3 * nb;
            if (!e->d_in.reserve(nc * B + 16) || !e->d_meta.reserve(rows * 8)) return CJ_E_OOM;
            uint8_t* d_tmp = (uint8_t*)e->d_in.p;
            uint64_t* d_meta = (uint64_t*)e->d_meta.p;
            std::vector<uint64_t>& m = e->h_meta;
            m.assign(rows, 0);
            size_t ci = 0;
            for (const Lz4Block& b : f.blocks) {
                if (b.word & 0x80000000u) continue;
                m[ci] = b.src_off; m[nc + ci] = b.word; m[2 * nc + ci] = ci * B; m[3 * nc + ci] = B;
                ci++;
            }
            std::vector<int64_t> cres(nc);
            if (nc) {
                HIP_TRY(hipMemcpyAsync(d_meta, m.data(), 4 * nc * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
                cj::BatchArgs a;
                cj::fill_args(a, 0, nc, d_in, d_meta, d_meta + nc, d_tmp, d_meta + 2 * nc, d_meta + 3 * nc, (int64_t*)(d_meta + 4 * nc));
                const int rc = cj::launch(e, CJ_CODEC_LZ4_BLOCK, CJ_OP_DECOMPRESS, a, s);
                if (rc != 0) return rc;
                HIP_TRY(hipMemcpyAsync(cres.data(), d_meta + 4 * nc, nc * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
            }
            HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
            ci = 0;
            uint64_t pos = 0;
            for (size_t i = 0; i < nb; i++) {
                const Lz4Block& b = f.blocks[i];
                const bool stored = (b.word & 0x80000000u) != 0;
                const int64_t r = stored ? (int64_t)(b.word & 0x7FFFFFFFu) : cres[ci];
                res[i] = r;
                m[r_g + i] = (uint64_t)(uintptr_t)(stored ? d_in + b.src_off : d_tmp + ci * B);
                m[r_g + nb + i] = pos;
                m[r_g + 2 * nb + i] = r > 0 ? (uint64_t)r : 0;
                if (r > 0) pos += (uint64_t)r;
                if (!stored) ci++;
            }
            if (!e->d_out.reserve(pos + 16)) return CJ_E_OOM;
            d_final = (uint8_t*)e->d_out.p;
            HIP_TRY(hipMemcpyAsync(d_meta + r_g, m.data() + r_g, 3 * nb * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
            cj::launch_copy_segments(d_meta + r_g, d_final, d_meta + r_g + nb, d_meta + r_g + 2 * nb, nullptr, 0, (uint32_t)nb, s);
            HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
        } else if (B == 65536 && !lz4f_chain_only() && lz4_frame_linked_lds(e, f, d_in, res, &d_final) == 0) {
            // linked 64 KiB blocks, decoded by the two-window LDS workgroup decoder (results in res, bytes at d_final)
        } else {
            // linked blocks: the chain kernel decodes straight into the contiguous output
            const uint64_t out_cap = nb * B;
            const size_t words = (nb * 4 + 7) / 8, rows = nb + words + nb;      // blk_off | word (u32) | result
            if (!e->d_out.reserve(out_cap + 16) || !e->d_meta.reserve(rows * 8)) return CJ_E_OOM;
            d_final = (uint8_t*)e->d_out.p;
            uint64_t* d_meta = (uint64_t*)e->d_meta.p;
            std::vector<uint64_t>& m = e->h_meta;
            m.assign(rows, 0);
            uint32_t* wv = reinterpret_cast<uint32_t*>(m.data() + nb);
            for (size_t i = 0; i < nb; i++) { m[i] = f.blocks[i].src_off; wv[i] = f.blocks[i].word; }
            HIP_TRY(hipMemcpyAsync(d_meta, m.data(), (nb + words) * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
            cj::launch_lz4_frame_chain(d_in, d_meta, (const uint32_t*)(d_meta + nb), (uint32_t)nb, d_final, out_cap, (uint32_t)B,
                                       (int64_t*)(d_meta + nb + words), s);
            HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
            HIP_TRY(hipMemcpyAsync(res.data(), d_meta + nb + words, nb * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
            HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
        }
        // first failure in stream order: malformed block, then the writer running out of room
        for (size_t i = 0; i < nb; i++) {
            if (res[i] < 0) return CJ_E_LZ4F_DECOMPRESS;
            if (f.csize && total + (uint64_t)res[i] > f.content_size) return CJ_E_LZ4F_CONTENT_SIZE;      // more than the header announced (LZ4F: ERROR_frameSize_wrong)
            if (out != nullptr && total + (uint64_t)res[i] > cap) return CJ_E_FRAME_WRITE;
            total += (uint64_t)res[i];
        }
        if (out != nullptr && total) {
            HIP_TRY(hipMemcpyAsync(out, d_final, total, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
            HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
        }
    }
    if (f.late_err && !f.complete) return f.late_err;                      // truncated / bad block word or checksum: after the blocks before it
    if (f.csize && f.content_size != total) return CJ_E_LZ4F_CONTENT_SIZE;
    if (f.late_err) return f.late_err;                                       // content checksum word missing
    if (f.csum && out != nullptr && f.content_sum != xxh32(out, total, 0)) return CJ_E_LZ4F_CONTENT_CHECKSUM;
    return (int64_t)total;
}

}  // extern "C"
