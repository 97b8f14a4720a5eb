// frame.hip — framed container formats on top of the batch engine.
//
// Snappy framing format: what the reference reaches at /root/reference/src/snappy.rs:24 (decompress), :38
// (compress), :82 (compress_into), :88 (decompress_into) through libcramjam::snappy::{compress,decompress} ->
// snap 1.1.1 read::FrameEncoder / read::FrameDecoder.  A framed stream is a sequence of independent <= 64 KiB
// pieces, which is exactly the batch the block engine wants: the host walks the 4-byte chunk headers (a serial
// but trivial scan), the GPU decodes/encodes every piece in one batch, checksums every piece (crc32c_pieces) and
// assembles the stream (copy_segments).  No codec or checksum arithmetic runs on the host.
#include "cj_engine.hpp"

namespace cj {
void launch_crc32c_pieces(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t* out, uint32_t n, hipStream_t s);
void launch_copy_segments(const uint64_t* src, uint8_t* dst_base, const uint64_t* dst_off, const uint64_t* len,
                          const uint64_t* hdr, uint32_t n, hipStream_t s);
}

namespace {

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
    cj::launch_copy_segments(d_meta + r_s, d_out, d_meta + r_s + ns, d_meta + r_s + 2 * ns, nullptr, (uint32_t)ns, s);
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
    cj::launch_copy_segments(d_meta + 5 * np, d_frame, d_meta + 6 * np, d_meta + 7 * np, d_meta + 8 * np, (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, d_frame, fpos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)fpos;
}

}  // extern "C"
