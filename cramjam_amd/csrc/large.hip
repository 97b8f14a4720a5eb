// large.hip — ONE large buffer through the single-buffer C-ABI (cj_lz4_block_compress, cj_snappy_raw_compress: what the
// reference reaches at /root/reference/src/lz4.rs:168 compress_block and /root/reference/src/snappy.rs:57 compress_raw).
// A batch gives the GPU one independent stream per wavefront; a single 100 MB buffer handed to compress_block would be
// ONE stream on ONE wavefront (0.04 GB/s measured).  Both raw formats allow cutting the INPUT into pieces that are
// compressed independently and joined into one valid stream:
//   * Snappy raw = varint(total length) + elements.  Elements never refer to anything but earlier output, so the element
//     streams of consecutive 64 KiB pieces (each compressed with its own hash table — which is also how snap's own
//     encoder works through a large input, 64 KiB block by block) simply concatenate.
//   * LZ4 block = sequences (literal run + match), the last one literal-only.  The pieces' streams are stitched: the
//     trailing literal-only sequence of piece k is dropped and its bytes are prepended to the literal run of the first
//     sequence of the next piece that has a match (literals are copies of the INPUT, so the merged run is one contiguous
//     input range); only that one token and its length bytes are rewritten.  Every piece obeys LZ4's end-of-block rules
//     on its own, so the joined stream does too.
// The result is a stream any LZ4 / Snappy decoder accepts (round trip through the oracle decoders: tests/test_large_gpu.py);
// like all compressed output of this library it is not byte-identical to liblz4's / snap's.
#include "cj_engine.hpp"
#include "lz4_lane_walk.hpp"
#include "big_parse.hpp"

namespace cj {

void launch_copy_segments(const uint64_t* src, uint8_t* dst_base, const uint64_t* dst_off, const uint64_t* len,
                          const uint64_t* hdr, uint32_t hdr_len, uint32_t n, hipStream_t s);      // frame_kernels.hip
void launch_crc32c_pieces(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t* out, uint32_t n, hipStream_t s);

namespace {

constexpr size_t kPiece = 65536;
constexpr size_t kLz4Stride = 65824;          // LZ4_compressBound(65536) = 65809, rounded up to 16
constexpr size_t kSnStride = 76512;           // snap max_compress_len(65536) = 76490, rounded up to 16
// buffers of up to kSplitMax bytes: every 64 KiB piece is cut into sub-pieces, one wavefront each (lz4_encode.hip, kSplit) —
// one wavefront needs ~1.7 ms for 64 KiB (~6 µs per round of 320 positions, whatever else runs), and below ~500 pieces most of the
// GPU's wavefront slots are idle anyway.  Up to kFineMax bytes: 16 sub-pieces of 4 KiB (13 rounds each), above: quarters of
// 16 KiB.  One call, quarters -> 4 KiB sub-pieces: 16 KiB 0.44 -> 0.20 ms, 64 KiB 0.59 -> 0.30, 1 MiB 0.68 -> 0.37 (text 1.08 ->
// 0.37), 4 MiB 0.74 -> 0.44, 16 MiB 1.6 -> 1.56 (copies); ratio 1.62 -> 1.60 (benchmark data), 4.79 -> 4.78 (text).
constexpr size_t kSplitMax = 32u << 20, kFineMax = 16u << 20;
struct SubPieces {
    size_t sub;              // bytes per sub-piece
    size_t per;              // sub-pieces per 64 KiB piece
    size_t lz4_stride;       // LZ4_compressBound(sub) rounded up to 16
    size_t sn_stride;        // snap max_compress_len(sub) rounded up to 16
    uint32_t flags;          // kFlagSplitPieces | log2(per) << kFlagSplitShift
};
inline SubPieces sub_pieces(size_t n) {
    if (n <= kFineMax) return {4096, 16, 4128, 4816, kFlagSplitPieces | (4u << kFlagSplitShift)};      // 4096 + 16 + 16; 32 + 4096 + 682 = 4810
    return {16384, 4, 16480, 19152, kFlagSplitPieces | (2u << kFlagSplitShift)};                       // 16464; 19146
}

__host__ __device__ inline uint32_t lz4_len_ext(uint32_t len) { return len < 15u ? 0u : (len - 15u) / 255u + 1u; }

// first_lit[i] = literal length of the first sequence of piece i's stream
__global__ __launch_bounds__(256) void lz4_stitch_plan_kernel(const uint8_t* tmp, uint32_t stride, uint32_t* first_lit, uint32_t np) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= np) return;
    const uint8_t* p = tmp + (size_t)i * stride;
    uint32_t lit = p[0] >> 4, q = 1;
    if (lit == 15u) {
        uint32_t b;
        do { b = p[q++]; lit += b; } while (b == 255u);
    }
    first_lit[i] = lit;
}

// One wavefront per descriptor: token + length bytes of the merged literal run, the run itself (from the input), then the
// rest of the piece's stream.  d = {dst offset, run start (input offset), run length, body source (device address), body
// length, address of the piece's first token (its low nibble = the match length code; 0 = literal-only sequence)};
// run length 0 and body length 0 = nothing to write.
struct Stitch { uint64_t dst, run_start, run, body_src, body_len, tok_src; };

__global__ __launch_bounds__(kBlockThreads) void lz4_stitch_kernel(const Stitch* d, const uint8_t* in, uint8_t* out, uint32_t n) {
    const uint32_t k = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (k >= n) return;
    const Stitch s = d[k];
    const uint32_t run = (uint32_t)s.run;
    if (run == 0u && s.body_len == 0ull) return;
    const uint32_t nib = s.tok_src ? (uint32_t)(*reinterpret_cast<const uint8_t*>(s.tok_src)) & 15u : 0u;
    uint8_t* o = out + s.dst;
    const uint32_t lane = lane_id();
    if (lane == 0) o[0] = (uint8_t)(((run < 15u ? run : 15u) << 4) | nib);
    const uint32_t ext = lz4_len_ext(run);
    if (ext) {
        const uint32_t rem = run - 15u;
        for (uint32_t q = lane; q < ext; q += 64u) o[1u + q] = (uint8_t)(q + 1u < ext ? 255u : rem - 255u * (ext - 1u));
    }
    wave_copy(o + 1u + ext, in + s.run_start, run);
    if (s.body_len) wave_copy(o + 1u + ext + run, reinterpret_cast<const uint8_t*>(s.body_src), (uint32_t)s.body_len);
}

inline uint32_t varint_len(uint64_t v) { uint32_t k = 1; while (v >= 0x80u) { v >>= 7; k++; } return k; }

}  // namespace

// compress the np pieces of d_in as one batch into d_tmp (stride bytes apart); results -> res
static int compress_pieces(cj_engine* e, cj_codec codec, uint32_t flags, const uint8_t* in, size_t n, size_t piece, size_t np, size_t stride,
                           std::vector<int64_t>& res) {
    hipStream_t s = e->stream;
    if (!e->d_in.reserve(n + 16) || !e->d_out.reserve(np * stride + 16) || !e->d_meta.reserve(12 * np * 8)) return CJ_E_OOM;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    std::vector<uint64_t>& m = e->h_meta;
    m.assign(12 * np, 0);
    for (size_t i = 0; i < np; i++) {
        m[i] = i * piece;
        m[np + i] = std::min(piece, n - i * piece);
        m[2 * np + i] = i * stride;
        m[3 * np + i] = stride;
    }
    HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), 4 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    BatchArgs a;
    fill_args(a, flags, np, d_in, d_meta, d_meta + np, (uint8_t*)e->d_out.p, d_meta + 2 * np, d_meta + 3 * np, (int64_t*)(d_meta + 4 * np));
    const int rc = launch(e, codec, CJ_OP_COMPRESS, a, s);
    if (rc != 0) return rc;
    res.resize(np);
    HIP_TRY(hipMemcpyAsync(res.data(), d_meta + 4 * np, np * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    return 0;
}

// Plan the join of the pieces [i0, i1) (consecutive pieces of `piece` bytes of an n-byte input) into ONE LZ4 block written
// at out_pos: fills plan[i] and returns the block's length.  res[i] = stream length | tail literal length << 32 (the
// encoder's kFlagReportTail), first[i] = literal length of the piece's first sequence.
static uint64_t plan_stitch(std::vector<Stitch>& plan, size_t i0, size_t i1, uint64_t out_pos, size_t n, size_t piece, size_t stride,
                            const uint8_t* d_tmp, const std::vector<int64_t>& res, const std::vector<uint32_t>& first) {
    uint64_t pos = out_pos, pending = 0;           // pending = literal bytes before this piece that no sequence carries yet
    for (size_t i = i0; i < i1; i++) {
        const uint64_t len = std::min(piece, n - i * piece);
        const uint32_t r = (uint32_t)((uint64_t)res[i] & 0xFFFFFFFFull), tail = (uint32_t)((uint64_t)res[i] >> 32);
        const uint32_t l2 = first[i];
        const bool last = i + 1 == i1, has_match = l2 < len;
        Stitch d = {pos, i * piece - pending, 0, 0, 0, 0};
        if (has_match) {
            const uint32_t skip = 1u + lz4_len_ext(l2) + l2;                         // token, length bytes and literals of the first sequence
            const uint32_t end = last ? r : r - (1u + lz4_len_ext(tail) + tail);     // non-final pieces lose their literal-only last sequence
            const uint64_t run = pending + l2;
            d.run = run;
            d.tok_src = (uint64_t)(uintptr_t)(d_tmp + i * stride);
            d.body_src = d.tok_src + skip;
            d.body_len = end - skip;
            pos += 1u + lz4_len_ext((uint32_t)run) + run + d.body_len;
            pending = last ? 0 : tail;
        } else {
            pending += len;
            if (last) {                                                               // the block ends with one literal-only sequence
                d.run = pending;
                pos += 1u + lz4_len_ext((uint32_t)pending) + pending;
                pending = 0;
            }
        }
        plan[i] = d;
    }
    return pos - out_pos;
}


int64_t large_snappy_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    cj_engine* e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    if (n > 0xFFFFFFFFull) return CJ_E_SNAPPY_TOO_BIG;
    const bool split = n <= kSplitMax;
    const SubPieces sp = sub_pieces(n);
    const size_t piece = split ? sp.sub : kPiece, stride = split ? sp.sn_stride : kSnStride;
    const size_t np = (n + piece - 1) / piece;
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    hipStream_t s = e->stream;
    std::vector<int64_t> res;
    int rc = compress_pieces(e, CJ_CODEC_SNAPPY_RAW, split ? sp.flags : 0u, in, n, piece, np, stride, res);
    if (rc != 0) return rc;
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    std::vector<uint64_t>& m = e->h_meta;
    uint8_t hdr[10];
    uint32_t hl = 0;
    for (uint64_t v = n;; ) { if (v < 0x80u) { hdr[hl++] = (uint8_t)v; break; } hdr[hl++] = (uint8_t)(v | 0x80u); v >>= 7; }
    uint64_t pos = hl;
    uint8_t* d_tmp = (uint8_t*)e->d_out.p;
    for (size_t i = 0; i < np; i++) {
        if (res[i] < 0) return res[i];
        const uint32_t ph = varint_len(m[np + i]);             // the piece's own length header is dropped
        const uint64_t body = (uint64_t)res[i] - ph;
        m[5 * np + i] = (uint64_t)(uintptr_t)(d_tmp + i * stride + ph);
        m[6 * np + i] = pos;
        m[7 * np + i] = body;
        pos += body;
    }
    if (pos > cap) return CJ_E_SNAPPY_BUF_SMALL;
    if (!e->d_frame.reserve(pos + 16)) return CJ_E_OOM;
    uint8_t* d_frame = (uint8_t*)e->d_frame.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    HIP_TRY(hipMemcpyAsync(d_frame, hdr, hl, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_meta + 5 * np, m.data() + 5 * np, 3 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    launch_copy_segments(d_meta + 5 * np, d_frame, d_meta + 6 * np, d_meta + 7 * np, nullptr, 0, (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, d_frame, pos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)pos;
}

int64_t large_lz4_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, bool prefix) {
    cj_engine* e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    const bool split = n <= kSplitMax;
    const SubPieces sp = sub_pieces(n);
    const size_t piece = split ? sp.sub : kPiece, stride = split ? sp.lz4_stride : kLz4Stride;
    const size_t np = (n + piece - 1) / piece;
    const size_t pre = prefix ? 4 : 0;
    if (cap < pre) return CJ_E_COMPRESS_FAILED;
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    hipStream_t s = e->stream;
    std::vector<int64_t> res;
    int rc = compress_pieces(e, CJ_CODEC_LZ4_BLOCK, kFlagReportTail | (split ? sp.flags : 0u), in, n, piece, np, stride, res);
    if (rc != 0) return rc;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    uint8_t* d_tmp = (uint8_t*)e->d_out.p;
    uint32_t* d_first = (uint32_t*)(d_meta + 5 * np);
    hipLaunchKernelGGL(lz4_stitch_plan_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, d_tmp, (uint32_t)stride, d_first, (uint32_t)np);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<uint32_t> first(np);
    HIP_TRY(hipMemcpyAsync(first.data(), d_first, np * 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    std::vector<Stitch> plan(np);
    for (size_t i = 0; i < np; i++) if (res[i] < 0) return res[i];
    const uint64_t pos = plan_stitch(plan, 0, np, 0, n, piece, stride, d_tmp, res, first);
    if (pos + pre > cap) return CJ_E_COMPRESS_FAILED;
    if (!e->d_frame.reserve(pos + np * sizeof(Stitch) + 64)) return CJ_E_OOM;
    uint8_t* d_frame = (uint8_t*)e->d_frame.p;
    Stitch* d_plan = reinterpret_cast<Stitch*>(d_frame + ((pos + 15u) & ~(uint64_t)15u));
    HIP_TRY(hipMemcpyAsync(d_plan, plan.data(), np * sizeof(Stitch), hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    hipLaunchKernelGGL(lz4_stitch_kernel, dim3((unsigned)((np + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlockThreads), 0, s,
                       d_plan, (const uint8_t*)e->d_in.p, d_frame, (uint32_t)np);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    if (prefix) { const uint32_t v = (uint32_t)n; std::memcpy(out, &v, 4); }
    HIP_TRY(hipMemcpyAsync(out + pre, d_frame, pos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)(pos + pre);
}


// The block sequence of an LZ4 frame (u32 size word + data per 64 KiB of input; frame.hip: cj_lz4_frame_compress_blocks) for
// inputs of up to kSplitMax bytes: every 64 KiB block is compressed by 16 / 4 wavefronts (sub-pieces) and joined into one
// LZ4 block; a block that does not shrink is stored.
int64_t large_lz4_frame_blocks(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    cj_engine* e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    const SubPieces sp = sub_pieces(n);
    const size_t sub = sp.sub, sub_stride = sp.lz4_stride;                  // (q = sub-piece: 16 or 4 per block)
    const size_t nq = (n + sub - 1) / sub, nb = (n + kPiece - 1) / kPiece;
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    hipStream_t s = e->stream;
    std::vector<int64_t> res;
    int rc = compress_pieces(e, CJ_CODEC_LZ4_BLOCK, kFlagReportTail | sp.flags, in, n, sub, nq, sub_stride, res);
    if (rc != 0) return rc;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    uint8_t* d_tmp = (uint8_t*)e->d_out.p;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint32_t* d_first = (uint32_t*)(d_meta + 5 * nq);
    hipLaunchKernelGGL(lz4_stitch_plan_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, d_tmp, (uint32_t)sub_stride, d_first, (uint32_t)nq);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<uint32_t> first(nq);
    HIP_TRY(hipMemcpyAsync(first.data(), d_first, nq * 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    for (size_t i = 0; i < nq; i++) if (res[i] < 0) return res[i];

    std::vector<Stitch> plan(nq);
    std::vector<uint64_t> seg(4 * nb);                     // src | dst_off | len | hdr per block (copy_segments)
    uint64_t fpos = 0;
    for (size_t b = 0; b < nb; b++) {
        const size_t q0 = sp.per * b, q1 = std::min(nq, q0 + sp.per);
        const uint64_t len = std::min(kPiece, n - b * kPiece);
        const uint64_t cl = plan_stitch(plan, q0, q1, fpos + 4, n, sub, sub_stride, d_tmp, res, first);
        const bool stored = cl >= len;                     // LZ4F_makeBlock: a block that does not shrink is stored
        if (stored) for (size_t q = q0; q < q1; q++) plan[q] = Stitch{0, 0, 0, 0, 0, 0};
        const uint64_t body = stored ? len : cl;
        seg[b] = (uint64_t)(uintptr_t)(d_in + b * kPiece);
        seg[nb + b] = fpos + 4;
        seg[2 * nb + b] = stored ? len : 0;               // compressed blocks: the size word only, the stitch kernel writes the body
        seg[3 * nb + b] = body | (stored ? 0x80000000ull : 0ull);
        fpos += 4 + body;
    }
    if (fpos > cap) return CJ_E_FRAME_WRITE;
    const size_t plan_bytes = nq * sizeof(Stitch), seg_bytes = 4 * nb * 8, tail_off = (fpos + 15u) & ~(uint64_t)15u;
    if (!e->d_frame.reserve(tail_off + plan_bytes + seg_bytes + 64)) return CJ_E_OOM;
    uint8_t* d_frame = (uint8_t*)e->d_frame.p;
    Stitch* d_plan = reinterpret_cast<Stitch*>(d_frame + tail_off);
    uint64_t* d_seg = reinterpret_cast<uint64_t*>(d_frame + tail_off + plan_bytes);
    HIP_TRY(hipMemcpyAsync(d_plan, plan.data(), plan_bytes, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_seg, seg.data(), seg_bytes, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    launch_copy_segments(d_seg, d_frame, d_seg + nb, d_seg + 2 * nb, d_seg + 3 * nb, 4, (uint32_t)nb, s);
    hipLaunchKernelGGL(lz4_stitch_kernel, dim3((unsigned)((nq + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlockThreads), 0, s,
                       d_plan, (const uint8_t*)d_in, d_frame, (uint32_t)nq);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, d_frame, fpos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)fpos;
}

// A Snappy framed stream (frame.hip: cj_snappy_frame_compress) for inputs of up to kSplitMax bytes: every 64 KiB piece is
// compressed by 16 / 4 wavefronts; its chunk = header (type, length, masked CRC-32C of the uncompressed piece) + varint(piece
// length) + the sub-pieces' element streams.  snap's rule: a piece is stored when it does not shrink by an eighth.
int64_t large_snappy_frame(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    static const uint8_t kIdent[10] = {0xff, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y'};
    cj_engine* e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    const SubPieces sp = sub_pieces(n);
    const size_t sub = sp.sub, sub_stride = sp.sn_stride;                   // (q = sub-piece: 16 or 4 per piece)
    const size_t nq = (n + sub - 1) / sub, np = (n + kPiece - 1) / kPiece;
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    hipStream_t s = e->stream;
    std::vector<int64_t> res;
    int rc = compress_pieces(e, CJ_CODEC_SNAPPY_RAW, sp.flags, in, n, sub, nq, sub_stride, res);
    if (rc != 0) return rc;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;             // 12 nq rows reserved; 0 .. 5 nq in use
    uint8_t* d_tmp = (uint8_t*)e->d_out.p;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    std::vector<uint64_t> pm(2 * np);
    for (size_t p = 0; p < np; p++) { pm[p] = p * kPiece; pm[np + p] = std::min(kPiece, n - p * kPiece); }
    HIP_TRY(hipMemcpyAsync(d_meta + 6 * nq, pm.data(), 2 * np * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    launch_crc32c_pieces(d_in, d_meta + 6 * nq, d_meta + 6 * nq + np, (uint32_t*)(d_meta + 9 * nq), (uint32_t)np, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<uint32_t> crc(np);
    HIP_TRY(hipMemcpyAsync(crc.data(), d_meta + 9 * nq, np * 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    for (size_t i = 0; i < nq; i++) if (res[i] < 0) return res[i];

    std::vector<uint64_t> sa(4 * np), sb(3 * nq, 0);       // with header: src | dst | len | hdr per piece; without: src | dst | len per quarter
    std::vector<uint32_t> vints(np, 0);
    uint64_t fpos = 10;
    for (size_t p = 0; p < np; p++) {
        const size_t q0 = sp.per * p, q1 = std::min(nq, q0 + sp.per);
        const uint64_t len = pm[np + p];
        uint32_t vl = 0, v = 0;
        for (uint64_t x = len;; ) { if (x < 0x80u) { v |= (uint32_t)x << (8 * vl); vl++; break; } v |= (uint32_t)((x & 0x7f) | 0x80u) << (8 * vl); vl++; x >>= 7; }
        vints[p] = v;
        uint64_t comp = vl;
        for (size_t q = q0; q < q1; q++) comp += (uint64_t)res[q] - varint_len(std::min(sub, n - q * sub));
        const bool stored = comp >= len - len / 8;
        const uint64_t body = stored ? len : comp;
        sa[p] = stored ? (uint64_t)(uintptr_t)(d_in + p * kPiece) : 0ull;       // compressed: the varint (address patched below)
        sa[np + p] = fpos + 8;
        sa[2 * np + p] = stored ? len : vl;
        sa[3 * np + p] = (stored ? 1ull : 0ull) | ((body + 4) << 8) | ((uint64_t)crc[p] << 32);
        if (!stored) {
            uint64_t pos = fpos + 8 + vl;
            for (size_t q = q0; q < q1; q++) {
                const uint32_t ph = varint_len(std::min(sub, n - q * sub));
                sb[q] = (uint64_t)(uintptr_t)(d_tmp + q * sub_stride + ph);
                sb[nq + q] = pos;
                sb[2 * nq + q] = (uint64_t)res[q] - ph;
                pos += sb[2 * nq + q];
            }
        }
        fpos += 8 + body;
    }
    if (fpos > cap) return CJ_E_FRAME_WRITE;
    const size_t tail_off = (fpos + 15u) & ~(uint64_t)15u, sa_bytes = sa.size() * 8, sb_bytes = sb.size() * 8;
    if (!e->d_frame.reserve(tail_off + sa_bytes + sb_bytes + np * 4 + 64)) return CJ_E_OOM;
    uint8_t* d_frame = (uint8_t*)e->d_frame.p;
    uint64_t* d_sa = reinterpret_cast<uint64_t*>(d_frame + tail_off);
    uint64_t* d_sb = d_sa + sa.size();
    uint32_t* d_v = reinterpret_cast<uint32_t*>(d_sb + sb.size());
    for (size_t p = 0; p < np; p++) if (sa[p] == 0ull) sa[p] = (uint64_t)(uintptr_t)(d_v + p);
    HIP_TRY(hipMemcpyAsync(d_frame, kIdent, 10, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_sa, sa.data(), sa_bytes, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_sb, sb.data(), sb_bytes, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(d_v, vints.data(), np * 4, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    launch_copy_segments(d_sa, d_frame, d_sa + np, d_sa + 2 * np, d_sa + 3 * np, 8, (uint32_t)np, s);
    launch_copy_segments(d_sb, d_frame, d_sb + nq, d_sb + 2 * nq, nullptr, 0, (uint32_t)nq, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, d_frame, fpos, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    return (int64_t)fpos;
}

size_t large_split_max() { return kSplitMax; }

// =====================================================================================================================
// decompress ONE large stream: parallel parse (big_parse.hip) -> one decoder workgroup per 64 KiB slab of output
// (lz4_decode_lds.hip, kSlab).  Error codes and capacity rules are those of the single-chunk kernels (lz4_block_prologue,
// snappy_parse_kernel's header checks), applied here on the host.
// =====================================================================================================================
// The checks in front of the decoder (lz4_block_prologue / the Snappy length header), on the host.  false: `early` is the
// call's result (an error, or 0 for an empty block); true: decode in[skip ..], element stream from `start`, capacity cap64.
static bool large_prologue(int codec, uint32_t flags, const uint8_t* in, size_t n, size_t cap, uint64_t& skip, uint64_t& start, uint64_t& cap64, int64_t& early) {
    const bool snappy = codec == CJ_CODEC_SNAPPY_RAW;
    skip = 0; start = 0; cap64 = cap;
    if (!snappy) {
        if (flags & CJ_FLAG_LZ4_SIZE_PREFIX) {
            if (n < 4) { early = CJ_E_NO_PREFIX; return false; }
            uint32_t u; std::memcpy(&u, in, 4);
            const int32_t size = (int32_t)u;
            if (size < 0) { early = CJ_E_NEG_PREFIX; return false; }
            if ((uint32_t)size > 0x7E000000u) { early = CJ_E_PREFIX_TOO_BIG; return false; }
            if ((uint64_t)size > cap64) { early = CJ_E_OUT_TOO_SMALL; return false; }
            skip = 4; cap64 = (uint64_t)size;
        } else {
            if (cap64 > 0xFFFFFFFFull || (int32_t)(uint32_t)cap64 < 0) { early = CJ_E_NEG_PREFIX; return false; }
            if ((uint32_t)cap64 > 0x7E000000u) { early = CJ_E_PREFIX_TOO_BIG; return false; }
        }
        if (n - skip > 0x7FFFFFF0ull) { early = CJ_E_CORRUPT; return false; }
        if (cap64 == 0) { early = (n - skip == 1 && in[skip] == 0) ? 0 : (int64_t)CJ_E_CORRUPT; return false; }
        if (n - skip == 0) { early = CJ_E_CORRUPT; return false; }
    } else {
        if (n == 0) { early = CJ_E_SNAPPY_EMPTY; return false; }
        if (n > 0x7FFFFFF0ull) { early = CJ_E_SNAPPY_CORRUPT; return false; }
        uint64_t ulen = 0; uint32_t shift = 0, i = 0, hdr = 0; bool ok = false;
        while (hdr < n && i < 10u) {
            const uint32_t b = in[hdr++];
            if (b < 0x80u) { if (!(i == 9u && b > 1u)) { ulen |= (uint64_t)b << shift; ok = true; } break; }
            ulen |= (uint64_t)(b & 0x7fu) << shift;
            shift += 7; i += 1;
        }
        if (!ok) { early = CJ_E_SNAPPY_HEADER; return false; }
        if (ulen > 0xFFFFFFFFull) { early = CJ_E_SNAPPY_TOO_BIG; return false; }
        if (ulen > cap64) { early = CJ_E_SNAPPY_BUF_SMALL; return false; }
        if (ulen == 0) { early = hdr == n ? 0 : (int64_t)CJ_E_SNAPPY_CORRUPT; return false; }
        if (hdr == n) { early = CJ_E_SNAPPY_CORRUPT; return false; }
        start = hdr; cap64 = ulen;
    }
    return true;
}

// Is the stream made of a handful of elements (incompressible data: one literal run; a run-length pattern: a literal and one
// long match)?  Host-side look at the first kFewElements element headers — it only decides WHICH device path decodes the
// stream (the one-wavefront kernel copies long runs cooperatively and needs no parse: 123 KB of JPEG 0.18 ms against 0.37;
// it is one wavefront, ~1 GB/s, so only streams of up to kFewMaxBytes take it); anything odd answers false and the large
// path reports the error.
constexpr uint32_t kFewElements = 64;          // (a 123 KB JPEG compressed by the sub-piece encoder is 31 literal elements)
constexpr size_t kFewMaxBytes = 256u << 10;
bool large_few_elements(int codec, uint32_t flags, const uint8_t* in, size_t n, size_t cap) {
    uint64_t skip = 0, start = 0, cap64 = cap; int64_t early = 0;
    if (!in || !large_prologue(codec, flags, in, n, cap, skip, start, cap64, early)) return false;
    if (cap64 > kFewMaxBytes || n - skip > kFewMaxBytes + 4096) return false;
    const uint8_t* p = in + skip;
    const uint64_t end = n - skip;
    uint64_t ip = start;
    for (uint32_t k = 0; k < kFewElements; k++) {
        if (ip >= end) return ip == end && codec == CJ_CODEC_SNAPPY_RAW;
        if (codec == CJ_CODEC_SNAPPY_RAW) {
            const uint32_t tag = p[ip];
            if ((tag & 3u) == 0u) {
                uint64_t len = (tag >> 2) + 1u, hdr = 1;
                if (len > 60u) {
                    const uint32_t nb = (uint32_t)len - 60u;
                    if (ip + 1 + nb > end) return false;
                    len = 0;
                    for (uint32_t j = 0; j < nb; j++) len |= (uint64_t)p[ip + 1 + j] << (8u * j);
                    len += 1; hdr = 1 + nb;
                }
                ip += hdr + len;
            } else ip += (tag & 3u) == 1u ? 2u : (tag & 3u) == 2u ? 3u : 5u;
        } else {
            const uint32_t token = p[ip++];
            uint64_t lit = token >> 4;
            if (lit == 15u) { uint32_t b; do { if (ip >= end) return false; b = p[ip++]; lit += b; } while (b == 255u); }
            ip += lit;
            if (ip == end) return true;                     // the last, literal-only sequence
            if (ip + 2 > end) return false;
            ip += 2;
            if ((token & 15u) == 15u) { uint32_t b; do { if (ip >= end) return false; b = p[ip++]; } while (b == 255u); }
        }
    }
    return false;
}

// debug aid (tuning builds, -DCJ_DEBUG_KNOBS): CJ_SLAB_PROFILE=1 switches the slab decoder's per-phase cycle counters on (read with cj_debug_lds_phase_cycles)
static uint32_t slab_profile_flag() {
#ifdef CJ_DEBUG_KNOBS
    static const uint32_t f = std::getenv("CJ_SLAB_PROFILE") ? CJ_FLAG_DEBUG_PROFILE : 0u;
    return f;
#else
    return 0u;
#endif
}

static std::vector<uint32_t>* g_dbg_sync = nullptr;       // set by cj_debug_big_parse only (single-threaded test hook)
static uint64_t g_dbg_nseq = 0;

int64_t large_decompress(int codec, uint32_t flags, const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    cj_engine* e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    const bool snappy = codec == CJ_CODEC_SNAPPY_RAW;
    const int64_t corrupt = snappy ? CJ_E_SNAPPY_CORRUPT : CJ_E_CORRUPT;
    uint64_t skip = 0, start = 0, cap64 = cap;
    {
        int64_t early = 0;
        if (!large_prologue(codec, flags, in, n, cap, skip, start, cap64, early)) return early;
    }
    const uint32_t iend = (uint32_t)(n - skip);
    const uint32_t piece = big_piece_for(iend);
    const uint32_t np = (uint32_t)((iend - start + piece - 1) / piece);

    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    hipStream_t s = e->stream;
    // parse scratch, 256 B aligned regions
    size_t off = 0;
    const auto region = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_status = region(64), o_bits = region((size_t)np * (piece / 8)), o_merge = region((size_t)np * 256),
                 o_exit = region((size_t)np * 256), o_next = region((size_t)np * 256), o_fe = region((size_t)np * 256), o_entry = region((size_t)np * 8), o_lidx = region((size_t)np * 256),
                 o_lop = region((size_t)np * 512), o_tot = region((size_t)np * 16), o_sync = region(((size_t)iend / 16 + 2) * 8);
    if (!e->d_in.reserve(n + 64) || !e->d_big.reserve(off)) return CJ_E_OOM;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint8_t* b = (uint8_t*)e->d_big.p;
    HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemsetAsync(b + o_status, 0, 64, s), CJ_E_NO_DEVICE);
    BigParse bp = {};
    bp.in = d_in + skip; bp.iend = iend; bp.start = (uint32_t)start; bp.piece = piece; bp.np = np; bp.cap = cap64;
    bp.bits = (uint32_t*)(b + o_bits); bp.merge = (uint32_t*)(b + o_merge); bp.exitp = (uint32_t*)(b + o_exit); bp.next_tab = (uint32_t*)(b + o_next); bp.fe_tab = (uint32_t*)(b + o_fe);
    bp.entry = (uint2*)(b + o_entry); bp.lane_idx = (uint32_t*)(b + o_lidx); bp.lane_op = (uint64_t*)(b + o_lop);
    bp.totals = (uint64_t*)(b + o_tot); bp.sync = (uint2*)(b + o_sync); bp.status = (uint32_t*)(b + o_status);
    launch_big_parse(bp, codec, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    uint32_t st[16];
    HIP_TRY(hipMemcpyAsync(st, b + o_status, 64, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    if (st[0] != 0 || st[1] != 0 || st[8] != 1) return corrupt;
    const uint64_t n_seq = ((uint64_t)st[3] << 32) | st[2], total = ((uint64_t)st[7] << 32) | st[6];
    if (total == 0) return 0;
    const uint32_t n_sync = (uint32_t)((n_seq + kSyncEvery - 1) / kSyncEvery);
    const uint32_t n_slabs = (uint32_t)((total + 65535) / 65536);
    if (g_dbg_sync) {                                       // test hook (cj_debug_big_parse): hand the sync points to the host
        g_dbg_nseq = n_seq;
        g_dbg_sync->resize((size_t)n_sync * 2);
        HIP_TRY(hipMemcpy(g_dbg_sync->data(), bp.sync, (size_t)n_sync * 8, hipMemcpyDeviceToHost), CJ_E_NO_DEVICE);
    }

    // slab descriptors: 5 u64 rows | meta | first | max_rec, counter | done flags
    const size_t r_meta = 5 * (size_t)n_slabs, r_first = r_meta + n_slabs, r_misc = r_first + n_slabs, r_done = r_misc + 2,
                 rows = r_done + (n_slabs + 1) / 2 + 1;
    if (!e->d_meta.reserve(rows * 8)) return CJ_E_OOM;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    HIP_TRY(hipMemsetAsync(d_meta + r_misc, 0, (rows - r_misc) * 8, s), CJ_E_NO_DEVICE);
    BigSlabs sd;
    sd.sync = bp.sync; sd.n_sync = n_sync; sd.n_seq = n_seq; sd.total = total; sd.iend = iend; sd.in_base_off = skip; sd.out_base_off = 0; sd.sync_index_base = 0; sd.n_slabs = n_slabs;
    sd.in_off = d_meta; sd.in_len = d_meta + n_slabs; sd.out_off = d_meta + 2 * (size_t)n_slabs; sd.out_cap = d_meta + 3 * (size_t)n_slabs;
    sd.result = (int64_t*)(d_meta + 4 * (size_t)n_slabs); sd.meta = (uint2*)(d_meta + r_meta); sd.first = (uint2*)(d_meta + r_first);
    sd.max_rec = (uint32_t*)(d_meta + r_misc);
    launch_big_slabs(sd, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    uint32_t max_rec = 0;
    HIP_TRY(hipMemcpyAsync(&max_rec, sd.max_rec, 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    if (e->n_cu == 0) HIP_TRY(hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, e->device), CJ_E_NO_DEVICE);
    const uint32_t grid = std::min<uint32_t>(2u * (uint32_t)e->n_cu, n_slabs);
    const uint32_t cross_stride = 3u * ((max_rec + 63u) & ~63u), tab_stride = 4u * ((max_rec + 63u) & ~63u);      // D1's entries + what D1f adds (forwarded copies, split straddlers)      // records + slab extras + forwarded literal copies
    const size_t tab_bytes = (size_t)grid * tab_stride * 16, cross_bytes = (size_t)grid * cross_stride * 16;
    if (!e->d_bigtab.reserve(tab_bytes + cross_bytes + (size_t)grid * (tab_stride + 512u) * 4) || !e->d_out.reserve(total + 256)) return CJ_E_OOM;
    BatchArgs a;
    fill_args(a, slab_profile_flag(), n_slabs, d_in, sd.in_off, sd.in_len, (uint8_t*)e->d_out.p, sd.out_off, sd.out_cap, sd.result);
    launch_lz4_decode_lds2_slabs(a, bp.sync, sd.meta, e->d_bigtab.p, (uint32_t*)(d_meta + r_misc) + 1, sd.first, iend,
                                 (uint32_t*)(d_meta + r_done), (uint8_t*)e->d_bigtab.p + tab_bytes, tab_stride, cross_stride, grid, s, codec);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<int64_t> res(n_slabs);
    HIP_TRY(hipMemcpyAsync(res.data(), sd.result, (size_t)n_slabs * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(out, e->d_out.p, total, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    for (uint32_t i = 0; i < n_slabs; i++)
        if (res[i] < 0) return corrupt;                     // the decoder's stall guard: cannot happen for a stream the parse accepted
    return (int64_t)total;
}

// Several streams (LZ4 blocks without their size prefix / Snappy raw streams with starts[j] = the first element's position),
// each a large stream of its own — the blocks of an LZ4 frame with large independent blocks (frame.hip), the large chunks of
// a host batch (engine.hip).  The parse kernels run over the pieces of ALL blocks at once (launch_big_parse_many) and the decoder
// over the slabs of all blocks (every block's first slab has no predecessor); the host waits twice in total.  result[j] = decoded size or
// CJ_E_CORRUPT; returns 0, or a CJ_E_* that concerns the call as a whole (CJ_E_BAD_ARG: not for this path).
int large_decompress_many(cj_engine* e, int codec, size_t nj, const uint8_t* const* ins, const size_t* lens, const size_t* starts, uint8_t* const* outs, const size_t* caps, int64_t* result) {
    const int64_t corrupt = codec == CJ_CODEC_SNAPPY_RAW ? CJ_E_SNAPPY_CORRUPT : CJ_E_CORRUPT;
    if (!e) e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    if (nj == 0) return 0;
    struct Job { size_t in_off, meta_off, out_off; BigSlabs sd; uint64_t n_seq, total; uint32_t n_sync, n_slabs; };
    std::vector<Job> jobs(nj);
    std::vector<BigParse> bps(nj);
    std::vector<uint2> pmap;
    size_t in_total = 0, big_total = 0;
    uint32_t max_piece = kBigPieceTiny;
    std::vector<size_t> big_off(nj);
    // the blocks of one frame lie in ONE host buffer, a few header bytes apart: one copy to the device instead of one per block
    const uint8_t* span_lo = ins[0]; const uint8_t* span_hi = ins[0] + lens[0];
    size_t sum_len = 0;
    for (size_t j = 0; j < nj; j++) { span_lo = std::min(span_lo, ins[j]); span_hi = std::max(span_hi, ins[j] + lens[j]); sum_len += lens[j]; }
    const bool one_span = (size_t)(span_hi - span_lo) <= sum_len + sum_len / 8 + 4096;
    size_t slab_bound = 0;                                                 // the descriptors of the decoder's chunk list reuse the tail of the parse scratch
    for (size_t j = 0; j < nj; j++) slab_bound += (caps[j] + 65535) / 65536;
    for (size_t j = 0; j < nj; j++) {
        if (lens[j] == 0 || lens[j] > 0x7FFFFFF0ull || caps[j] == 0 || caps[j] > 0xFFFFFFFFull || (starts && starts[j] >= lens[j])) return CJ_E_BAD_ARG;
        BigParse& bp = bps[j];
        bp = BigParse{};
        bp.iend = (uint32_t)lens[j]; bp.start = starts ? (uint32_t)starts[j] : 0u; bp.cap = caps[j];
        bp.piece = big_piece_for(bp.iend);
        bp.np = (bp.iend - bp.start + bp.piece - 1) / bp.piece;
        max_piece = std::max(max_piece, bp.piece);
        for (uint32_t p = 0; p < bp.np; p++) pmap.push_back(make_uint2((uint32_t)j, p));
        jobs[j].in_off = one_span ? (size_t)(ins[j] - span_lo) : in_total; in_total += (lens[j] + 64 + 255) & ~(size_t)255;
        const size_t np = bp.np;
        big_off[j] = big_total;
        big_total += (((np * (bp.piece / 8)) + 255) & ~(size_t)255) + 6 * ((np * 256 + 255) & ~(size_t)255) + ((np * 8 + 255) & ~(size_t)255)
                   + ((np * 512 + 255) & ~(size_t)255) + ((np * 16 + 255) & ~(size_t)255) + ((((size_t)bp.iend / 16 + 2) * 8 + 255) & ~(size_t)255);
    }
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    hipStream_t s = e->stream;
    const size_t o_status = big_total, o_jobs = o_status + 64 * nj, o_pmap = (o_jobs + nj * sizeof(BigParse) + 255) & ~(size_t)255;
    if (!e->d_in.reserve((one_span ? (size_t)(span_hi - span_lo) : in_total) + 64) || !e->d_big.reserve(o_status + std::max(o_pmap - o_status + pmap.size() * 8, ((nj * sizeof(BigSlabs) + 255) & ~(size_t)255) + slab_bound * 8) + 64)) return CJ_E_OOM;
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint8_t* b = (uint8_t*)e->d_big.p;
    uint32_t* d_status = (uint32_t*)(b + o_status);                      // nj * 16 words, contiguous: one copy back
    HIP_TRY(hipMemsetAsync(d_status, 0, 64 * nj, s), CJ_E_NO_DEVICE);
    for (size_t j = 0; j < nj; j++) {
        BigParse& bp = bps[j];
        const size_t np = bp.np;
        size_t off = big_off[j];
        const auto region = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        bp.in = d_in + jobs[j].in_off;
        bp.bits = (uint32_t*)(b + region(np * (bp.piece / 8))); bp.merge = (uint32_t*)(b + region(np * 256)); bp.exitp = (uint32_t*)(b + region(np * 256));
        bp.next_tab = (uint32_t*)(b + region(np * 256)); bp.fe_tab = (uint32_t*)(b + region(np * 256)); bp.entry = (uint2*)(b + region(np * 8));
        bp.lane_idx = (uint32_t*)(b + region(np * 256)); bp.lane_op = (uint64_t*)(b + region(np * 512)); bp.totals = (uint64_t*)(b + region(np * 16));
        bp.sync = (uint2*)(b + region(((size_t)bp.iend / 16 + 2) * 8)); bp.status = d_status + 16 * j;
        if (!one_span) HIP_TRY(hipMemcpyAsync(d_in + jobs[j].in_off, ins[j], lens[j], hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    }
    if (one_span) HIP_TRY(hipMemcpyAsync(d_in, span_lo, (size_t)(span_hi - span_lo), hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(b + o_jobs, bps.data(), nj * sizeof(BigParse), hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(b + o_pmap, pmap.data(), pmap.size() * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    launch_big_parse_many((const BigParse*)(b + o_jobs), (uint32_t)nj, (const uint2*)(b + o_pmap), (uint32_t)pmap.size(), max_piece, codec, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<uint32_t> st(16 * nj);
    HIP_TRY(hipMemcpyAsync(st.data(), d_status, 64 * nj, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    // the slabs of all blocks form ONE chunk list for the decoder (every block's first slab has no predecessor)
    size_t n_slabs = 0, out_total = 0;
    for (size_t j = 0; j < nj; j++) {
        Job& J = jobs[j];
        const uint32_t* t = st.data() + 16 * j;
        J.n_slabs = 0; J.total = 0;
        if (t[0] != 0 || t[1] != 0 || t[8] != 1) { result[j] = corrupt; continue; }
        J.n_seq = ((uint64_t)t[3] << 32) | t[2]; J.total = ((uint64_t)t[7] << 32) | t[6];
        result[j] = (int64_t)J.total;
        if (J.total == 0) continue;
        J.n_sync = (uint32_t)((J.n_seq + kSyncEvery - 1) / kSyncEvery);
        J.n_slabs = (uint32_t)((J.total + 65535) / 65536);
        J.meta_off = n_slabs; n_slabs += J.n_slabs;                       // index of the block's first slab
        J.out_off = out_total; out_total += (J.total + 256 + 255) & ~(size_t)255;
    }
    // outputs that follow each other in the caller's buffer (the blocks of a frame, all full but the last): same layout on the
    // device, one copy back
    bool out_contig = true;
    {
        const uint8_t* expect = nullptr;
        for (size_t j = 0; j < nj; j++) {
            if (jobs[j].n_slabs == 0) { if (result[j] < 0) out_contig = false; continue; }
            if (expect != nullptr && outs[j] != expect) out_contig = false;
            expect = outs[j] + jobs[j].total;
        }
        if (out_contig) { out_total = 0; for (size_t j = 0; j < nj; j++) if (jobs[j].n_slabs) { jobs[j].out_off = out_total; out_total += jobs[j].total; } }
    }
    if (n_slabs == 0) return 0;
    if (n_slabs > 0xFFFFFFF0ull) return CJ_E_BAD_ARG;
    const size_t r_misc = 7 * n_slabs, r_done = r_misc + 2, rows = r_done + (n_slabs + 1) / 2 + 1;
    if (!e->d_meta.reserve(rows * 8) || !e->d_out.reserve(out_total + 256)) return CJ_E_OOM;
    uint64_t* m = (uint64_t*)e->d_meta.p;
    HIP_TRY(hipMemsetAsync(m + r_misc, 0, (rows - r_misc) * 8, s), CJ_E_NO_DEVICE);
    const uint2* sync_base = reinterpret_cast<const uint2*>(b);
    std::vector<BigSlabs> sds(nj);
    std::vector<uint2> slab_job(n_slabs);
    for (size_t j = 0; j < nj; j++) {
        Job& J = jobs[j];
        BigSlabs& sd = sds[j];
        sd = BigSlabs{};
        if (J.n_slabs == 0) continue;
        const size_t f = J.meta_off;
        sd.sync = bps[j].sync; sd.n_sync = J.n_sync; sd.n_seq = J.n_seq; sd.total = J.total; sd.iend = bps[j].iend; sd.in_base_off = J.in_off;
        sd.out_base_off = J.out_off; sd.sync_index_base = (uint32_t)(bps[j].sync - sync_base); sd.n_slabs = J.n_slabs;
        sd.in_off = m + f; sd.in_len = m + n_slabs + f; sd.out_off = m + 2 * n_slabs + f; sd.out_cap = m + 3 * n_slabs + f;
        sd.result = (int64_t*)(m + 4 * n_slabs + f); sd.meta = (uint2*)(m + 5 * n_slabs) + f; sd.first = (uint2*)(m + 6 * n_slabs) + f;
        sd.max_rec = (uint32_t*)(m + r_misc);
        for (uint32_t i = 0; i < J.n_slabs; i++) slab_job[f + i] = make_uint2((uint32_t)j, i);
    }
    // the descriptors and the slab -> block map travel in the parse scratch (the verdict words and the job structs are done with)
    const size_t o_sds = o_status, o_sj = (o_sds + nj * sizeof(BigSlabs) + 255) & ~(size_t)255;
    HIP_TRY(hipMemcpyAsync(b + o_sds, sds.data(), nj * sizeof(BigSlabs), hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipMemcpyAsync(b + o_sj, slab_job.data(), n_slabs * 8, hipMemcpyHostToDevice, s), CJ_E_NO_DEVICE);
    launch_big_slabs_many((const BigSlabs*)(b + o_sds), (const uint2*)(b + o_sj), (uint32_t)n_slabs, s);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    uint32_t max_rec = 0;
    HIP_TRY(hipMemcpyAsync(&max_rec, m + r_misc, 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);

    if (e->n_cu == 0) HIP_TRY(hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, e->device), CJ_E_NO_DEVICE);
    const uint32_t grid = (uint32_t)std::min<size_t>(2u * (size_t)e->n_cu, n_slabs);
    const uint32_t cross_stride = 3u * ((max_rec + 63u) & ~63u), tab_stride = 4u * ((max_rec + 63u) & ~63u);
    const size_t tab_bytes = (size_t)grid * tab_stride * 16, cross_bytes = (size_t)grid * cross_stride * 16;
    if (!e->d_bigtab.reserve(tab_bytes + cross_bytes + (size_t)grid * (tab_stride + 512u) * 4)) return CJ_E_OOM;
    BatchArgs a;
    fill_args(a, slab_profile_flag(), n_slabs, d_in, m, m + n_slabs, (uint8_t*)e->d_out.p, m + 2 * n_slabs, m + 3 * n_slabs, (int64_t*)(m + 4 * n_slabs));
    launch_lz4_decode_lds2_slabs(a, sync_base, m + 5 * n_slabs, e->d_bigtab.p, (uint32_t*)(m + r_misc) + 1, m + 6 * n_slabs, 0u,
                                 (uint32_t*)(m + r_done), (uint8_t*)e->d_bigtab.p + tab_bytes, tab_stride, cross_stride, grid, s, codec);
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    std::vector<int64_t> res(n_slabs);
    HIP_TRY(hipMemcpyAsync(res.data(), m + 4 * n_slabs, n_slabs * 8, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    if (out_contig) {
        for (size_t j = 0; j < nj; j++)
            if (jobs[j].n_slabs) { HIP_TRY(hipMemcpyAsync(outs[j], e->d_out.p, out_total, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE); break; }
    } else {
        for (size_t j = 0; j < nj; j++)
            if (jobs[j].n_slabs) HIP_TRY(hipMemcpyAsync(outs[j], (uint8_t*)e->d_out.p + jobs[j].out_off, jobs[j].total, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    }
    HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
    for (size_t j = 0; j < nj; j++)
        for (uint32_t i = 0; i < jobs[j].n_slabs; i++) if (res[jobs[j].meta_off + i] < 0) result[j] = corrupt;
    return 0;
}

// The large chunks idx[0 .. n_listed) of a host batch (cj_batch_host): the checks in front of the decoder on the host, then all
// of them together through large_decompress_many.
int large_decompress_listed(cj_engine* e, int codec, uint32_t flags, size_t n_listed, const size_t* idx, const uint8_t* const* in_ptrs,
                            const size_t* in_lens, uint8_t* const* out_ptrs, const size_t* out_caps, int64_t* result) {
    std::vector<const uint8_t*> ins; std::vector<size_t> lens, starts, caps, which; std::vector<uint8_t*> outs;
    for (size_t k = 0; k < n_listed; k++) {
        const size_t i = idx[k];
        uint64_t skip, start, cap64; int64_t early = 0;
        if (!large_prologue(codec, flags, in_ptrs[i], in_lens[i], out_caps[i], skip, start, cap64, early)) { result[i] = early; continue; }
        ins.push_back(in_ptrs[i] + skip); lens.push_back(in_lens[i] - skip); starts.push_back(start); outs.push_back(out_ptrs[i]); caps.push_back(cap64);
        which.push_back(i);
    }
    if (ins.empty()) return 0;
    std::vector<int64_t> r(ins.size());
    const int rc = large_decompress_many(e, codec, ins.size(), ins.data(), lens.data(), starts.data(), outs.data(), caps.data(), r.data());
    if (rc != 0) return rc;
    for (size_t k = 0; k < which.size(); k++) result[which[k]] = r[k];
    return 0;
}

}  // namespace cj

// Test hook: decompress one stream through the large path and return the parse stage's absolute sync points
// ((ip, op) of every 8th sequence, as pairs of u32) and the sequence count.  Returns the decoded size or CJ_E_*.
extern "C" int64_t cj_debug_big_parse(int codec, uint32_t flags, const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                      uint32_t* sync_pairs, size_t max_pairs, uint64_t* n_seq) {
    std::vector<uint32_t> v;
    cj::g_dbg_sync = &v;
    const int64_t r = cj::large_decompress(codec, flags, in, n, out, cap);
    cj::g_dbg_sync = nullptr;
    if (n_seq) *n_seq = cj::g_dbg_nseq;
    if (r >= 0 && sync_pairs) std::memcpy(sync_pairs, v.data(), std::min(v.size() / 2, max_pairs) * 8);
    return r;
}

