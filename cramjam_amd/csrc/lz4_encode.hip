// lz4_encode.hip — LZ4 *block* encoder for gfx950, one wavefront per independent chunk (matcher: cj_enc2.hpp).
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/lz4.rs:127,206:
// libcramjam::lz4::block::compress_into -> lz4 crate compress_to_buffer -> LZ4_compress_default,
// including the optional u32-LE uncompressed-size prefix (`store_size`, default on).
// Stream rules honoured (lz4_Block_format.md): min match 4, offset 1..65535, the last match starts
// at least 12 bytes before the end, the last 5 bytes are literals, inputs < 13 bytes are one literal
// run.  Every emitted block decodes with LZ4_decompress_safe at exact capacity (tests check this
// against the CPU oracle).
#include "cj_enc2.hpp"

namespace cj {

// writes the 255-run length extension for value v (v = len - 15): v/255 bytes of 255 then v%255
__device__ __forceinline__ uint32_t emit_len_ext(uint8_t* out, uint32_t op, uint32_t v) {
    const uint32_t full = v / 255u, rem = v - full * 255u, lane = lane_id();
    for (uint32_t k = 0; k < full; k += 64u)
        if (k + lane < full) out[op + k + lane] = 255u;
    if (lane == 0) out[op + full] = (uint8_t)rem;
    return op + full + 1u;
}

// ---- round-based matcher (cj_enc2.hpp): the encoder of every batch and of the split pieces of large buffers ----
struct Lz4Fmt {
    static __device__ __forceinline__ uint32_t last_start(uint32_t n) { return n - 12u; }      // a match may start here at the latest
    static __device__ __forceinline__ uint32_t limit(uint32_t n) { return n - 5u; }            // and must end here at the latest
    static __device__ __forceinline__ uint32_t seq_size(uint32_t lit, uint32_t code, uint32_t) {
        return 3u + lit + (lit >= 15u ? 1u + (lit - 15u) / 255u : 0u) + (code >= 15u ? 1u + (code - 15u) / 255u : 0u);
    }
    static __device__ __forceinline__ uint32_t emit_lane(enc2::gcptr in, enc2::gptr out, uint32_t o, uint32_t lit0, uint32_t lit, uint32_t code, uint32_t off) {
#ifdef CJ_EXP_NO_EMIT
        return 0u;          // (experiment x01: what the emission's scattered stores and literal loads cost)
#endif
        enc2::s8(out, o, ((lit < 15u ? lit : 15u) << 4) | (code < 15u ? code : 15u));
        uint32_t q = o + 1u;
        if (lit >= 15u) {
            uint32_t v = lit - 15u;
            while (v >= 255u) { enc2::s8(out, q, 255u); q += 1u; v -= 255u; }
            enc2::s8(out, q, v); q += 1u;
        }
        const uint32_t lit_at = q;
        if (lit < enc2::kLaneLit) enc2::lane_copy(out, q, in, lit0, lit);
        q += lit;
        enc2::s8(out, q, off); enc2::s8(out, q + 1u, off >> 8);
        if (code >= 15u) {
            uint32_t v = code - 15u;
            q += 2u;
            while (v >= 255u) { enc2::s8(out, q, 255u); q += 1u; v -= 255u; }
            enc2::s8(out, q, v);
        }
        return lit_at;
    }
    static __device__ __forceinline__ uint32_t emit_wave(enc2::gcptr gin, enc2::gptr gout, uint32_t op, uint32_t lit0, uint32_t lit, uint32_t off, uint32_t mlen) {
        const uint8_t* in = (const uint8_t*)gin; uint8_t* out = (uint8_t*)gout;
        const uint32_t lane = lane_id(), mcode = mlen - 4u;
        if (lane == 0) out[op] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
        op += 1;
        if (lit >= 15u) op = emit_len_ext(out, op, lit - 15u);
        wave_copy(out + op, in + lit0, lit);
        op += lit;
        if (lane < 2) out[op + lane] = (uint8_t)(off >> (8u * lane));
        op += 2;
        if (mcode >= 15u) op = emit_len_ext(out, op, mcode - 15u);
        return op;
    }
};

// Split pieces (large.hip, buffers of up to 32 MiB, where one wavefront per 64 KiB piece would leave most of the GPU idle): every 64 KiB
// piece is cut into split_per(flags) = 16 or 4 consecutive sub-pieces, one wavefront each — the chunks of the batch ARE the sub-pieces,
// position 0 of a walk = the start of its piece.  Each wavefront first indexes the data BEFORE its sub-piece (HashTab::preindex), so it
// finds what a serial walk over the piece would; each writes its own stream, and the streams are stitched like any other pieces.
template <bool kSplit, int kW>
__device__ __forceinline__ void lz4_encode_chunk(const BatchArgs& a, uint32_t chunk, const HashTab& ht, uint32_t* scr, uint32_t wave) {
    const uint64_t base_off = kSplit ? a.in_off[chunk & ~(split_per(a.flags) - 1u)] : a.in_off[chunk];      // the piece's first sub-piece
    const uint8_t* in = a.in_base + base_off;               // position 0 = start of the piece
    const uint32_t q0 = (uint32_t)(a.in_off[chunk] - base_off);      // this wave's range = [q0, n)
    const uint64_t n64 = q0 + a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    const uint64_t cap64 = a.out_cap[chunk];
    const uint32_t lane = lane_id();
    const bool prefix = (a.flags & CJ_FLAG_LZ4_SIZE_PREFIX) != 0;

    const bool first = wave == 0u;              // the chunk's first wavefront writes everything outside the rounds
    if (n64 > 0x7E000000ull) { if (first && lane == 0) a.result[chunk] = CJ_E_INPUT_TOO_LARGE; return; }
    const uint32_t n = (uint32_t)n64;
    // the engine only launches with capacity >= LZ4_compressBound(n) (+4); anything smaller is refused here
    const uint64_t need = (uint64_t)(n - q0) + (n - q0) / 255u + 16u + (prefix ? 4u : 0u);
    if (cap64 < need) { if (first && lane == 0) a.result[chunk] = CJ_E_COMPRESS_FAILED; return; }
    if (prefix) {
        if (first && lane < 4) out[lane] = (uint8_t)(n >> (8u * lane));
        out += 4;
    }
    uint32_t anchor = q0, op = 0;
    if (n - q0 >= 13u) {
        ht.clear(threadIdx.x, 64u * kW);
        if constexpr (kSplit) ht.preindex(in, q0);
        ht.settle();
        enc2::Walk<Lz4Fmt, kW> w{enc2::uniform_gptr(in), (enc2::gptr)enc2::uniform_gptr(out), n, Lz4Fmt::last_start(n), Lz4Fmt::limit(n), scr, ht, 0u, 0u, wave};
        anchor = w.run(q0);
        op = w.op;
    }
    if (!first) return;
    uint64_t tail_report = 0;
    {   // last literals
        const uint32_t lit = n - anchor;
        if (a.flags & kFlagReportTail) tail_report = (uint64_t)lit << 32;
        if (lane == 0) out[op] = (uint8_t)((lit < 15u ? lit : 15u) << 4);
        op += 1;
        if (lit >= 15u) op = emit_len_ext(out, op, lit - 15u);
        wave_copy(out + op, in + anchor, lit);
        op += lit;
    }
    if (lane == 0) a.result[chunk] = (int64_t)(((uint64_t)op + (prefix ? 4u : 0u)) | tail_report);
}

template <bool kSplit, int kW>
__global__ __launch_bounds__(64 * kW) void lz4_encode_kernel(BatchArgs a) {
    __shared__ uint16_t ht_lds[kHashSize];
    __shared__ uint32_t scr[enc2::Walk<Lz4Fmt, kW>::kWords];
    const uint32_t chunk = blockIdx.x;
    if (chunk >= a.n_chunks) return;
    lz4_encode_chunk<kSplit, kW>(a, chunk, HashTab{ht_lds}, scr, uni(threadIdx.x >> 6));
}

// Two wavefronts per chunk (cj_enc2.hpp): a CU's LDS holds nine tables, a wavefront issues one instruction per ~5 cycles — two per
// table give the CU eighteen instruction streams.  One workgroup per chunk, launched plainly: the hardware dispatches the next
// workgroup when one finishes (a persistent grid with a chunk counter measured 152 GB/s against 170: its loop costs the kernel its
// register budget).  The sub-pieces of split pieces (large.hip: 4 / 16 KiB each, pre-indexed tables) take one wavefront.
constexpr int kEncWaves = 2;

#ifdef CJ_ENC_PROFILE
extern "C" __attribute__((visibility("default"))) int cj_debug_enc_profile(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(enc2::g_enc_prof), 16 * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(enc2::g_enc_prof), z, sizeof z); }
    return 0;
}
#endif

// One workgroup per chunk, in grids of at most kEncGrid chunks: HIP refuses a grid of 2^32 threads or more (33.5 M chunks of 128), and the
// batch API takes up to 0xFFFFFFF0.  Returns hipSuccess or the launch's error (round-5 advisor: the launch result was never looked at).
hipError_t launch_lz4_encode(const BatchArgs& a, hipStream_t s) {
    constexpr uint32_t kEncGrid = 1u << 24;
    for (uint32_t start = 0; start < a.n_chunks; start += kEncGrid) {
        BatchArgs b = a;
        b.in_off += start; b.in_len += start; b.out_off += start; b.out_cap += start; b.result += start;
        b.n_chunks = a.n_chunks - start < kEncGrid ? a.n_chunks - start : kEncGrid;
        // (sub-pieces of a split piece are indexed from their piece's first one: a slice starts on a piece boundary — kEncGrid is a multiple of every split)
        if (a.flags & kFlagSplitPieces) hipLaunchKernelGGL((lz4_encode_kernel<true, 1>), dim3(b.n_chunks), dim3(64), 0, s, b);
        else hipLaunchKernelGGL((lz4_encode_kernel<false, kEncWaves>), dim3(b.n_chunks), dim3(64 * kEncWaves), 0, s, b);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

}  // namespace cj
