// snappy_encode.hip — Snappy *raw* encoder for gfx950, one wavefront per independent chunk (matcher: cj_enc2.hpp).
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/snappy.rs:75,97:
// libcramjam::snappy::raw::compress -> snap 1.1.1 raw::Encoder::compress.  Output = varint(len)
// preamble + literal / copy-1 / copy-2 elements (format_description.txt); copies longer than 64
// are split exactly like the CPU encoders do (64,64,...,[60],rest) so every element is canonical.
// The 64 KiB window of the match finder keeps every offset < 65536, so copy-4 is never emitted.
#include "cj_enc2.hpp"

namespace cj {

__device__ __forceinline__ uint32_t emit_snappy_literal(uint8_t* out, uint32_t op, const uint8_t* src, uint32_t len) {
    const uint32_t lane = lane_id();
    const uint32_t n1 = len - 1u;
    uint32_t hdr;
    if (n1 < 60u) {
        if (lane == 0) out[op] = (uint8_t)(n1 << 2);
        hdr = 1;
    } else {
        const uint32_t nb = n1 < 256u ? 1u : n1 < 65536u ? 2u : n1 < 16777216u ? 3u : 4u;
        if (lane == 0) out[op] = (uint8_t)((59u + nb) << 2);
        if (lane >= 1u && lane <= nb) out[op + lane] = (uint8_t)(n1 >> (8u * (lane - 1u)));
        hdr = 1u + nb;
    }
    wave_copy(out + op + hdr, src, len);
    return op + hdr + len;
}

__device__ __forceinline__ uint32_t emit_snappy_copy(uint8_t* out, uint32_t op, uint32_t off, uint32_t len) {
    const uint32_t lane = lane_id();
    // 64-byte pieces while len >= 68 (keeps >= 4 for the tail)
    const uint32_t n64 = len >= 68u ? (len - 4u) / 64u : 0u;
    for (uint32_t k = 0; k < n64; k += 64u) {
        const uint32_t i = k + lane;
        if (i < n64) {
            uint8_t* p = out + op + 3u * i;
            p[0] = (uint8_t)(2u | (63u << 2));
            p[1] = (uint8_t)off;
            p[2] = (uint8_t)(off >> 8);
        }
    }
    op += 3u * n64;
    len -= 64u * n64;
    if (len > 64u) {
        if (lane == 0) { out[op] = (uint8_t)(2u | (59u << 2)); out[op + 1] = (uint8_t)off; out[op + 2] = (uint8_t)(off >> 8); }
        op += 3; len -= 60u;
    }
    if (len < 12u && off < 2048u) {
        if (lane == 0) { out[op] = (uint8_t)(1u | ((len - 4u) << 2) | ((off >> 8) << 5)); out[op + 1] = (uint8_t)off; }
        op += 2;
    } else {
        if (lane == 0) { out[op] = (uint8_t)(2u | ((len - 1u) << 2)); out[op + 1] = (uint8_t)off; out[op + 2] = (uint8_t)(off >> 8); }
        op += 3;
    }
    return op;
}

__device__ __forceinline__ uint32_t snappy_literal_size(uint32_t lit) {     // header + data bytes of a literal element (0 for none)
    if (lit == 0u) return 0u;
    const uint32_t n1 = lit - 1u;
    return lit + (n1 < 60u ? 1u : n1 < 256u ? 2u : n1 < 65536u ? 3u : n1 < 16777216u ? 4u : 5u);
}

__device__ __forceinline__ uint32_t snappy_copy_size(uint32_t off, uint32_t len) {   // mirrors emit_snappy_copy's splitting
    const uint32_t n64 = len >= 68u ? (len - 4u) / 64u : 0u;
    uint32_t bytes = 3u * n64;
    len -= 64u * n64;
    if (len > 64u) { bytes += 3u; len -= 60u; }
    return bytes + ((len < 12u && off < 2048u) ? 2u : 3u);
}

// ---- round-based matcher (cj_enc2.hpp): the encoder of every batch and of the split pieces of large buffers ----
struct SnappyFmt {
    // a copy may start in the last 8 bytes of the input neither here nor (for its last 15) in the CPU encoders: the matcher's
    // position lanes read 8 bytes at a time
    static __device__ __forceinline__ uint32_t last_start(uint32_t n) { return n - 8u; }
    static __device__ __forceinline__ uint32_t limit(uint32_t n) { return n; }
    static __device__ __forceinline__ uint32_t seq_size(uint32_t lit, uint32_t code, uint32_t off) {
        return snappy_literal_size(lit) + snappy_copy_size(off, code + 4u);
    }
    static __device__ __forceinline__ uint32_t emit_lane(enc2::gcptr in, enc2::gptr out, uint32_t o, uint32_t lit0, uint32_t lit, uint32_t code, uint32_t off) {
        uint32_t lit_at = o;
        if (lit) {          // lit < 65536: one to three header bytes
            const uint32_t n1 = lit - 1u;
            if (n1 < 60u) { enc2::s8(out, o, n1 << 2); o += 1u; }
            else if (n1 < 256u) { enc2::s8(out, o, 60u << 2); enc2::s8(out, o + 1u, n1); o += 2u; }
            else { enc2::s8(out, o, 61u << 2); enc2::s8(out, o + 1u, n1); enc2::s8(out, o + 2u, n1 >> 8); o += 3u; }
            lit_at = o;
            if (lit < enc2::kLaneLit) enc2::lane_copy(out, o, in, lit0, lit);
            o += lit;
        }
        uint32_t len = code + 4u;
        while (len >= 68u) { enc2::s8(out, o, 2u | (63u << 2)); enc2::s8(out, o + 1u, off); enc2::s8(out, o + 2u, off >> 8); o += 3u; len -= 64u; }
        if (len > 64u) { enc2::s8(out, o, 2u | (59u << 2)); enc2::s8(out, o + 1u, off); enc2::s8(out, o + 2u, off >> 8); o += 3u; len -= 60u; }
        if (len < 12u && off < 2048u) { enc2::s8(out, o, 1u | ((len - 4u) << 2) | ((off >> 8) << 5)); enc2::s8(out, o + 1u, off); }
        else { enc2::s8(out, o, 2u | ((len - 1u) << 2)); enc2::s8(out, o + 1u, off); enc2::s8(out, o + 2u, off >> 8); }
        return lit_at;
    }
    static __device__ __forceinline__ uint32_t emit_wave(enc2::gcptr gin, enc2::gptr gout, uint32_t op, uint32_t lit0, uint32_t lit, uint32_t off, uint32_t mlen) {
        const uint8_t* in = (const uint8_t*)gin; uint8_t* out = (uint8_t*)gout;
        if (lit) op = emit_snappy_literal(out, op, in + lit0, lit);
        return emit_snappy_copy(out, op, off, mlen);
    }
};

// kSplit: the chunks are consecutive sub-pieces of 64 KiB pieces, each wavefront with its own pre-indexed hash table — see lz4_encode.hip
template <bool kSplit, int kW>
__device__ __forceinline__ void snappy_encode_chunk(const BatchArgs& a, uint32_t chunk, const HashTab& ht, uint32_t* scr, uint32_t wave) {
    const uint64_t base_off = kSplit ? a.in_off[chunk & ~(split_per(a.flags) - 1u)] : a.in_off[chunk];      // the piece's first sub-piece
    const uint8_t* in = a.in_base + base_off;               // position 0 = start of the piece
    const uint32_t q0 = (uint32_t)(a.in_off[chunk] - base_off);      // this wave's range = [q0, n)
    const uint64_t n64 = q0 + a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    const uint64_t cap64 = a.out_cap[chunk];
    const uint32_t lane = lane_id();

    // snap: TooBig above u32::MAX (we also keep positions in 32 bits); BufferTooSmall below max_compress_len
    const bool first = wave == 0u;              // the chunk's first wavefront writes everything outside the rounds
    if (n64 > 0xFFFFFFFFull - 64u) { if (first && lane == 0) a.result[chunk] = CJ_E_SNAPPY_TOO_BIG; return; }
    const uint64_t need = 32u + (n64 - q0) + (n64 - q0) / 6u;
    if (need > 0xFFFFFFFFull) { if (first && lane == 0) a.result[chunk] = CJ_E_SNAPPY_TOO_BIG; return; }
    if (cap64 < need) { if (first && lane == 0) a.result[chunk] = CJ_E_SNAPPY_BUF_SMALL; return; }
    const uint32_t n = (uint32_t)n64;

    uint32_t op = 0;
    {   // varint preamble
        uint32_t v = n - q0;
        while (v >= 0x80u) { if (first && lane == 0) out[op] = (uint8_t)(v | 0x80u); op += 1; v >>= 7; }
        if (first && lane == 0) out[op] = (uint8_t)v;
        op += 1;
    }
    uint32_t anchor = q0;
    if (n - q0 >= 8u) {
        ht.clear(threadIdx.x, 64u * kW);
        if constexpr (kSplit) ht.preindex(in, q0);
        ht.settle();
        enc2::Walk<SnappyFmt, kW> w{enc2::uniform_gptr(in), (enc2::gptr)enc2::uniform_gptr(out), n, SnappyFmt::last_start(n), SnappyFmt::limit(n), scr, ht, op, 0u, wave};
        anchor = w.run(q0);
        op = w.op;
    }
    if (!first) return;
    if (anchor < n) op = emit_snappy_literal(out, op, in + anchor, n - anchor);
    if (lane == 0) a.result[chunk] = (int64_t)op;
}

template <bool kSplit, int kW>
__global__ __launch_bounds__(64 * kW) void snappy_encode_kernel(BatchArgs a) {
    __shared__ uint16_t ht_lds[kHashSize];
    __shared__ uint32_t scr[enc2::Walk<SnappyFmt, kW>::kWords];
    const uint32_t chunk = blockIdx.x;
    if (chunk >= a.n_chunks) return;
    snappy_encode_chunk<kSplit, kW>(a, chunk, HashTab{ht_lds}, scr, uni(threadIdx.x >> 6));
}

// Two wavefronts per chunk (cj_enc2.hpp): a CU's LDS holds nine tables, a wavefront issues one instruction per ~5 cycles — two per
// table give the CU eighteen instruction streams.  One workgroup per chunk, launched plainly: the hardware dispatches the next
// workgroup when one finishes (a persistent grid with a chunk counter measured 152 GB/s against 170: its loop costs the kernel its
// register budget).  The sub-pieces of split pieces (large.hip: 4 / 16 KiB each, pre-indexed tables) take one wavefront.
constexpr int kEncWaves = 2;

// One workgroup per chunk, in grids of at most kEncGrid chunks: HIP refuses a grid of 2^32 threads or more (33.5 M chunks of 128), and the
// batch API takes up to 0xFFFFFFF0.  Returns hipSuccess or the launch's error (round-5 advisor: the launch result was never looked at).
hipError_t launch_snappy_encode(const BatchArgs& a, hipStream_t s) {
    constexpr uint32_t kEncGrid = 1u << 24;
    for (uint32_t start = 0; start < a.n_chunks; start += kEncGrid) {
        BatchArgs b = a;
        b.in_off += start; b.in_len += start; b.out_off += start; b.out_cap += start; b.result += start;
        b.n_chunks = a.n_chunks - start < kEncGrid ? a.n_chunks - start : kEncGrid;
        // (sub-pieces of a split piece are indexed from their piece's first one: a slice starts on a piece boundary — kEncGrid is a multiple of every split)
        if (a.flags & kFlagSplitPieces) hipLaunchKernelGGL((snappy_encode_kernel<true, 1>), dim3(b.n_chunks), dim3(64), 0, s, b);
        else hipLaunchKernelGGL((snappy_encode_kernel<false, kEncWaves>), dim3(b.n_chunks), dim3(64 * kEncWaves), 0, s, b);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

}  // namespace cj
