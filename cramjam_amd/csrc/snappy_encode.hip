// snappy_encode.hip — Snappy *raw* encoder for gfx950, one wavefront per independent chunk.
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/snappy.rs:75,97:
// libcramjam::snappy::raw::compress -> snap 1.1.1 raw::Encoder::compress.  Output = varint(len)
// preamble + literal / copy-1 / copy-2 elements (format_description.txt); copies longer than 64
// are split exactly like the CPU encoders do (64,64,...,[60],rest) so every element is canonical.
// The 64 KiB window of the match finder keeps every offset < 65536, so copy-4 is never emitted.
#include "cj_match.hpp"

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

__global__ __launch_bounds__(kEncThreads) void snappy_encode_kernel(BatchArgs a) {
    __shared__ uint16_t ht_all[kEncWaves][kHashSize];
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t chunk = uni(blockIdx.x * kEncWaves + wave);
    if (chunk >= a.n_chunks) return;
    uint16_t* ht = ht_all[wave];
    const uint8_t* in = a.in_base + a.in_off[chunk];
    const uint64_t n64 = a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    const uint64_t cap64 = a.out_cap[chunk];
    const uint32_t lane = lane_id();

    // snap: TooBig above u32::MAX (we also keep positions in 32 bits); BufferTooSmall below max_compress_len
    if (n64 > 0xFFFFFFFFull - 64u) { if (lane == 0) a.result[chunk] = CJ_E_SNAPPY_TOO_BIG; return; }
    const uint64_t need = 32u + n64 + n64 / 6u;
    if (need > 0xFFFFFFFFull) { if (lane == 0) a.result[chunk] = CJ_E_SNAPPY_TOO_BIG; return; }
    if (cap64 < need) { if (lane == 0) a.result[chunk] = CJ_E_SNAPPY_BUF_SMALL; return; }
    const uint32_t n = (uint32_t)n64;

    uint32_t op = 0;
    {   // varint preamble
        uint32_t v = n;
        while (v >= 0x80u) { if (lane == 0) out[op] = (uint8_t)(v | 0x80u); op += 1; v >>= 7; }
        if (lane == 0) out[op] = (uint8_t)v;
        op += 1;
    }
    uint32_t anchor = 0;
    if (n >= 8u) {
        ht_clear(ht);
        const uint32_t last_start = n - 4u;
        uint32_t pos = 0;
        while (pos <= last_start) {
            uint32_t cand, hslot;
            uint64_t mask = probe_round(in, ht, pos, last_start, cand, hslot);
            const uint32_t batch_end = pos + 64u;
            uint64_t covered = 0ull;
            while (mask) {
                const uint32_t first = ctz64(mask);
                uint32_t mpos = pos + first;
                uint32_t mc = rdlane(cand, first);
                uint32_t mlen = 4u + wave_extend(in, mpos + 4u, mc + 4u, n);
                const uint32_t back = wave_extend_back(in, mpos, mc, mpos - anchor);
                mpos -= back; mc -= back; mlen += back;
                if (mpos > anchor) op = emit_snappy_literal(out, op, in + anchor, mpos - anchor);
                op = emit_snappy_copy(out, op, mpos - mc, mlen);
                anchor = mpos + mlen;
                covered |= covered_bits(pos, mpos, anchor - pos);
                if (anchor >= batch_end) mask = 0;
                else mask &= ~0ull << (anchor - pos);
            }
            insert_uncovered(ht, pos, hslot, covered);
            pos = anchor > batch_end ? anchor : batch_end;
        }
    }
    if (anchor < n) op = emit_snappy_literal(out, op, in + anchor, n - anchor);
    if (lane == 0) a.result[chunk] = (int64_t)op;
}

void launch_snappy_encode(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + kEncWaves - 1) / kEncWaves), block(kEncThreads);
    hipLaunchKernelGGL(snappy_encode_kernel, grid, block, 0, s, a);
}

}  // namespace cj
