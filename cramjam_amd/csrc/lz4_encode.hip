// lz4_encode.hip — LZ4 *block* encoder for gfx950, one wavefront per independent chunk.
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/lz4.rs:127,206:
// libcramjam::lz4::block::compress_into -> lz4 crate compress_to_buffer -> LZ4_compress_default,
// including the optional u32-LE uncompressed-size prefix (`store_size`, default on).
// Stream rules honoured (lz4_Block_format.md): min match 4, offset 1..65535, the last match starts
// at least 12 bytes before the end, the last 5 bytes are literals, inputs < 13 bytes are one literal
// run.  Every emitted block decodes with LZ4_decompress_safe at exact capacity (tests check this
// against the CPU oracle).
#include "cj_match.hpp"

namespace cj {

// writes the 255-run length extension for value v (v = len - 15): v/255 bytes of 255 then v%255
__device__ __forceinline__ uint32_t emit_len_ext(uint8_t* out, uint32_t op, uint32_t v) {
    const uint32_t full = v / 255u, rem = v - full * 255u, lane = lane_id();
    for (uint32_t k = 0; k < full; k += 64u)
        if (k + lane < full) out[op + k + lane] = 255u;
    if (lane == 0) out[op + full] = (uint8_t)rem;
    return op + full + 1u;
}

__global__ __launch_bounds__(kEncThreads) void lz4_encode_kernel(BatchArgs a) {
    __shared__ uint16_t ht_all[kEncWaves][kHashSize];
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t chunk = uni(blockIdx.x * kEncWaves + wave);
    if (chunk >= a.n_chunks) return;
    uint16_t* ht = ht_all[wave];
    const uint8_t* in = a.in_base + a.in_off[chunk];
    const uint64_t n64 = a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    uint64_t cap64 = a.out_cap[chunk];
    const uint32_t lane = lane_id();
    const bool prefix = (a.flags & CJ_FLAG_LZ4_SIZE_PREFIX) != 0;

    if (n64 > 0x7E000000ull) { if (lane == 0) a.result[chunk] = CJ_E_INPUT_TOO_LARGE; return; }
    const uint32_t n = (uint32_t)n64;
    // the engine only launches with capacity >= LZ4_compressBound(n) (+4); anything smaller is refused here
    const uint64_t need = (uint64_t)n + n / 255u + 16u + (prefix ? 4u : 0u);
    if (cap64 < need) { if (lane == 0) a.result[chunk] = CJ_E_COMPRESS_FAILED; return; }
    if (prefix) {
        if (lane < 4) out[lane] = (uint8_t)(n >> (8u * lane));
        out += 4;
    }

    uint32_t anchor = 0, op = 0;
    if (n >= 13u) {
        ht_clear(ht);
        const uint32_t last_start = n - 12u;    // a match may start here at the latest
        const uint32_t matchlimit = n - 5u;     // and must end here at the latest
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
                uint32_t mlen = 4u + wave_extend(in, mpos + 4u, mc + 4u, matchlimit);
                const uint32_t back = wave_extend_back(in, mpos, mc, mpos - anchor);
                mpos -= back; mc -= back; mlen += back;
                const uint32_t lit = mpos - anchor;
                const uint32_t mcode = mlen - 4u;
                // token
                if (lane == 0) out[op] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
                op += 1;
                if (lit >= 15u) op = emit_len_ext(out, op, lit - 15u);
                wave_copy(out + op, in + anchor, lit);
                op += lit;
                const uint32_t off = mpos - mc;
                if (lane < 2) out[op + lane] = (uint8_t)(off >> (8u * lane));
                op += 2;
                if (mcode >= 15u) op = emit_len_ext(out, op, mcode - 15u);
                anchor = mpos + mlen;
                covered |= covered_bits(pos, mpos, anchor - pos);
                if (anchor >= batch_end) mask = 0;
                else mask &= ~0ull << (anchor - pos);
            }
            insert_uncovered(ht, pos, hslot, covered);
            pos = anchor > batch_end ? anchor : batch_end;
        }
    }
    {   // last literals
        const uint32_t lit = n - anchor;
        if (lane == 0) out[op] = (uint8_t)((lit < 15u ? lit : 15u) << 4);
        op += 1;
        if (lit >= 15u) op = emit_len_ext(out, op, lit - 15u);
        wave_copy(out + op, in + anchor, lit);
        op += lit;
    }
    if (lane == 0) a.result[chunk] = (int64_t)op + (prefix ? 4 : 0);
}

void launch_lz4_encode(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + kEncWaves - 1) / kEncWaves), block(kEncThreads);
    hipLaunchKernelGGL(lz4_encode_kernel, grid, block, 0, s, a);
}

}  // namespace cj
