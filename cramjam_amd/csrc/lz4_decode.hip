// lz4_decode.hip — LZ4 *block* decoder for gfx950, one wavefront per independent chunk.
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/lz4.rs:88,90,164,168:
// libcramjam::lz4::block::decompress_into -> lz4 crate -> LZ4_decompress_safe.  Accept/reject rules
// follow the liblz4 1.10.0 safe decoder (end-of-block parsing restrictions relative to the decode
// capacity, variable-length field limits); offset 0 is rejected (spec-invalid).
//
// Shape: the sequence grammar is parsed wave-uniformly on the scalar unit out of a 512-byte register
// window (v_readlane), the vector lanes only move bytes: literals HBM->HBM, matches from the
// chunk's own freshly written output (same-wave L1-coherent), byte per lane, 16 B/lane on long runs.
#include "lz4_lane_walk.hpp"
#include "snappy_records.hpp"

namespace cj {

// Decode one block (wave-uniform arguments).  hist: bytes of valid output directly before `out` that matches may reach
// into (0 for an independent block; min(position, 64 KiB) for a linked LZ4-frame block).  Returns the decoded size or
// CJ_E_CORRUPT.
__device__ __forceinline__ int64_t lz4_wave_decode(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap, uint32_t hist) {
    if (cap == 0) return (n == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT;
    if (n == 0) return CJ_E_CORRUPT;

    InWindow w;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u);
    w.base = in - mis;
    w.iend = mis + n;
    w.anchor(mis);
    const uint32_t iend = w.iend;
    uint32_t ip = mis;      // input position relative to w.base
    uint32_t op = 0;        // output position
    bool bad = false;

    for (;;) {
        w.ensure(ip);
        const uint32_t t4 = w.fetch32(ip);
        const uint32_t token = t4 & 0xffu;
        ip += 1;
        uint64_t lit = token >> 4;
        if (lit == 15u) {
            // variable-length literal count: bytes may not reach into the last 15 input bytes
            if (ip + 15u >= iend) { bad = true; break; }
            uint32_t b = (t4 >> 8) & 0xffu;
            ip += 1; lit += b;
            if (ip + 15u > iend) { bad = true; break; }
            while (b == 255u) {
                b = w.fetch32_any(ip) & 0xffu;
                ip += 1; lit += b;
                if (ip + 15u > iend) { bad = true; break; }
            }
            if (bad) break;
        }
        const uint32_t rem_out = cap - op, rem_in = iend - ip;
        if ((uint64_t)rem_out < lit + 12u || (uint64_t)rem_in < lit + 8u) {
            // must be the final sequence: consumes the input exactly, fits the output
            if (rem_in != lit || rem_out < lit) { bad = true; break; }
            wave_copy(out + op, w.base + ip, (uint32_t)lit);
            op += (uint32_t)lit;
            break;
        }
        wave_copy(out + op, w.base + ip, (uint32_t)lit);
        ip += (uint32_t)lit; op += (uint32_t)lit;

        const uint32_t o4 = w.fetch32_any(ip);
        const uint32_t offset = o4 & 0xffffu;
        ip += 2;
        uint64_t mlen = token & 15u;
        if (mlen == 15u) {
            uint32_t b = (o4 >> 16) & 0xffu;
            ip += 1; mlen += b;
            if (ip + 4u > iend) { bad = true; break; }
            while (b == 255u) {
                b = w.fetch32_any(ip) & 0xffu;
                ip += 1; mlen += b;
                if (ip + 4u > iend) { bad = true; break; }
            }
            if (bad) break;
        }
        mlen += 4u;
        if (offset == 0u || (uint64_t)offset > (uint64_t)op + hist) { bad = true; break; }
        if ((uint64_t)(cap - op) < mlen + 5u) { bad = true; break; }   // last 5 bytes are literals
        wave_order();
        wave_match_copy(out + op, offset, (uint32_t)mlen);
        wave_order();
        op += (uint32_t)mlen;
    }
    return bad ? (int64_t)CJ_E_CORRUPT : (int64_t)op;
}

// route: nullptr = decode every chunk; else only chunks the parse stage flagged kRouteWave.
__global__ __launch_bounds__(kBlockThreads) void lz4_decode_kernel(BatchArgs a, const ParseMeta* route) {
    const uint32_t chunk = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (chunk >= a.n_chunks) return;
    if (route != nullptr && (route[chunk].in_skip & kRouteWave) == 0u) return;
    const uint8_t* in = a.in_base + a.in_off[chunk];
    uint64_t n64 = a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    uint64_t cap64 = a.out_cap[chunk];
    int64_t status = 0;

    if (a.flags & CJ_FLAG_LZ4_SIZE_PREFIX) {
        // lz4 crate decompress_to_buffer(src, None, buffer): u32-LE size prefix (reference src/lz4.rs:90,164)
        if (n64 < 4) { status = CJ_E_NO_PREFIX; }
        else {
            int32_t size = (int32_t)((uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24));
            if (size < 0) status = CJ_E_NEG_PREFIX;
            else if ((uint32_t)size > 0x7E000000u) status = CJ_E_PREFIX_TOO_BIG;
            else if ((uint64_t)size > cap64) status = CJ_E_OUT_TOO_SMALL;
            else { in += 4; n64 -= 4; cap64 = (uint64_t)size; }
        }
    } else {
        // capacity is handed to liblz4 as an i32
        int32_t size = (int32_t)(uint32_t)cap64;
        if (cap64 > 0xFFFFFFFFull || size < 0) status = CJ_E_NEG_PREFIX;
        else if ((uint32_t)size > 0x7E000000u) status = CJ_E_PREFIX_TOO_BIG;
    }
    if (status == 0 && n64 > 0x7FFFFFF0ull) status = CJ_E_CORRUPT;
    if (status != 0) { if (lane_id() == 0) a.result[chunk] = status; return; }

    const int64_t r = lz4_wave_decode(in, (uint32_t)n64, out, (uint32_t)cap64, 0u);
    if (lane_id() == 0) a.result[chunk] = r;
}

// ---------------------------------------------------------------------------------------------------
// LZ4 FRAME with LINKED blocks (frame.hip): block k may copy from the previous 64 KiB of output, so the blocks of a
// frame form a chain.  One wavefront walks them in order and decodes straight into the contiguous output; stored
// blocks are copied.  word[k] = block size | bit 31 (stored).  result[k] = decoded size or CJ_E_CORRUPT; the walk stops
// at the first bad block.  (Slow by construction — one wave, one dependency chain; frames with independent blocks take
// the batch path instead.)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lz4_frame_chain_kernel(const uint8_t* in, const uint64_t* blk_off, const uint32_t* word,
                                                             uint32_t nblk, uint8_t* out, uint64_t out_cap, uint32_t block_max,
                                                             int64_t* result) {
    uint64_t pos = 0;
    for (uint32_t k = 0; k < nblk; k++) {
        const uint32_t wd = uni(word[k]);
        const uint32_t sz = wd & 0x7FFFFFFFu;
        const uint8_t* src = in + blk_off[k];
        const uint64_t room = out_cap - pos;
        int64_t r;
        if (wd & 0x80000000u) {
            if (sz > room) r = CJ_E_CORRUPT;
            else { wave_copy(out + pos, src, sz); r = (int64_t)sz; }
        } else {
            const uint32_t cap = room < block_max ? (uint32_t)room : block_max;
            r = lz4_wave_decode(src, sz, out + pos, cap, pos < 65536u ? (uint32_t)pos : 65536u);
        }
        if (lane_id() == 0) result[k] = r;
        if (r < 0) {
            for (uint32_t j = k + 1 + lane_id(); j < nblk; j += 64u) result[j] = CJ_E_CORRUPT;
            return;
        }
        pos += (uint64_t)r;
        wave_order();
    }
}

void launch_lz4_frame_chain(const uint8_t* in, const uint64_t* blk_off, const uint32_t* word, uint32_t nblk, uint8_t* out,
                            uint64_t out_cap, uint32_t block_max, int64_t* result, hipStream_t s) {
    if (nblk == 0) return;
    hipLaunchKernelGGL(lz4_frame_chain_kernel, dim3(1), dim3(64), 0, s, in, blk_off, word, nblk, out, out_cap, block_max, result);
}

void launch_lz4_decode(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlockThreads);
    hipLaunchKernelGGL(lz4_decode_kernel, grid, block, 0, s, a, (const ParseMeta*)nullptr);
}

void launch_lz4_decode_routed(const BatchArgs& a, const void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlockThreads);
    hipLaunchKernelGGL(lz4_decode_kernel, grid, block, 0, s, a, (const ParseMeta*)meta);
}


}  // namespace cj
