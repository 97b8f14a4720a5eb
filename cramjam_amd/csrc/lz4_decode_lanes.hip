// lz4_decode_lanes.hip — LZ4 *block* decoding with ONE LANE per independent chunk, and the parse-only
// variant that feeds the workgroup-per-chunk LDS decoder.
//
// Same contract and accept/reject rules as lz4_decode.hip (reference call sites
// /root/reference/src/lz4.rs:88,90,164,168 -> LZ4_decompress_safe), different mapping: the
// wave-per-chunk kernel is bound by the serial token chain of each chunk (one dependent memory round
// trip per copy, ~3.7k cycles per sequence measured).  Here every lane walks its own chunk, so a
// wavefront retires 64 sequences per step; the price is uncoalesced 16 B accesses.
//   lz4_decode_lanes_kernel : full decode, 16 B "wild" copies with exact tails
//   lz4_parse_kernel        : walk + validate only (reads nothing but the compressed stream), emits
//                             (ip, op) sync points every 8 sequences + the decoded size; chunks the LDS
//                             decoder cannot take (capacity > 64 KiB, > 8192 sequences) are decoded here.
#include "lz4_lane_walk.hpp"

namespace cj {

__global__ __launch_bounds__(64) void lz4_decode_lanes_kernel(BatchArgs a) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= a.n_chunks) return;
    const uint8_t* in = a.in_base + a.in_off[c];
    uint64_t n64 = a.in_len[c];
    uint8_t* out = a.out_base + a.out_off[c];
    uint64_t cap64 = a.out_cap[c];
    const int64_t status = lz4_block_prologue(a.flags, in, n64, cap64);
    if (status != 0) { a.result[c] = status; return; }
    const uint32_t cap = (uint32_t)cap64, iend = (uint32_t)n64;
    if (cap == 0) { a.result[c] = (iend == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT; return; }
    if (iend == 0) { a.result[c] = CJ_E_CORRUPT; return; }
    a.result[c] = lz4_lane_walk<true>(in, iend, out, cap, nullptr, 0, nullptr);
}

__global__ __launch_bounds__(64) void lz4_parse_kernel(BatchArgs a, uint2* sync, ParseMeta* meta) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= a.n_chunks) return;
    const uint8_t* in0 = a.in_base + a.in_off[c];
    const uint8_t* in = in0;
    uint64_t n64 = a.in_len[c];
    uint8_t* out = a.out_base + a.out_off[c];
    uint64_t cap64 = a.out_cap[c];
    ParseMeta pm = {0u, 0u};
    const int64_t status = lz4_block_prologue(a.flags, in, n64, cap64);
    if (status != 0) { a.result[c] = status; meta[c] = pm; return; }
    const uint32_t cap = (uint32_t)cap64, iend = (uint32_t)n64;
    int64_t r;
    if (cap == 0) r = (iend == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT;
    else if (iend == 0) r = CJ_E_CORRUPT;
    else if (cap > kLdsOutMax || iend > kLdsInMax) { r = 0; pm.in_skip = kRouteWave; }   // too big for LDS: wave-per-chunk kernel decodes and validates
    else {
        uint32_t nseq = 0;
        r = lz4_lane_walk<false>(in, iend, nullptr, cap, sync + (size_t)c * kSyncStride, kSyncStride, &nseq);
        if (r > 0) {
            if ((nseq + kSyncEvery - 1u) / kSyncEvery > kSyncStride) r = lz4_lane_walk<true>(in, iend, out, cap, nullptr, 0, nullptr);
            else if (nseq < kLdsMinSeq) pm.in_skip = kRouteWave;            // few, long sequences: wave-per-chunk kernel
            else { pm.nseq = nseq; pm.in_skip = (uint32_t)(in - in0); }
        }
    }
    a.result[c] = r;
    meta[c] = pm;
}

void launch_lz4_decode_lanes(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + 63u) / 64u), block(64);
    hipLaunchKernelGGL(lz4_decode_lanes_kernel, grid, block, 0, s, a);
}

void launch_lz4_parse(const BatchArgs& a, void* sync, void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + 63u) / 64u), block(64);
    hipLaunchKernelGGL(lz4_parse_kernel, grid, block, 0, s, a, (uint2*)sync, (ParseMeta*)meta);
}

}  // namespace cj
