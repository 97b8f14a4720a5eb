// lz4_lane_walk.hpp — the per-lane LZ4 block walker shared by
//   * lz4_decode_lanes_kernel  (kCopy = true : one lane decodes one whole chunk), and
//   * lz4_parse_kernel         (kCopy = false: one lane only walks + validates the token chain and
//                               records an (ip, op) sync point every kSyncEvery sequences for the
//                               workgroup-per-chunk LDS decoder, lz4_decode_lds.hip).
// Accept/reject rules: liblz4 1.10.0 LZ4_decompress_safe (reference src/lz4.rs:88,164,168 -> lz4 crate),
// offset 0 rejected.
#pragma once
#include "cj_common.hpp"

namespace cj {
#if defined(__HIPCC__)

constexpr uint32_t kSyncEvery = 8;        // sequences per sync point
constexpr uint32_t kSyncPitch = 2048 + 16;      // distance (entries) between two chunks' sync points: 129 x 128 bytes, not a power of two
                                              // (the 64 lanes of a parse wave would otherwise store into the same memory channel)
constexpr uint32_t kSyncStride = 2048;    // sync points reserved per chunk (=> at most 16 384 sequences on the LDS path: a 64 KiB chunk of LZ4 has at most
                                          // 16 384; real text has ~10 000, which the 1 024 points of rounds 1-2 sent to the one-wavefront kernel)
#ifndef CJ_L2_WINDOW
#define CJ_L2_WINDOW 65536                // (tuning variants only: a smaller window for experiments with more workgroups per CU)
#endif
constexpr uint32_t kLdsOutMax = CJ_L2_WINDOW;    // the LDS decoder holds at most this much output ...
constexpr uint32_t kLdsInMax = CJ_L2_WINDOW - 32;     // ... and this much compressed input: variant 2 stages it in the 64 KiB output window (<= 15 B misalignment + 15 B round-up)

// The window of the workgroup decoder for a batch (CJ_FLAG_CHUNKS_LE_32K / _16K, set by the engine for the parse + decode pipeline only):
// what the parse kernels route by — capacity and compressed size the window holds, the sequence counts its record table is sized for
// (an LZ4 sequence with a match covers four bytes) and below which a chunk of few long runs is quicker on one wavefront.
__host__ __device__ inline uint32_t lds_window(uint32_t flags) { return (flags & CJ_FLAG_CHUNKS_LE_16K) ? 16384u : (flags & CJ_FLAG_CHUNKS_LE_32K) ? 32768u : kLdsOutMax; }
__host__ __device__ constexpr uint32_t lds_window_max_seq(uint32_t win) { return win >= 65536u ? kSyncStride * kSyncEvery : win / 4u; }
__host__ __device__ constexpr uint32_t lds_window_min_seq(uint32_t win) { return win >= 65536u ? 256u : win / 256u; }

struct ParseMeta {       // one per chunk, written by lz4_parse_kernel
    uint32_t nseq;       // sequences incl. the final literal-only one; 0 = nothing left for the LDS decoder
    uint32_t in_skip;    // low bits: 4 when a size prefix was consumed; kRouteWave: decode with the wave-per-chunk kernel
};
constexpr uint32_t kRouteWave = 0x80000000u;
constexpr uint32_t kRouteLane = 0x40000000u;   // decoded by the lane-per-chunk kernel on the auxiliary stream (listed in lane_list)
constexpr uint32_t kLaneShareDen = 20;         // lane_share/20 of the LDS-class chunks go to the lane kernel
// Routing (measured, tools/mode_sweep.py): chunks made of few long runs (RLE, incompressible data) decode at
// 2.2-2.6 TB/s with the wave-per-chunk kernel (16 B/lane cooperative copies) but crawl through the LDS path;
// chunks with many short sequences are ~2x faster through parse + LDS.  The parse kernel knows the count.
constexpr uint32_t kLdsMinSeq = 256;
constexpr uint32_t kRouteStored = 0x20000000u;   // linked-frame parse: the block is stored uncompressed (copied, not decoded)

// Sync points of the lane-per-chunk parse kernels, gathered four at a time in registers and stored as ONE aligned 32-byte group:
// a wave's 64 lanes belong to 64 chunks, so every 8-byte store used to dirty a 32-byte sector of its own (PMC: 1.86 GB written per
// 100 k chunks for 0.27 GB of sync points).  A chunk's region starts 128-byte aligned (kSyncPitch) and holds a multiple of 4 slots.
struct SyncBatch {
    uint32_t x0, y0, x1, y1, x2, y2, x3, y3;
    __device__ __forceinline__ void put(uint2* csync, uint32_t slot, uint2 v) {
        // a shift register: no slot selection (put runs in nearly every step of a wave: some lane is at a multiple of 8)
        x0 = x1; y0 = y1; x1 = x2; y1 = y2; x2 = x3; y2 = y3; x3 = v.x; y3 = v.y;
        if ((slot & 3u) == 3u && slot < kSyncStride) {
            uint4* g = reinterpret_cast<uint4*>(csync + (slot - 3u));
            g[0] = make_uint4(x0, y0, x1, y1);
            g[1] = make_uint4(x2, y2, x3, y3);
        }
    }
    // n = sync points put so far: the group that was not completed (its k newest entries sit at the register's end: 3 | 2 3 | 1 2 3)
    __device__ __forceinline__ void flush(uint2* csync, uint32_t n) {
        const uint32_t k = n & 3u, base = n & ~3u;
        if (base >= kSyncStride) return;
        // (values first, then the selects: a conditional between MEMBERS becomes a select of addresses, and a batch that lives inside a
        //  larger object is then never split into registers — seen as 552 bytes of scratch in the dual parse kernel)
        const uint32_t a1 = x1, b1 = y1, a2 = x2, b2 = y2, a3 = x3, b3 = y3;
        if (k >= 1u) csync[base] = make_uint2(k == 1u ? a3 : k == 2u ? a2 : a1, k == 1u ? b3 : k == 2u ? b2 : b1);
        if (k >= 2u) csync[base + 1u] = make_uint2(k == 2u ? a3 : a2, k == 2u ? b3 : b2);
        if (k >= 3u) csync[base + 2u] = make_uint2(a3, b3);
    }
};

__device__ __forceinline__ uint4 ld16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16u(uint8_t* p, const uint4& v) { __builtin_memcpy(p, &v, 16); }
#if defined(CJ_HOST_SIM)
__device__ __forceinline__ uint4 ld16u_nt(const uint8_t* p) { return ld16u(p); }
__device__ __forceinline__ void st16u_nt(uint8_t* p, const uint4& v) { st16u(p, v); }
__device__ __forceinline__ void st16u_wt(uint8_t* p, const uint4& v) { st16u(p, v); }
#else
// Non-temporal 16 B load for the match sources: those reads land anywhere in the last 64 KiB of the chunk's output
// and are never reused, but through the normal path they evict the partially written output lines of every lane
// from L2 before they fill (PMC: 30.8 GB of HBM writes for 6.5 GB of output).
typedef uint32_t cj_u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint4 ld16u_nt(const uint8_t* p) {
    const cj_u32x4_unaligned v = __builtin_nontemporal_load(reinterpret_cast<const cj_u32x4_unaligned*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// Non-temporal 16 B store: finished output that this kernel never reads back (keeps L2 for data that is reused)
__device__ __forceinline__ void st16u_nt(uint8_t* p, const uint4& v) {
    cj_u32x4_unaligned t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<cj_u32x4_unaligned*>(p));
}
// Write-through 16 B store (sc0 sc1): the bytes are in memory once the store is acknowledged (s_waitcnt vmcnt(0)), visible to
// the other XCDs without an L2 write-back (buffer_wbl2 costs several µs with 64 KiB freshly written).  Used where another
// workgroup waits for exactly these bytes (slab mode of the LDS decoder).
typedef uint32_t cj_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16u_wt(uint8_t* p, const uint4& v) {
    cj_u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(t) : "memory");
}
#endif

// up to 4 bytes at in[ip..], zero-filled past iend
__device__ __forceinline__ uint32_t ld_le_tail(const uint8_t* in, uint32_t ip, uint32_t iend) {
    if (ip + 4u <= iend) return ld32u(in + ip);
    uint32_t v = 0;
    for (uint32_t i = 0; i < 4u && ip + i < iend; i++) v |= (uint32_t)in[ip + i] << (8u * i);
    return v;
}

// dst[0..n) = src[0..n), non-overlapping; room_* = bytes that may be touched from dst/src
__device__ __forceinline__ void lane_copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t room_dst, uint32_t room_src) {
    uint32_t k = 0;
    const uint32_t room = room_dst < room_src ? room_dst : room_src;
    for (; k < n && k + 16u <= room; k += 16u) st16u(dst + k, ld16u(src + k));
    for (; k < n; k++) dst[k] = src[k];
}

// 16-byte vector whose byte i is pat[i % d], for 1 <= d < 16
__device__ __forceinline__ uint4 splat_pattern(const uint8_t* pat, uint32_t d) {
    uint32_t w[4];
    if (d == 1u) {
        uint32_t b = pat[0] * 0x01010101u;
        w[0] = w[1] = w[2] = w[3] = b;
    } else if (d == 2u) {
        uint32_t h = (uint32_t)pat[0] | ((uint32_t)pat[1] << 8);
        h |= h << 16;
        w[0] = w[1] = w[2] = w[3] = h;
    } else if (d == 4u) {
        uint32_t v = ld32u(pat);
        w[0] = w[1] = w[2] = w[3] = v;
    } else if (d == 8u) {
        w[0] = w[2] = ld32u(pat);
        w[1] = w[3] = ld32u(pat + 4);
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        uint32_t r = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16u; i++) {
            w[i >> 2] |= (uint32_t)pat[r] << (8u * (i & 3u));
            r += 1u;
            if (r == d) r = 0;
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// dst[j] = dst[j - d], j in [0, m); room = bytes that may be touched from dst
__device__ __forceinline__ void lane_match(uint8_t* dst, uint32_t d, uint32_t m, uint32_t room) {
    uint32_t k = 0;
    const uint8_t* src = dst - d;
    if (d >= 16u) {
        for (; k < m && k + 16u <= room; k += 16u) st16u(dst + k, ld16u_nt(src + k));
    } else if (m >= 16u && room >= 32u) {
        const uint4 p = splat_pattern(src, d);
        const uint32_t s = (16u / d) * d;          // advance by whole periods so the phase stays aligned
        for (; k < m && k + 16u <= room; k += s) st16u(dst + k, p);
        if (k > m) k = m;
    }
    for (; k < m; k++) dst[k] = src[k];      // src[k] == dst[k - d]; the pointer form avoids u32 wrap
}

// Walks one LZ4 block.  Returns decoded size or CJ_E_CORRUPT.  kCopy=false touches only `in`.
// sync: uint2 slots (ip, op) written every kSyncEvery sequences (may be null when kCopy);
// nseq_out: number of sequences walked (incl. the final literal-only one).
template <bool kCopy>
__device__ __forceinline__ int64_t lz4_lane_walk(const uint8_t* in, uint32_t iend, uint8_t* out, uint32_t cap,
                                                 uint2* sync, uint32_t sync_cap, uint32_t* nseq_out) {
    uint32_t ip = 0, op = 0, nseq = 0;
    for (;;) {
        if (!kCopy) {
            if ((nseq % kSyncEvery) == 0u) {
                const uint32_t slot = nseq / kSyncEvery;
                if (slot < sync_cap) sync[slot] = make_uint2(ip, op);
            }
        }
        nseq += 1;
        const uint32_t t4 = ld_le_tail(in, ip, iend);
        const uint32_t token = t4 & 0xffu;
        ip += 1;
        uint64_t lit = token >> 4;
        if (lit == 15u) {
            if (ip + 15u >= iend) return CJ_E_CORRUPT;
            uint32_t b = (t4 >> 8) & 0xffu;
            ip += 1; lit += b;
            if (ip + 15u > iend) return CJ_E_CORRUPT;
            while (b == 255u) {
                b = in[ip];
                ip += 1; lit += b;
                if (ip + 15u > iend) return CJ_E_CORRUPT;
            }
        }
        const uint32_t rem_out = cap - op, rem_in = iend - ip;
        if ((uint64_t)rem_out < lit + 12u || (uint64_t)rem_in < lit + 8u) {
            if ((uint64_t)rem_in != lit || (uint64_t)rem_out < lit) return CJ_E_CORRUPT;
            if (kCopy) lane_copy(out + op, in + ip, (uint32_t)lit, (uint32_t)lit, (uint32_t)lit);   // exact tail
            op += (uint32_t)lit;
            break;
        }
        if (kCopy) lane_copy(out + op, in + ip, (uint32_t)lit, rem_out, rem_in);
        ip += (uint32_t)lit; op += (uint32_t)lit;

        const uint32_t o4 = ld_le_tail(in, ip, iend);       // >= 8 input bytes remain here
        const uint32_t offset = o4 & 0xffffu;
        ip += 2;
        uint64_t mlen = token & 15u;
        if (mlen == 15u) {
            uint32_t b = (o4 >> 16) & 0xffu;
            ip += 1; mlen += b;
            if (ip + 4u > iend) return CJ_E_CORRUPT;
            while (b == 255u) {
                b = in[ip];
                ip += 1; mlen += b;
                if (ip + 4u > iend) return CJ_E_CORRUPT;
            }
        }
        mlen += 4u;
        if (offset == 0u || offset > op) return CJ_E_CORRUPT;
        if ((uint64_t)(cap - op) < mlen + 5u) return CJ_E_CORRUPT;
        if (kCopy) lane_match(out + op, offset, (uint32_t)mlen, cap - op);
        op += (uint32_t)mlen;
    }
    if (nseq_out) *nseq_out = nseq;
    return (int64_t)op;
}

// shared prologue: size-prefix / capacity rules (lz4 crate decompress_to_buffer). Returns 0 or CJ_E_*;
// on success in/n/cap describe the raw block.
__device__ __forceinline__ int64_t lz4_block_prologue(uint32_t flags, const uint8_t*& in, uint64_t& n64, uint64_t& cap64) {
    if (flags & CJ_FLAG_LZ4_SIZE_PREFIX) {
        if (n64 < 4) return CJ_E_NO_PREFIX;
        int32_t size = (int32_t)ld32u(in);
        if (size < 0) return CJ_E_NEG_PREFIX;
        if ((uint32_t)size > 0x7E000000u) return CJ_E_PREFIX_TOO_BIG;
        if ((uint64_t)size > cap64) return CJ_E_OUT_TOO_SMALL;
        in += 4; n64 -= 4; cap64 = (uint64_t)size;
    } else {
        int32_t size = (int32_t)(uint32_t)cap64;
        if (cap64 > 0xFFFFFFFFull || size < 0) return CJ_E_NEG_PREFIX;
        if ((uint32_t)size > 0x7E000000u) return CJ_E_PREFIX_TOO_BIG;
    }
    if (n64 > 0x7FFFFFF0ull) return CJ_E_CORRUPT;
    return 0;
}

#endif
}  // namespace cj
