// lds_shared.hpp — device helpers shared by the two workgroup decoders (lz4_decode_lds.hip: bitmap resolver, slab and linked
// modes; lz4_decode_lvl.hip: level-ordered dense resolver): LDS accessors at any byte alignment, the tiered global/LDS -> LDS
// copies, the ready bitmap, and the PARSE STAGE INSIDE the decoder (fused_parse).
#pragma once
#include "lz4_lane_walk.hpp"
#include "snappy_records.hpp"
#include "parse_grammar.hpp"
#include <type_traits>

namespace cj {

constexpr uint32_t kLongRun = 512;                   // runs at least this long are copied by the whole wavefront
constexpr uint32_t kSpinLimit = 1u << 18;

// record: x = literal source (position in the compressed stream), y = literal length, z = match destination
//         (= op after literals), w = offset | match length << 16 (0 = no match: LZ4's final sequence, a Snappy literal)

// ---- unaligned LDS accessors ------------------------------------------------------------------
// gfx950 executes ds_read_b32/ds_write_b32 at any byte alignment (tools/lds_unaligned_probe.hip:
// bit-exact, ~140 vs ~80 cycles dependent latency).  hipcc will not emit them for align-1 LDS accesses
// (it splits into bytes), hence inline asm.  Every read carries its own s_waitcnt, so no result is
// consumed early; writes need no wait (DS ops of one wave execute in order).  Addresses are LDS byte
// offsets (low 32 bits of the flat address of a __shared__ object).
__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// 4 bytes at ANY byte address through two ALIGNED dwords + v_alignbyte: an unaligned ds_read is replayed once per active lane
// (~64 LDS cycles per wave-instruction with a full wave), this pair costs 4 when conflict-free.  D1 walks the token chain with it.
__device__ __forceinline__ uint32_t lds_ld32a(uint32_t a) {
    uint64_t v;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a & ~3u) : "memory");
    return __builtin_amdgcn_alignbyte((uint32_t)(v >> 32), (uint32_t)v, a & 3u);
}
// 8 bytes at any byte address as three aligned dwords (x = bytes 0..3, y = bytes 4..7); may read up to 4 bytes past them
__device__ __forceinline__ uint2 lds_ld64a(uint32_t a) {
    uint64_t v;
    uint32_t v2;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(v2) : "v"(a & ~3u) : "memory");
    return make_uint2(__builtin_amdgcn_alignbyte((uint32_t)(v >> 32), (uint32_t)v, a & 3u), __builtin_amdgcn_alignbyte(v2, (uint32_t)(v >> 32), a & 3u));
}
__device__ __forceinline__ uint2 lds_ld64(uint32_t a) {
    uint2 v;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v.x), "=&v"(v.y) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st32(uint32_t a, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_ld8(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st8(uint32_t a, uint32_t v) {
    asm volatile("ds_write_b8 %0, %1" :: "v"(a), "v"(v) : "memory");
}

// The whole wavefront copies n bytes (wave-uniform) from global memory to the LDS byte address a_dst: a long literal run, or a long
// copy from an earlier slab.  <= 3 head bytes until the destination is dword aligned, then lane l takes dwords l, l + 64, ... with
// eight loads in flight (2 KiB per round trip, conflict-free stores), then the tail bytes.  Reads exactly [src, src + n).
// (A byte per lane and round trip — the first version — took ~0.5 ms for a 64 KiB run: 360 k cycles per slab on data that is mostly
//  literals, tests/perf/slab_phase_profile.py.)
__device__ __forceinline__ void wave_copy_to_lds(uint32_t a_dst, const uint8_t* src, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h = (0u - a_dst) & 3u;
    h = h < n ? h : n;
    if (lane < h) lds_st8(a_dst + lane, src[lane]);
    const uint8_t* s = src + h;
    const uint32_t d = a_dst + h, m = n - h, nd = m >> 2;
    uint32_t i = lane;
    for (; i + 7u * 64u < nd; i += 8u * 64u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) v[j] = ld32u(s + 4u * (i + 64u * j));
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) lds_st32(d + 4u * (i + 64u * j), v[j]);
    }
    for (; i < nd; i += 64u) lds_st32(d + 4u * i, ld32u(s + 4u * i));
    const uint32_t t = m & 3u;
    if (lane < t) lds_st8(d + 4u * nd + lane, s[4u * nd + lane]);
}

// ---- single-wait tiered copy --------------------------------------------------------------------
// A misaligned DS access is replayed lane by lane (~64 LDS cycles per wave-instruction, measured), and
// every scattered DS wave-instruction costs ~8 cycles of the CU's LDS pipe, so the copies are built to
// need FEW instructions: aligned dword reads only (over-reading is harmless) + v_alignbyte to undo the
// source misalignment, ONE s_waitcnt per batch, and writes as <=3 head bytes + aligned dwords + <=3 tail
// bytes.  Writes must be exact: a lane whose element is past its length writes to a private dummy
// slot instead of being masked off (no exec-mask churn).  T = tier (max bytes per lane), wave-uniform.
template <int ND> struct DW { uint32_t w[ND]; };

__device__ __forceinline__ DW<6> lds_ld_aligned6(uint32_t a) {
    DW<6> r;
    asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:4\n\tds_read_b32 %2, %6 offset:8\n\tds_read_b32 %3, %6 offset:12\n\tds_read_b32 %4, %6 offset:16\n\tds_read_b32 %5, %6 offset:20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ DW<10> lds_ld_aligned10(uint32_t a) {
    DW<10> r;
    asm volatile("ds_read_b32 %0, %10\n\tds_read_b32 %1, %10 offset:4\n\tds_read_b32 %2, %10 offset:8\n\tds_read_b32 %3, %10 offset:12\n\tds_read_b32 %4, %10 offset:16\n\tds_read_b32 %5, %10 offset:20\n\tds_read_b32 %6, %10 offset:24\n\tds_read_b32 %7, %10 offset:28\n\tds_read_b32 %8, %10 offset:32\n\tds_read_b32 %9, %10 offset:36\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]), "=&v"(r.w[7]), "=&v"(r.w[8]), "=&v"(r.w[9]) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ DW<18> lds_ld_aligned18(uint32_t a) {
    DW<18> r;
    asm volatile("ds_read_b32 %0, %18\n\tds_read_b32 %1, %18 offset:4\n\tds_read_b32 %2, %18 offset:8\n\tds_read_b32 %3, %18 offset:12\n\tds_read_b32 %4, %18 offset:16\n\tds_read_b32 %5, %18 offset:20\n\tds_read_b32 %6, %18 offset:24\n\tds_read_b32 %7, %18 offset:28\n\tds_read_b32 %8, %18 offset:32\n\tds_read_b32 %9, %18 offset:36\n\tds_read_b32 %10, %18 offset:40\n\tds_read_b32 %11, %18 offset:44\n\tds_read_b32 %12, %18 offset:48\n\tds_read_b32 %13, %18 offset:52\n\tds_read_b32 %14, %18 offset:56\n\tds_read_b32 %15, %18 offset:60\n\tds_read_b32 %16, %18 offset:64\n\tds_read_b32 %17, %18 offset:68\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]), "=&v"(r.w[7]), "=&v"(r.w[8]), "=&v"(r.w[9]), "=&v"(r.w[10]), "=&v"(r.w[11]), "=&v"(r.w[12]), "=&v"(r.w[13]), "=&v"(r.w[14]), "=&v"(r.w[15]), "=&v"(r.w[16]), "=&v"(r.w[17]) : "v"(a) : "memory");
    return r;
}

struct Dummies { uint32_t b, w; };     // per-lane dummy byte address / aligned dummy dword address

template <int T, class Loaded>
__device__ __forceinline__ void lds_store_tier(const Loaded& r, uint32_t dst, uint32_t sh_bytes, uint32_t n, Dummies dm) {
    constexpr int NV = T / 4 + 1;
    uint32_t v[NV];                                    // v[i] = source bytes 4i .. 4i+3
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = __builtin_amdgcn_alignbyte(r.w[i + 1], r.w[i], sh_bytes);
    const uint32_t h = (0u - dst) & 3u;               // bytes until dst is dword aligned
    const uint32_t hh = h < n ? h : n;
    const uint32_t nm = (n - hh) >> 2, t = (n - hh) & 3u;
#pragma unroll
    for (int q = 0; q < 3; q++) lds_st8((uint32_t)q < hh ? dst + q : dm.b, v[0] >> (8 * q));
    uint32_t tv = 0;
#pragma unroll
    for (int i = 0; i < T / 4; i++) {
        const uint32_t mi = __builtin_amdgcn_alignbyte(v[i + 1], v[i], h);     // bytes h+4i .. h+4i+3
        lds_st32((uint32_t)i < nm ? dst + h + 4u * i : dm.w, mi);
        tv = (uint32_t)i == nm ? mi : tv;
    }
    const uint32_t tpos = dst + hh + 4u * nm;
#pragma unroll
    for (int q = 0; q < 3; q++) lds_st8((uint32_t)q < t ? tpos + q : dm.b, tv >> (8 * q));
}

// ---- D2 for a sequence that OWNS the bytes behind its literals --------------------------------------------------------
// A sequence whose match this lane writes later (D3, same wavefront, program order) may spill up to 3 bytes past the end of its
// literals: they land in its own match region, which no other lane reads before the bitmap says so.  That removes what
// lds_store_tier spends on the exact END: the <= 3 bytes in front of the first aligned address go out as one byte and one
// halfword store, the rest as aligned dwords, each predicated on holding at least one literal byte — no tail byte stores, no tail
// dword selection (14 -> 6 or 10 stores per batch; the LDS pipe is what bounds this kernel).  n <= T; n = 0: nothing is stored.
template <int OFF>
__device__ __forceinline__ void lds_st32_off(uint32_t a, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(OFF) : "memory");
}
// (predicated by the exec mask, not by a dummy address: the LDS pipe charges a scattered store by its ACTIVE lanes — 13.8 cycles
//  with 64, 7.7 with 32, 5.1 with 16 (tools/lds_throughput_probe.hip) — and the later dwords of a batch belong to few lanes)
template <int I, int N>
__device__ __forceinline__ void own_dwords(const uint32_t* v, uint32_t h, int32_t r, uint32_t ah) {
    if constexpr (I < N) {
        const uint32_t mi = __builtin_amdgcn_alignbyte(v[I + 1], v[I], h);
        if (r > 4 * I) lds_st32_off<4 * I>(ah, mi);
        own_dwords<I + 1, N>(v, h, r, ah);
    }
}
template <int T>
__device__ __forceinline__ void lds_store_own(const uint32_t* w, uint32_t a, uint32_t n) {      // w: source bytes 0 .. T-1 (T / 4 dwords)
    uint32_t v[T / 4 + 1];
#pragma unroll
    for (int i = 0; i < T / 4; i++) v[i] = w[i];
    v[T / 4] = 0u;
    const uint32_t h = (0u - a) & 3u;
    const int32_t r = (int32_t)n - (int32_t)h;              // literal bytes from the first aligned address on
    const uint32_t ah = a + h;
    // the <= 3 bytes in front of the first aligned address as one byte and one halfword store, both aligned (a misaligned dword
    // store of a full wave is replayed lane by lane: 64 cycles of the LDS pipe against 14 for each of these)
    if ((h & 1u) && n) asm volatile("ds_write_b8 %0, %1" :: "v"(a), "v"(v[0]) : "memory");
    if ((h & 2u) && n) asm volatile("ds_write_b16 %0, %1" :: "v"(a + (h & 1u)), "v"(v[0] >> (8u * (h & 1u))) : "memory");
    own_dwords<0, T / 4>(v, h, r, ah);
}
// the ready bits of [lo, lo + n), n <= 32: two words, no loop (n = 0: two ORs of nothing)
__device__ __forceinline__ void bits_set32(uint32_t* bits, uint32_t lo, uint32_t n) {
    const uint64_t m = ((1ull << n) - 1ull) << (lo & 31u);
    const uint32_t wa = (uint32_t)(uintptr_t)bits + ((lo >> 5) << 2);
    if (n) asm volatile("ds_or_b32 %0, %1" :: "v"(wa), "v"((uint32_t)m) : "memory");
    if ((uint32_t)(m >> 32)) asm volatile("ds_or_b32 %0, %1 offset:4" :: "v"(wa), "v"((uint32_t)(m >> 32)) : "memory");
}

// copy n (<= tier, tier in {16,32,64} wave-uniform) bytes src -> dst, both LDS byte addresses; [src, src+n) is
// final and does not overlap [dst, dst+n)
__device__ __forceinline__ void lds_copy_tier(uint32_t tier, uint32_t dst, uint32_t src, uint32_t n, Dummies dm) {
    const uint32_t sa = src & ~3u, sh = src & 3u;
    if (tier <= 16u) lds_store_tier<16>(lds_ld_aligned6(sa), dst, sh, n, dm);
    else if (tier <= 32u) lds_store_tier<32>(lds_ld_aligned10(sa), dst, sh, n, dm);
    else lds_store_tier<64>(lds_ld_aligned18(sa), dst, sh, n, dm);
}

// ---- sparse exact copy (D3) --------------------------------------------------------------------------------
// D3 copies with a handful of lanes active: the matches that became ready in this poll.  gfx950 executes LDS accesses at any
// byte alignment for about one extra cycle per ACTIVE misaligned lane (tools/lds_unaligned_probe.hip: +64 cycles with 64
// lanes, +16 with 16, +4 with 4), so there a copy of m <= 32 bytes is the first and the last 8 (16, 4) bytes of the match —
// overlapping in the middle — read and stored at their exact addresses: 2-4 reads and 2-4 stores instead of the 6-10 aligned
// dword reads, byte shifts and 10-14 head / dword / tail stores of lds_store_tier (which is built for 64 active lanes, D2).
// [src, src + m) is final and does not overlap [dst, dst + m); up to 7 bytes past the source may be read (never stored).
__device__ __forceinline__ void lds_copy_sparse(uint32_t dst, uint32_t src, uint32_t m) {
    if (m > 16u) {
        uint64_t r0, r1, r2, r3;
        const uint32_t s2 = src + m - 16u, d2 = dst + m - 16u;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %5\n\tds_read_b64 %3, %5 offset:8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(src), "v"(s2) : "memory");
        asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %0, %3 offset:8\n\tds_write_b64 %1, %4\n\tds_write_b64 %1, %5 offset:8"
                     :: "v"(dst), "v"(d2), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
    } else if (m >= 8u) {
        uint64_t r0, r1;
        const uint32_t s2 = src + m - 8u, d2 = dst + m - 8u;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(src), "v"(s2) : "memory");
        asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" :: "v"(dst), "v"(d2), "v"(r0), "v"(r1) : "memory");
    } else if (m >= 4u) {
        uint64_t r0;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(src) : "memory");
        const uint32_t lo = (uint32_t)r0, hi = (uint32_t)(r0 >> (8u * (m - 4u)));
        asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3" :: "v"(dst), "v"(dst + m - 4u), "v"(lo), "v"(hi) : "memory");
    } else {                                               // 1..3 bytes (Snappy copies)
        uint32_t r;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(src) : "memory");
        if (m & 2u) asm volatile("ds_write_b16 %0, %1" :: "v"(dst), "v"(r) : "memory");
        if (m & 1u) asm volatile("ds_write_b8 %0, %1" :: "v"(dst + (m & 2u)), "v"(r >> (8u * (m & 2u))) : "memory");
    }
}

// ---- dense exact copy (D3, many lanes ready at once) -----------------------------------------------------------
// m = 8 .. 32 bytes from LDS address src to dst, [src, src + m) final, no overlap, with ALIGNED accesses only: the pipe charges an
// aligned 8-byte access of n scattered lanes 3 - 9 cycles (16.5 for a full wave's store), a misaligned one n + 1 whatever its width
// (tools/lds_throughput_probe.hip).  Up to five aligned 8-byte reads around the source; the bytes in front of the destination's
// first 8-byte boundary as byte / halfword / dword pieces; whole aligned 8-byte words (source shifted by (src & 7) + (-dst & 7)
// bytes: whole dwords by two conditional register moves, the rest by v_alignbyte); the tail as dword / halfword / byte pieces.
// Nothing outside [dst, dst + m) is written, up to 7 bytes outside [src, src + m) are read.  Lanes with on = false do nothing.
__device__ __forceinline__ void lds_copy_dense(uint32_t src, uint32_t dst, uint32_t m, bool on) {
    const uint32_t sa = src & ~7u, so = src & 7u;
    uint32_t S[12];
#pragma unroll
    for (int i = 0; i < 12; i++) S[i] = 0u;
    const uint32_t span = so + m;                        // bytes from sa that matter (<= 39)
    if (on) {
        uint64_t r0, r1;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(r0), "=&v"(r1) : "v"(sa) : "memory");
        uint64_t r2 = 0, r3 = 0, r4 = 0;
        if (span > 16u) asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(r2) : "v"(sa) : "memory");
        if (span > 24u) asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(r3) : "v"(sa) : "memory");
        if (span > 32u) asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(r4) : "v"(sa) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) :: "memory");
        S[0] = (uint32_t)r0; S[1] = (uint32_t)(r0 >> 32); S[2] = (uint32_t)r1; S[3] = (uint32_t)(r1 >> 32); S[4] = (uint32_t)r2; S[5] = (uint32_t)(r2 >> 32);
        S[6] = (uint32_t)r3; S[7] = (uint32_t)(r3 >> 32); S[8] = (uint32_t)r4; S[9] = (uint32_t)(r4 >> 32);
    }
    const uint32_t hb = (0u - dst) & 7u;                 // bytes in front of the destination's first 8-byte boundary (< m)
    const uint32_t rem = m - hb, K = rem >> 3, tb = rem & 7u;
    // head: stream bytes 0 .. 7 = the 8 bytes at byte `so` of S
    {
        const uint32_t q = 0u - (so >> 2);
        const uint32_t h0 = (q & S[1]) | (~q & S[0]), h1 = (q & S[2]) | (~q & S[1]), h2 = (q & S[3]) | (~q & S[2]);
        const uint32_t lo = __builtin_amdgcn_alignbyte(h1, h0, so & 3u), hi = __builtin_amdgcn_alignbyte(h2, h1, so & 3u);
        if (on && (hb & 1u)) asm volatile("ds_write_b8 %0, %1" :: "v"(dst), "v"(lo) : "memory");
        if (on && (hb & 2u)) asm volatile("ds_write_b16 %0, %1" :: "v"(dst + (hb & 1u)), "v"(lo >> (8u * (hb & 1u))) : "memory");
        if (on && (hb & 4u)) asm volatile("ds_write_b32 %0, %1" :: "v"(dst + (hb & 3u)), "v"(__builtin_amdgcn_alignbyte(hi, lo, hb & 3u)) : "memory");
    }
    // body: stream byte hb + 8 k = byte T + 8 k of S
    const uint32_t T = so + hb;                          // 0 .. 14
    uint32_t B[11], C[9], E[8];
    // (bit selects, not ?: — the compiler turns a run of conditional element picks into a dynamically indexed scratch array)
    const uint32_t m4 = 0u - ((T >> 2) & 1u), m8 = 0u - ((T >> 3) & 1u);
#pragma unroll
    for (int i = 0; i < 11; i++) B[i] = (m4 & S[i + 1]) | (~m4 & S[i]);
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = (m8 & B[i + 2]) | (~m8 & B[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) E[i] = __builtin_amdgcn_alignbyte(C[i + 1], C[i], T & 3u);
    const uint32_t d8 = dst + hb;
    if (on && K > 0u) asm volatile("ds_write_b64 %0, %1" :: "v"(d8), "v"(((uint64_t)E[1] << 32) | E[0]) : "memory");
    if (on && K > 1u) asm volatile("ds_write_b64 %0, %1 offset:8" :: "v"(d8), "v"(((uint64_t)E[3] << 32) | E[2]) : "memory");
    if (on && K > 2u) asm volatile("ds_write_b64 %0, %1 offset:16" :: "v"(d8), "v"(((uint64_t)E[5] << 32) | E[4]) : "memory");
    if (on && K > 3u) asm volatile("ds_write_b64 %0, %1 offset:24" :: "v"(d8), "v"(((uint64_t)E[7] << 32) | E[6]) : "memory");
    // tail: word K (K <= 3 when tb > 0)
    {
        const uint32_t k0 = 0u - (uint32_t)(K == 0u), k1 = 0u - (uint32_t)(K == 1u), k2 = 0u - (uint32_t)(K == 2u), k3 = 0u - (uint32_t)(K >= 3u);
        const uint32_t tl = (E[0] & k0) | (E[2] & k1) | (E[4] & k2) | (E[6] & k3);
        const uint32_t th = (E[1] & k0) | (E[3] & k1) | (E[5] & k2) | (E[7] & k3);
        const uint32_t ta = d8 + 8u * K;
        const uint32_t w = (tb & 4u) ? th : tl;           // the word the 2- and 1-byte pieces come from
        if (on && (tb & 4u)) asm volatile("ds_write_b32 %0, %1" :: "v"(ta), "v"(tl) : "memory");
        if (on && (tb & 2u)) asm volatile("ds_write_b16 %0, %1" :: "v"(ta + (tb & 4u)), "v"(w) : "memory");
        if (on && (tb & 1u)) asm volatile("ds_write_b8 %0, %1" :: "v"(ta + (tb & 6u)), "v"(w >> (8u * (tb & 2u))) : "memory");
    }
}

// ---- the general path's copy (D3: matches above 32 bytes, self-overlapping ones; below kLongRun) ----------------------------
// m bytes from LDS address src to dst, `off` bytes apart in the output (off < 8: dst = src + off in the same window), the source final
// up to dst, the lane alone on this match.  In 8-byte pieces at
// their exact addresses — a handful of lanes, so the misaligned accesses are cheap — instead of byte by byte (a 200-byte match
// was 400 dependent LDS instructions; the corpus has 38 matches of 65 - 511 bytes per chunk, on the chain).  The DS queue of a
// wavefront executes in order, so a piece may read what the piece before it wrote:
//   off >= 8   pieces at 0, 8, 16 ... and a last one at m - 8 (m >= 8);
//   off <  8   the bytes repeat with period off: one read, the pieces generated in registers (below).
// Fewer than 8 bytes from 8 or more back: byte by byte.
__device__ __forceinline__ void lds_copy_serial(uint32_t src, uint32_t dst, uint32_t off, uint32_t m) {
    const auto piece = [](uint32_t s, uint32_t d) {
        uint64_t r;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(s) : "memory");
        asm volatile("ds_write_b64 %0, %1" :: "v"(d), "v"(r) : "memory");
    };
    const auto byte1 = [](uint32_t s, uint32_t d) {
        uint32_t r;
        asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(s) : "memory");
        asm volatile("ds_write_b8 %0, %1" :: "v"(d), "v"(r) : "memory");
    };
    if (m < 8u && off >= 8u) { for (uint32_t k = 0; k < m; k++) byte1(src + k, dst + k); return; }
    uint32_t k = 0;
    if (off < 8u) {
        // a run with period p = off < 8 (byte runs, short patterns: 0.6 % of the corpus's matches, 2.8 % of kppkn.gtb's): ONE read of
        // the p bytes, the repetition built in registers (the unit OR-ed onto itself at doubling distances), pieces at phase
        // 0, 8 mod p, ... cut out of its first 16 bytes — stores only, no read waits for a store
        const uint32_t p = off, pb = 8u * p;
        uint64_t b;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(src) : "memory");
        const uint64_t um = (1ull << pb) - 1ull;
        const uint64_t unit = b & um;
        const auto repeat = [pb](uint64_t x) {
            x |= x << pb;
            if (2u * pb < 64u) x |= x << (2u * pb);
            if (4u * pb < 64u) x |= x << (4u * pb);
            return x;
        };
        const uint32_t e = p == 3u ? 2u : p == 5u ? 3u : p == 6u ? 2u : p == 7u ? 1u : 0u;      // 8 mod p
        const uint64_t unit2 = e ? ((unit >> (8u * e)) | (unit << (8u * (p - e)))) & um : unit;
        const uint64_t q0 = repeat(unit), q1 = repeat(unit2);                               // bytes 0..7 and 8..15 of the run
        const uint32_t w0 = (uint32_t)q0, w1 = (uint32_t)(q0 >> 32), w2 = (uint32_t)q1, w3 = (uint32_t)(q1 >> 32);
        uint32_t r = 0;                                                                  // phase of the piece at k
        uint32_t lo, hi;
        const auto cut = [&]() {                                                         // the 8 bytes at phase r (< 8)
            const uint32_t hi4 = 0u - (r >> 2), s = r & 3u;
            const uint32_t a = (hi4 & w1) | (~hi4 & w0), bb = (hi4 & w2) | (~hi4 & w1), c = (hi4 & w3) | (~hi4 & w2);
            lo = __builtin_amdgcn_alignbyte(bb, a, s); hi = __builtin_amdgcn_alignbyte(c, bb, s);
        };
        for (; k + 8u <= m; k += 8u) {
            cut();
            asm volatile("ds_write_b64 %0, %1" :: "v"(dst + k), "v"(((uint64_t)hi << 32) | lo) : "memory");
            r += e; r = r >= p ? r - p : r;
        }
        const uint32_t rem = m - k;
        if (rem) {
            cut();
            const uint32_t t = dst + k, w = (rem & 4u) ? hi : lo;
            if (rem & 4u) asm volatile("ds_write_b32 %0, %1" :: "v"(t), "v"(lo) : "memory");
            if (rem & 2u) asm volatile("ds_write_b16 %0, %1" :: "v"(t + (rem & 4u)), "v"(w) : "memory");
            if (rem & 1u) asm volatile("ds_write_b8 %0, %1" :: "v"(t + (rem & 6u)), "v"(w >> (8u * (rem & 2u))) : "memory");
        }
        return;
    }
    if (off >= 32u) {                                            // four pieces per wait: what they read was written before they started
        for (; k + 32u <= m; k += 32u) {
            uint64_t r0, r1, r2, r3;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(src + k) : "memory");
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8\n\tds_write_b64 %0, %3 offset:16\n\tds_write_b64 %0, %4 offset:24"
                         :: "v"(dst + k), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
        }
    }
    for (; k + 8u <= m; k += 8u) piece(src + k, dst + k);
    if (k < m) piece(src + m - 8u, dst + m - 8u);                // (m - 8 >= the bytes done one by one: m - k < 8 was handled above)
}

__device__ __forceinline__ uint32_t wave_tier(uint32_t n, bool active) {       // wave-uniform tier for the active lanes
    if (ballot64(active && n > 32u)) return 64u;
    if (ballot64(active && n > 16u)) return 32u;
    return 16u;
}


// ---- ready bitmap: one bit per output byte ---------------------------------------------------
__device__ __forceinline__ void bits_set(uint32_t* bits, uint32_t lo, uint32_t hi) {     // [lo, hi), hi > lo
    uint32_t w0 = lo >> 5, w1 = (hi - 1u) >> 5;
    for (uint32_t w = w0; w <= w1; w++) {
        uint32_t m = ~0u;
        if (w == w0) m &= ~0u << (lo & 31u);
        if (w == w1) m &= ~0u >> (31u - ((hi - 1u) & 31u));
        __hip_atomic_fetch_or(&bits[w], m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__device__ __forceinline__ bool bits_ready(uint32_t* bits, uint32_t lo, uint32_t hi) {   // [lo, hi), hi > lo
    if (hi - lo <= 32u) {                 // the common case fits a 64-bit window: one double read
        const uint2 v = lds_ld64((uint32_t)(uintptr_t)(bits + (lo >> 5)));     // may read one word past the bitmap: harmless
        const uint64_t win = (((uint64_t)v.y << 32) | v.x) >> (lo & 31u);
        const uint64_t m = ~0ull >> (64u - (hi - lo));
        return (win & m) == m;
    }
    uint32_t w0 = lo >> 5, w1 = (hi - 1u) >> 5;
    for (uint32_t w = w0; w <= w1; w++) {
        uint32_t m = ~0u;
        if (w == w0) m &= ~0u << (lo & 31u);
        if (w == w1) m &= ~0u >> (31u - ((hi - 1u) & 31u));
        uint32_t v = __hip_atomic_load(&bits[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((v & m) != m) return false;
    }
    return true;
}

// whole-wave version for long ranges (all lanes call with the same lo/hi)
__device__ __forceinline__ void wave_bits_set(uint32_t* bits, uint32_t lo, uint32_t hi) {
    const uint32_t w0 = lo >> 5, w1 = (hi - 1u) >> 5;
    for (uint32_t w = w0 + lane_id(); w <= w1; w += 64u) {
        uint32_t m = ~0u;
        if (w == w0) m &= ~0u << (lo & 31u);
        if (w == w1) m &= ~0u >> (31u - ((hi - 1u) & 31u));
        __hip_atomic_fetch_or(&bits[w], m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <int ND>
__device__ __forceinline__ DW<ND> gl_ld_aligned(const uint8_t* pa, const uint8_t* last) {   // ND aligned dwords, clamped to the last valid one
    DW<ND> r;
#pragma unroll
    for (int i = 0; i < ND; i++) {
        const uint8_t* q = pa + 4 * i;
        r.w[i] = *reinterpret_cast<const uint32_t*>(q < last ? q : last);
    }
    return r;
}

// ND dwords (ND = 4k + 2) from the exact, arbitrarily aligned pointer g as k dwordx4 + one dwordx2 load: a scattered
// wave load costs the texture-address unit about one cycle per lane whatever its width, so 6 dword loads per lane
// (the aligned form above) take 3x as long as these 2.  May read up to 4*ND bytes from g: the caller checks that
// this stays inside the chunk's last 16 B granule.
template <int ND>
__device__ __forceinline__ DW<ND> gl_ld_vec(const uint8_t* g) {
    static_assert(ND % 4 == 2, "ND = 4k + 2");
    DW<ND> r;
#pragma unroll
    for (int i = 0; i + 4 <= ND; i += 4) {
        uint4 v;
        __builtin_memcpy(&v, g + 4 * i, 16);
        r.w[i] = v.x; r.w[i + 1] = v.y; r.w[i + 2] = v.z; r.w[i + 3] = v.w;
    }
    uint2 t;
    __builtin_memcpy(&t, g + 4 * (ND - 2), 8);
    r.w[ND - 2] = t.x; r.w[ND - 1] = t.y;
    return r;
}

// The same for a source that needs no shift (exact pointer): lds_store_tier<T> stores source bytes < n <= T only, and with a
// zero shift those come from the first T / 4 dwords — the two dwords after them only feed bytes that are never stored.  So T
// bytes = T / 16 dwordx4 loads are enough: one scattered load instead of two for the 16-byte tier, two instead of three for 32
// (D2 is bound by the address unit's ~1 cycle per lane per scattered load instruction, not by latency: requesting the next
// batch's bytes a batch ahead made it slower, 22.7 k -> 28 k cycles per chunk, because it took a third instruction per record).
template <int T>
__device__ __forceinline__ DW<T / 4 + 2> gl_ld_exact(const uint8_t* g) {
    DW<T / 4 + 2> r;
#pragma unroll
    for (int i = 0; i < T / 4; i += 4) {
        uint4 v;
        __builtin_memcpy(&v, g + 4 * i, 16);
        r.w[i] = v.x; r.w[i + 1] = v.y; r.w[i + 2] = v.z; r.w[i + 3] = v.w;
    }
    r.w[T / 4] = 0u; r.w[T / 4 + 1] = 0u;
    return r;
}

// =====================================================================================================
// The PARSE STAGE INSIDE the decoder (kFused): no separate parse kernel, no sync points in memory, the compressed chunk is
// read from HBM once.  S0 has staged the chunk in LDS; 256 lanes then walk 256 SEGMENTS of it at once — the segmented
// speculative walk of parse_spec.hip (one wavefront per chunk there) on four wavefronts:
//   P1a  lane l starts at the guessed position l * seg and walks its own segment, marking every position it visits
//        (one bit per input byte, in the ready bitmap's space: it is not needed before D2);
//   P1b  it walks on through the following segments until it steps on a marked position: from there its path IS that
//        segment owner's path (the next-element function depends on the bytes only); merge[l] = that position;
//   P2   the true path = lane 0's piece, then the piece of the lane it merged into, ...: a pointer chain over the lanes,
//        marked from lane 0 by pointer doubling (8 rounds for 256 lanes) instead of 256 dependent hops;
//   P3   the lanes on the chain count the sequences and output bytes of their piece, an exclusive scan over the lanes
//        gives every piece its first sequence index and output position;
//   P4   they walk their piece once more with the true (index, output position): the decoder's own validation rules
//        (the same Lz4Grammar / SnappyGrammar functions as the parse kernels) and the 16-byte records of D1, written
//        straight to the workgroup's record table.
// Anything that is not a clean chunk of kLdsMinSeq .. kSyncStride * kSyncEvery (256 .. 16 384) sequences (any violation, too few or too many sequences) is handed to
// the wavefront-per-chunk kernel, which decodes every valid chunk and names every error exactly: returns false then.
// =====================================================================================================
#ifndef CJ_FUSED_LANES
#define CJ_FUSED_LANES 512
#endif
constexpr uint32_t kFusedLanes = CJ_FUSED_LANES;          // every thread of the workgroup walks a segment (256, wavefronts 0-3 only: 137 k cycles per chunk)
constexpr uint32_t kFusedAux = 6144u;                               // merge, next, mark, entry (16 bits per lane each) + totals: behind the decoder's LDS
static_assert(4u * kFusedLanes * 2u + 128u <= kFusedAux, "aux");
// (round 6: the parse runs on every window of the workgroup decoder — 512 lanes on 64 KiB, 256 on 32 KiB, 128 on 16 KiB: the workgroup's threads)
constexpr uint32_t fused_aux_bytes(uint32_t lanes) { return lanes >= 512u ? kFusedAux : 8u * lanes + 128u; }

// The walked elements are LISTED (round 6, f04): every step of P1a / P1b also stores what it read — a 12-byte cell { lit_at | lit << 16,
// mlen | offset << 16, output bytes of the lane's walk before the step } — into the workgroup's table slot, row = the loop's iteration,
// column = the lane (one coalesced store per wavefront and step; P1b's rows follow the longest P1a of the workgroup).  A piece of the true
// path is a suffix of its lane's list: it begins at the P1a step that stood on the piece's entry position, and that step's number is the
// count of the lane's own marks below the entry.  P3 therefore is a subtraction, and P4 needs no walk either: a listed element's output
// position is the piece's base + the difference of two "output before" words, so every cell is checked and written on its own.  The pieces
// are cut into GROUPS of four cells, the groups go to a work list in LDS, and every thread takes groups off it — four loads that depend
// on nothing, then four checks — whatever lane walked them.  (r05 f02 / r06 f04: the walks run at the pace of the workgroup's slowest
// lane, and a lane whose neighbours' guesses never meet the true path walks 40 .. 90 elements: P1a 11 k, P1b 29 k, P3 24 k, P4 57 k
// cycles of the fused kernel's 200 k per chunk; listed: P1a 16 k, P1b 41 k, P3 4.5 k, P4 31 k — profiles/r06/experiments f04.)  A chunk
// whose walks outgrow the rows takes the walking P3 / P4 below, one with more groups than the work list holds (16 384 sequences in
// pieces of 4 k + 1: out of reach in practice) the walking P4: slower, same records.
#ifndef CJ_FUSED_LIST
#define CJ_FUSED_LIST 1
#endif
#ifndef CJ_FUSED_ROWS
#define CJ_FUSED_ROWS 144u
#endif
constexpr uint32_t kFlRows = CJ_FUSED_ROWS;                         // (at most 255: step numbers travel in 8 bits)
// the lists' place in the slot, in 4-byte units: behind the chunk's records + sentinel (64 KiB window: 132 KiB; the forwarding phase's extras come later, when the lists are dead)
constexpr uint32_t fl_base(uint32_t win) { return win >= 65536u ? 33792u : ((lds_window_max_seq(win) + 1u) * 2u + 255u) & ~255u; }
constexpr uint32_t fl_slot_units(uint32_t win, uint32_t lanes) { return (fl_base(win) + 3u * kFlRows * lanes + 3u) / 4u; }      // 16-byte units of a slot that holds the lists
#ifndef CJ_FUSED_GROUP
#define CJ_FUSED_GROUP 4u
#endif
#ifndef CJ_FUSED_MAX_GROUPS
#define CJ_FUSED_MAX_GROUPS 4096u
#endif
constexpr uint32_t kFlGroup = CJ_FUSED_GROUP, kFlMaxGroups = CJ_FUSED_MAX_GROUPS;   // cells per group; the work list (16-bit entries: lane | group of its piece << 9) lives in the bitmap's 8 KiB
static_assert(kFlRows / kFlGroup < 128u, "a piece's group number fits 7 bits");
static_assert(kFlRows < 256u && fl_slot_units(65536u, kFusedLanes) <= 4u * 16384u, "the lists fit the table slot of a 64 KiB window");
struct FlCell { uint32_t x, y, ob; };
__device__ unsigned long long g_fused_paths[4];          // test hook (counted under CJ_FLAG_DEBUG_PROFILE only: an atomic per chunk on one address is waited for with the thread's next load): chunks whose P3 / P4 came from the lists, walked P4 only, walked both

__device__ __forceinline__ uint32_t wave_excl_scan_add32(uint32_t v, uint32_t& total) {
    const uint32_t lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)x, d, 64);
        if (lane >= (uint32_t)d) x += t;
    }
    total = rdlane(x, 63);
    return x - v;
}

// a_in: LDS address of stream position 0; aux: kFusedAux bytes of LDS; bits: 8 KiB, zeroed; table2: the record table (8-byte records).
// On success: nseq_out / U_out, *near_out += matches with an offset below kFwdNear.  Every thread of the workgroup calls it.
template <class G, uint32_t kThreads, uint32_t kWin = 65536u>
__device__ __forceinline__ bool fused_parse(uint32_t a_in, uint32_t iend, uint32_t cap, uint32_t* bits, uint32_t* aux, uint2* table2,
                                            uint32_t* s_near, uint32_t& nseq_out, uint32_t& U_out, uint32_t* sub_prof = nullptr) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // (these shadow the file's 64 KiB values: the window's lanes, list base, work-list capacity — the bitmap's space —, sequence limits, "near")
    constexpr uint32_t kFusedLanes = kThreads, kFlBase = fl_base(kWin), kFlMaxGroups = CJ_FUSED_MAX_GROUPS < kWin / 16u ? CJ_FUSED_MAX_GROUPS : kWin / 16u;
    constexpr uint32_t kMinSeq = kWin >= 65536u ? kLdsMinSeq : lds_window_min_seq(kWin), kMaxSeq = lds_window_max_seq(kWin), kNear = kWin / 16u;
    constexpr uint32_t kBitWordsW = kWin / 32u;
    // (CJ_FLAG_DEBUG_PROFILE: cycles of P1a, P1b, P2, P3 + scan, P4 as thread 0 sees them, barriers included -> sub-marks 6 .. 10)
    unsigned long long t_sub = sub_prof ? __builtin_readcyclecounter() : 0ull;
    const auto sub_mark = [&](uint32_t k) {
        if (sub_prof && tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); sub_prof[k] += (uint32_t)(now - t_sub); t_sub = now; }
    };
    // (positions fit 16 bits: a staged chunk ends below 65 520; 0xFFFF / 0xFFFE stand for kPosErr / kPosEnd)
    uint16_t* s_merge = reinterpret_cast<uint16_t*>(aux);
    uint16_t* s_next = s_merge + kFusedLanes;
    uint16_t* s_mark = s_merge + 2u * kFusedLanes;
    uint16_t* s_entry = s_merge + 3u * kFusedLanes;
    uint32_t* s_tot = aux + 2u * kFusedLanes;            // [0..8): per-wave counts, [8..16): per-wave bytes, [16..24): verdict words
    const uint32_t a_bits = (uint32_t)(uintptr_t)bits;
    const auto rd = [a_in](uint32_t q) { return lds_ld32a(a_in + q); };
    const auto rd8 = [a_in](uint32_t q) { return lds_ld64a(a_in + q); };
    const bool plane = tid < kFusedLanes;                  // the parse lanes (wavefronts 0-3); everyone takes the barriers

    uint32_t nl = (iend + 63u) / 64u;
    nl = nl > kFusedLanes ? kFusedLanes : nl;
    const uint32_t seg = (((iend + nl - 1u) / nl + 3u) & ~3u) | 4u;          // 4 x odd: the lanes' start positions spread over the banks
    const bool active = plane && tid < nl && tid * seg < iend;
    const uint32_t seg_end = (tid + 1u) * seg;
    if (tid < 24u) s_tot[tid] = 0u;                       // ([24..32): every wavefront's P1a iterations, stored unconditionally)

    // ---- P1a ----
    uint32_t p = active ? tid * seg : kPosEnd;
    WalkCarry wc = {0u, 0xFFFFFFFFu};                     // (the walks carry the next element's first bytes: one dependent read per element)
    uint32_t* const fl_c = reinterpret_cast<uint32_t*>(table2) + kFlBase;      // cell (row, lane) at fl_c + 3 * (row * kFusedLanes + lane)
    uint32_t fl_na = 0, fl_nb = 0, fl_op = 0;             // listed steps of P1a / P1b, output bytes of the walk so far
    uint32_t fl_ita = 0, fl_itb = 0, fl_rowb = 0;         // this wavefront's iterations of P1a / P1b, P1b's first row
    const auto fl_put = [&](uint32_t row, const Seq& sq) {
        const bool rep = sq.lit <= 0xffffu && sq.mlen <= 0xffffu && sq.offset <= 0xffffu;      // anything else cannot be part of a chunk of at most 64 KiB: P4 refuses it
        FlCell c;
        c.x = rep ? sq.lit_at | (sq.lit << 16) : 0xFFFFFFFFu; c.y = rep ? sq.mlen | (sq.offset << 16) : 0xFFFFFFFFu; c.ob = fl_op;
        __builtin_memcpy(fl_c + 3u * (row * kFusedLanes + tid), &c, 12);
        fl_op += sq.lit + sq.mlen;
    };
    if (plane) {
        uint32_t it = 0;
        while (ballot64(p < seg_end && p < iend) != 0ull) {
            if (p < seg_end && p < iend) {
                asm volatile("ds_or_b32 %0, %1" :: "v"(a_bits + 4u * (p >> 5)), "v"(1u << (p & 31u)) : "memory");
                Seq sq;
                const bool ok = walk_step_carry<G>(rd, rd8, p, iend, sq, wc, true);
#if CJ_FUSED_LIST
                if (ok && it < kFlRows) { fl_put(it, sq); fl_na += 1; }
#endif
                p = ok ? sq.next : kPosErr;
            }
            it += 1;
        }
        fl_ita = it;
        if (lane == 0) s_tot[24u + wave] = it;
    }
    __syncthreads();
    sub_mark(6u);
#if CJ_FUSED_LIST
    for (uint32_t w = 0; w < kFusedLanes / 64u; w++) fl_rowb = fl_rowb > s_tot[24u + w] ? fl_rowb : s_tot[24u + w];
#endif
    // ---- P1b ----
    uint32_t merge_pos = p;
    if (plane) {
        bool going = active && p < iend;
        uint32_t it = 0;
        while (ballot64(going) != 0ull) {
            if (going) {
                // the mark word and the element travel together (the step is thrown away where the position is marked)
                uint32_t w;
                asm volatile("ds_read_b32 %0, %1" : "=v"(w) : "v"(a_bits + 4u * (p >> 5)) : "memory");
                Seq sq;
                const bool ok = walk_step_carry<G>(rd, rd8, p, iend, sq, wc, true);
                const uint32_t nx = ok ? sq.next : kPosErr;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w) :: "memory");
                if ((w >> (p & 31u)) & 1u) { merge_pos = p; going = false; }
                else {
#if CJ_FUSED_LIST
                    if (ok && fl_rowb + it < kFlRows) { fl_put(fl_rowb + it, sq); fl_nb += 1; }
#endif
                    p = nx;
                    if (p >= iend) { merge_pos = p; going = false; }
                }
            }
            it += 1;
        }
#if CJ_FUSED_LIST
        fl_itb = it;
        if (fl_rowb + it > kFlRows && lane == 0) atomicOr(&s_tot[20], 1u);      // (s_tot was zeroed in front of the barrier behind P1a)
#endif
        if (active && merge_pos >= iend && merge_pos != kPosEnd) merge_pos = kPosErr;
        const uint32_t nx0 = merge_pos < iend ? merge_pos / seg : tid;       // a piece that ends the stream points at itself
        s_merge[tid] = (uint16_t)(merge_pos >= 0xFFFEu ? (merge_pos == kPosEnd ? 0xFFFEu : 0xFFFFu) : merge_pos);
        s_next[tid] = (uint16_t)(active ? nx0 : tid);
        s_mark[tid] = tid == 0u ? 1u : 0u;
        s_entry[tid] = 0u;
    }
    __syncthreads();
    sub_mark(7u);
    // ---- P2: mark the chain from lane 0 by pointer doubling ----
    for (uint32_t round = 0; (1u << round) < kFusedLanes; round++) {
        uint32_t n1 = 0, n2 = 0;
        if (plane) {
            n1 = s_next[tid];
            if (s_mark[tid]) s_mark[n1] = 1u;
            n2 = s_next[n1];
        }
        __syncthreads();
        if (plane) s_next[tid] = n2;
        __syncthreads();
    }
    bool on_chain = false;
    if (plane) {
        on_chain = active && s_mark[tid] != 0u;
        if (on_chain && merge_pos < iend) s_entry[merge_pos / seg] = (uint16_t)merge_pos;
    }
    __syncthreads();
    sub_mark(8u);
    const uint32_t entry = plane ? s_entry[tid] : 0u;       // lane 0 enters at 0
    const uint32_t piece_end = merge_pos;

    // ---- P3: count ----
    uint32_t cnt = 0, outb = 0;
    uint32_t base_idx = 0, base_op = 0, total_seq = 0;
    bool p4_done = false;
#if CJ_FUSED_LIST
    const bool use_list = s_tot[20] == 0u;                  // (uniform; written before the barrier in front of P2)
    if (sub_prof && tid == 0 && use_list) sub_prof[11] += 1000u;       // (sub-mark 11: chunks on the list path, per mille)
    if (use_list) {
        // the lane's list: steps 0 .. fl_na - 1 in rows 0 .., steps fl_na .. in rows fl_rowb ..
        const uint32_t fl_n = fl_na + fl_nb;
        const auto fl_at = [&](uint32_t l2, uint32_t jj, uint32_t na) { return fl_c + 3u * ((jj < na ? jj : fl_rowb + (jj - na)) * kFusedLanes + l2); };
        uint32_t fl_k = 0, fl_ob = 0;                        // the piece's first step, the walk's output bytes before it
        if (plane) {
            if (on_chain) {
                // the step that stood on `entry` = the number of this lane's marks below it (only the segment's own lane marks in it, in walk order)
                const uint32_t lo = tid * seg, hi = entry, w0 = lo >> 5;
                uint32_t mk[6];
#pragma unroll
                for (uint32_t u = 0; u < 6u; u++) mk[u] = bits[(w0 + u) < kBitWordsW - 1u ? (w0 + u) : kBitWordsW - 1u];
#pragma unroll
                for (uint32_t u = 0; u < 6u; u++) {
                    const uint32_t b0 = (w0 + u) * 32u;
                    uint32_t m = b0 < lo ? ~0u << (lo - b0) : ~0u;
                    if (b0 + 32u > hi) m &= hi > b0 ? (1u << (hi - b0)) - 1u : 0u;
                    fl_k += (uint32_t)__builtin_popcount(mk[u] & m);
                }
                if (fl_k > fl_n) fl_k = fl_n;              // (cannot happen: every mark below the entry has its listed step)
                cnt = fl_n - fl_k;
                if (cnt) { fl_ob = fl_at(tid, fl_k, fl_na)[2]; outb = fl_op - fl_ob; }
            }
            uint32_t tc, tb;
            base_idx = wave_excl_scan_add32(cnt, tc);
            base_op = wave_excl_scan_add32(outb, tb);
            if (lane == 0) { s_tot[wave] = tc; s_tot[8u + wave] = tb; }
        }
        __syncthreads();
        sub_mark(9u);
        if (plane) {
            for (uint32_t w = 0; w < wave; w++) { base_idx += s_tot[w]; base_op += s_tot[8u + w]; }
        }
        for (uint32_t w = 0; w < kFusedLanes / 64u; w++) total_seq += s_tot[w];
        if (total_seq < kMinSeq || total_seq > kMaxSeq) return false;       // uniform: too few / too many sequences for this decoder
        // P4 from the lists.  LDS: the work list in the bitmap's space (every lane has counted its marks: the barrier above), the pieces'
        // parameters in the lane arrays of P1 / P2 (dead: merge_pos and entry are in registers) — { first record (15) | first step (8) |
        // steps of P1a (8) | ends the stream (1), output position of the piece - output before its first step (17) | steps (8) }
        uint16_t* wl = reinterpret_cast<uint16_t*>(bits);
        uint2* info = reinterpret_cast<uint2*>(aux);
        if (plane) {
            const uint32_t ng = (cnt + kFlGroup - 1u) / kFlGroup;
            if (ng != 0u) {
                const uint32_t at = atomicAdd(&s_tot[21], ng);
                if (at + ng <= kFlMaxGroups) { for (uint32_t g = 0; g < ng; g++) wl[at + g] = (uint16_t)(tid | (g << 9)); }
            }
            info[tid] = make_uint2(base_idx | (fl_k << 15) | (fl_na << 23) | ((merge_pos == kPosEnd ? 1u : 0u) << 31), ((base_op - fl_ob) & 0x1ffffu) | (fl_n << 17));
        }
        __syncthreads();
        sub_mark(12u);
        const uint32_t n_groups = s_tot[21];
        if (n_groups <= kFlMaxGroups) {                      // (uniform; more groups than the list holds: the walking P4 below)
            p4_done = true;
            if (sub_prof && tid == 0) atomicAdd(&g_fused_paths[0], 1ull);
            if (plane) {
                bool bad = false;
                uint32_t near = 0;
                for (uint32_t gi = tid; gi < n_groups; gi += kFusedLanes) {
                    const uint32_t we = wl[gi], l2 = we & 511u;
                    const uint2 pi = info[l2];
                    const uint32_t bidx = pi.x & 0x7fffu, pk = (pi.x >> 15) & 0xffu, pna = (pi.x >> 23) & 0xffu, pn = pi.y >> 17, opd = pi.y & 0x1ffffu;
                    const bool ends = (pi.x >> 31) != 0u;            // the lane's last step consumed the input exactly
                    const uint32_t j0 = pk + kFlGroup * (we >> 9);
                    FlCell c[kFlGroup] = {};
#pragma unroll
                    for (uint32_t u = 0; u < kFlGroup; u++) if (j0 + u < pn) __builtin_memcpy(&c[u], fl_at(l2, j0 + u, pna), 12);
#pragma unroll
                    for (uint32_t u = 0; u < kFlGroup; u++) {
                        const uint32_t jj = j0 + u;
                        if (jj < pn) {
                            Seq sq;
                            sq.lit_at = c[u].x & 0xffffu; sq.lit = c[u].x >> 16; sq.mlen = c[u].y & 0xffffu; sq.offset = c[u].y >> 16;
                            sq.last = ends && jj + 1u == pn; sq.next = 0u;
                            // (mod 2^17: exact wherever the position is at most `cap`; the first cell of a chunk that is not fails its own check)
                            const uint32_t idx = bidx + (jj - pk), op = (opd + c[u].ob) & 0x1ffffu;
                            bool fin = false;
                            uint32_t op2 = op;
                            if ((c[u].x & c[u].y) == 0xFFFFFFFFu || op > cap || !G::check(sq, op2, cap, fin) || idx >= total_seq) bad = true;
                            else {
                                const uint32_t w = sq.mlen == 0u ? 0u : (sq.offset | (sq.mlen << 16));
                                table2[idx] = make_uint2(c[u].x, (op & 0xffffu) | (w << 16));         // 8-byte record (lds2_body)
                                near += (w != 0u && sq.offset < kNear) ? 1u : 0u;
                                if (fin) { atomicAdd(&s_tot[17], 1u); s_tot[18] = op2; s_tot[19] = idx + 1u; }
                            }
                        }
                    }
                }
                sub_mark(13u);
                if (near) atomicAdd(s_near, near);
                if (bad || (on_chain && piece_end == kPosErr)) atomicOr(&s_tot[16], 1u);
            }
        } else if (sub_prof && tid == 0) atomicAdd(&g_fused_paths[1], 1ull);
    } else
#endif
    {
    if (sub_prof && tid == 0) atomicAdd(&g_fused_paths[2], 1ull);
    if (plane) {
        uint32_t q = on_chain ? entry : kPosEnd;
        WalkCarry c3 = {0u, 0xFFFFFFFFu};
        while (ballot64(q < iend && q != piece_end) != 0ull) {
            if (q < iend && q != piece_end) {
                Seq sq;
                if (walk_step_carry<G>(rd, rd8, q, iend, sq, c3, true)) { cnt += 1; outb += sq.lit + sq.mlen; q = sq.next; }
                else q = kPosErr;
            }
        }
    }
    if (plane) {
        uint32_t tc, tb;
        base_idx = wave_excl_scan_add32(cnt, tc);
        base_op = wave_excl_scan_add32(outb, tb);
        if (lane == 0) { s_tot[wave] = tc; s_tot[8u + wave] = tb; }
    }
    __syncthreads();
    sub_mark(9u);
    if (plane) {
        for (uint32_t w = 0; w < wave; w++) { base_idx += s_tot[w]; base_op += s_tot[8u + w]; }
    }
    for (uint32_t w = 0; w < kFusedLanes / 64u; w++) total_seq += s_tot[w];
    // (a valid chunk's pieces add up to at most `cap` output bytes; a wild count is caught by the checks of P4)
    if (total_seq < kMinSeq || total_seq > kMaxSeq) return false;       // uniform: too few / too many sequences for this decoder
    }

    // ---- P4: validate + write the records ----
    if (p4_done) {
        // (from the lists, above)
    } else
    if (plane) {
        bool bad = false, saw_last = false;
        uint32_t final_op = 0, near = 0;
        uint32_t q = on_chain ? entry : kPosEnd, idx = base_idx, op = base_op;
        WalkCarry c4 = {0u, 0xFFFFFFFFu};
        while (ballot64(q < iend && q != piece_end && !bad) != 0ull) {
            if (q < iend && q != piece_end && !bad) {
                Seq sq;
                bool fin = false;
                uint32_t op2 = op;
                if (!walk_step_carry<G>(rd, rd8, q, iend, sq, c4, true) || !G::check(sq, op2, cap, fin) || idx >= total_seq) bad = true;
                else {
                    const uint32_t w = sq.mlen == 0u ? 0u : ((sq.offset & 0xffffu) | (sq.mlen << 16));       // (a Snappy stream may END with a copy)
                    table2[idx] = make_uint2(sq.lit_at | (sq.lit << 16), (op & 0xffffu) | (w << 16));      // 8-byte record (lds2_body)
                    near += (w != 0u && sq.offset < kNear) ? 1u : 0u;
                    op = op2;
                    if (fin) { final_op = op; saw_last = true; q = kPosEnd; }
                    else { q = sq.next; idx += 1; }
                }
            }
        }
        if (near) atomicAdd(s_near, near);
        if (on_chain && (bad || piece_end == kPosErr)) atomicOr(&s_tot[16], 1u);
        if (on_chain && saw_last) { atomicAdd(&s_tot[17], 1u); s_tot[18] = final_op; s_tot[19] = idx + 1u; }
    }
    __syncthreads();
    sub_mark(10u);
    const uint32_t op_end = s_tot[18];
    if (s_tot[16] != 0u || s_tot[17] != 1u || s_tot[19] != total_seq || !G::result_ok(op_end, cap) || op_end == 0u) return false;
    nseq_out = total_seq;
    U_out = op_end;
    return true;
}

}  // namespace cj
