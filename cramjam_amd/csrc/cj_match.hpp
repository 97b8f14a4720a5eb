// cj_match.hpp — wave-parallel LZ77 match finder shared by the LZ4-block and Snappy-raw encoders.
//
// The CPU encoders the reference links (LZ4_compress_default, snap's compress_block) probe ONE hash
// slot per step on one core.  Here a wavefront probes 64 consecutive positions at once against a
// per-wave hash table in LDS (4096 x u16, 8 KiB), ballots the lanes whose candidate verifies, and
// consumes the ballot greedily left to right (first match wins, lanes covered by it are dropped),
// so one table round serves several sequences.  Output is a valid stream for the format's decoder;
// it is NOT byte-identical to the CPU encoders (nor required to be: the reference pins compressed
// bytes only for the 14-byte all-literal case, /root/reference/tests/test_variants.py:329-334).
#pragma once
#include "cj_common.hpp"

namespace cj {
#if defined(__HIPCC__)

// 13 bits (8192 x u16 = 16 KiB per wave, the table size liblz4 uses for <= 64 KiB inputs) together with the backward
// extension reproduces the CPU encoders' ratio on the benchmark data (LZ4 1.636 vs 1.632, Snappy 1.630 vs 1.618);
// 12 bits: 1.47 at 1.75x the speed, 14 bits: 1.70 at 0.6x (occupancy is LDS-limited).
#ifndef CJ_HASH_BITS
#define CJ_HASH_BITS 13
#endif
constexpr uint32_t kHashBits = CJ_HASH_BITS;
constexpr uint32_t kHashSize = 1u << kHashBits;
// waves per encoder block: keep the block's tables within the 64 KiB static-LDS limit
constexpr int kEncWaves = 1;   // one wave per block: 160 KiB / 16 KiB = 10 resident waves per CU (4-wave blocks would round down to 8)
constexpr int kEncThreads = 64 * kEncWaves;

__device__ __forceinline__ void ht_clear(uint16_t* ht) {
    uint32_t* p = reinterpret_cast<uint32_t*>(ht);
    for (uint32_t i = lane_id(); i < kHashSize / 2; i += 64u) p[i] = 0u;
}

// count equal bytes of in[a..] vs in[b..] (b < a), stopping at position `limit` for a
__device__ __forceinline__ uint32_t wave_extend(const uint8_t* in, uint32_t a, uint32_t b, uint32_t limit) {
    uint32_t cnt = 0;
    const uint32_t lane = lane_id();
    for (;;) {
        uint32_t j = a + cnt + lane;
        bool eq = false;
        if (j < limit) eq = in[j] == in[b + cnt + lane];
        uint64_t mm = ballot64(eq);
        if (mm == ~0ull) { cnt += 64u; continue; }
        cnt += ctz64(~mm);
        break;
    }
    return cnt;
}

// backward extension ("catch-up"): how many bytes before a / b also match, limited to `room` (bytes back to the
// anchor) and to b itself.  The probe often hits a repeated region a few bytes after its start; without this the
// head of every such match is emitted as literals (synth-v1: ratio 1.37 -> see DESIGN.md).
__device__ __forceinline__ uint32_t wave_extend_back(const uint8_t* in, uint32_t a, uint32_t b, uint32_t room) {
    const uint32_t lim = room < b ? room : b;
    uint32_t cnt = 0;
    const uint32_t lane = lane_id();
    while (cnt < lim) {
        const uint32_t i = cnt + lane;
        bool eq = false;
        if (i < lim) eq = in[a - 1u - i] == in[b - 1u - i];
        const uint64_t mm = ballot64(eq);
        if (mm == ~0ull) { cnt += 64u; continue; }
        cnt += ctz64(~mm);
        break;
    }
    return cnt < lim ? cnt : lim;
}

// One probe round over positions [pos, pos+64).  last_start: last position where a match may start
// (needs 4 readable bytes).  Returns the ballot of verified lanes; cand (per lane) is the match source,
// hslot the lane's table slot (kNoSlot when the lane is past last_start).  The table is NOT updated here:
// the caller inserts, after it has consumed the ballot, only the positions that did not end up inside an
// emitted match (insert_uncovered).  Positions inside a match repeat content whose source is already
// indexed; inserting them too evicts distant sources from the small table ~4x faster on match-heavy
// data (measured on synth-v1: ratio 1.37 with dense insertion vs the CPU encoder's 1.63).
constexpr uint32_t kNoSlot = 0xffffffffu;

__device__ __forceinline__ uint64_t probe_round(const uint8_t* in, const uint16_t* ht, uint32_t pos,
                                                uint32_t last_start, uint32_t& cand, uint32_t& hslot) {
    const uint32_t my = pos + lane_id();
    const bool valid = my <= last_start;
    uint32_t v = 0, h = kNoSlot, c = 0;
    bool ok = false;
    if (valid) {
        v = ld32u(in + my);
        h = (v * 2654435761u) >> (32 - kHashBits);
        c = (my & 0xFFFF0000u) | ht[h];
        if (c >= my) c -= 65536u;           // slot belongs to the previous 64 KiB lap (or is stale)
        if (c < my && my - c <= 65535u) ok = ld32u(in + c) == v;
    }
    cand = c;
    hslot = h;
    return ballot64(ok);
}

// covered: bit l set = position pos + l lies strictly inside an emitted match (not its first byte)
__device__ __forceinline__ void insert_uncovered(uint16_t* ht, uint32_t pos, uint32_t hslot, uint64_t covered) {
    const uint32_t lane = lane_id();
    if (hslot != kNoSlot && ((covered >> lane) & 1ull) == 0ull) ht[hslot] = (uint16_t)(pos + lane);
}

// lanes of a round strictly inside a match that starts at position mstart (possibly before the round, after a
// backward extension) and ends before lane `end_lane`
__device__ __forceinline__ uint64_t covered_bits(uint32_t pos, uint32_t mstart, uint32_t end_lane) {
    const uint32_t lo = mstart >= pos ? mstart - pos + 1u : 0u, hi = end_lane < 64u ? end_lane : 64u;
    if (hi <= lo) return 0ull;
    const uint64_t upto_hi = hi >= 64u ? ~0ull : ((1ull << hi) - 1ull);
    return upto_hi & ~((1ull << lo) - 1ull);
}

#endif
}  // namespace cj
