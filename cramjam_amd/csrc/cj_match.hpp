// cj_match.hpp — what the LZ4-block and Snappy-raw encoders share below the matcher (cj_enc2.hpp): the per-wavefront hash table in
// LDS, whole-wave backward extension, and the wave scans the selection and the emission use.
//
// The CPU encoders the reference links (LZ4_compress_default, snap's compress_block) probe ONE hash slot per step on one core.
// Here a wavefront probes a round of consecutive positions at once against its own 8192 x u16 table; the output is a valid stream
// for the format's decoder, NOT byte-identical to the CPU encoders (nor required to be: the reference pins compressed bytes only
// for the 14-byte all-literal case, /root/reference/tests/test_variants.py:329-334).
#pragma once
// The matcher's byte-determinism rests on gfx9 semantics: DPP row_bcast / wave_shr exist, and among the lanes of ONE DS store
// instruction that hit the same address the highest lane wins (HashTab).  Another architecture would compile and give other bytes.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "cj_match.hpp: the encoders are written for gfx950 (gfx9 wave64: DPP row_bcast / wave_shr, DS same-address store order)"
#endif
#include "cj_common.hpp"

namespace cj {
#if defined(__HIPCC__)

// 13 bits (8192 x u16 = 16 KiB per wave, the table size liblz4 uses for <= 64 KiB inputs) reproduce the CPU encoders' ratio on the
// benchmark data; 12 bits: 1.47 at 1.75x the speed, 14 bits: 1.70 at 0.6x (occupancy is LDS-limited) — round 1's sweep.
constexpr uint32_t kHashBits = 13;
constexpr uint32_t kHashSize = 1u << kHashBits;
__device__ __forceinline__ uint32_t hash_slot(uint32_t v) { return (v * 2654435761u) >> (32 - kHashBits); }

// The hash table of one wavefront: 16-bit positions in LDS.  DS operations of a wavefront execute in order, and a DS write with
// equal addresses keeps the highest lane's data — "the most recent position wins a slot" rests on both.
struct HashTab {
    uint16_t* p;
    __device__ __forceinline__ uint32_t get(uint32_t h) const { return p[h]; }
    __device__ __forceinline__ void set(uint32_t h, uint32_t v) const { p[h] = (uint16_t)v; }
    // (by all `threads` threads of the chunk's workgroup)
    __device__ __forceinline__ void clear(uint32_t tid, uint32_t threads) const {
        uint32_t* q = reinterpret_cast<uint32_t*>(p);
        for (uint32_t i = tid; i < kHashSize / 2; i += threads) q[i] = 0u;
    }
    // after a group of stores, before the lookups that follow: only the COMPILER has to be told — clear() stores dwords through a
    // punned pointer, and nothing else keeps the 16-bit lookups from being scheduled above them
    __device__ __forceinline__ void settle() const { asm volatile("" ::: "memory"); }
    // split pieces (large.hip): index the data BEFORE this wave's sub-piece (positions [0, q0), ascending so that the most recent
    // position wins a slot), so that the sub-piece finds the matches a serial walk over the piece would
    __device__ __forceinline__ void preindex(const uint8_t* in, uint32_t q0) const {
        for (uint32_t pos = lane_id(); pos < q0; pos += 64u) set(hash_slot(ld32u(in + pos)), pos);
    }
};

// backward extension ("catch-up"): how many bytes before a / b also match, limited to `room` (bytes back to the anchor) and to b
// itself.  The probe often hits a repeated region a few bytes after its start; without this the head of every such match is
// emitted as literals (synth-v1: ratio 1.37).
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

// Wave scans over the 64 lanes with DPP moves (one VALU instruction per step, no LDS): row_shr 1, 2, 4, 8 scan the rows of 16
// lanes; across rows row_bcast:15 / row_bcast:31 (gfx9 family).  A lane whose DPP source lies outside its row (or whose row is
// masked off) receives `identity`.
template <uint32_t kCtrl, uint32_t kRowMask = 0xfu>
__device__ __forceinline__ uint32_t dpp_from(uint32_t identity, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, (int)kCtrl, (int)kRowMask, 0xf, false);
}
constexpr uint32_t kDppRowShr = 0x110u, kDppBcast15 = 0x142u, kDppBcast31 = 0x143u, kDppWaveShr1 = 0x138u;

__device__ __forceinline__ uint32_t wave_excl_add(uint32_t v, uint32_t& total) {
    uint32_t x = v;
    x += dpp_from<kDppRowShr + 1u>(0u, x);
    x += dpp_from<kDppRowShr + 2u>(0u, x);
    x += dpp_from<kDppRowShr + 4u>(0u, x);
    x += dpp_from<kDppRowShr + 8u>(0u, x);
    x += dpp_from<kDppBcast15, 0xau>(0u, x);
    x += dpp_from<kDppBcast31, 0xcu>(0u, x);
    total = rdlane(x, 63);
    return x - v;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// max over the lanes BELOW this one and `first` (lane 0 receives `first`); total = max over first and all 64 lanes
__device__ __forceinline__ uint32_t wave_excl_max(uint32_t v, uint32_t first, uint32_t& total) {
    uint32_t x = v;
    x = umax(x, dpp_from<kDppRowShr + 1u>(0u, x));
    x = umax(x, dpp_from<kDppRowShr + 2u>(0u, x));
    x = umax(x, dpp_from<kDppRowShr + 4u>(0u, x));
    x = umax(x, dpp_from<kDppRowShr + 8u>(0u, x));
    x = umax(x, dpp_from<kDppBcast15, 0xau>(0u, x));
    x = umax(x, dpp_from<kDppBcast31, 0xcu>(0u, x));
    total = umax(rdlane(x, 63), first);
    return umax(dpp_from<kDppWaveShr1>(first, x), first);
}

// number of set bits of m below this lane
__device__ __forceinline__ uint32_t bits_below_lane(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#endif
}  // namespace cj
