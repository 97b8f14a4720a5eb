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
// The matcher's byte-determinism rests on gfx9 semantics: stores are acknowledged under vmcnt (settle()), DPP row_bcast / wave_shr
// exist, and among the lanes of ONE DS store instruction that hit the same address the highest lane wins (HashTab).  Another
// architecture would compile and give other bytes per block kind — refuse it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "cj_match.hpp: the encoders are written for gfx950 (gfx9 wave64: vmcnt-acknowledged stores, DPP row_bcast, DS same-address store order)"
#endif
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
// slots actually kept per wave: any count <= 2^kHashBits (the 32-bit hash is scaled onto [0, kHashSlots) with a
// multiply-high, so the load stays uniform for counts that are not a power of two)
#ifndef CJ_HASH_SLOTS
#define CJ_HASH_SLOTS (1u << CJ_HASH_BITS)
#endif
constexpr uint32_t kHashSize = CJ_HASH_SLOTS;
__device__ __forceinline__ uint32_t hash_slot(uint32_t v) {
    if constexpr ((kHashSize & (kHashSize - 1u)) == 0u) return (v * 2654435761u) >> (32 - kHashBits);
    else return __umulhi(v * 2654435761u, kHashSize);
}
// waves per encoder block: keep the block's tables within the 64 KiB static-LDS limit
constexpr int kEncWaves = 1;   // one wave per block: 160 KiB / 16 KiB = 10 resident waves per CU (4-wave blocks would round down to 8)
constexpr int kEncThreads = 64 * kEncWaves;

#ifndef CJ_PRE_STEP
#define CJ_PRE_STEP 1u
#endif
constexpr uint32_t kPreStep = CJ_PRE_STEP;

// The hash table of one wavefront: 16-bit positions in LDS, or (kGlobal, the table blocks of large batches) in the block's slot of
// a global scratch array.  The slot is private to the wavefront, but its lanes are different work-items and DS-style in-order
// execution does not hold for global memory: a load may be served before an earlier store of ANOTHER lane to that slot has
// landed (seen on hardware: the first lookups of a chunk read the previous chunk's entries instead of the zeros just stored, and
// copies of a chunk compressed to different bytes depending on which kind of block took them).  So the global table is accessed
// with agent-scope atomics (loads and stores that go to L2, not the CU's vector L1) and `settle()` — wait until the stores are
// acknowledged — separates every group of stores from the lookups that follow.
template <bool kGlobal>
struct HashTab {
    uint16_t* p;
    __device__ __forceinline__ uint32_t get(uint32_t h) const {
        if constexpr (kGlobal) return __hip_atomic_load(p + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return p[h];
    }
    __device__ __forceinline__ void set(uint32_t h, uint32_t v) const {
        if constexpr (kGlobal) __hip_atomic_store(p + h, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[h] = (uint16_t)v;
    }
    __device__ __forceinline__ void clear() const {
        uint32_t* q = reinterpret_cast<uint32_t*>(p);
        for (uint32_t i = lane_id(); i < kHashSize / 2; i += 64u) {
            if constexpr (kGlobal) __hip_atomic_store(q + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else q[i] = 0u;
        }
    }
    // after a group of stores, before the lookups that follow.  Global table: wait until the stores are acknowledged.  LDS table:
    // DS operations of a wavefront execute in order, so only the COMPILER has to be told — clear() stores dwords through a
    // punned pointer, and nothing else keeps the 16-bit lookups from being scheduled above them.
    __device__ __forceinline__ void settle() const {
        if constexpr (kGlobal) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("" ::: "memory");
    }
    // kSplit encoders: index the data BEFORE this wave's sub-piece (positions [0, q0), every kPreStep-th one, ascending so that
    // the most recent position wins a slot), so that the sub-piece finds the matches a serial walk over the piece would
    __device__ __forceinline__ void preindex(const uint8_t* in, uint32_t q0) const {
        for (uint32_t pos = lane_id() * kPreStep; pos < q0; pos += 64u * kPreStep) set(hash_slot(ld32u(in + pos)), pos);
    }
};

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

// ---- per-lane match extension ------------------------------------------------------------------------------
// Every verified lane measures ITS OWN candidate right after the probe — forward up to kLaneFwdCap bytes, backward
// up to 16 — with 16 B vector compares, all lanes in parallel.  The greedy selection that follows then runs on
// registers only (v_readlane), instead of one cooperative extension = several dependent global round trips per
// selected match: the encoders were bound by exactly those round trips (~13 per 64-position round, ~9k cycles).
// Matches that hit a cap (long runs) are finished cooperatively by wave_extend / wave_extend_back.
// (fwd: equal bytes after the 4 verified ones; back: equal bytes before the position / candidate, limited to the
//  pending literal run and to the candidate's own position.)
#ifndef CJ_LANE_FWD_CAP
#define CJ_LANE_FWD_CAP 64
#endif
constexpr uint32_t kLaneFwdCap = CJ_LANE_FWD_CAP;

__device__ __forceinline__ uint4 ld16m(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

__device__ __forceinline__ uint32_t first_diff16(const uint4& x, const uint4& y) {     // index of the first differing byte, 16 if none
    const uint32_t d0 = x.x ^ y.x, d1 = x.y ^ y.y, d2 = x.z ^ y.z, d3 = x.w ^ y.w;
    if (d0) return (uint32_t)__builtin_ctz(d0) >> 3;
    if (d1) return 4u + ((uint32_t)__builtin_ctz(d1) >> 3);
    if (d2) return 8u + ((uint32_t)__builtin_ctz(d2) >> 3);
    if (d3) return 12u + ((uint32_t)__builtin_ctz(d3) >> 3);
    return 16u;
}

__device__ __forceinline__ uint32_t last_same16(const uint4& x, const uint4& y) {      // equal bytes counted from the END of the 16, 16 if all
    const uint32_t d0 = x.x ^ y.x, d1 = x.y ^ y.y, d2 = x.z ^ y.z, d3 = x.w ^ y.w;
    if (d3) return (uint32_t)__builtin_clz(d3) >> 3;
    if (d2) return 4u + ((uint32_t)__builtin_clz(d2) >> 3);
    if (d1) return 8u + ((uint32_t)__builtin_clz(d1) >> 3);
    if (d0) return 12u + ((uint32_t)__builtin_clz(d0) >> 3);
    return 16u;
}

// One probe ROUND covers kSub x 64 consecutive positions: lane l owns positions pos + 64 j + l, j < kSub.  All kSub
// sub-rounds are probed against the table as it was at the start of the round, and every kind of memory access is
// issued for all sub-rounds before the first result is used — the round costs the same ~4 dependent global round
// trips as a 64-position round did (own dwords, candidate dwords, extension blocks, literal sources of the emitted
// sequences), which is what bounds these kernels (10 waves per CU: the 16 KiB table per wave fills the LDS).
// Probing a round of positions against a table that is up to a round stale loses < 1 % of ratio on the benchmark
// data (256 positions, simulated: 1.650 -> 1.645; measured 256 / 320 / 384 positions: 1.628 / 1.625 / 1.623) — near repeats are
// still found through older table entries and the backward extension; the caller starts a chunk with short rounds while
// the table is empty.  Round size: kSub = 5 (320 positions) since the selection left the serial chain — 4: 81 GB/s, 5: 87,
// 6: 68 (past 168 registers only two wavefronts fit a SIMD).  last_start: last position where a match may start (needs 4 readable bytes); limit: a match must end
// here at the latest; anchor: start of the pending literal run.  The table is NOT updated here: the caller inserts,
// after it has consumed the ballots, only the positions that did not end up inside an emitted match
// (insert_uncovered).  Positions inside a match repeat content whose source is already indexed; inserting them
// too evicts distant sources from the small table ~4x faster on match-heavy data (measured on synth-v1: ratio
// 1.37 with dense insertion vs the CPU encoder's 1.63).
constexpr uint32_t kNoSlot = 0xffffffffu;
#ifndef CJ_ENC_SUB
#define CJ_ENC_SUB 5
#endif
constexpr int kSub = CJ_ENC_SUB;
constexpr uint32_t kRoundPositions = 64u * kSub;

struct Round {
    uint32_t cand[kSub], hslot[kSub];
    uint32_t ext[kSub];          // fwd | back << 8 | fwd_more << 14 | back_more << 15 (per lane)
    uint64_t mask[kSub];         // verified lanes of each sub-round (wave-uniform)
};

// the round's own dwords (position pos + 64 j + lane); the caller may request them one round ahead (OwnDwords::load at the
// expected next position while the current round is selected and emitted: one of the round's four dependent round trips)
struct OwnDwords {
    uint32_t v[kSub];
    uint32_t pos;
    __device__ __forceinline__ void load(const uint8_t* in, uint32_t p, uint32_t last_start) {
        pos = p;
#pragma unroll
        for (int j = 0; j < kSub; j++) {
            const uint32_t my = p + 64u * j + lane_id();
            v[j] = 0u;
            if (my <= last_start) v[j] = ld32u(in + my);
        }
    }
};

template <bool kGlobal>
__device__ __forceinline__ void probe_round(const uint8_t* in, const HashTab<kGlobal>& ht, uint32_t pos, uint32_t last_start,
                                            uint32_t limit, uint32_t anchor, Round& r, const OwnDwords& own) {
    const uint32_t lane = lane_id();
    uint32_t v[kSub];
    bool ok[kSub];
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        v[j] = own.v[j];
        r.hslot[j] = kNoSlot;
    }
    // table lookups, then the candidate dwords: kSub divergent loads in flight
    uint32_t w[kSub];
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        const uint32_t my = pos + 64u * j + lane;
        uint32_t c = 0;
        ok[j] = false;
        if (my <= last_start) {
            const uint32_t h = hash_slot(v[j]);
            r.hslot[j] = h;
            c = (my & 0xFFFF0000u) | ht.get(h);
            if (c >= my) c -= 65536u;       // slot belongs to the previous 64 KiB lap (or is stale)
            ok[j] = c < my && my - c <= 65535u;
        }
        r.cand[j] = c;
        w[j] = 0u;
        if (ok[j]) w[j] = ld32u(in + c);
    }
    // first forward block and the backward block of every verified candidate: up to 4 kSub vector loads in flight
    uint4 fx[kSub], fy[kSub], bx[kSub], by[kSub];
    bool f16[kSub], b16[kSub];
    uint32_t blim[kSub];
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        const uint32_t my = pos + 64u * j + lane, c = r.cand[j];
        ok[j] = ok[j] && w[j] == v[j];
        const uint32_t room = my > anchor ? my - anchor : 0u;
        blim[j] = room < c ? room : c;
        f16[j] = ok[j] && my + 20u <= limit;
        b16[j] = ok[j] && c >= 16u && blim[j] > 0u;
        fx[j] = fy[j] = bx[j] = by[j] = make_uint4(0, 0, 0, 0);
        if (f16[j]) { fx[j] = ld16m(in + my + 4u); fy[j] = ld16m(in + c + 4u); }
        if (b16[j]) { bx[j] = ld16m(in + my - 16u); by[j] = ld16m(in + c - 16u); }
    }
    // first blocks -> lengths; lanes whose first 16 bytes all matched continue, ALL sub-rounds together per extra
    // round trip (a per-sub-round loop would serialise up to kSub x 3 more round trips)
    uint32_t fwd[kSub], back[kSub];
    bool more[kSub], back_more[kSub];
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        const uint32_t my = pos + 64u * j + lane, c = r.cand[j];
        fwd[j] = 0u; back[j] = 0u; more[j] = false; back_more[j] = false;
        if (ok[j]) {
            const uint32_t a = my + 4u, b = c + 4u;
            if (f16[j]) {
                fwd[j] = first_diff16(fx[j], fy[j]);
                more[j] = fwd[j] == 16u;
            } else {
                while (a + fwd[j] < limit && in[a + fwd[j]] == in[b + fwd[j]]) fwd[j] += 1u;   // within 16 bytes of the limit: chunk tail only
            }
            if (b16[j]) {
                const uint32_t sm = last_same16(bx[j], by[j]);
                back[j] = sm < blim[j] ? sm : blim[j];
                back_more[j] = sm == 16u && blim[j] > 16u;
            } else {
                while (back[j] < blim[j] && in[my - 1u - back[j]] == in[c - 1u - back[j]]) back[j] += 1u;   // candidate in the chunk's first 16 bytes
            }
        }
    }
    for (uint32_t it = 1; it < kLaneFwdCap / 16u; it++) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < kSub; j++) any = any || more[j];
        if (ballot64(any) == 0ull) break;
#pragma unroll
        for (int j = 0; j < kSub; j++) {        // loads of all sub-rounds first ...
            const uint32_t a = pos + 64u * j + lane + 4u + fwd[j];
            f16[j] = more[j] && a + 16u <= limit;
            if (f16[j]) { fx[j] = ld16m(in + a); fy[j] = ld16m(in + r.cand[j] + 4u + fwd[j]); }
        }
#pragma unroll
        for (int j = 0; j < kSub; j++) {        // ... then the compares
            if (!more[j]) continue;
            const uint32_t a = pos + 64u * j + lane + 4u, b = r.cand[j] + 4u;
            if (f16[j]) {
                const uint32_t d = first_diff16(fx[j], fy[j]);
                fwd[j] += d;
                more[j] = d == 16u;
            } else {
                while (a + fwd[j] < limit && in[a + fwd[j]] == in[b + fwd[j]]) fwd[j] += 1u;
                more[j] = false;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        const uint32_t a = pos + 64u * j + lane + 4u;
        const bool fwd_more = more[j] && a + fwd[j] < limit;
        r.ext[j] = fwd[j] | (back[j] << 8) | ((uint32_t)fwd_more << 14) | ((uint32_t)back_more[j] << 15);
        r.mask[j] = ballot64(ok[j]);
    }
}

// the selected lane's measurements -> final (mpos, mc, mlen) of the match, finishing capped extensions cooperatively
__device__ __forceinline__ void finish_match(const uint8_t* in, uint32_t ext_lane, uint32_t first, uint32_t anchor, uint32_t limit,
                                             uint32_t& mpos, uint32_t& mc, uint32_t& mlen) {
    const uint32_t e = rdlane(ext_lane, first);
    mlen = 4u + (e & 0xffu);
    if (e & (1u << 14)) mlen += wave_extend(in, mpos + mlen, mc + mlen, limit);
    const uint32_t room = mpos - anchor;
    uint32_t back = (e >> 8) & 0x3fu;
    if (back > room) back = room;
    else if ((e & (1u << 15)) && back == 16u) back += wave_extend_back(in, mpos - 16u, mc - 16u, room - 16u);
    mpos -= back; mc -= back; mlen += back;
}

// ---- lane-parallel greedy selection ---------------------------------------------------------------------------
// The greedy leftmost rule is a serial chain over the selected matches, but only ONE thing has to be carried along
// it: where the previous selected match ended.  select_walk does exactly that (≈12 instructions per match: ctz,
// one v_readlane, a predicated move); everything that used to ride on the chain (backward clamp, literal length,
// encoded size, output position, queue slot, coverage) is computed afterwards for all candidates at once — sizes
// and coverage through wave prefix scans, the queue through ds_permute.  Rounds that contain a capped extension
// (very long match) or need whole-wave emission (long literal run) fall back to the serial loop.
struct Selection {
    uint64_t sel[kSub];          // selected lanes per sub-round (wave-uniform)
    uint32_t prev_end[kSub];     // per lane: end of the previous selected match (start of this sequence's literals)
    uint32_t out_pos[kSub];      // per lane: output position of this sequence
    bool covered[kSub];          // per lane: position lies inside a selected match (not its first byte)
    uint32_t anchor;             // end of the last selected match (wave-uniform)
    uint32_t op;                 // output position after the last selected sequence (wave-uniform)
    uint32_t count;
    bool coop;                   // a selected sequence needs whole-wave handling: the caller redoes the round serially
};

// Wave scans over the 64 lanes with DPP moves (one VALU instruction per step, no LDS): row_shr / row_shl 1, 2, 4, 8 scan the
// rows of 16 lanes; across rows the prefix scans use row_bcast:15 / row_bcast:31 (gfx9 family), the suffix scan three
// v_readlane.  A lane whose DPP source lies outside its row (or whose row is masked off) receives `identity`.
template <uint32_t kCtrl, uint32_t kRowMask = 0xfu>
__device__ __forceinline__ uint32_t dpp_from(uint32_t identity, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, (int)kCtrl, (int)kRowMask, 0xf, false);
}
constexpr uint32_t kDppRowShr = 0x110u, kDppRowShl = 0x100u, kDppBcast15 = 0x142u, kDppBcast31 = 0x143u, kDppWaveShr1 = 0x138u;

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
// max over the lanes BELOW this one (first = the value for lane 0); total = max over first and all 64 lanes
__device__ __forceinline__ uint32_t wave_excl_max(uint32_t v, uint32_t first, uint32_t& total) {
    uint32_t x = v;
    x = umax(x, dpp_from<kDppRowShr + 1u>(0u, x));
    x = umax(x, dpp_from<kDppRowShr + 2u>(0u, x));
    x = umax(x, dpp_from<kDppRowShr + 4u>(0u, x));
    x = umax(x, dpp_from<kDppRowShr + 8u>(0u, x));
    x = umax(x, dpp_from<kDppBcast15, 0xau>(0u, x));
    x = umax(x, dpp_from<kDppBcast31, 0xcu>(0u, x));
    total = umax(rdlane(x, 63), first);
    return umax(dpp_from<kDppWaveShr1>(first, x), first);              // lane 0 receives `first`
}
// min over this lane and the lanes ABOVE it; total = min over all 64 lanes
__device__ __forceinline__ uint32_t wave_suffix_min(uint32_t v, uint32_t& total) {
    constexpr uint32_t kNone = 0xffffffffu;
    uint32_t x = v;
    x = umin(x, dpp_from<kDppRowShl + 1u>(kNone, x));
    x = umin(x, dpp_from<kDppRowShl + 2u>(kNone, x));
    x = umin(x, dpp_from<kDppRowShl + 4u>(kNone, x));
    x = umin(x, dpp_from<kDppRowShl + 8u>(kNone, x));                  // lane 16 r = min of row r
    const uint32_t r1 = rdlane(x, 16), r2 = rdlane(x, 32), r3 = rdlane(x, 48);
    const uint32_t m3 = r3, m2 = umin(r2, r3), m1 = umin(r1, m2);      // min over the rows from 3 / 2 / 1 up
    const uint32_t row = lane_id() >> 4;
    const uint32_t above = row == 0u ? m1 : row == 1u ? m2 : row == 2u ? m3 : kNone;
    total = umin(rdlane(x, 0), m1);
    return umin(x, above);
}

// size_fn(lit, mcode, off) = encoded bytes of one sequence; coop_fn(lit, mcode) = needs whole-wave emission
template <class SizeFn, class CoopFn>
__device__ __forceinline__ void select_walk(const Round& r, uint32_t pos, uint32_t anchor, uint32_t op, Selection& s,
                                            SizeFn size_fn, CoopFn coop_fn) {
    const uint32_t lane = lane_id();
    // ---- the serial part: which candidates are selected and where the previous selected match ended ----
    uint32_t cur = anchor, cnt = 0;
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        const uint32_t pj = pos + 64u * j;
        const uint32_t e_lane = pj + lane + 4u + (r.ext[j] & 0xffu);       // end of this lane's match if it is selected
        uint64_t mask = r.mask[j], sel = 0ull;
        uint32_t pe = 0;
        if (cur > pj) mask = cur - pj >= 64u ? 0ull : mask & (~0ull << (cur - pj));
        while (mask) {
            const uint32_t first = ctz64(mask);
            const uint32_t e = rdlane(e_lane, first);
            sel |= 1ull << first;
            pe = lane == first ? cur : pe;
            cur = e; cnt += 1;
            mask = e - pj >= 64u ? 0ull : mask & (~0ull << (e - pj));
        }
        s.sel[j] = sel;
        s.prev_end[j] = pe;
    }
    s.anchor = cur; s.count = cnt;
    // ---- everything else for all selected candidates at once: backward clamp, sizes -> output positions (prefix sum),
    //      coverage = inside [start + 1, end) of a selected match (prefix max of the ends, suffix min of the starts) ----
    uint32_t start[kSub];
    uint32_t run_op = op, run_end = anchor;
    bool coop = false;
#pragma unroll
    for (int j = 0; j < kSub; j++) {
        const uint32_t p = pos + 64u * j + lane, ext = r.ext[j];
        const bool sel = ((s.sel[j] >> lane) & 1ull) != 0ull;
        const uint32_t room = p - s.prev_end[j];
        uint32_t bk = (ext >> 8) & 0x3fu;
        bk = bk < room ? bk : room;
        const uint32_t lit = room - bk, mcode = (ext & 0xffu) + bk;
        // capped extensions: forward -> the match end above is too short; backward -> only if the literal run leaves room
        coop = coop || (sel && (coop_fn(lit, mcode) || (ext & 0x4000u) != 0u || ((ext & 0x8000u) != 0u && room > 16u)));
        uint32_t total;
        const uint32_t before = wave_excl_add(sel ? size_fn(lit, mcode, p - r.cand[j]) : 0u, total);
        s.out_pos[j] = run_op + before;
        run_op += total;
        uint32_t top;
        const uint32_t end_before = wave_excl_max(sel ? p + 4u + (ext & 0xffu) : 0u, run_end, top);     // ends of the selected matches at lower positions (and of earlier rounds)
        run_end = top;
        s.covered[j] = p < end_before;
        start[j] = sel ? p - bk : 0xffffffffu;
    }
    uint32_t run_start = 0xffffffffu;
#pragma unroll
    for (int j = kSub - 1; j >= 0; j--) {
        const uint32_t p = pos + 64u * j + lane;
        uint32_t low;
        const uint32_t st = wave_suffix_min(start[j], low);              // start of the next selected match at this or a higher position
        s.covered[j] = s.covered[j] || p > (st < run_start ? st : run_start);
        run_start = low < run_start ? low : run_start;
    }
    s.op = run_op;
    s.coop = ballot64(coop) != 0ull;
}

// number of set bits of m below this lane
__device__ __forceinline__ uint32_t bits_below_lane(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// push `v` of the selected lanes of a sub-round to queue lanes base + rank; the other lanes push to the remaining lanes
// (a permutation, so nothing collides); lanes whose queue slot belongs to this sub-round take the received value
__device__ __forceinline__ uint32_t queue_push(uint32_t q, uint32_t v, uint32_t dest, uint32_t base, uint32_t cnt) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)v);
    const uint32_t lane = lane_id();
    return (lane >= base && lane < base + cnt) ? t : q;
}

// exact n-byte copy by ONE lane: 64 B batches with the four loads in flight together (a load -> store loop costs one
// global round trip per 16 bytes), then 16 B blocks and an 8/4/2/1 tail: lane-parallel emission of short literal runs
__device__ __forceinline__ void lane_copy_exact(uint8_t* dst, const uint8_t* src, uint32_t n) {
    uint32_t k = 0;
    for (; k + 64u <= n; k += 64u) {
        const uint4 t0 = ld16m(src + k), t1 = ld16m(src + k + 16u), t2 = ld16m(src + k + 32u), t3 = ld16m(src + k + 48u);
        __builtin_memcpy(dst + k, &t0, 16); __builtin_memcpy(dst + k + 16u, &t1, 16);
        __builtin_memcpy(dst + k + 32u, &t2, 16); __builtin_memcpy(dst + k + 48u, &t3, 16);
    }
    const uint32_t rem = n - k;                     // < 64: up to three 16 B blocks + tail, all loads first
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0, t2 = t0;
    uint2 t8 = make_uint2(0, 0);
    uint32_t t4 = 0, t21 = 0;
    const uint32_t nb = rem >> 4;
    if (nb > 0u) t0 = ld16m(src + k);
    if (nb > 1u) t1 = ld16m(src + k + 16u);
    if (nb > 2u) t2 = ld16m(src + k + 32u);
    uint32_t q = k + 16u * nb;
    if (rem & 8u) { __builtin_memcpy(&t8, src + q, 8); q += 8u; }
    if (rem & 4u) { __builtin_memcpy(&t4, src + q, 4); q += 4u; }
    if (rem & 2u) { t21 = (uint32_t)src[q] | ((uint32_t)src[q + 1u] << 8); q += 2u; }
    if (rem & 1u) t21 |= (uint32_t)src[q] << 16;
    if (nb > 0u) __builtin_memcpy(dst + k, &t0, 16);
    if (nb > 1u) __builtin_memcpy(dst + k + 16u, &t1, 16);
    if (nb > 2u) __builtin_memcpy(dst + k + 32u, &t2, 16);
    q = k + 16u * nb;
    if (rem & 8u) { __builtin_memcpy(dst + q, &t8, 8); q += 8u; }
    if (rem & 4u) { __builtin_memcpy(dst + q, &t4, 4); q += 4u; }
    if (rem & 2u) { dst[q] = (uint8_t)t21; dst[q + 1u] = (uint8_t)(t21 >> 8); q += 2u; }
    if (rem & 1u) dst[q] = (uint8_t)(t21 >> 16);
}

// covered (per lane): the lane's position lies strictly inside an emitted match (not its first byte).  Tracked with
// two VALU compares per sub-round and match — the scalar unit is the busiest pipe of these kernels (one per CU,
// shared by all waves), 64-bit mask arithmetic there cost ~60 scalar instructions per match.
template <bool kGlobal>
__device__ __forceinline__ void insert_uncovered(const HashTab<kGlobal>& ht, uint32_t pos, uint32_t hslot, bool covered) {
    if (hslot != kNoSlot && !covered) ht.set(hslot, pos + lane_id());
}
// All insertions of a round (sub-round j: positions pos + 64 j + lane).  Several lanes of a round may hash to the same slot; the
// table must end up with the HIGHEST of their positions ("the most recent position wins").  In LDS that is what happens: a DS
// write with equal addresses keeps the highest lane's data, and the sub-rounds are written in ascending order.  Global stores
// make no such promise, so a table in global memory is read back and every lane whose position is higher than the one it finds
// writes again, until nothing changes (a slot is contested by a handful of lanes at most: one or two passes).
template <bool kGlobal>
__device__ __forceinline__ void insert_round(const HashTab<kGlobal>& ht, uint32_t pos, const uint32_t (&hslot)[kSub], const bool (&covered)[kSub]) {
#pragma unroll
    for (int j = 0; j < kSub; j++) insert_uncovered(ht, pos + 64u * j, hslot[j], covered[j]);
    ht.settle();
    if constexpr (kGlobal) {
        for (;;) {
            bool again = false;
#pragma unroll
            for (int j = 0; j < kSub; j++) {
                if (hslot[j] != kNoSlot && !covered[j]) {
                    const uint32_t mine = 64u * j + lane_id();                          // relative to pos: < kRoundPositions
                    const uint32_t there = (ht.get(hslot[j]) - pos) & 0xffffu;          // the slot holds a position of this round
                    if (there < mine) { ht.set(hslot[j], pos + mine); again = true; }
                }
            }
            if (ballot64(again) == 0ull) break;
            ht.settle();
        }
    }
}

// ---- persistent encoder blocks ------------------------------------------------------------------------------
// One wavefront per block, chunks from a shared counter.  kGlobalTable = false: hash table in LDS — ten such blocks fill a
// CU's 160 KiB and leave 6 of its 16 wave slots (at <= 128 VGPRs) empty; true: hash table in the block's slot of a global
// scratch array (L2-resident), no LDS — these blocks take the empty slots.  Alone, a wavefront with its table in L2 is as
// fast as one with an LDS table (7.9 vs 8.2 GB/s, 3 resp. 5 per CU); per-wavefront rates fall as the CU fills (LZ4: 10 LDS
// wavefronts 66 GB/s, + 3 global 75 GB/s, + 6 global 75 GB/s), so three per CU are launched.  Enc::chunk<kGlobal>(a, c, table) =
// one chunk; HashTab<true> makes the global table behave exactly like the LDS one (same bytes out whichever block takes a chunk).
static_assert(kHashSize * 2u <= kEncTableBytes, "table slot");
#ifndef CJ_ENC_TABLE_WAVES_PER_EU
#define CJ_ENC_TABLE_WAVES_PER_EU 3
#endif
template <class Enc, bool kGlobalTable>
__device__ __forceinline__ void encode_persistent_body(const BatchArgs& a, uint32_t* counter, uint16_t* ht) {
    for (;;) {
        uint32_t c = 0;
        if (threadIdx.x == 0) c = atomicAdd(counter, 1u);
        const uint32_t chunk = uni(c);                               // lane 0's value (one wavefront per block)
        if (chunk >= a.n_chunks) return;
        Enc::template chunk<kGlobalTable>(a, chunk, HashTab<kGlobalTable>{ht});
    }
}
// (four wavefronts per SIMD as the register target.  The compiler notes that 16 KiB of LDS per block allow only 2.5 per SIMD —
//  -Wpass-failed, expected — and takes up to 168 registers; measured alternatives: the table as DYNAMIC shared memory keeps the
//  target, 128 registers, and is 10 % slower (68 vs 77 GB/s); amdgpu_num_vgpr is not honoured here)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
#ifndef CJ_ENC_LDS_WAVES_PER_EU
#define CJ_ENC_LDS_WAVES_PER_EU 4
#endif
template <class Enc>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CJ_ENC_LDS_WAVES_PER_EU, CJ_ENC_LDS_WAVES_PER_EU))) void encode_lds_blocks_kernel(BatchArgs a, uint32_t* counter) {
    __shared__ uint16_t ht_lds[kHashSize];                           // exactly 16 KiB: one more word and only nine blocks fit a CU
    encode_persistent_body<Enc, false>(a, counter, ht_lds);
}
#pragma clang diagnostic pop
template <class Enc>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CJ_ENC_TABLE_WAVES_PER_EU, CJ_ENC_TABLE_WAVES_PER_EU)))
void encode_table_blocks_kernel(BatchArgs a, uint32_t* counter, uint16_t* tables) {
    encode_persistent_body<Enc, true>(a, counter, tables + (size_t)blockIdx.x * (kEncTableBytes / 2u));
}

template <class Enc>
inline void launch_encode_filled(const BatchArgs& a, hipStream_t s, const EncFill& f) {
    (void)hipMemsetAsync(f.counter, 0, 4, s);
    (void)hipEventRecord(f.fork, s);
    hipLaunchKernelGGL((encode_lds_blocks_kernel<Enc>), dim3(f.lds_blocks), dim3(64), 0, s, a, f.counter);
    if (f.table_blocks == 0u) return;
    (void)hipStreamWaitEvent(f.aux, f.fork, 0);
    hipLaunchKernelGGL((encode_table_blocks_kernel<Enc>), dim3(f.table_blocks), dim3(64), 0, f.aux, a, f.counter, f.tables);
    (void)hipEventRecord(f.join, f.aux);
    (void)hipStreamWaitEvent(s, f.join, 0);
}

#endif
}  // namespace cj
