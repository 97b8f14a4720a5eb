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

constexpr uint32_t kHashBits = 12;
constexpr uint32_t kHashSize = 1u << kHashBits;

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

// One probe round over positions [pos, pos+64).  last_start: last position where a match may start
// (needs 4 readable bytes).  Returns the ballot of verified lanes; cand (per lane) is the match source.
__device__ __forceinline__ uint64_t probe_round(const uint8_t* in, uint16_t* ht, uint32_t pos,
                                                uint32_t last_start, uint32_t& cand) {
    const uint32_t my = pos + lane_id();
    const bool valid = my <= last_start;
    uint32_t v = 0, h = 0, c = 0;
    bool ok = false;
    if (valid) {
        v = ld32u(in + my);
        h = (v * 2654435761u) >> (32 - kHashBits);
        c = (my & 0xFFFF0000u) | ht[h];
    }
    // all lanes have read the table before any lane updates it (one instruction stream)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (valid) {
        ht[h] = (uint16_t)my;
        if (c >= my) c -= 65536u;           // slot belongs to the previous 64 KiB lap (or is stale)
        if (c < my && my - c <= 65535u) ok = ld32u(in + c) == v;
    }
    cand = c;
    return ballot64(ok);
}

#endif
}  // namespace cj
