// crc32c_lanes.hpp — CRC-32C (Castagnoli, reflected 0x82F63B78) of one piece by 64 lanes with coalesced loads.
//
// Needed by the Snappy framing format (framing_format.txt §3: every data chunk carries the masked CRC-32C of its
// UNCOMPRESSED bytes); the reference reaches it through snap 1.1.1's frame.rs `crc32c_masked`
// (/root/reference/src/snappy.rs:24,38 -> libcramjam::snappy::{compress,decompress}).
//
// A CRC is linear over GF(2): the register after feeding n zero bytes from state v is v * x^(8n) mod P.  So
// instead of one serial byte chain per piece, lane l owns dwords l, l+64, l+128, ... (one coalesced 256 B wave
// load per step) and treats all other bytes as zero:
//     state = ADV256(state ^ dword)            // 4 table lookups: "advance 256 bytes", the wave's stride
// and its last (possibly partial) dword is advanced to the end of the piece by one GF(2) multiplication with
// x^(8 * bytes_to_end).  The XOR of the 64 lane results is the CRC register.  Lane 0 starts from 0xFFFFFFFF
// (the CRC's init), the others from 0.
//
// Everything here is plain integer code usable on host and device: tests/hostsim runs crc32c_lane() for
// lanes 0..63 against a bytewise CRC.  Tables are generated at compile time.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CJ_HD __host__ __device__
#else
#define CJ_HD
#endif

namespace cj {

constexpr uint32_t kCrc32cPoly = 0x82F63B78u;

// a * b mod P in the reflected representation (x^0 is bit 31)
CJ_HD constexpr uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? kCrc32cPoly : 0u);
    }
    return p;
}

constexpr uint32_t kCrcTailEntries = 260;   // x^(8n), n = 0 .. 259 (a lane's last dword starts at most 259 bytes before the end)

struct Crc32cTables {
    uint32_t adv256[4][256];            // adv256[j][b] = (b << 8j) * x^(8*256)
    uint32_t xpow8[kCrcTailEntries];    // x^(8n) mod P
};

constexpr Crc32cTables make_crc32c_tables() {
    Crc32cTables t{};
    t.xpow8[0] = 0x80000000u;                         // x^0
    for (uint32_t n = 1; n < kCrcTailEntries; n++) t.xpow8[n] = gf_mul(t.xpow8[n - 1], 0x00800000u);   // * x^8
    const uint32_t m = t.xpow8[256];
    for (uint32_t j = 0; j < 4; j++)
        for (uint32_t b = 0; b < 256; b++) t.adv256[j][b] = gf_mul(b << (8u * j), m);
    return t;
}

// One lane's share.  p/len: the piece; adv: 4x256 table (LDS on the device); xpow8: tail multipliers.
// Returns the lane's contribution to the (un-inverted) CRC register.
template <class Ld32>
CJ_HD inline uint32_t crc32c_lane(const uint8_t* p, uint32_t len, uint32_t lane, const uint32_t* adv,
                                  const uint32_t* xpow8, Ld32 ld32) {
    uint32_t s = lane == 0u ? 0xFFFFFFFFu : 0u;
    uint32_t pos = 4u * lane;
    while (pos + 4u <= len) {
        const uint32_t v = s ^ ld32(p + pos);
        const uint32_t next = pos + 256u;
        if (next >= len) return gf_mul(v, xpow8[len - pos]);          // last dword of this lane: 4 .. 259 bytes to the end
        s = adv[v & 0xffu] ^ adv[256u + ((v >> 8) & 0xffu)] ^ adv[512u + ((v >> 16) & 0xffu)] ^ adv[768u + (v >> 24)];
        pos = next;
    }
    if (pos < len) {                                                  // 1..3 trailing bytes
        uint32_t d = 0;
        for (uint32_t k = 0; k < len - pos; k++) d |= (uint32_t)p[pos + k] << (8u * k);
        return gf_mul(s ^ d, xpow8[len - pos]);
    }
    return s;                                                         // lane owns nothing (only when len <= 4*lane)
}

CJ_HD inline uint32_t crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

}  // namespace cj
