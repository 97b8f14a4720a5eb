// Host build of the lane-parallel CRC-32C (cramjam_amd/csrc/crc32c_lanes.hpp): XOR of the 64 lane shares.
#include "crc32c_lanes.hpp"
#include <cstring>
static constexpr cj::Crc32cTables T = cj::make_crc32c_tables();
extern "C" uint32_t sim_crc32c(const uint8_t* p, uint32_t len) {
    uint32_t x = 0;
    for (uint32_t l = 0; l < 64; l++)
        x ^= cj::crc32c_lane(p, len, l, &T.adv256[0][0], T.xpow8, [](const uint8_t* q) { uint32_t v; std::memcpy(&v, q, 4); return v; });
    return ~x;
}
extern "C" uint32_t sim_crc32c_mask(uint32_t c) { return cj::crc32c_mask(c); }
