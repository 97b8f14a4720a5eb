// xxh32_host.hpp — XXH32 (the LZ4 frame format's checksum: header byte, optional block checksums, content checksum),
// one-shot and streaming, HOST side.  Written from the published XXH32 algorithm description.  It runs on the host on
// purpose: the algorithm is a serial recurrence of four 32-bit multiply-rotate accumulators per stream.
#pragma once
#include <cstdint>
#include <cstring>

namespace cj {

struct Xxh32 {
    static constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    uint32_t v[4];
    uint64_t total = 0;
    uint8_t tail[16];
    uint32_t ntail = 0;
    uint32_t seed;

    explicit Xxh32(uint32_t s = 0) : seed(s) { v[0] = s + P1 + P2; v[1] = s + P2; v[2] = s; v[3] = s - P1; }
    static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
    static uint32_t rd(const uint8_t* p) { uint32_t x; std::memcpy(&x, p, 4); return x; }
    void stripe(const uint8_t* p) {
        for (int i = 0; i < 4; i++) v[i] = rotl(v[i] + rd(p + 4 * i) * P2, 13) * P1;
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        if (ntail) {
            while (n && ntail < 16) { tail[ntail++] = *p++; n--; }
            if (ntail < 16) return;
            stripe(tail);
            ntail = 0;
        }
        // four independent chains: keep them in registers across the loop
        uint32_t a = v[0], b = v[1], c = v[2], d = v[3];
        while (n >= 16) {
            a = rotl(a + rd(p) * P2, 13) * P1;
            b = rotl(b + rd(p + 4) * P2, 13) * P1;
            c = rotl(c + rd(p + 8) * P2, 13) * P1;
            d = rotl(d + rd(p + 12) * P2, 13) * P1;
            p += 16; n -= 16;
        }
        v[0] = a; v[1] = b; v[2] = c; v[3] = d;
        while (n) { tail[ntail++] = *p++; n--; }
    }
    uint32_t digest() const {
        uint32_t h = total >= 16 ? rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18) : seed + P5;
        h += (uint32_t)total;
        uint32_t i = 0;
        for (; i + 4 <= ntail; i += 4) h = rotl(h + rd(tail + i) * P3, 17) * P4;
        for (; i < ntail; i++) h = rotl(h + (uint32_t)tail[i] * P5, 11) * P1;
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
        return h;
    }
};

inline uint32_t xxh32(const uint8_t* p, size_t n, uint32_t seed = 0) {
    Xxh32 x(seed);
    x.update(p, n);
    return x.digest();
}

}  // namespace cj
