// Host build of the Snappy record grammar used by the GPU parse kernel and the LDS decoder's record expansion
// (cramjam_amd/csrc/snappy_records.hpp): decode a raw block by applying the records, with the kernels' prologue.
#include "snappy_records.hpp"
#include <cstring>
extern "C" int64_t sim_snappy_decode(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint32_t* nrec_out) {
    if (n == 0) return -8;
    uint64_t ulen = 0; uint32_t shift = 0, i = 0, hdr = 0; bool ok = false;
    while (hdr < n && i < 10) {
        uint32_t b = in[hdr]; hdr++;
        if (b < 0x80) { if (!(i == 9 && b > 1)) { ulen |= (uint64_t)b << shift; ok = true; } break; }
        ulen |= (uint64_t)(b & 0x7f) << shift; shift += 7; i++;
    }
    if (!ok) return -9;
    if (ulen > 0xFFFFFFFFull) return -10;
    if (ulen > cap) return -11;
    const uint8_t* s = in + hdr;
    const uint32_t iend = (uint32_t)(n - hdr), dn = (uint32_t)ulen;
    auto rd = [s, iend](uint32_t p) { uint32_t v = 0; for (uint32_t k = 0; k < 4 && p + k < iend; k++) v |= (uint32_t)s[p + k] << (8 * k); return v; };
    uint32_t ip = 0, op = 0, nrec = 0;
    while (ip < iend) {
        cj::SnRecord rec;
        const uint32_t op0 = op;
        if (cj::snappy_record_step(rd, ip, op, iend, dn, rec) != 0) return -12;
        nrec++;
        if (rec.lit_len) { if (rec.dst - rec.lit_len != op0) return -1000; std::memcpy(out + op0, s + rec.lit_src, rec.lit_len); }
        const uint32_t off = rec.w & 0xffffu, m = rec.w >> 16;
        if (m) { if (dn > 65536) return -1001; for (uint32_t k = 0; k < m; k++) out[rec.dst + k] = out[rec.dst + k - off]; }
        if (rec.dst + m != op) return -1002;
    }
    if (nrec_out) *nrec_out = nrec;
    return op == dn ? (int64_t)dn : -12;
}
