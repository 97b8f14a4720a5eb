// snappy_records.hpp — one step of the Snappy raw element grammar, shared by the lane-per-chunk parse kernel, the LDS
// decoder's record expansion and the host simulation (tests/hostsim).  A RECORD is an optional literal element followed
// by an optional copy element — the same shape as an LZ4 sequence, so the LDS decoder's literal (D2) and match (D3)
// phases are codec independent.  Accept/reject rules are snap 1.1.1's raw::Decoder (same as snappy_decode.hip;
// reference call sites /root/reference/src/snappy.rs:57,106).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CJ_SN_HD __host__ __device__ __forceinline__
#else
#define CJ_SN_HD inline
#endif

namespace cj {

struct SnRecord { uint32_t lit_src, lit_len, dst, w; };     // dst = output position after the literal; w = offset | copy length << 16

// rd(p): 4 bytes (little endian) at stream offset p, zero-filled past iend.  ip/op advance over one record.
// Returns 0, or -1 for any malformed element (snap: Literal / CopyRead / CopyWrite / Offset errors).
template <class Rd>
CJ_SN_HD int snappy_record_step(const Rd& rd, uint32_t& ip, uint32_t& op, uint32_t iend, uint32_t dn, SnRecord& rec) {
    uint32_t t4 = rd(ip);
    uint32_t tag = t4 & 0xffu;
    rec.lit_src = 0u; rec.lit_len = 0u; rec.w = 0u;
    if ((tag & 3u) == 0u) {
        ip += 1u;
        uint64_t len = (tag >> 2) + 1u;
        if (len > 60u) {
            const uint32_t nb = (uint32_t)len - 60u;        // 1..4 length bytes
            if (iend - ip < nb) return -1;
            uint32_t v = rd(ip);
            if (nb < 4u) v &= (1u << (8u * nb)) - 1u;
            ip += nb;
            len = (uint64_t)v + 1u;
        }
        if (len > (uint64_t)(iend - ip) || len > (uint64_t)(dn - op)) return -1;
        rec.lit_src = ip; rec.lit_len = (uint32_t)len;
        ip += (uint32_t)len; op += (uint32_t)len;
        rec.dst = op;
        if (ip >= iend) return 0;
        t4 = rd(ip);
        tag = t4 & 0xffu;
        if ((tag & 3u) == 0u) return 0;                     // another literal follows: it starts the next record
    }
    rec.dst = op;
    const uint32_t kind = tag & 3u;
    ip += 1u;
    uint32_t len, offset;
    if (kind == 1u) {
        if (iend - ip < 1u) return -1;
        len = 4u + ((tag >> 2) & 7u);
        offset = ((tag >> 5) << 8) | ((t4 >> 8) & 0xffu);
        ip += 1u;
    } else if (kind == 2u) {
        if (iend - ip < 2u) return -1;
        len = 1u + (tag >> 2);
        offset = (t4 >> 8) & 0xffffu;
        ip += 2u;
    } else {
        if (iend - ip < 4u) return -1;
        len = 1u + (tag >> 2);
        offset = rd(ip);
        ip += 4u;
    }
    if (offset == 0u || offset > op) return -1;
    if (len > dn - op) return -1;
    rec.w = (offset & 0xffffu) | (len << 16);               // callers that keep w only take chunks <= 64 KiB (offset <= 65535)
    op += len;
    return 0;
}

}  // namespace cj
