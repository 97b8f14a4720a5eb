// parse_grammar.hpp — the two element grammars of the parallel parse kernels (parse_spec.hip: one chunk per
// wavefront; big_parse.hip: one large stream cut into pieces).  `at` reads the stream through rd(p) = the 4 bytes at
// stream offset p (little endian; bytes past the end may be anything — every length is bounds-checked against iend).
#pragma once
#include "cj_common.hpp"
#include <type_traits>

namespace cj {

constexpr uint32_t kPosEnd = 0xFFFFFFFEu;                // the walk consumed the input exactly (last sequence)
constexpr uint32_t kPosErr = 0xFFFFFFFFu;                // the walk ran into a malformed field

// One sequence / record at position ip (relative to the start of the element stream).  `at` computes, FROM THE INPUT
// BYTES ONLY, the literal length, the match length (0 = none), the offset and the position of the next one; it returns
// false when a field or the item itself runs past the input (malformed on whatever path it lies).  `check` applies the
// decoder's rules that need the output position (phase 4).
struct Seq { uint32_t lit, mlen, offset, next, lit_at; bool last; };      // lit_at = position of the literal bytes

// number of 0xFF bytes at g[ip..] counted in whole 64-byte steps (the caller's byte loop finishes the run); stays 16 bytes
// clear of iend
__device__ __forceinline__ uint32_t count_ff(const uint8_t* g, uint32_t ip, uint32_t iend) {
    uint32_t k = 0;
    while (ip + k + 80u <= iend) {
        uint4 a, b, c, d;
        __builtin_memcpy(&a, g + ip + k, 16); __builtin_memcpy(&b, g + ip + k + 16u, 16);
        __builtin_memcpy(&c, g + ip + k + 32u, 16); __builtin_memcpy(&d, g + ip + k + 48u, 16);
        if ((a.x & a.y & a.z & a.w & b.x & b.y & b.z & b.w & c.x & c.y & c.z & c.w & d.x & d.y & d.z & d.w) != 0xFFFFFFFFu) break;
        k += 64u;
    }
    return k;
}

struct Lz4Grammar {
    template <class Rd>
    static __device__ __forceinline__ bool at(const Rd& rd, uint32_t ip, uint32_t iend, Seq& s, const uint8_t* g = nullptr) {
        // g (optional) = global pointer to stream position 0: lets a long run of 0xFF length bytes (a literal run or a match
        // of megabytes: one length byte per 255 bytes) be skipped 64 bytes at a time instead of one dependent read each
        const uint32_t t4 = rd(ip);
        const uint32_t token = t4 & 0xffu;
        ip += 1;
        uint32_t lit = token >> 4;
        if (lit == 15u) {
            if (ip + 15u >= iend) return false;
            uint32_t b = (t4 >> 8) & 0xffu;
            ip += 1; lit += b;
            if (ip + 15u > iend) return false;
            uint32_t streak = 0;
            while (b == 255u) {
                if (g != nullptr && ++streak == 8u) {
                    const uint32_t k = count_ff(g, ip, iend);
                    if (k > 0x00800000u) return false;            // > 2 GiB of literals: no valid block
                    lit += 255u * k; ip += k;
                }
                b = rd(ip) & 0xffu;
                ip += 1; lit += b;
                if (ip + 15u > iend) return false;
            }
        }
        s.lit = lit;
        s.lit_at = ip;
        const uint32_t rem_in = iend - ip;
        if (rem_in < lit + 8u) {                     // can only be the final sequence: it must consume the input exactly
            s.last = true; s.mlen = 0; s.offset = 0; s.next = kPosEnd;
            return rem_in == lit;
        }
        s.last = false;
        ip += lit;
        const uint32_t o4 = rd(ip);
        s.offset = o4 & 0xffffu;
        ip += 2;
        uint32_t mlen = token & 15u;
        if (mlen == 15u) {
            uint32_t b = (o4 >> 16) & 0xffu;
            ip += 1; mlen += b;
            if (ip + 4u > iend) return false;
            uint32_t streak = 0;
            while (b == 255u) {
                if (g != nullptr && ++streak == 8u) {
                    const uint32_t k = count_ff(g, ip, iend);
                    if (k > 0x00800000u) return false;
                    mlen += 255u * k; ip += k;
                }
                b = rd(ip) & 0xffu;
                ip += 1; mlen += b;
                if (ip + 4u > iend) return false;
            }
        }
        s.mlen = mlen + 4u;
        s.next = ip;
        return true;
    }
    // LZ4_decompress_safe's rules with the output capacity (same as lz4_parse_kernel).  Returns false = malformed;
    // `fin` is set when this was the final sequence.
    template <class T>
    static __device__ __forceinline__ bool check(const Seq& s, T& op, T cap, bool& fin) {
        const T rem_out = cap - op;
        fin = false;
        if (s.last || rem_out < s.lit + 12u) {
            // must be the final sequence: consumes the input exactly (`at` checked that when s.last), fits the output
            if (!s.last || rem_out < s.lit) return false;
            op += s.lit; fin = true;
            return true;
        }
        op += s.lit;
        if (s.offset == 0u || s.offset > op) return false;
        if (cap - op < s.mlen + 5u) return false;
        op += s.mlen;
        return true;
    }
    template <class T>
    static __device__ __forceinline__ bool result_ok(T, T) { return true; }
};

// Snappy: a record = optional literal element + optional copy element (snappy_records.hpp); cap = the decoded length dn
struct SnappyGrammar {
    template <class Rd>
    static __device__ __forceinline__ bool at(const Rd& rd, uint32_t ip, uint32_t iend, Seq& s, const uint8_t* = nullptr) {
        uint32_t t4 = rd(ip);
        uint32_t tag = t4 & 0xffu;
        s.lit = 0; s.mlen = 0; s.offset = 0; s.last = false; s.lit_at = ip;
        if ((tag & 3u) == 0u) {
            ip += 1;
            uint64_t len = (tag >> 2) + 1u;
            if (len > 60u) {
                const uint32_t nb = (uint32_t)len - 60u;
                if (iend - ip < nb) return false;
                uint32_t v = rd(ip);
                if (nb < 4u) v &= (1u << (8u * nb)) - 1u;
                ip += nb;
                len = (uint64_t)v + 1u;
            }
            if (len > (uint64_t)(iend - ip)) return false;
            s.lit = (uint32_t)len;
            s.lit_at = ip;
            ip += (uint32_t)len;
            if (ip >= iend) { s.last = true; s.next = kPosEnd; return true; }
            t4 = rd(ip);
            tag = t4 & 0xffu;
            if ((tag & 3u) == 0u) { s.next = ip; return true; }       // another literal follows: it starts the next record
        }
        const uint32_t kind = tag & 3u;
        ip += 1;
        if (kind == 1u) {
            if (iend - ip < 1u) return false;
            s.mlen = 4u + ((tag >> 2) & 7u);
            s.offset = ((tag >> 5) << 8) | ((t4 >> 8) & 0xffu);
            ip += 1;
        } else if (kind == 2u) {
            if (iend - ip < 2u) return false;
            s.mlen = 1u + (tag >> 2);
            s.offset = (t4 >> 8) & 0xffffu;
            ip += 2;
        } else {
            if (iend - ip < 4u) return false;
            s.mlen = 1u + (tag >> 2);
            s.offset = rd(ip);
            ip += 4;
        }
        if (ip >= iend) { s.last = true; s.next = kPosEnd; }
        else s.next = ip;
        return true;
    }
    template <class T>
    static __device__ __forceinline__ bool check(const Seq& s, T& op, T dn, bool& fin) {
        fin = s.last;
        if (s.lit > dn - op) return false;
        op += s.lit;
        if (s.mlen) {
            if (s.offset == 0u || s.offset > op) return false;
            if (s.mlen > dn - op) return false;
            op += s.mlen;
        }
        return true;
    }
    template <class T>
    static __device__ __forceinline__ bool result_ok(T op_end, T dn) { return op_end == dn; }
};

#if defined(__HIPCC__)
// One element at position ip, straight-line: the common LZ4 sequence (length extensions of at most one byte, not within 8 bytes of
// the end) costs two round trips and ~25 instructions; G::at — the general function with its loops — runs only where a lane
// meets anything else (a wavefront walks at the pace of its slowest lane: the general function alone is ~1.4 k cycles per step).
// Used by the walks of the fused parse (lz4_decode_lds.hip) and of the large-stream parse (big_parse.hip).
template <class G, class Rd>
__device__ __forceinline__ bool walk_step(const Rd& rd, uint32_t ip, uint32_t iend, Seq& s, const uint8_t* g = nullptr) {
    if constexpr (std::is_same<G, Lz4Grammar>::value) {
        const uint32_t t4 = rd(ip);
        const uint32_t token = t4 & 0xffu, e1 = (t4 >> 8) & 0xffu;
        const bool x1 = (token >> 4) == 15u;
        const uint32_t lit = (token >> 4) + (x1 ? e1 : 0u);
        const uint32_t ip1 = ip + 1u + (x1 ? 1u : 0u), ip2 = ip1 + lit;
        const uint32_t o4 = rd(ip2 < iend ? ip2 : ip);
        const uint32_t mc = token & 15u, e2 = (o4 >> 16) & 0xffu;
        const bool x2 = mc == 15u;
        // rem_in >= lit + 8 keeps every bound G::at checks while it reads one-byte extensions
        const bool fast = !(x1 && e1 == 255u) && !(x2 && e2 == 255u) && ip1 + 16u <= iend && iend - ip1 >= lit + 8u;
        if (ballot64(!fast) == 0ull) {
            s.lit = lit; s.lit_at = ip1; s.last = false; s.offset = o4 & 0xffffu;
            s.mlen = mc + (x2 ? e2 : 0u) + 4u;
            s.next = ip2 + 2u + (x2 ? 1u : 0u);
            return true;
        }
        bool ok = true;
        if (fast) {
            s.lit = lit; s.lit_at = ip1; s.last = false; s.offset = o4 & 0xffffu;
            s.mlen = mc + (x2 ? e2 : 0u) + 4u;
            s.next = ip2 + 2u + (x2 ? 1u : 0u);
        } else ok = G::at(rd, ip, iend, s, g);
        return ok;
    } else {
        // Snappy record = optional literal element (header of 1 .. 4 bytes) + optional copy element, everything at least 4 bytes clear of
        // the end: the same two round trips (a copy-4 element: a third, for its offset); a 5-byte literal header and the stream's last
        // record take G::at.  (Round 6: a walk from a guessed start meets copy-4 tags and long literal headers at every fourth position —
        // with only the common shapes here 112 of a chunk's 134 wavefront-steps of P1a took G::at for some lane, 3 do now.)
        const uint32_t t4 = rd(ip);
        const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
        const bool is_lit = (tag & 3u) == 0u;
        const uint32_t lhdr = is_lit ? (l6 < 60u ? 1u : l6 - 58u) : 0u;
        const uint32_t lit = is_lit ? (l6 < 60u ? l6 + 1u : ((t4 >> 8) & (0xffffffu >> (8u * (62u - (l6 > 62u ? 62u : l6))))) + 1u) : 0u;
        const uint32_t ip2 = ip + lhdr + lit;
        const bool in2 = ip2 + 4u <= iend && ip2 >= ip;
        const uint32_t c4 = is_lit ? rd(in2 ? ip2 : ip) : t4;
        const uint32_t ctag = c4 & 0xffu, kind = ctag & 3u;
        const uint32_t clen = kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
        uint32_t off = kind == 1u ? ((ctag >> 5) << 8) | ((c4 >> 8) & 0xffu) : (c4 >> 8) & 0xffffu;
        const uint32_t ip3 = kind == 0u ? ip2 : ip2 + (kind == 1u ? 2u : kind == 2u ? 3u : 5u);
        const bool fast = !(is_lit && l6 > 62u) && in2 && ip3 < iend && (kind != 0u || is_lit);
        if (ballot64(fast && kind == 3u) != 0ull) { if (fast && kind == 3u) off = rd(ip2 + 1u); }
        bool ok = true;
        if (fast) {
            s.lit = lit; s.lit_at = ip + lhdr; s.last = false; s.next = ip3;
            s.mlen = kind == 0u ? 0u : clen; s.offset = kind == 0u ? 0u : off;
        } else ok = G::at(rd, ip, iend, s, g);
        return ok;
    }
}

// The same step with ONE dependent read per Snappy record where the stream allows it: the copy element behind a literal is read as
// 8 bytes, and the first 4 bytes of the NEXT record come with it (its tag sits 2 or 3 bytes behind a copy-1 / copy-2 tag) — carried
// to the next call in `c`.  A wavefront whose lanes all carry their element
// skips the first read altogether.  rd8(p) = the 8 bytes at p.  (The fused parse walks every piece three times, a wavefront at the
// pace of its slowest lane: 129 k of the fused kernel's 201 k cycles per chunk were these walks, r05 f02.)
struct WalkCarry { uint32_t t4, at; };          // the 4 bytes at position `at` (0xFFFFFFFF: nothing carried)
template <class G, class Rd, class Rd8>
__device__ __forceinline__ bool walk_step_carry(const Rd& rd, const Rd8& rd8, uint32_t ip, uint32_t iend, Seq& s, WalkCarry& c, bool on) {
    if constexpr (std::is_same<G, Lz4Grammar>::value) {
        // (LZ4: the two 4-byte reads of walk_step stay — with the offset field as a misaligned 8-byte read and the first read under a
        //  branch the fused kernel lost 4 %, 379 -> 363 GB/s on 8 192 chunks: these walks are bound by the LDS pipe's misaligned accesses,
        //  not by its latency)
        (void)rd8; (void)c; (void)on;
        return walk_step<G>(rd, ip, iend, s);
    } else {
        const bool have = c.at == ip;
        uint32_t t4 = c.t4;
        if (ballot64(on && !have) != 0ull) { if (!have) t4 = rd(ip); }
        c.at = 0xFFFFFFFFu;
        const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
        const bool is_lit = (tag & 3u) == 0u;
        const uint32_t lhdr = is_lit ? (l6 < 60u ? 1u : l6 - 58u) : 0u;                // (headers of 1 .. 4 bytes: walk_step above)
        const uint32_t lit = is_lit ? (l6 < 60u ? l6 + 1u : ((t4 >> 8) & (0xffffffu >> (8u * (62u - (l6 > 62u ? 62u : l6))))) + 1u) : 0u;
        const uint32_t ip2 = ip + lhdr + lit;
        const bool in2 = ip2 + 4u <= iend && ip2 >= ip;
        uint2 c8 = make_uint2(t4, 0u);
        if (ballot64(on && is_lit) != 0ull) { if (is_lit) c8 = rd8(in2 ? ip2 : ip); }
        const uint32_t c4 = c8.x;
        const uint32_t ctag = c4 & 0xffu, kind = ctag & 3u;
        const uint32_t clen = kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
        uint32_t off = kind == 1u ? ((ctag >> 5) << 8) | ((c4 >> 8) & 0xffu) : kind == 3u ? (c8.x >> 8) | (c8.y << 24) : (c4 >> 8) & 0xffffu;
        const uint32_t ip3 = kind == 0u ? ip2 : ip2 + (kind == 1u ? 2u : kind == 2u ? 3u : 5u);
        const bool fast = !(is_lit && l6 > 62u) && in2 && ip3 < iend && (kind != 0u || is_lit);
        // (a copy-4 element that is the whole record: its offset's last byte lies behind the four bytes at ip)
        if (ballot64(on && fast && kind == 3u && !is_lit) != 0ull) { if (fast && kind == 3u && !is_lit) off = rd(ip + 1u); }
        bool ok = true;
        if (fast) {
            s.lit = lit; s.lit_at = ip + lhdr; s.last = false; s.next = ip3;
            s.mlen = kind == 0u ? 0u : clen; s.offset = kind == 0u ? 0u : off;
            // the element behind a copy-1 / copy-2 that followed a literal came with the 8 bytes; behind a literal alone the 4 bytes at ip2 ARE the next element
            if (is_lit && kind != 3u) { c.t4 = kind == 0u ? c4 : __builtin_amdgcn_alignbyte(c8.y, c8.x, kind == 1u ? 2u : 3u); c.at = ip3; }
        }
        if (ballot64(on && !fast) != 0ull) { if (!fast) ok = G::at(rd, ip, iend, s, nullptr); }
        return ok;
    }
}
#endif

}  // namespace cj
