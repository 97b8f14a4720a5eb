// lvl_shared.hpp — device helpers of the level-ordered workgroup decoders (lz4_decode_lvl.hip: records and sorted descriptors
// in a global table, two workgroups per CU, any chunk; lz4_decode_lvl1.hip: everything in LDS, one workgroup per CU, chunks of
// up to 4032 sequences): byte-masked dword-grid stores, the dense LDS -> LDS copies, the position index lookup.
#pragma once
#include "lds_shared.hpp"

namespace cj {

constexpr uint32_t kLvMaxLevel = 1023;
constexpr uint32_t kLvUnknown = 0xffffu;

__device__ __forceinline__ void lds_mskor(uint32_t a, uint32_t mask, uint32_t v) {      // MEM[a] = (MEM[a] & ~mask) | v, atomically; v inside mask
    asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(a), "v"(mask), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_st16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" :: "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t lds_ld16(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

// v[k] = the bytes for dword k of the destination's dword grid starting at the aligned LDS address da; the lane owns bytes
// [hb, hb + n) of that run (n >= 1, hb < 4, hb + n <= 4 NV).  First and last dword: byte-masked atomic stores; between them whole
// dwords (a lane whose run is shorter stores to its private dummy dword instead of being masked off).
template <int NV>
__device__ __forceinline__ void lds_store_grid(const uint32_t (&v)[NV], uint32_t da, uint32_t hb, uint32_t n, uint32_t dummy_w) {
    const uint32_t e = hb + n, nd = (e + 3u) >> 2;                       // dwords touched: 1 .. NV
    const uint32_t hm = 0xffffffffu << (8u * hb);
    const uint32_t tm = 0xffffffffu >> (8u * (3u - ((e - 1u) & 3u)));
    const uint32_t m0 = nd == 1u ? (hm & tm) : hm;
    lds_mskor(da, m0, v[0] & m0);
    uint32_t tv = 0;
#pragma unroll
    for (int i = 1; i < NV; i++) {
        if (i < NV - 1) lds_st32((uint32_t)i + 1u < nd ? da + 4u * (uint32_t)i : dummy_w, v[i]);
        tv = (uint32_t)i + 1u == nd ? v[i] : tv;
    }
    lds_mskor(nd >= 2u ? da + 4u * (nd - 1u) : dummy_w, tm, tv & tm);
}

// n (1 .. 16 / 1 .. 32) bytes LDS -> LDS, both byte addresses arbitrary, [as, as + n) final and not overlapping [ad, ad + n).
// Reads aligned dwords from up to 3 bytes in front of the source to 7 behind it (never stored).
__device__ __forceinline__ void lvl_copy16(uint32_t ad, uint32_t as, uint32_t n, uint32_t dummy_w) {
    const uint32_t hb = ad & 3u, sp = as - hb, sa = sp & ~3u, sh = sp & 3u;
    uint64_t p0, p1, p2;
    asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read2_b32 %2, %3 offset0:4 offset1:5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(p0), "=&v"(p1), "=&v"(p2) : "v"(sa) : "memory");
    const uint32_t w[6] = {(uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32), (uint32_t)p2, (uint32_t)(p2 >> 32)};
    uint32_t v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
    lds_store_grid<5>(v, ad & ~3u, hb, n, dummy_w);
}
__device__ __forceinline__ void lvl_copy32(uint32_t ad, uint32_t as, uint32_t n, uint32_t dummy_w) {
    const uint32_t hb = ad & 3u, sp = as - hb, sa = sp & ~3u, sh = sp & 3u;
    uint64_t p0, p1, p2, p3, p4;
    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2_b32 %1, %5 offset0:2 offset1:3\n\tds_read2_b32 %2, %5 offset0:4 offset1:5\n\t"
                 "ds_read2_b32 %3, %5 offset0:6 offset1:7\n\tds_read2_b32 %4, %5 offset0:8 offset1:9\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4) : "v"(sa) : "memory");
    const uint32_t w[10] = {(uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32), (uint32_t)p2, (uint32_t)(p2 >> 32),
                            (uint32_t)p3, (uint32_t)(p3 >> 32), (uint32_t)p4, (uint32_t)(p4 >> 32)};
    uint32_t v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
    lds_store_grid<9>(v, ad & ~3u, hb, n, dummy_w);
}

// A run of n bytes LDS -> LDS whose source is final and does not overlap the destination (a literal run from the staged input, the
// non-overlapping part of a long match).  One lane per run in pieces of <= 32 bytes; a run of kLongRun bytes or more is copied by
// the whole wavefront, 64 pieces of 32 bytes per round (neighbouring pieces meet in byte-masked dwords).  `act` lanes take part;
// every lane of the wavefront calls.
__device__ __forceinline__ void lvl_wave_pieces(uint32_t ad, uint32_t as, uint32_t n, uint32_t dummy_w) {     // wave-uniform arguments
    const uint32_t lane = lane_id();
    for (uint32_t base = 0; base < n; base += 2048u) {
        const uint32_t o = base + 32u * lane;
        if (o < n) lvl_copy32(ad + o, as + o, n - o < 32u ? n - o : 32u, dummy_w);
    }
}
__device__ __forceinline__ void lvl_run_copy(bool act, uint32_t ad, uint32_t as, uint32_t n, uint32_t dummy_w) {
    const uint32_t lane = lane_id();
    uint32_t rem = act ? n : 0u;
    uint64_t longm = ballot64(rem >= kLongRun);
    while (longm) {
        const uint32_t l = ctz64(longm);
        longm &= longm - 1ull;
        lvl_wave_pieces(rdlane(ad, l), rdlane(as, l), rdlane(rem, l), dummy_w);
        if (lane == l) rem = 0u;
    }
    while (ballot64(rem > 0u) != 0ull) {
        const uint32_t k = rem < 32u ? rem : 32u;
        if (ballot64(k > 16u) != 0ull) { if (k > 0u) lvl_copy32(ad, as, k, dummy_w); }
        else if (k > 0u) lvl_copy16(ad, as, k, dummy_w);
        ad += k; as += k; rem -= k;
    }
}

// One lane per match: dst[0, m) = dst[-off ...], every source byte in front of dst final.  Pieces of at most 32 bytes that do
// not overlap their source; a self-overlapping match (off < m) reads from a distance that doubles while the copied region is
// still shorter than it, so the distance stays a multiple of the period (model: tests/test_level_decoder_model.py).  Matches of
// kLongRun bytes or more are copied by the whole wavefront.  `act` lanes take part; every lane of the wavefront calls.
__device__ __forceinline__ void lvl_match_copy(bool act, uint32_t a_out, uint8_t* s_out, uint32_t dst, uint32_t off, uint32_t m, uint32_t dummy_w) {
    const uint32_t lane = lane_id();
    uint32_t rem = act ? m : 0u;
    uint64_t longm = ballot64(rem >= kLongRun);
    while (longm) {
        const uint32_t l = ctz64(longm);
        longm &= longm - 1ull;
        const uint32_t lmm = rdlane(m, l), lo = rdlane(off, l), ld = rdlane(dst, l);
        const uint8_t* sb = s_out + (ld - lo);
        if (lo >= lmm || lo >= 2048u) {                      // every round of 2 KiB reads bytes that earlier rounds (or levels) wrote
            lvl_wave_pieces(a_out + ld, a_out + ld - lo, lmm, dummy_w);
        } else
        if (lo == 1u || lo == 2u || lo == 4u) {              // run of a 1/2/4-byte pattern: 16 bytes per lane
            const uint32_t h = (0u - ld) & 15u, hh = h < lmm ? h : lmm;
            if (lane < hh) s_out[ld + lane] = sb[lane % lo];
            uint32_t wv = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++) wv |= (uint32_t)sb[(hh + i) % lo] << (8u * i);
            const uint32_t nv = (lmm - hh) >> 4;
            uint4* dv = reinterpret_cast<uint4*>(s_out + ld + hh);
            for (uint32_t q = lane; q < nv; q += 64u) dv[q] = make_uint4(wv, wv, wv, wv);
            const uint32_t t0 = hh + (nv << 4);
            if (t0 + lane < lmm) s_out[ld + t0 + lane] = sb[(t0 + lane) % lo];
        } else {
            uint32_t rr = lane, step = 64u;
            if (lo <= 64u) { rr = lane % lo; step = 64u % lo; }
            for (uint32_t k = lane; k < lmm; k += 64u) {
                s_out[ld + k] = sb[lo >= lmm ? k : rr];
                rr += step;
                if (rr >= lo) rr -= lo;
            }
        }
        if (lane == l) rem = 0u;
    }
    uint32_t d = off, at = a_out + dst;
    while (ballot64(rem > 0u) != 0ull) {
        uint32_t n = rem < d ? rem : d;
        n = n < 32u ? n : 32u;
        if (ballot64(n > 16u) != 0ull) { if (n > 0u) lvl_copy32(at, at - d, n, dummy_w); }
        else if (n > 0u) lvl_copy16(at, at - d, n, dummy_w);
        d = (n == d) ? 2u * d : d;
        at += n; rem -= n;
    }
}

// the same for a FULL wavefront of a large level: when every active lane holds a short match that does not overlap its source (the
// rule on the benchmark data) it is one straight dword-grid copy, no loop and no long-run test
__device__ __forceinline__ void lvl_match_copy_dense(bool act, uint32_t a_out, uint8_t* s_out, uint32_t dst, uint32_t off, uint32_t m, uint32_t dummy_w) {
    if (ballot64(act && (off < m || m > 32u)) != 0ull) { lvl_match_copy(act, a_out, s_out, dst, off, m, dummy_w); return; }
    const uint32_t at = a_out + dst;
    if (ballot64(act && m > 16u) != 0ull) { if (act) lvl_copy32(at, at - off, m, dummy_w); }
    else if (act) lvl_copy16(at, at - off, m, dummy_w);
}

// the same for a level of a few matches (the tail of the dependency DAG, copied by one wavefront): with a handful of active lanes an
// LDS access at its exact byte address costs about one cycle per active lane, so a match of up to 32 bytes is 2-4 reads and 2-4
// stores (lds_copy_sparse) instead of the ~40 instructions of the dword-grid copy.  Everything else takes lvl_match_copy.
__device__ __forceinline__ void lvl_match_copy_sparse(bool act, uint32_t a_out, uint8_t* s_out, uint32_t dst, uint32_t off, uint32_t m, uint32_t dummy_w) {
    const bool fast = act && off >= m && m <= 32u;
    if (fast) lds_copy_sparse(a_out + dst, a_out + dst - off, m);
    if (ballot64(act && !fast) != 0ull) lvl_match_copy(act && !fast, a_out, s_out, dst, off, m, dummy_w);
}

// position index lookup: record that holds output byte p, and whether p lies in its match part
__device__ __forceinline__ void lvl_lookup(uint32_t a_A, uint32_t a_C, uint32_t p, uint32_t& q, bool& inm) {
    const uint32_t g = p >> 4, i = p & 15u;
    uint32_t a, c;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_u16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(c) : "v"(a_A + 4u * g), "v"(a_C + 2u * g) : "memory");
    const uint32_t mask = (2u << i) - 1u;
    const uint32_t s = a & mask, d = (a >> 16) & mask;
    q = (c & 0x7fffu) + (uint32_t)__popc(s) - 1u;
    inm = (s | d) == 0u ? (c >> 15) != 0u : (d != 0u && __clz((int)d) <= __clz((int)s));
}

}  // namespace cj
