// big_chunks.hip — the parse stage for chunks of 64 KiB .. 256 KiB in a device batch (big_chunks.hpp; BASELINE configs[4]).
// Accept / reject rules: those of the other mappings (liblz4 1.10.0 LZ4_decompress_safe, snap 1.1.1 raw::Decoder; reference call
// sites /root/reference/src/lz4.rs:88,164,168, src/snappy.rs:57,106); wherever this kernel is not sure — any violation, a Snappy
// copy that reaches further back than 65 535 bytes, more records than a region holds — the chunk stays with the wavefront-per-
// chunk kernel, which decodes every valid chunk and names every error exactly.
//
//   big_list_kernel    one thread per chunk of the batch: a chunk whose capacity (LZ4) / announced length (Snappy) lies in
//                      (64 KiB, 256 KiB] is appended to the list (the small-chunk pipeline has flagged it kRouteWave);
//   big_parse_kernel   32 lanes per listed chunk.  Lane j starts 1 KiB in front of its boundary j * seg of the compressed
//                      bytes and walks until it has crossed it (a malformed element there = "not a token": one byte further):
//                      that position is its CANDIDATE.  From it the lane walks for real — the checks that do not need the
//                      absolute output position, one 16-byte record per sequence into its region, positions counted from its
//                      own start — until it stands EXACTLY on the candidate of the segment it has reached: from there that
//                      lane's records are the chain's (the next-token function depends on the bytes only; a wrong candidate
//                      costs time, never correctness).  Lane 0 starts on the first token, so the lanes reached from it — the
//                      LIVE lanes — hold exactly the chunk's sequences, each once.  Epilogue: what lies in front of each live
//                      lane (records, output bytes), the deferred checks (every offset reaches back at most to output byte 0,
//                      LZ4's end-of-block margins, Snappy's announced length), and for every 64 KiB boundary of the output the
//                      record that holds it (a binary search by the lane that owns it).
#include "big_chunks.hpp"
#include "parse_grammar.hpp"

namespace cj {

#ifndef CJ_BIG_MS
#define CJ_BIG_MS 16
#endif
constexpr uint32_t kBigMs = CJ_BIG_MS;                     // milestones per segment
// the distance of a stream's milestone lines: a 16th of a 32nd of its length, at least 128 bytes, a multiple of 16
__device__ __forceinline__ uint32_t big_ms_spacing(uint32_t n) { const uint32_t sp = (((n + kBigLanes * kBigMs - 1u) / (kBigLanes * kBigMs)) + 15u) & ~15u; return sp < 128u ? 128u : sp; }
constexpr uint32_t kBigNoLink = 0xFFFFu;
constexpr uint32_t kBigListHdr = 4;
constexpr uint32_t kBigWaves = 4;                   // wavefronts per block of the parse kernel

// prologue shared by the listing and the parse kernel: the element stream of chunk c and its output bound, or false
template <int kCodec>
__device__ __forceinline__ bool big_prologue(const BatchArgs& a, uint32_t c, const uint8_t*& in, uint32_t& n, uint32_t& cap, uint32_t& skip) {
    const uint8_t* in0 = a.in_base + a.in_off[c];
    uint64_t n64 = a.in_len[c], cap64 = a.out_cap[c];
    if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
        const uint8_t* inp = in0;
        if (lz4_block_prologue(a.flags, inp, n64, cap64) != 0) return false;
        if (cap64 <= kLdsOutMax && n64 <= kLdsInMax) return false;             // the small-chunk pipeline's
        if (cap64 > kBigOutMax || n64 > kBigInMax || n64 == 0) return false;
        in = inp; n = (uint32_t)n64; cap = (uint32_t)cap64; skip = (uint32_t)(inp - in0);
        return true;
    } else {
        if (n64 == 0 || n64 > 0xFFFFFFF0ull) return false;
        uint64_t ulen = 0;
        uint32_t shift = 0, i = 0, hdr = 0;
        bool ok = false;
        while (hdr < (uint32_t)n64 && i < 10u) {
            const uint32_t b = in0[hdr];
            hdr += 1;
            if (b < 0x80u) { if (!(i == 9u && b > 1u)) { ulen |= (uint64_t)b << shift; ok = true; } break; }
            ulen |= (uint64_t)(b & 0x7fu) << shift;
            shift += 7; i += 1;
        }
        if (!ok || ulen > cap64 || ulen == 0 || hdr == (uint32_t)n64) return false;
        if (ulen <= kLdsOutMax && n64 - hdr <= kLdsInMax) return false;
        if (ulen > kBigOutMax || n64 - hdr > kBigInMax) return false;
        in = in0 + hdr; n = (uint32_t)n64 - hdr; cap = (uint32_t)ulen; skip = hdr;
        return true;
    }
}

template <int kCodec>
__global__ __launch_bounds__(256) void big_list_kernel(BatchArgs a, uint32_t* list) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= a.n_chunks) return;
    const uint8_t* in; uint32_t n, cap, skip;
    if (!big_prologue<kCodec>(a, c, in, n, cap, skip)) return;
    const uint32_t idx = atomicAdd(&list[0], 1u);
    if (idx < list[1]) list[kBigListHdr + idx] = c;
}

// The lanes' view of their streams: a 128-byte ring per lane in LDS (two 64-byte units), refilled cooperatively like the 256-byte
// rings of the small-chunk parse kernels (lane_stream.hpp) — 4 lanes fetch one lane's next unit with aligned 16-byte loads, 16
// units per load instruction — at half the size: 9 KiB per wavefront, sixteen wavefronts per CU.  A lane reads FIELDS from it: a
// sequence is field A (token + literal length bytes; Snappy: a literal's tag + length bytes, or nothing) at ip and field B (offset +
// match length bytes; Snappy: the copy element) behind the literals at ip2.  Where the literals are longer than what the ring holds
// ahead — the streams of 256 KiB chunks are literal-heavy, 80 bytes per sequence on the benchmark data — the lane parks field A,
// lets the ring jump to ip2 and reads field B in the next iteration: two iterations for such a sequence, never the general path.
// (Measured against it: a 16-byte window in registers loaded straight from global memory at every field, no LDS — one round trip
//  per sequence, 3x the time per iteration: each load pulls a whole line for 16 bytes, and 262 144 lanes' lines do not stay in L2
//  between two steps.  tools/experiments/big_chunks/big_chunks_window_walk.hip)
constexpr uint32_t kSsRing = 128, kSsUnit = 64;
constexpr uint32_t kSsStride = kSsRing + 16u;           // 16-byte aligned rings (one ds_write_b128 per fetched piece)
constexpr uint32_t kSsWaveBytes = 64u * kSsStride;
constexpr uint32_t kSsLanesPerUnit = kSsUnit / 16u, kSsTargets = 64u / kSsLanesPerUnit, kSsLoads = 64u / kSsTargets;

struct SegStream {
    const uint8_t* base;    // 128-byte aligned address at or below the first stream byte
    uint32_t lo, hi;        // cached window [lo, hi): multiples of 64, hi - lo <= 128
    uint32_t end;           // offset of the end of the stream
    uint32_t ring;          // LDS byte offset of this lane's ring
    __device__ __forceinline__ uint32_t ring32(uint32_t p) const {        // the 4 bytes at p, read from the ring whether or not they are cached
        const uint32_t a0 = ring + (p & (kSsRing - 4u)), a1 = ring + ((p + 4u) & (kSsRing - 4u));
        uint32_t w0, w1;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0), "v"(a1) : "memory");
        return __builtin_amdgcn_alignbyte(w1, w0, p & 3u);
    }
    __device__ __forceinline__ bool in_window(uint32_t p) const { return p >= lo && p + 4u <= hi && p + 4u <= end; }
    __device__ __forceinline__ uint32_t ld32(uint32_t p) const {           // anywhere in the stream (zero-filled past its end)
        if (in_window(p)) return ring32(p);
        const uint32_t v = ld_le_tail(base, p, end);
        __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): here, not on the common path behind the branch
        return v;
    }
};

// what a lane needs to know about the lanes it fetches for (lane t = 16 r + lane / 4 in load r): their stream base and end never change
struct SegPlan { uint32_t blo[kSsLoads], bhi[kSsLoads], end[kSsLoads]; };
__device__ __forceinline__ SegPlan seg_plan(const SegStream& st) {
    SegPlan p;
    const uint32_t lane = lane_id();
    const uint32_t blo = (uint32_t)(uintptr_t)st.base, bhi = (uint32_t)((uintptr_t)st.base >> 32);
#pragma unroll
    for (uint32_t r = 0; r < kSsLoads; r++) {
        const int t = (int)(kSsTargets * r + lane / kSsLanesPerUnit);
        p.blo[r] = (uint32_t)__shfl((int)blo, t); p.bhi[r] = (uint32_t)__shfl((int)bhi, t); p.end[r] = (uint32_t)__shfl((int)st.end, t);
    }
    return p;
}
// one wave-convergent refill round: every lane that has room gets its next 64-byte unit
__device__ __forceinline__ void seg_refill(SegStream& st, bool want, uint32_t wave_ring, const SegPlan& plan) {
    const uint32_t lane = lane_id(), piece = lane % kSsLanesPerUnit;
    const uint32_t mine = st.hi | (want ? 1u : 0u);                    // hi is a multiple of 64
    uint4 v[kSsLoads];
    uint32_t dsta[kSsLoads];
#pragma unroll
    for (uint32_t r = 0; r < kSsLoads; r++) {
        const uint32_t t = kSsTargets * r + lane / kSsLanesPerUnit;
        const uint32_t th = (uint32_t)__shfl((int)mine, (int)t);
        const uint32_t off = (th & ~1u) + 16u * piece;
        v[r] = make_uint4(0, 0, 0, 0);
        dsta[r] = 0xffffffffu;
        if ((th & 1u) && off < plan.end[r]) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(((uint64_t)plan.bhi[r] << 32) | plan.blo[r]) + off;
            v[r] = *reinterpret_cast<const uint4*>(src);               // 16-byte aligned, never crosses into a page past the stream
            dsta[r] = wave_ring + t * kSsStride + (off & (kSsRing - 1u));
        }
    }
#pragma unroll
    for (uint32_t r = 0; r < kSsLoads; r++) {
        if (dsta[r] != 0xffffffffu) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 q = {v[r].x, v[r].y, v[r].z, v[r].w};
            asm volatile("ds_write_b128 %0, %1" :: "v"(dsta[r]), "v"(q) : "memory");
        }
    }
    if (want) {
        if (st.hi - st.lo >= kSsRing) st.lo += kSsUnit;
        st.hi += kSsUnit;
    }
}


constexpr uint32_t kBigAhead = 32;                  // cached bytes a lane must have ahead of the field it reads next

struct BigElem { uint32_t lit, lit_at, mlen, offset, next; bool ok, last; };

// field A at ip -> e.lit, e.lit_at, e.mlen (LZ4: the token's match nibble), ip2; false: not the common shape (general path)
template <int kCodec>
__device__ __forceinline__ bool big_field_a(const SegStream& st, uint32_t ip, uint32_t iend, BigElem& e, uint32_t& ip2) {
    if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
        const uint32_t t4 = st.ring32(ip);
        const uint32_t token = t4 & 0xffu, e1 = (t4 >> 8) & 0xffu, e2 = (t4 >> 16) & 0xffu, e3 = t4 >> 24;
        const bool x1 = (token >> 4) == 15u, x2 = x1 && e1 == 255u, x3 = x2 && e2 == 255u;
        const uint32_t lit = (token >> 4) + (x1 ? e1 : 0u) + (x2 ? e2 : 0u) + (x3 ? e3 : 0u);
        const uint32_t ip1 = ip + 1u + (x1 ? 1u : 0u) + (x2 ? 1u : 0u) + (x3 ? 1u : 0u);
        ip2 = ip1 + lit;
        e.lit = lit; e.lit_at = ip1; e.mlen = token & 15u;
        return st.in_window(ip) && !(x3 && e3 == 255u) && ip2 + 8u <= iend;      // (iend - ip1 >= lit + 8: not the block's last sequences)
    } else {
        // a literal element of any header size (1 .. 5 bytes), or none: the element at ip is the copy itself
        const uint32_t t4 = st.ring32(ip), t5 = st.ring32(ip + 4u);
        const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
        const bool is_lit = (tag & 3u) == 0u;
        const uint32_t nb = l6 < 60u ? 0u : l6 - 59u;                   // length bytes behind the tag
        const uint32_t lv = nb == 4u ? (t4 >> 8) | (t5 << 24) : (t4 >> 8) & (0x00ffffffu >> (8u * (3u - (nb ? nb : 1u))));
        const uint32_t lhdr = is_lit ? 1u + nb : 0u;
        const uint32_t lit = is_lit ? (nb ? lv + 1u : l6 + 1u) : 0u;
        const bool ok = st.in_window(ip) && st.in_window(ip + 4u) && !(is_lit && (lit == 0u || lit > kBigInMax)) && ip + 8u <= iend;      // (lit == 0: a 4-byte length of 2^32 - 1)
        ip2 = ok ? ip + lhdr + lit : ip;
        e.lit = lit; e.lit_at = ip + lhdr; e.mlen = 0u;
        return ok && ip2 + 8u <= iend;
    }
}
// field B at ip2 (inside the ring with the 4 bytes behind it) completes the element; false: general path
template <int kCodec>
__device__ __forceinline__ bool big_field_b(const SegStream& st, uint32_t ip2, BigElem& e) {
    const uint32_t o4 = st.ring32(ip2), o5 = st.ring32(ip2 + 4u);
    if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
        const uint32_t f1 = (o4 >> 16) & 0xffu, f2 = o4 >> 24, f3 = o5 & 0xffu;
        const bool y1 = e.mlen == 15u, y2 = y1 && f1 == 255u, y3 = y2 && f2 == 255u;
        e.offset = o4 & 0xffffu;
        e.mlen = e.mlen + (y1 ? f1 : 0u) + (y2 ? f2 : 0u) + (y3 ? f3 : 0u) + 4u;
        e.next = ip2 + 2u + (y1 ? 1u : 0u) + (y2 ? 1u : 0u) + (y3 ? 1u : 0u);
        return !(y3 && f3 == 255u);
    } else {
        // a copy of any kind behind the literal — or another literal: then this element is the literal alone
        const uint32_t ctag = o4 & 0xffu, kind = ctag & 3u;
        e.mlen = kind == 0u ? 0u : kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
        e.offset = kind == 1u ? ((ctag >> 5) << 8) | ((o4 >> 8) & 0xffu) : kind == 2u ? (o4 >> 8) & 0xffffu : kind == 3u ? (o4 >> 8) | (o5 << 24) : 0u;
        e.next = ip2 + (kind == 0u ? 0u : kind == 1u ? 2u : kind == 2u ? 3u : 5u);
        return kind != 0u || e.lit != 0u;
    }
}

// what a lane of the walk leaves for the epilogue
struct BigLane {
    uint32_t cnt, r;          // records written, output bytes they describe (counted from the lane's start, modulo 2^32 while it walks beside the chain)
    uint32_t lim, lim_cnt;    // LZ4's end-of-block margin behind the last sequence with a match, and that sequence's record number + 1
    uint32_t link;            // the milestone (global number) where the lane met another lane's walk, or kBigNoLink
    int32_t iv_bad;           // the last interval of its walk in which it met something malformed (-1: none)
    uint32_t flags, nms;      // kBigLaneLast / kBigLaneFull; milestones it published
};
constexpr uint32_t kBigLaneFull = 1u, kBigLaneLast = 2u;
// a milestone word: st-position (20 bits) | records before it (12 bits) | output bytes before it (32 bits); all ones = not published
__device__ __forceinline__ uint64_t ms_word(uint32_t pos, uint32_t cnt, uint32_t r) { return (uint64_t)pos | ((uint64_t)cnt << 20) | ((uint64_t)r << 32); }

// THE WALK.  Work unit = (listed chunk bi, segment j); a wavefront takes the SAME segment of 64 consecutive chunks (unit u -> j = u /
// capr, bi = u % capr, capr = the list's capacity rounded up to 64): the lanes of a wavefront run in lock step, and the segments of
// one chunk differ — on the benchmark data the first 7 KiB of a stream hold 370 sequences, a later 7 KiB eighty — while the same
// segment of different chunks does not.  The lanes of a chunk therefore meet through GLOBAL memory (agent-scope stores / loads).
//
// Lane j starts AT its boundary j * seg, wherever in an element that is, and walks: an element that cannot be one means "this was no
// token" (one byte further, the interval is marked bad).  Walking from a wrong position falls into step with the chain with some
// probability per element of the chain it passes (1 / its own step length: a handful of elements); from there on its records ARE the
// chain's.  Every seg / 16 bytes the lane publishes a MILESTONE: the first element position at or behind that line, and how many
// records / output bytes it had written before it.  A lane that walks into the next segment compares its position with the
// owner's milestone at every line it crosses: equal = from there on the owner's records continue its own, it stops (link).  So a
// lane walks its own segment plus the stretch the next lane needed to fall into step — one or two milestones — and a wavefront is
// not held up for whole segments by the few lanes whose neighbour started badly (the first version walked a 1 KiB lead-in and
// compared once per segment: 3 - 13 % of the lanes missed, their neighbours walked a second segment, and with 64 lanes per wavefront
// every wavefront had one: 1 700 iterations per wavefront instead of 300).  A milestone not yet published = walk on: never wrong.
template <int kCodec>
__global__ __launch_bounds__(64 * kBigWaves) void big_walk_kernel(BatchArgs a, const uint32_t* list, uint32_t capr, uint4* recs, unsigned long long* cands, int32_t* need_iv, BigLane* lanes) {
    __shared__ __attribute__((aligned(16))) uint8_t rings[kBigWaves * kSsWaveBytes];
    constexpr uint32_t k = kBigLanes;
    const uint32_t gl = blockIdx.x * (64u * kBigWaves) + threadIdx.x;
    const uint32_t j = gl / capr, bi = gl % capr;                // (capr is a multiple of 64: a wavefront has one j)
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t wave_ring = (uint32_t)(uintptr_t)rings + wave * kSsWaveBytes;
    const uint32_t listed = list[0] < list[1] ? list[0] : list[1];
    const bool exists = bi < listed && j < k;
    const uint32_t c = exists ? list[kBigListHdr + bi] : 0u;
    unsigned long long* cms = cands + (size_t)bi * k * kBigMs;   // the chunk's milestones, numbered g = 16 j + q, line g at mis + g * sp
    int32_t* my_need = need_iv + ((size_t)bi * k + j) * kBigMs;

    const uint8_t* in = nullptr;
    uint32_t n = 0, cap = 0, skip = 0;
    const bool walk = exists && big_prologue<kCodec>(a, c, in, n, cap, skip);          // (true for every listed chunk)
    const uint32_t sp = big_ms_spacing(n), seg = sp * kBigMs;
    const uint32_t bj = j * seg;
    bool done = !(walk && bj < n);
    const uint32_t mis = done ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(in) & 127u);
    SegStream st;
    st.base = done ? nullptr : in - mis;                            // 128-byte aligned
    st.end = done ? 0u : mis + n;
    st.ring = wave_ring + lane * kSsStride;
    const uint32_t iend = st.end;
    // positions below are st-positions (offsets from st.base): stream position + mis
    uint32_t ip = mis + bj;
    st.lo = st.hi = ip & ~(kSsUnit - 1u);
    const SegPlan plan = seg_plan(st);
    // a sequence whose field B lies beyond the ring: field A parked here, the lane stands on ip2 (`at_b`)
    bool at_b = false;
    uint32_t a_lit = 0, a_at = 0, a_mc = 0, a_ip = 0;
    uint32_t own_g = j * kBigMs, own_line = mis + bj;             // the next milestone of its own to publish, and its line
    const uint32_t own_end = (j + 1u) * kBigMs;
    uint32_t chk_g = own_end, chk_line = j + 1u < k ? mis + (j + 1u) * seg : 0xFFFFFFFFu;   // the next foreign milestone to compare with
    uint32_t link = kBigNoLink;
    uint32_t cnt = 0, r = 0, lim = 0, lim_cnt = 0, nms = 0;
    int32_t need = INT32_MIN, iv_bad = -1;                        // need: how far in front of the lane's first output byte the matches of the current interval reach
    bool full = false, saw_last = false, fin = false;
    uint4* region = recs + (size_t)bi * kBigRecPitch + (size_t)j * kBigRegion;
    // records leave in groups of two slots = one aligned 32-byte store (iteration `it` fills slot it & 1)
    uint4 p0 = make_uint4(0, 0, 0, 0), p1 = p0;
    uint32_t it = 0;
    bool grp = false;

    for (;;) {
        {   // the ring holds the field the lane reads next (pos) and kBigAhead bytes behind it
            const uint32_t pos = ip;
            if (!done && (pos >= st.hi || pos < st.lo)) st.lo = st.hi = pos & ~(kSsUnit - 1u);      // jumped out of the window (long literal run; back to the byte behind an element that was none): re-anchor
            for (;;) {
                const bool want = !done && st.hi < iend && (st.hi - st.lo < kSsRing || pos >= st.lo + kSsUnit);
                const bool urgent = want && pos + kBigAhead > st.hi;
                if (ballot64(urgent) == 0ull) break;
                seg_refill(st, want, wave_ring, plan);
            }
        }
        if (ballot64(!done) == 0ull) break;
        bool go = !done && !fin;
        bool emit = false;
        uint4 slot = make_uint4(0u, 0u, r, 0u);                  // the region's sentinel
        if (ballot64(!done && (fin || (!at_b && (ip >= own_line || ip >= chk_line)))) != 0ull) {
            if (!done && fin) { emit = true; done = true; }      // the sentinel behind the last record
            else if (go && !at_b) {
                // its own lines: the first element at or behind line g is here (a long literal run may cross several)
                while (own_g < own_end && ip >= own_line) {
                    if (nms != 0u) my_need[nms - 1u] = need;
                    need = INT32_MIN;
                    __hip_atomic_store(cms + own_g, ms_word(ip, cnt, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    own_g += 1u; own_line += sp; nms += 1u;
                }
                if (own_g == own_end) own_line = 0xFFFFFFFFu;
                if (ip >= chk_line) {                             // in another lane's segment, on an element: the owner's milestone at the last line crossed
                    while (chk_g + 1u < k * kBigMs && ip >= chk_line + sp) { chk_g += 1u; chk_line += sp; }
                    const unsigned long long cm = __hip_atomic_load(cms + chk_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cm != ~0ull && (uint32_t)(cm & 0xFFFFFu) == ip) { link = chk_g; done = true; go = false; emit = true; }      // the region's sentinel
                    else { chk_g += 1u; chk_line = chk_g < k * kBigMs ? chk_line + sp : 0xFFFFFFFFu; }      // another position, or not published yet: walk on
                }
            }
        }
        if (go && ip >= iend) { done = true; go = false; }      // ran off the stream beside the chain (a stream that ends without its last element: nobody links here)
        // one element (or its first half): fields from the ring for the common shapes, the grammar's general function for the rest
        BigElem e;
        e.lit = a_lit; e.lit_at = a_at; e.mlen = a_mc; e.offset = 0u; e.next = ip + 1u; e.ok = true; e.last = false;
        uint32_t ip2 = ip;
        bool fa = at_b;
        if (!at_b) fa = big_field_a<kCodec>(st, ip, iend, e, ip2);
        const bool near_b = ip2 >= st.lo && ip2 + 8u <= st.hi;
        bool half = false, fast = false;
        if (go && fa && !near_b && !at_b) { half = true; a_lit = e.lit; a_at = e.lit_at; a_mc = e.mlen; a_ip = ip; }      // park field A, stand on field B
        else if (fa) fast = big_field_b<kCodec>(st, ip2, e) && near_b;
        const uint32_t elem_ip = at_b ? a_ip : ip;              // where the element began
        if (ballot64(go && !half && !fast) != 0ull) {
            if (go && !half && !fast) {
                using G = typename std::conditional<kCodec == CJ_CODEC_SNAPPY_RAW, SnappyGrammar, Lz4Grammar>::type;
                const auto rd = [&st](uint32_t q) { return st.ld32(q); };
                Seq sq;
                e.ok = G::at(rd, elem_ip, iend, sq, st.base);
                e.lit = sq.lit; e.lit_at = sq.lit_at; e.mlen = sq.mlen; e.offset = sq.offset; e.next = sq.next; e.last = sq.last;
            }
        }
        if (go && half) { at_b = true; ip = ip2; go = false; }
        else if (go) { if (at_b) { at_b = false; ip = elem_ip; } }      // (the element is complete: the code below sees it at its start)
        if (go) {
            const uint32_t op2 = r + e.lit;
            const bool has_match = kCodec == CJ_CODEC_LZ4_BLOCK ? !e.last : e.mlen != 0u;
            const uint32_t off16 = e.offset & 0xffffu;
            // (a Snappy copy that reaches further back than 65 535 bytes needs more than the previous slab: wavefront kernel)
            const bool bad_now = !e.ok || (has_match && (e.offset == 0u || e.offset > 0xffffu)) || (!e.last && e.next >= iend) || e.lit > kBigInMax || e.mlen > 0x00ffffffu;
            // (beside the chain the output count runs on garbage lengths: kept below 2^30 so that the interval maxima stay comparable)
            if (bad_now || op2 + e.mlen > 0x40000000u) { iv_bad = (int32_t)nms - 1; ip += 1u; if (op2 + e.mlen > 0x40000000u) r = 0u; }      // no element here: one byte further
            else if (cnt + 4u > kBigRegion) { full = true; done = true; }                 // more records than a region holds
            else {
                const int32_t reach = has_match ? (int32_t)off16 - (int32_t)op2 : INT32_MIN;
                need = reach > need ? reach : need;
                const uint32_t mlen = has_match ? e.mlen : 0u;
                if (kCodec == CJ_CODEC_LZ4_BLOCK && has_match) { lim = op2 + (mlen + 5u > 12u ? mlen + 5u : 12u); lim_cnt = cnt + 1u; }
                emit = true;
                slot = make_uint4((e.lit_at - mis) | ((mlen >> 16) << 24), e.lit, r, (has_match ? off16 : 0u) | ((mlen & 0xffffu) << 16));
                cnt += 1u;
                r = op2 + mlen;
                ip = e.next > ip ? e.next : ip + 1u;
                saw_last = e.last;
                fin = e.last;
            }
        }
        // a lane that emits nothing in an iteration (no element here, finished) leaves a hole in the wave's phase: it keeps its slot
        // number by its own count, so the pair it stores is (cnt - 1) & ~1 .. | 1 — per-lane slots, one store per two of ITS records
        if (emit) {
            const uint32_t sl = (done && !full ? cnt : cnt - 1u);    // the sentinel sits behind the last record
            if ((sl & 1u) == 0u) p0 = slot; else p1 = slot;
            if ((sl & 1u) == 1u || done) {
                if ((sl & 1u) == 1u) { region[sl - 1u] = p0; region[sl] = p1; }
                else region[sl] = p0;
            }
        }
        it += 1u;
    }

#ifdef CJ_BIG_DEBUG
    if (exists && bi < 64u) printf("W j=%u bi=%u it=%u cnt=%u link=%d nms=%u ivbad=%d full=%d last=%d ipend=%u seg=%u\n", j, bi, it, cnt, link == kBigNoLink ? -1 : (int)link, nms, iv_bad, (int)full, (int)saw_last, ip - mis, seg);
#endif
    if (exists) {
        if (nms != 0u) my_need[nms - 1u] = need;
        const BigLane bl = {cnt, r, lim, lim_cnt, link, iv_bad, (full ? kBigLaneFull : 0u) | (saw_last ? kBigLaneLast : 0u), nms};
        lanes[(size_t)bi * k + j] = bl;
    }
}

// THE EPILOGUE, 32 lanes per listed chunk: which lanes does the chain run through and from which of their milestones, what lies in
// front of each, the checks that needed absolute positions, and for every 64 KiB boundary of the output the record that holds it
template <int kCodec>
__global__ __launch_bounds__(256) void big_sum_kernel(BatchArgs a, const uint32_t* list, const uint4* recs, const unsigned long long* cands, const int32_t* need_iv, const BigLane* lanes, BigMeta* bigmeta, ParseMeta* meta) {
    __shared__ volatile uint32_t s_live[256];
    constexpr uint32_t k = kBigLanes;
    const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
    const uint32_t bi = gl >> kBigLanesLog, j = gl & (k - 1u);
    const uint32_t lane = lane_id();
    const uint32_t g0 = threadIdx.x & ~(k - 1u);
    const uint32_t listed = list[0] < list[1] ? list[0] : list[1];
    const bool exists = bi < listed;
    const uint32_t c = exists ? list[kBigListHdr + bi] : 0u;
    const uint8_t* in = nullptr;
    uint32_t n = 0, cap = 0, skip = 0;
    const bool walk = exists && big_prologue<kCodec>(a, c, in, n, cap, skip);
    BigLane bl = {0u, 0u, 0u, 0u, kBigNoLink, -1, 0u, 0u};
    if (walk) bl = lanes[(size_t)bi * k + j];
    const uint32_t link = bl.link;
    const bool saw_last = (bl.flags & kBigLaneLast) != 0u;
    const uint4* region = recs + (size_t)bi * kBigRecPitch + (size_t)j * kBigRegion;
    // live = the chain runs through this lane, from its milestone q_in on (stored as q_in + 1); lane 0 from its first
    volatile uint32_t* my_live = s_live + threadIdx.x;
    uint32_t lv = walk && j == 0u && bl.nms != 0u ? 1u : 0u;
    *my_live = lv;
    for (uint32_t round = 1; round < k; round++) {               // (the 32 lanes of a chunk sit in one wavefront: in order, no barrier)
        if (lv != 0u && link != kBigNoLink) s_live[g0 + link / kBigMs] = 1u + link % kBigMs;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        lv = *my_live;
    }
    const bool live = lv != 0u;
    const uint32_t q_in = live ? lv - 1u : 0u;
    uint32_t cnt_q = 0, r_q = 0;
    if (live) {
        const unsigned long long mw = __hip_atomic_load(cands + ((size_t)bi * k + j) * kBigMs + q_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cnt_q = (uint32_t)(mw >> 20) & 0xFFFu; r_q = (uint32_t)(mw >> 32);
    }
    const uint32_t cnt = bl.cnt - cnt_q, r = bl.r - r_q;           // what the lane adds to the chain
    const uint32_t v_cnt = live ? cnt : 0u, v_out = live ? r : 0u;
    uint32_t s_cnt = v_cnt, s_out = v_out;
    for (uint32_t d = 1; d < k; d <<= 1) {
        const uint32_t t0 = (uint32_t)__shfl_up((int)s_cnt, d, 64), t1 = (uint32_t)__shfl_up((int)s_out, d, 64);
        if (j >= d) { s_cnt += t0; s_out += t1; }
    }
    const uint32_t first = s_cnt - v_cnt, opb = s_out - v_out;
    const uint32_t last_lane = (lane & ~(k - 1u)) + k - 1u;
    const uint32_t nseq = (uint32_t)__shfl((int)s_cnt, (int)last_lane, 64), total = (uint32_t)__shfl((int)s_out, (int)last_lane, 64);
    bool lane_ok = true;
    if (live) {
        int32_t need = INT32_MIN;
        for (uint32_t q = q_in; q < bl.nms; q++) { const int32_t v = need_iv[((size_t)bi * k + j) * kBigMs + q]; need = v > need ? v : need; }
        // (reach counted from the lane's start -> from its milestone: + r_q; 64-bit: the sums may leave 32 bits while nothing is wrong)
        const bool need_ok = need == INT32_MIN || (int64_t)need + (int64_t)(int32_t)r_q <= (int64_t)opb;
        lane_ok = !(bl.flags & kBigLaneFull) && bl.iv_bad < (int32_t)q_in && (link != kBigNoLink || saw_last) && need_ok && r <= kBigOutMax && (uint64_t)opb + r <= kBigOutMax;
        if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
            if (bl.lim_cnt > cnt_q) lane_ok = lane_ok && (uint64_t)opb + (bl.lim - r_q) <= cap;      // its last sequence with a match
            if (saw_last) lane_ok = lane_ok && (uint64_t)opb + r <= cap;
        }
    }
    const uint64_t okm = ballot64(lane_ok), lastm = ballot64(live && saw_last);
    const uint64_t gmask = (k >= 64u ? ~0ull : ((1ull << k) - 1ull)) << (lane & ~(k - 1u));
    bool chunk_ok = walk && (okm & gmask) == gmask && __popcll(lastm & gmask) == 1;
    if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) chunk_ok = chunk_ok && total == cap;
    else chunk_ok = chunk_ok && total <= cap;
    chunk_ok = chunk_ok && total > kLdsOutMax && nseq >= 1u;      // (a chunk that decodes to at most 64 KiB: the wavefront kernel — rare, and the slab walk below assumes two slabs)

    // ---- the record that holds output byte 65536 * s: the live lane whose output range contains it searches its region ----
    BigMeta* bm = bigmeta + bi;
    if (exists && chunk_ok && live) {
        for (uint32_t sb = 1; sb < kBigSlabs; sb++) {
            const uint32_t b = sb * 65536u;
            if (b >= total || b < opb || b >= opb + r) continue;      // not in this chunk / not in this lane's part (r > 0 here)
            const uint32_t rel = b - opb + r_q;                     // in the lane's own count
            uint32_t lo = 0, hi = cnt;                             // largest idx in [0, cnt) with lit_start[idx] <= rel (idx 0 has lit_start r_q)
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t z = __hip_atomic_load(&reinterpret_cast<const uint32_t*>(region + cnt_q + mid)[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int32_t)(z - rel) <= 0) lo = mid; else hi = mid;
            }
            bm->slab_first[sb] = first + lo;
        }
    }
    if (exists) {
        if (chunk_ok) {
            bm->first[j] = first;
            bm->slot0[j] = cnt_q;
            bm->opb[j] = opb - r_q;                                // record position + this = output position (modulo 2^32)
            if (j == 0u) {
                bm->chunk = c; bm->nseq = nseq; bm->in_skip = skip; bm->U = total; bm->slab_first[0] = 0u;
                a.result[c] = (int64_t)total;
                meta[c] = ParseMeta{0u, 0u};                       // handled here: the wavefront kernel skips it
            }
        } else if (j == 0u) { bm->chunk = c; bm->nseq = 0u; }      // stays with the wavefront kernel (the small-chunk pipeline flagged it)
    }
}

__global__ __launch_bounds__(256) void big_items_kernel(BatchArgs a, const uint32_t* list, const BigMeta* bigmeta, uint32_t cap,
                                                        uint64_t* rows, ParseMeta* item_meta, uint32_t* done) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x, items = kBigSlabs * cap;
    if (w >= items) return;
    const uint32_t bi = w % cap, sl = w / cap;
    const uint32_t listed = list[0] < list[1] ? list[0] : list[1];
    uint64_t in_off = 0, in_len = 0, out_off = 0, out_cap = 0, res = 0;
    uint32_t nrec = 0;
    if (bi < listed) {
        const BigMeta* bm = bigmeta + bi;
        const uint32_t U = bm->U, nseq = bm->nseq;
        if (nseq != 0u && sl * 65536u < U) {
            const uint32_t c = bm->chunk;
            const uint32_t R0 = bm->slab_first[sl], R1 = (sl + 1u) * 65536u < U ? bm->slab_first[sl + 1u] : nseq - 1u;
            nrec = R1 - R0 + 1u;
            in_off = a.in_off[c] + bm->in_skip; in_len = a.in_len[c] - bm->in_skip;
            out_off = a.out_off[c] + (uint64_t)sl * 65536u; out_cap = (uint64_t)sl * 65536u;      // (out_cap = the slab's first output position in its chunk: what the slab mode calls the stream position)
            res = U - sl * 65536u < 65536u ? U - sl * 65536u : 65536u;
        }
    }
    rows[w] = in_off; rows[items + w] = in_len; rows[2 * (size_t)items + w] = out_off; rows[3 * (size_t)items + w] = out_cap; rows[4 * (size_t)items + w] = res;
    item_meta[w] = ParseMeta{nrec, 0u};
    done[w] = 0u;
}

void launch_big_items(const BatchArgs& a, const uint32_t* list, const void* bigmeta, uint32_t cap, uint64_t* rows, void* item_meta, uint32_t* done, hipStream_t s) {
    if (cap == 0) return;
    hipLaunchKernelGGL(big_items_kernel, dim3((kBigSlabs * cap + 255u) / 256u), dim3(256), 0, s, a, list, (const BigMeta*)bigmeta, cap, rows, (ParseMeta*)item_meta, done);
}

size_t big_recs_bytes(size_t cap) { return cap * (size_t)kBigRecPitch * sizeof(uint4); }
size_t big_meta_bytes(size_t cap) { return cap * sizeof(BigMeta); }

size_t big_walk_scratch_bytes(size_t cap) { const size_t capr = (cap + 63) & ~(size_t)63; return capr * kBigLanes * (kBigMs * 12 + sizeof(BigLane)); }

// scratch: big_walk_scratch_bytes(cap) bytes (the lanes' milestones, interval maxima and summaries)
void launch_big_parse(const BatchArgs& a, int codec, uint32_t* list, uint32_t cap, void* recs, void* bigmeta, void* meta, void* scratch, hipStream_t s) {
    if (a.n_chunks == 0 || cap == 0) return;
    const uint32_t capr = (cap + 63u) & ~63u;
    const size_t nms = (size_t)capr * kBigLanes * kBigMs;
    unsigned long long* cands = (unsigned long long*)scratch;
    int32_t* need_iv = (int32_t*)((uint8_t*)scratch + nms * 8);
    BigLane* lanes = (BigLane*)((uint8_t*)scratch + nms * 12);
    (void)hipMemsetAsync(cands, 0xFF, nms * 8, s);          // all ones = not published
    const dim3 lgrid((a.n_chunks + 255u) / 256u);
    const dim3 wgrid((capr * kBigLanes + 64u * kBigWaves - 1u) / (64u * kBigWaves)), wblock(64u * kBigWaves);
    const dim3 sgrid((cap * kBigLanes + 255u) / 256u);
    if (codec == CJ_CODEC_SNAPPY_RAW) {
        hipLaunchKernelGGL((big_list_kernel<CJ_CODEC_SNAPPY_RAW>), lgrid, dim3(256), 0, s, a, list);
        hipLaunchKernelGGL((big_walk_kernel<CJ_CODEC_SNAPPY_RAW>), wgrid, wblock, 0, s, a, (const uint32_t*)list, capr, (uint4*)recs, cands, need_iv, lanes);
        hipLaunchKernelGGL((big_sum_kernel<CJ_CODEC_SNAPPY_RAW>), sgrid, dim3(256), 0, s, a, (const uint32_t*)list, (const uint4*)recs, (const unsigned long long*)cands, (const int32_t*)need_iv, (const BigLane*)lanes, (BigMeta*)bigmeta, (ParseMeta*)meta);
    } else {
        hipLaunchKernelGGL((big_list_kernel<CJ_CODEC_LZ4_BLOCK>), lgrid, dim3(256), 0, s, a, list);
        hipLaunchKernelGGL((big_walk_kernel<CJ_CODEC_LZ4_BLOCK>), wgrid, wblock, 0, s, a, (const uint32_t*)list, capr, (uint4*)recs, cands, need_iv, lanes);
        hipLaunchKernelGGL((big_sum_kernel<CJ_CODEC_LZ4_BLOCK>), sgrid, dim3(256), 0, s, a, (const uint32_t*)list, (const uint4*)recs, (const unsigned long long*)cands, (const int32_t*)need_iv, (const BigLane*)lanes, (BigMeta*)bigmeta, (ParseMeta*)meta);
    }
}

}  // namespace cj
