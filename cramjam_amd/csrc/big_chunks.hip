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

constexpr uint32_t kBigLead = 1024;                 // bytes in front of its boundary where a lane starts looking for the chain
constexpr uint32_t kBigSegMin = 2048;               // shortest segment
constexpr uint32_t kBigNone = 0xFFFFFFFEu, kBigNoLink = 0xFFu;      // (0xFFFFFFFF, the scratch's memset value: no candidate yet)
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
    list[kBigListHdr + atomicAdd(&list[0], 1u)] = c;            // (the list holds one slot per chunk of the batch)
}

// the listed chunks of one GROUP: entries [base, base + cap) of the list (the engine walks the list in groups of as many chunks as
// it has record areas for)
__device__ __forceinline__ uint32_t big_group_count(const uint32_t* list, uint32_t base, uint32_t cap) {
    const uint32_t listed = list[0];
    return listed > base ? (listed - base < cap ? listed - base : cap) : 0u;
}

// The lanes' view of their streams: a 16-byte WINDOW in registers, loaded straight from global memory at the position of the field
// the lane reads next.  A sequence is two fields — token + literal length bytes at ip, offset + match length bytes behind the
// literals at ip2 — and the window loaded at ip2 also holds the next token (at most 5 + 4 bytes): ONE round trip per sequence, none
// when the literals are so short that the next fields still lie inside the window.  (The first version walked on 128-byte rings in
// LDS refilled cooperatively, like the small-chunk parse kernels: that pays where a lane walks tens of KiB in short steps.  Here a
// lane walks 7 KiB, the streams of 256 KiB chunks are literal-heavy — 80 bytes per sequence on the benchmark data, so nearly every
// step left its ring — and the refill bookkeeping cost 700 - 2 000 wave-instructions per step: 2.9 ms per 8 192 chunks.  What covers
// a round trip is the other fifteen wavefronts of the CU.)
struct Win {
    uint32_t d0, d1, d2, d3;   // the bytes [wb, wb + 16) of the stream
    uint32_t wb;               // st-position of the window (0xFFFFFFFF: nothing loaded)
    __device__ __forceinline__ bool covers(uint32_t p, uint32_t nbytes) const { return p >= wb && p + nbytes <= wb + 16u; }
    // [p, p + 16), moved back where that would leave the last 16-byte granule of the stream (st-positions: base is 128-byte aligned)
    __device__ __forceinline__ void load(const uint8_t* base, uint32_t p, uint32_t safe_end) {
        wb = p + 16u <= safe_end ? p : safe_end - 16u;
        const uint4 v = ld16u(base + wb);
        d0 = v.x; d1 = v.y; d2 = v.z; d3 = v.w;
    }
    // the 4 / 8 bytes at p (p >= wb; bytes past the window read as zero)
    __device__ __forceinline__ uint32_t x32(uint32_t p) const {
        const uint32_t o = p - wb, k = o >> 2;
        const uint32_t lo = k == 0u ? d0 : k == 1u ? d1 : k == 2u ? d2 : k == 3u ? d3 : 0u;
        const uint32_t hi = k == 0u ? d1 : k == 1u ? d2 : k == 2u ? d3 : 0u;
        return __builtin_amdgcn_alignbyte(hi, lo, o & 3u);
    }
    __device__ __forceinline__ uint2 x64(uint32_t p) const {
        const uint32_t o = p - wb, k = o >> 2;
        const uint32_t a = k == 0u ? d0 : k == 1u ? d1 : k == 2u ? d2 : k == 3u ? d3 : 0u;
        const uint32_t b = k == 0u ? d1 : k == 1u ? d2 : k == 2u ? d3 : 0u;
        const uint32_t c = k == 0u ? d2 : k == 1u ? d3 : 0u;
        return make_uint2(__builtin_amdgcn_alignbyte(b, a, o & 3u), __builtin_amdgcn_alignbyte(c, b, o & 3u));
    }
};

struct BigElem { uint32_t lit, lit_at, mlen, offset, next; bool ok, last; };

// one element at st-position ip (straight-line for the common shape, the grammar's general function for the lanes that meet
// anything else): the step of lz4_parse_kernel / snappy_parse_kernel without the checks that need the output position
template <int kCodec>
__device__ __forceinline__ BigElem big_elem(Win& w, const uint8_t* base, uint32_t ip, uint32_t iend, uint32_t safe_end, bool going) {
    BigElem e;
    bool fast = false;
    {   // field A at ip: normally inside the window that was loaded for the previous sequence's field B
        const bool cold = going && !w.covers(ip, 4u);
        if (ballot64(cold) != 0ull) { if (cold) w.load(base, ip, safe_end); }
    }
    const uint32_t t4 = w.x32(ip);
    uint32_t ip2, nxf;                       // field B's position; bytes of field B
    bool fa;                                 // field A has the common shape
    if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
        const uint32_t token = t4 & 0xffu, e1 = (t4 >> 8) & 0xffu, e2 = (t4 >> 16) & 0xffu, e3 = t4 >> 24;
        const bool x1 = (token >> 4) == 15u, x2 = x1 && e1 == 255u, x3 = x2 && e2 == 255u;
        const uint32_t lit = (token >> 4) + (x1 ? e1 : 0u) + (x2 ? e2 : 0u) + (x3 ? e3 : 0u);
        const uint32_t ip1 = ip + 1u + (x1 ? 1u : 0u) + (x2 ? 1u : 0u) + (x3 ? 1u : 0u);
        ip2 = ip1 + lit;
        fa = !(x3 && e3 == 255u) && ip2 + 8u <= iend;                   // (iend - ip1 >= lit + 8: not the block's last sequences)
        e.lit = lit; e.lit_at = ip1; e.mlen = token & 15u;
    } else {
        const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
        const bool is_lit = (tag & 3u) == 0u;
        // (literal headers of up to 4 bytes, and below copy-4 elements and a literal behind a literal, are straight-line too: a lane's lead-in
        //  reads every byte as a tag — a quarter of them copy-4 tags —, and the general function reads through global memory: f06)
        const uint32_t lhdr = is_lit ? (l6 < 60u ? 1u : l6 - 58u) : 0u;
        const uint32_t lit = is_lit ? (l6 < 60u ? l6 + 1u : ((t4 >> 8) & (0xffffffu >> (8u * (62u - (l6 > 62u ? 62u : l6))))) + 1u) : 0u;
        ip2 = ip + lhdr + lit;
        fa = !(is_lit && l6 > 62u) && ip2 + 8u <= iend && ip2 >= ip;
        e.lit = lit; e.lit_at = ip + lhdr; e.mlen = is_lit ? 1u : 0u;      // (mlen: "a literal came first", for the shape test below)
    }
    {   // field B at ip2 and the field A behind it (at most 5 + 4 bytes)
        const bool far = going && fa && !w.covers(ip2, 9u);
        if (ballot64(far) != 0ull) { if (far) w.load(base, ip2, safe_end); }
    }
    const uint2 o8 = w.x64(ip2);
    if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
        const uint32_t f1 = (o8.x >> 16) & 0xffu, f2 = o8.x >> 24, f3 = o8.y & 0xffu;
        const bool y1 = e.mlen == 15u, y2 = y1 && f1 == 255u, y3 = y2 && f2 == 255u;
        fast = fa && !(y3 && f3 == 255u);
        e.offset = o8.x & 0xffffu;
        e.mlen = e.mlen + (y1 ? f1 : 0u) + (y2 ? f2 : 0u) + (y3 ? f3 : 0u) + 4u;
        e.next = ip2 + 2u + (y1 ? 1u : 0u) + (y2 ? 1u : 0u) + (y3 ? 1u : 0u);
        nxf = 0u;
    } else {
        const uint32_t c4 = o8.x, ctag = c4 & 0xffu, kind = ctag & 3u;
        const bool had_lit = e.mlen != 0u;
        const uint32_t ip3 = kind == 0u ? ip2 : ip2 + (kind == 1u ? 2u : kind == 2u ? 3u : 5u);
        fast = fa && (kind != 0u || had_lit) && ip3 < iend;
        e.mlen = kind == 0u ? 0u : kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
        e.offset = kind == 0u ? 0u : kind == 1u ? ((ctag >> 5) << 8) | ((c4 >> 8) & 0xffu) : kind == 2u ? (c4 >> 8) & 0xffffu : (o8.x >> 8) | (o8.y << 24);
        e.next = ip3;
        nxf = 0u;
    }
    (void)nxf;
    e.ok = true; e.last = false;
    if (ballot64(going && !fast) != 0ull) {
        if (going && !fast) {
            using G = typename std::conditional<kCodec == CJ_CODEC_SNAPPY_RAW, SnappyGrammar, Lz4Grammar>::type;
            const auto rd = [base, iend](uint32_t q) {                  // anywhere in the stream (zero-filled past its end)
                const uint32_t v = ld_le_tail(base, q, iend);
                __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
                return v;
            };
            Seq sq;
            e.ok = G::at(rd, ip, iend, sq, base);
            e.lit = sq.lit; e.lit_at = sq.lit_at; e.mlen = sq.mlen; e.offset = sq.offset; e.next = sq.next; e.last = sq.last;
        }
    }
    return e;
}

// what a lane of the walk leaves for the epilogue
struct BigLane { uint32_t cnt, r, lim, link; int32_t need; uint32_t flags, pad, spare; };      // flags: 1 = bad, 2 = saw the stream's last element
constexpr uint32_t kBigLaneBad = 1u, kBigLaneLast = 2u;

// THE WALK.  Work unit = (listed chunk bi, segment j); a wavefront takes the SAME segment of 64 consecutive chunks (unit u -> j = u /
// capr, bi = u % capr, capr = the list's capacity rounded up to 64): the lanes of a wavefront run in lock step, and the segments of
// one chunk differ — on the benchmark data the first 7 KiB of a stream hold 370 sequences, a later 7 KiB eighty — while the same
// segment of different chunks does not; with a chunk's lanes side by side in one wavefront every wavefront ran as long as its
// densest segment (3.2 ms for 8 192 chunks).  The lanes of a chunk therefore meet through GLOBAL memory: a lane publishes its
// candidate (agent-scope store), a lane that has crossed a boundary looks the candidate of the segment it stands in up (agent-
// scope load) — and if that lane has not got that far yet (its wavefront may not even have started), it simply WALKS ON as it
// does past a wrong candidate: never wrong, only double work.
template <int kCodec>
__global__ __launch_bounds__(64 * kBigWaves) void big_walk_kernel(BatchArgs a, const uint32_t* list, uint32_t base_i, uint32_t cap_g, uint32_t capr, uint4* recs, uint32_t* cands, BigLane* lanes) {
    constexpr uint32_t k = kBigLanes;
    const uint32_t gl = blockIdx.x * (64u * kBigWaves) + threadIdx.x;
    const uint32_t j = gl / capr, bi = gl % capr;                // (capr is a multiple of 64: a wavefront has one j)
    const uint32_t listed = big_group_count(list, base_i, cap_g);
    const bool exists = bi < listed && j < k;
    const uint32_t c = exists ? list[kBigListHdr + base_i + bi] : 0u;
    uint32_t* ccand = cands + (size_t)bi * k;

    const uint8_t* in = nullptr;
    uint32_t n = 0, cap = 0, skip = 0;
    const bool walk = exists && big_prologue<kCodec>(a, c, in, n, cap, skip);          // (true for every listed chunk)
    uint32_t seg = (((n + k - 1u) >> kBigLanesLog) + 15u) & ~15u;
    seg = seg < kBigSegMin ? kBigSegMin : seg;
    const uint32_t bj = j * seg;
    bool done = !(walk && bj < n);
    const uint32_t mis = done ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(in) & 127u);
    const uint8_t* base = done ? nullptr : in - mis;               // 128-byte aligned
    const uint32_t iend = done ? 0u : mis + n;
    const uint32_t safe_end = (iend + 15u) & ~15u;                  // reads stay inside the stream's last 16-byte granule (cramjam_hip.h)
    Win w = {0u, 0u, 0u, 0u, 0xFFFFFFFFu};
    // positions below are st-positions (offsets from base): stream position + mis
    const uint32_t my_b = mis + bj;
    uint32_t ip = mis + (j == 0u || bj <= kBigLead ? 0u : bj - kBigLead);
    bool lead = !done && j != 0u, retried = false;
    uint32_t lsteps = 0;                                         // elements walked in the lead-in
    uint32_t tb = j + 1u;
    uint32_t next_b = tb < k ? mis + tb * seg : 0xFFFFFFFFu;
    if (exists && (done || j == 0u)) __hip_atomic_store(ccand + j, done ? kBigNone : mis, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t link = kBigNoLink;
    uint32_t cnt = 0, r = 0, lim = 0;                            // records written, output bytes so far (from 0), LZ4's end-of-block margin
    int32_t need = 0;                                            // how far in front of this lane's first output byte its matches reach
    bool bad = false, saw_last = false, fin = false;
    uint4* region = recs + (size_t)bi * kBigRecPitch + (size_t)j * kBigRegion;
    // records leave in groups of two slots = one aligned 32-byte store, per lane: record k of the lane in slot k, the region's
    // sentinel behind the last one
    uint4 p0 = make_uint4(0, 0, 0, 0);
    uint32_t nput = 0;
    const auto put = [&](const uint4& v, bool last_one) {
        if ((nput & 1u) == 0u) { p0 = v; if (last_one) region[nput] = v; }
        else { region[nput - 1u] = p0; region[nput] = v; }
        nput += 1u;
    };
    // one element of the real walk: the checks that do not need the absolute output position, its record
    const auto take = [&](const BigElem& e) {
        const uint32_t op2 = r + e.lit;
        const bool has_match = kCodec == CJ_CODEC_LZ4_BLOCK ? !e.last : e.mlen != 0u;
        const uint32_t off16 = e.offset & 0xffffu;
        // (a Snappy copy that reaches further back than 65 535 bytes needs more than the previous slab: wavefront kernel)
        const bool bad_now = !e.ok || (has_match && (e.offset == 0u || e.offset > 0xffffu)) || cnt + 4u > kBigRegion
                             || (!e.last && e.next >= iend) || e.lit > kBigInMax || e.mlen > 0x00ffffffu;
        const int32_t reach = has_match ? (int32_t)off16 - (int32_t)op2 : need;
        need = reach > need ? reach : need;
        if (kCodec == CJ_CODEC_LZ4_BLOCK && has_match) lim = op2 + (e.mlen + 5u > 12u ? e.mlen + 5u : 12u);
        const uint32_t mlen = has_match ? e.mlen : 0u;
        put(make_uint4((e.lit_at - mis) | ((mlen >> 16) << 24), e.lit, r, (has_match ? off16 : 0u) | ((mlen & 0xffffu) << 16)), false);
        cnt += 1u;
        r = op2 + mlen;
        ip = e.next;
        saw_last = e.last;
        fin = e.last;
        if (bad_now || r > kBigOutMax) { bad = true; done = true; }
    };

    for (;;) {
        if (ballot64(!done) == 0ull) break;
        bool go = !done && !fin;
        if (ballot64(!done && (fin || (!lead && ip >= next_b))) != 0ull) {
            if (!done && fin) { put(make_uint4(0u, 0u, r, 0u), true); done = true; }      // the sentinel behind the last record
            else if (go && !lead && ip >= next_b) {              // across a boundary, on a token: the candidate of the segment it is in now?
                while (tb + 1u < k && ip >= next_b + seg) { tb += 1u; next_b += seg; }
                const uint32_t cm = __hip_atomic_load(ccand + tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cm == ip) { link = tb; done = true; go = false; put(make_uint4(0u, 0u, r, 0u), true); }      // the region's sentinel
                else { tb += 1u; next_b = tb < k ? next_b + seg : 0xFFFFFFFFu; }        // another position, or not known yet: walk on
            }
        }
        const BigElem e = big_elem<kCodec>(w, base, ip, iend, safe_end, go);
        if (ballot64(go && lead) != 0ull) {
            if (go && lead) {
                uint32_t nx = e.ok && !e.last ? e.next : ip + 1u;          // malformed here = this was no token: try the next byte
                nx = nx <= ip ? ip + 1u : nx;
                ip = nx;
                lsteps += 1u;
                // A walk from a wrong position falls into step with the chain with some probability per ELEMENT of the chain it passes, so
                // the lead-in has to cover enough elements, not bytes: where 1 KiB held fewer than 24 (long literal runs), the lane
                // starts over further back, at 32 elements' worth by the density it has just seen.
                if (ip >= my_b && lsteps < 24u && !retried) {
                    retried = true;
                    uint32_t back = 32u * kBigLead / lsteps;
                    back = back < 2u * kBigLead ? 2u * kBigLead : (back > 8u * kBigLead ? 8u * kBigLead : back);
                    ip = mis + (bj > back ? bj - back : 0u);
                    lsteps = 0u;
                } else
                if (ip >= my_b) {
                    lead = false;
                    const uint32_t cand = ip < iend ? ip : kBigNone;
                    __hip_atomic_store(ccand + j, cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cand == kBigNone) done = true;
                }
                go = false;
            }
        }
        if (go) take(e);
        // (more elements per trip, the parse kernels' trick, LOSE here: 1 / 2 / 3 / 7 more 2.39 / 2.71 / 2.78 / 3.73 ms against 2.25 — this
        //  element function carries two window tests with their ballots, it is no cheaper than the trip)
    }
    if ((nput & 1u) != 0u && !done) region[nput - 1u] = p0;      // (cannot happen: every lane that wrote a record ends with its sentinel or as bad)

    if (exists) {
        if (bad && (nput & 1u) != 0u) region[nput - 1u] = p0;      // (a bad lane's records are never used; kept complete for debugging)
        const BigLane bl = {cnt, r, lim, link, need, (bad ? kBigLaneBad : 0u) | (saw_last ? kBigLaneLast : 0u), 0u, 0u};
        lanes[(size_t)bi * k + j] = bl;
    }
}

// THE EPILOGUE, 32 lanes per listed chunk: which lanes does the chain run through, what lies in front of each, the checks that
// needed absolute positions, and for every 64 KiB boundary of the output the record that holds it
template <int kCodec>
__global__ __launch_bounds__(256) void big_sum_kernel(BatchArgs a, const uint32_t* list, uint32_t base_i, uint32_t cap_g, const uint4* recs, const BigLane* lanes, BigMeta* bigmeta, ParseMeta* meta) {
    __shared__ volatile uint32_t s_live[256];
    constexpr uint32_t k = kBigLanes;
    const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
    const uint32_t bi = gl >> kBigLanesLog, j = gl & (k - 1u);
    const uint32_t lane = lane_id();
    const uint32_t g0 = threadIdx.x & ~(k - 1u);
    const uint32_t listed = big_group_count(list, base_i, cap_g);
    const bool exists = bi < listed;
    const uint32_t c = exists ? list[kBigListHdr + base_i + bi] : 0u;
    const uint8_t* in = nullptr;
    uint32_t n = 0, cap = 0, skip = 0;
    const bool walk = exists && big_prologue<kCodec>(a, c, in, n, cap, skip);
    BigLane bl = {0u, 0u, 0u, kBigNoLink, 0, kBigLaneBad, 0u, 0u};
    if (walk) bl = lanes[(size_t)bi * k + j];
    const uint32_t cnt = bl.cnt, r = bl.r, lim = bl.lim, link = bl.link, pad = bl.pad;
    const int32_t need = bl.need;
    const bool bad = (bl.flags & kBigLaneBad) != 0u, saw_last = (bl.flags & kBigLaneLast) != 0u;
    const uint4* region = recs + (size_t)bi * kBigRecPitch + (size_t)j * kBigRegion;
    volatile uint32_t* my_live = s_live + threadIdx.x;
    bool live = walk && j == 0u;
    *my_live = live ? 1u : 0u;
    for (uint32_t round = 1; round < k; round++) {               // (the 32 lanes of a chunk sit in one wavefront: in order, no barrier)
        if (live && link != kBigNoLink) s_live[g0 + link] = 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        live = *my_live != 0u;
    }
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
        lane_ok = !bad && (link != kBigNoLink || saw_last) && need <= (int32_t)opb;
        if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
            if (lim != 0u) lane_ok = lane_ok && (uint64_t)opb + lim <= cap;          // its last sequence with a match
            if (saw_last) lane_ok = lane_ok && (uint64_t)opb + r <= cap;
        }
    }
    const uint64_t okm = ballot64(lane_ok), lastm = ballot64(live && saw_last);
    const uint64_t gmask = (k >= 64u ? ~0ull : ((1ull << k) - 1ull)) << (lane & ~(k - 1u));
    bool chunk_ok = walk && (okm & gmask) == gmask && __popcll(lastm & gmask) == 1;
    if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) chunk_ok = chunk_ok && total == cap;
    else chunk_ok = chunk_ok && total <= cap;
    chunk_ok = chunk_ok && total > kLdsOutMax && nseq >= 1u;      // (a chunk that decodes to at most 64 KiB: the wavefront kernel — rare, and the slab walk below assumes two slabs)

    // ---- the record that holds output byte kBigSlabBytes * s: the live lane whose output range contains it searches its region ----
    BigMeta* bm = bigmeta + bi;
    uint32_t sf[kBigSlabs];                                      // slab s's first record, known to all 32 lanes of the chunk after the reduction
    sf[0] = 0u;
    for (uint32_t sb = 1; sb < kBigSlabs; sb++) {
        uint32_t mine = 0xFFFFFFFFu;
        const uint32_t b = sb * kBigSlabBytes;
        if (exists && chunk_ok && live && b < total && b >= opb && b < opb + r) {      // in this lane's part (r > 0 here)
            const uint32_t rel = b - opb;
            uint32_t lo = 0, hi = cnt;                             // largest idx in [0, cnt) with lit_start[idx] <= rel (idx 0 has lit_start 0)
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t z = __hip_atomic_load(&reinterpret_cast<const uint32_t*>(region + pad + mid)[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (z <= rel) lo = mid; else hi = mid;
            }
            mine = first + lo;
        }
        for (uint32_t d = 1; d < k; d <<= 1) {                   // minimum over the chunk's 32 lanes (they sit side by side in one wavefront)
            const uint32_t o = (uint32_t)__shfl_xor((int)mine, (int)d, 64);
            mine = o < mine ? o : mine;
        }
        sf[sb] = mine;
    }
    // The slab decoder's record table and cross list hold kBigSlabRecs records per slab: an LZ4 sequence with a match covers at least
    // four output bytes, so a 64 KiB slab never has more — a Snappy stream may (copies and literals of one byte): such a chunk
    // stays with the wavefront kernel.
    for (uint32_t sb = 0; sb < kBigSlabs; sb++) {
        if (sb * kBigSlabBytes >= total) break;
        const uint32_t r0 = sf[sb], r1 = (sb + 1u) * kBigSlabBytes < total ? sf[sb + 1u] : nseq - 1u;
        if (r0 == 0xFFFFFFFFu || r1 == 0xFFFFFFFFu || r1 < r0 || r1 - r0 + 1u > kBigSlabRecs) chunk_ok = false;
    }
    if (exists && chunk_ok && j == 0u)
        for (uint32_t sb = 1; sb < kBigSlabs; sb++) if (sb * kBigSlabBytes < total) bm->slab_first[sb] = sf[sb];
    if (exists) {
        if (chunk_ok) {
            bm->first[j] = first;
            bm->opb[j] = opb | (pad << 28);
            if (j == 0u) {
                bm->chunk = c; bm->nseq = nseq; bm->in_skip = skip; bm->U = total; bm->slab_first[0] = 0u;
                a.result[c] = (int64_t)total;
                meta[c] = ParseMeta{0u, 0u};                       // handled here: the wavefront kernel skips it
            }
        } else if (j == 0u) { bm->chunk = c; bm->nseq = 0u; }      // stays with the wavefront kernel (the small-chunk pipeline flagged it)
#ifdef CJ_BIG_DEBUG
        if (!chunk_ok && j == 0u && bi < 4096u) printf("reject bi=%u c=%u walk=%d okm=%llx lastm=%llx livem=%llx total=%u cap=%u nseq=%u\n", bi, c, (int)walk, (unsigned long long)((okm & gmask) >> (lane & ~(k-1u))), (unsigned long long)((lastm & gmask) >> (lane & ~(k-1u))), (unsigned long long)((ballot64(live) & gmask) >> (lane & ~(k-1u))), total, cap, nseq);
#endif
    }
}

__global__ __launch_bounds__(256) void big_items_kernel(BatchArgs a, const uint32_t* list, uint32_t base_i, const BigMeta* bigmeta, const uint4* recs, uint32_t cap,
                                                        uint64_t* rows, ParseMeta* item_meta, uint32_t* done) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x, items = kBigSlabs * cap;
    if (w >= items) return;
    const uint32_t bi = w % cap, sl = w / cap;
    const uint32_t listed = big_group_count(list, base_i, cap);
    uint64_t in_off = 0, in_len = 0, out_off = 0, out_cap = 0, res = 0;
    uint32_t nrec = 0, pf = 0;
    if (bi < listed) {
        const BigMeta* bm = bigmeta + bi;
        const uint32_t U = bm->U, nseq = bm->nseq;
        if (nseq != 0u && sl * kBigSlabBytes < U) {
            const uint32_t c = bm->chunk;
            const uint32_t R0 = bm->slab_first[sl], R1 = (sl + 1u) * kBigSlabBytes < U ? bm->slab_first[sl + 1u] : nseq - 1u;
            nrec = R1 - R0 + 1u;
            in_off = a.in_off[c] + bm->in_skip; in_len = a.in_len[c] - bm->in_skip;
            out_off = a.out_off[c] + (uint64_t)sl * kBigSlabBytes; out_cap = (uint64_t)sl * kBigSlabBytes;      // (out_cap = the slab's first output position in its chunk: what the slab mode calls the stream position)
            res = U - sl * kBigSlabBytes < kBigSlabBytes ? U - sl * kBigSlabBytes : kBigSlabBytes;
            // the 128-byte lines (relative to the element stream) that hold the slab's literals: from its first record's literal source to
            // the end of its last record's literals — the slab decoder touches them while it expands the records (kRecFeed)
            const auto rec_at = [&](uint32_t gi) {
                uint32_t t = 0;
                for (uint32_t step = kBigLanes / 2u; step != 0u; step >>= 1) t += gi >= bm->first[t + step] ? step : 0u;
                return recs[(size_t)bi * kBigRecPitch + gi + t * kBigRegion + (bm->opb[t] >> 28) - (t ? bm->first[t] : 0u)];
            };
            const uint4 ra = rec_at(R0), rb = rec_at(R1);
            const uint32_t lo = (ra.x & 0x00ffffffu) >> 7, hi = ((rb.x & 0x00ffffffu) + rb.y + 127u) >> 7;
            const uint32_t nl = hi > lo ? hi - lo + 1u : 1u;
            pf = (lo & 0xfffu) | ((nl < 1023u ? nl : 1023u) << 12);
        }
    }
    rows[w] = in_off; rows[items + w] = in_len; rows[2 * (size_t)items + w] = out_off; rows[3 * (size_t)items + w] = out_cap; rows[4 * (size_t)items + w] = res;
    item_meta[w] = ParseMeta{nrec, pf};
    done[w] = 0u;
}

void launch_big_items(const BatchArgs& a, const uint32_t* list, uint32_t base, const void* bigmeta, const void* recs, uint32_t cap, uint64_t* rows, void* item_meta, uint32_t* done, hipStream_t s) {
    if (cap == 0) return;
    hipLaunchKernelGGL(big_items_kernel, dim3((kBigSlabs * cap + 255u) / 256u), dim3(256), 0, s, a, list, base, (const BigMeta*)bigmeta, (const uint4*)recs, cap, rows, (ParseMeta*)item_meta, done);
}

size_t big_recs_bytes(size_t cap) { return cap * (size_t)kBigRecPitch * sizeof(uint4); }
size_t big_meta_bytes(size_t cap) { return cap * sizeof(BigMeta); }

size_t big_walk_scratch_bytes(size_t cap) { const size_t capr = (cap + 63) & ~(size_t)63; return capr * kBigLanes * (4 + sizeof(BigLane)); }

// list[0] = number of listed chunks (zeroed here, counted on the device), list[4 + i] = chunk index; room for every chunk of the batch
void launch_big_list(const BatchArgs& a, int codec, uint32_t* list, hipStream_t s) {
    if (a.n_chunks == 0) return;
    (void)hipMemsetAsync(list, 0, 16, s);
    const dim3 lgrid((a.n_chunks + 255u) / 256u);
    if (codec == CJ_CODEC_SNAPPY_RAW) hipLaunchKernelGGL((big_list_kernel<CJ_CODEC_SNAPPY_RAW>), lgrid, dim3(256), 0, s, a, list);
    else hipLaunchKernelGGL((big_list_kernel<CJ_CODEC_LZ4_BLOCK>), lgrid, dim3(256), 0, s, a, list);
}

// one group of the list: entries [base, base + cap).  scratch: big_walk_scratch_bytes(cap) bytes (the lanes' candidates and summaries)
void launch_big_parse(const BatchArgs& a, int codec, const uint32_t* list, uint32_t base, uint32_t cap, void* recs, void* bigmeta, void* meta, void* scratch, hipStream_t s) {
    if (a.n_chunks == 0 || cap == 0) return;
    const uint32_t capr = (cap + 63u) & ~63u;
    uint32_t* cands = (uint32_t*)scratch;
    BigLane* lanes = (BigLane*)((uint8_t*)scratch + (size_t)capr * kBigLanes * 4);
    (void)hipMemsetAsync(cands, 0xFF, (size_t)capr * kBigLanes * 4, s);          // 0xFFFFFFFF = no candidate yet
    const dim3 wgrid((capr * kBigLanes + 64u * kBigWaves - 1u) / (64u * kBigWaves)), wblock(64u * kBigWaves);
    const dim3 sgrid((cap * kBigLanes + 255u) / 256u);
    if (codec == CJ_CODEC_SNAPPY_RAW) {
        hipLaunchKernelGGL((big_walk_kernel<CJ_CODEC_SNAPPY_RAW>), wgrid, wblock, 0, s, a, list, base, cap, capr, (uint4*)recs, cands, lanes);
        hipLaunchKernelGGL((big_sum_kernel<CJ_CODEC_SNAPPY_RAW>), sgrid, dim3(256), 0, s, a, list, base, cap, (const uint4*)recs, (const BigLane*)lanes, (BigMeta*)bigmeta, (ParseMeta*)meta);
    } else {
        hipLaunchKernelGGL((big_walk_kernel<CJ_CODEC_LZ4_BLOCK>), wgrid, wblock, 0, s, a, list, base, cap, capr, (uint4*)recs, cands, lanes);
        hipLaunchKernelGGL((big_sum_kernel<CJ_CODEC_LZ4_BLOCK>), sgrid, dim3(256), 0, s, a, list, base, cap, (const uint4*)recs, (const BigLane*)lanes, (BigMeta*)bigmeta, (ParseMeta*)meta);
    }
}

}  // namespace cj
