// seg_parse.hip — the SEGMENTED PARSE of a batch of independent chunks: LZ4 block / Snappy raw token chains walked by k lanes
// per chunk instead of one, and written out as the workgroup decoder's own 8-byte records (seg_parse.hpp).
// Accept / reject rules: those of lz4_parse_kernel / snappy_parse_kernel (liblz4 1.10.0 LZ4_decompress_safe, snap 1.1.1
// raw::Decoder; reference call sites /root/reference/src/lz4.rs:88,164,168, src/snappy.rs:57,106) — and where this kernel is
// not sure (ANY violation, too few / too many sequences) it hands the chunk to the wavefront-per-chunk kernel, which decodes
// every valid chunk and names every error exactly.
//
// Why: one chunk is ONE serial chain of ~2 700 sequences (benchmark data; text ~10 000), and a lane-per-chunk walk of 100 000
// chunks is 1 563 wavefronts on 1 024 SIMDs that each retire a dependent step every ~0.9 µs: 2.45 ms with the GPU almost
// empty (profiles/r03).  The chain has to be cut.
//   * A chunk gets k = 2^klog lanes.  Lane j owns the sequences whose TOKEN lies in [c_j, c_{j+1}), where c_0 = 0 and c_j is the
//     first token position at or behind the boundary b_j = j * seg of the compressed bytes.
//   * Nobody knows c_j in advance: lane j finds a CANDIDATE by starting kLead bytes in front of b_j and walking (the
//     next-token function depends on the bytes only, and a walk from a wrong position falls into step with the true chain
//     after a few elements: tools/models/seg_parse_sync.py — 0 of 640 boundaries wrong after 1 KiB on the benchmark data).
//     A malformed element in that lead-in means "not a token": the walk restarts one byte further.
//   * From its candidate on the lane walks for real: validates what can be validated without the absolute output position,
//     writes records with positions relative to its own start, and stops when it stands EXACTLY on the candidate of the
//     segment it has reached — from there that lane's records are the chain's.  If the candidate there is another position
//     (the lead-in had not synchronised, or the chain jumps over the segment in one long literal run) it walks on: a
//     wrong candidate costs time, never correctness.  Lane 0 starts on the true first token, so by induction the lanes it
//     reaches — the LIVE lanes — hold exactly the chain's sequences, each once.
//   * Epilogue: the live lanes' counts and output bytes are summed up in order (SegMeta), the checks that needed absolute
//     positions are made (every offset reaches back at most to output position 0, LZ4's end-of-block margins, Snappy's
//     announced length), and lane 0 writes the chunk's verdict.
#include "seg_parse.hpp"
#include "parse_grammar.hpp"

namespace cj {

#ifndef CJ_SEG_LEAD
#define CJ_SEG_LEAD 1024u
#endif
constexpr uint32_t kSegLead = CJ_SEG_LEAD;          // bytes in front of its boundary where a lane starts looking for the chain
constexpr uint32_t kSegMinBytes = 1024;             // shortest segment: a small chunk is walked by fewer lanes
constexpr uint32_t kSegWaves = 4;                   // wavefronts per block
constexpr uint32_t kCandPending = 0xFFFFFFFFu, kCandNone = 0xFFFFFFFEu;
constexpr uint32_t kLinkNone = 0xFFu;

// ---------------------------------------------------------------------------------------------------------------------------
// The lanes' view of their streams: a 128-byte ring per lane in LDS (two 64-byte units), refilled cooperatively like the
// older parse kernels' 256-byte rings (lane_stream.hpp) — 4 lanes fetch one lane's next unit with aligned 16-byte loads, 16
// units per load instruction — but half the size: 9 KiB per wavefront instead of 17, so that sixteen wavefronts share a CU
// instead of nine.  A lane-per-chunk walk retires an instruction every ~12 cycles per wavefront (dependent reads, short
// dependent chains): what fills the SIMDs is MORE WAVEFRONTS, and the segments supply them.  A refill round waits for its
// loads (one HBM round trip), which the other wavefronts of the CU cover.
// (Measured alternative, profiles/r04/experiments: asynchronous LDS-DMA refills — global_load_lds_dwordx4 into a piece-major
//  ring — need one instruction per ring slot and lane group, the M0 base being wave-uniform: ~85 instructions per round
//  against ~60 here, and rounds come every other step whatever the mechanism, because 64 unsynchronised lanes share them.)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSsRing = 128, kSsUnit = 64;
constexpr uint32_t kSsStride = kSsRing + 16u;           // 16-byte aligned rings (one ds_write_b128 per fetched piece)
constexpr uint32_t kSsWaveBytes = 64u * kSsStride;
constexpr uint32_t kSsLanesPerUnit = kSsUnit / 16u, kSsTargets = 64u / kSsLanesPerUnit, kSsLoads = 64u / kSsTargets;
#ifndef CJ_SEG_AHEAD
#define CJ_SEG_AHEAD 32u
#endif
constexpr uint32_t kSegAhead = CJ_SEG_AHEAD;        // cached bytes a lane must have ahead before a step

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

// one element at st-position ip: literal length / position, match length (0 = none), offset, position of the next element.
// ok = false: malformed on whatever path it lies.  Straight-line for the common shape (both reads inside the lane's ring, no
// long length extension, not near the end); the lanes that meet anything else take the grammar's general function.
struct Elem { uint32_t lit, lit_at, mlen, offset, next; bool ok, last; };

template <int kCodec>
__device__ __forceinline__ Elem seg_elem(const SegStream& st, uint32_t ip, uint32_t iend, bool going) {
    Elem e;
    bool fast = false;
    if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
        const uint32_t t4 = st.ring32(ip);
        const uint32_t token = t4 & 0xffu, e1 = (t4 >> 8) & 0xffu;
        const bool x1 = (token >> 4) == 15u;
        const uint32_t lit = (token >> 4) + (x1 ? e1 : 0u);
        const uint32_t ip1 = ip + 1u + (x1 ? 1u : 0u), ip2 = ip1 + lit;
        const bool w1 = st.in_window(ip), w2 = ip2 + 4u <= st.hi && ip2 + 4u <= iend;
        const uint32_t o4 = st.ring32(ip2);
        const uint32_t mc = token & 15u, e2 = (o4 >> 16) & 0xffu;
        const bool x2 = mc == 15u;
        // rem_in >= lit + 8 implies every bound the general walk checks while it reads one-byte extensions
        fast = w1 && w2 && !(x1 && e1 == 255u) && !(x2 && e2 == 255u) && iend - ip1 >= lit + 8u;
        e.lit = lit; e.lit_at = ip1; e.offset = o4 & 0xffffu; e.mlen = mc + (x2 ? e2 : 0u) + 4u;
        e.next = ip2 + 2u + (x2 ? 1u : 0u); e.ok = true; e.last = false;
    } else {
        const uint32_t t4 = st.ring32(ip);
        const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
        const bool is_lit = (tag & 3u) == 0u;
        const uint32_t lhdr = is_lit ? (l6 == 60u ? 2u : 1u) : 0u;
        const uint32_t lit = is_lit ? (l6 == 60u ? ((t4 >> 8) & 0xffu) + 1u : l6 + 1u) : 0u;
        const uint32_t ip2 = ip + lhdr + lit;                                     // the copy element
        const bool w1 = st.in_window(ip), w2 = ip2 + 4u <= st.hi && ip2 + 4u <= iend;
        const uint32_t c4 = st.ring32(ip2);
        const uint32_t ctag = c4 & 0xffu, kind = ctag & 3u;
        const uint32_t ip3 = ip2 + (kind == 1u ? 2u : 3u);
        fast = w1 && w2 && !(is_lit && l6 > 60u) && (kind == 1u || kind == 2u) && ip3 < iend;
        e.lit = lit; e.lit_at = ip + lhdr;
        e.mlen = kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
        e.offset = kind == 1u ? ((ctag >> 5) << 8) | ((c4 >> 8) & 0xffu) : (c4 >> 8) & 0xffffu;
        e.next = ip3; e.ok = true; e.last = false;
    }
    if (ballot64(going && !fast) != 0ull) {
        if (going && !fast) {
            using G = typename std::conditional<kCodec == CJ_CODEC_SNAPPY_RAW, SnappyGrammar, Lz4Grammar>::type;
            const auto rd = [&st](uint32_t p) { return st.ld32(p); };
            Seq s;
            e.ok = G::at(rd, ip, iend, s);
            e.lit = s.lit; e.lit_at = s.lit_at; e.mlen = s.mlen; e.offset = s.offset; e.next = s.next; e.last = s.last;
        }
    }
    return e;
}

template <int kCodec>
__global__ __launch_bounds__(64 * kSegWaves) void seg_parse_kernel(BatchArgs a, uint32_t klog, uint2* recs, SegMeta* segmeta, ParseMeta* meta) {
    __shared__ __attribute__((aligned(16))) uint8_t rings[kSegWaves * kSsWaveBytes];
    __shared__ volatile uint32_t s_cand[kSegWaves * 64];
    __shared__ volatile uint32_t s_live[kSegWaves * 64];
    const uint32_t k = 1u << klog;
    const uint32_t gl = blockIdx.x * (64u * kSegWaves) + threadIdx.x;
    const uint32_t c = gl >> klog, j = gl & (k - 1u);
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t g0 = wave * 64u + (lane & ~(k - 1u));           // index (in the block) of lane 0 of my chunk
    const uint32_t wave_ring = (uint32_t)(uintptr_t)rings + wave * kSsWaveBytes;
    const bool exists = c < a.n_chunks;

    // ---- per-chunk prologue, computed by each of the chunk's lanes (the prologues of lz4_parse_kernel / snappy_parse_kernel) ----
    const uint8_t* in = nullptr;                // the element stream (behind a size prefix / the length preamble)
    uint32_t n = 0, cap = 0, skip = 0;          // its length; LZ4: output capacity, Snappy: announced length; bytes in front of it
    ParseMeta pm = {0u, 0u};
    int64_t res = 0;
    bool walk = false;                          // the chunk is walked (else res / pm say what becomes of it)
    if (exists) {
        const uint8_t* in0 = a.in_base + a.in_off[c];
        uint64_t n64 = a.in_len[c], cap64 = a.out_cap[c];
        if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
            const uint8_t* inp = in0;
            res = lz4_block_prologue(a.flags, inp, n64, cap64);
            if (res == 0) {
                const uint32_t cap0 = (uint32_t)cap64, iend0 = (uint32_t)n64;
                if (cap0 == 0) res = (iend0 == 1 && inp[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT;
                else if (iend0 == 0) res = CJ_E_CORRUPT;
                else if (cap0 > kLdsOutMax || iend0 > kLdsInMax) pm.in_skip = kRouteWave;      // too big for the LDS window
                else { walk = true; in = inp; n = iend0; cap = cap0; skip = (uint32_t)(inp - in0); }
            }
        } else {
            if (n64 == 0) res = CJ_E_SNAPPY_EMPTY;
            else if (n64 > 0xFFFFFFF0ull) res = CJ_E_SNAPPY_CORRUPT;
            else {
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
                if (!ok) res = CJ_E_SNAPPY_HEADER;
                else if (ulen > 0xFFFFFFFFull) res = CJ_E_SNAPPY_TOO_BIG;
                else if (ulen > cap64) res = CJ_E_SNAPPY_BUF_SMALL;
                else if (ulen == 0) res = (hdr == (uint32_t)n64) ? 0 : (int64_t)CJ_E_SNAPPY_CORRUPT;   // nothing to decode: trailing elements are errors
                else if (ulen > kLdsOutMax || n64 - hdr > kLdsInMax || hdr == (uint32_t)n64) pm.in_skip = kRouteWave;
                else { walk = true; in = in0 + hdr; n = (uint32_t)n64 - hdr; cap = (uint32_t)ulen; skip = hdr; }
            }
        }
    }
    // ---- segments: seg bytes each (a multiple of 16, at least kSegMinBytes); lane j's boundary b_j = j * seg ----
    uint32_t seg = (((n + k - 1u) >> klog) + 15u) & ~15u;
    seg = seg < kSegMinBytes ? kSegMinBytes : seg;
    const uint32_t bj = j * seg;
    bool done = !(walk && bj < n);                               // (lane 0: n > 0)
    const uint32_t mis = done ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(in) & 127u);
    SegStream st;
    st.base = done ? nullptr : in - mis;
    st.end = done ? 0u : mis + n;
    st.ring = wave_ring + lane * kSsStride;
    const uint32_t iend = st.end;
    // positions below are st-positions (offsets from st.base): stream position + mis
    const uint32_t my_b = mis + bj;
    uint32_t ip = mis + (j == 0u || bj <= kSegLead ? 0u : bj - kSegLead);
    st.lo = st.hi = ip & ~(kSsUnit - 1u);
    const SegPlan plan = seg_plan(st);
    bool lead = !done && j != 0u;                                // still looking for the chain
    uint32_t tb = j + 1u;                                        // next boundary this lane will cross ...
    uint32_t next_b = tb < k ? mis + tb * seg : 0xFFFFFFFFu;     // ... and its position
    s_cand[wave * 64u + lane] = done ? kCandNone : (j == 0u ? mis : kCandPending);
    uint32_t link = kLinkNone;                                   // the segment whose candidate this lane's walk ended on
    uint32_t cnt = 0, r = 0, near = 0, lim = 0;                  // records written, output bytes so far (from 0), near matches, end-of-block margin
    int32_t need = 0;                                            // how far in front of this lane's first output byte its matches reach
    bool bad = false, saw_last = false, fin = false;             // fin: the last record is out, the region's sentinel follows
    const uint32_t region_slots = seg_region_slots(k);
    uint2* region = recs + (size_t)c * kRecPitch + (size_t)j * region_slots;
    // Records leave in groups of four slots = one aligned 32-byte store (a wave's lanes write to unrelated addresses; single
    // 8-byte stores would each dirty a sector of their own).  The group phase is the WAVE's: iteration `it` fills slot it & 3 and
    // every fourth iteration all lanes store, so a lane that turns to its real walk in iteration it0 begins its region at slot
    // it0 & 3 (`pad`, handed to the decoder in SegMeta) and from then on puts exactly one slot per iteration: a record, an empty
    // record while it waits for a neighbour's candidate, or — last — the region's sentinel.
    uint2 p0 = make_uint2(0, 0), p1 = p0, p2 = p0, p3 = p0;
    uint32_t it = 0, it_base = 0, pad = 0;                       // it_base: iteration of the lane's slot 0 (a multiple of 4); pad: slot of its first record
    bool grp = false;                                            // the lane has put a slot into the current group

#ifdef CJ_SEG_TIMING
    unsigned long long t_all = __builtin_readcyclecounter(), t_ref = 0, t_el = 0; uint32_t n_rounds = 0;
#endif
    for (;;) {
        // ---- the ring: a refill round when some lane is about to run dry; it tops up EVERY lane that has room ----
        if (!done && ip >= st.hi) st.lo = st.hi = ip & ~(kSsUnit - 1u);            // jumped past the window (long literal run): re-anchor
        for (;;) {
            const bool want = !done && st.hi < iend && (st.hi - st.lo < kSsRing || ip >= st.lo + kSsUnit);
            const bool urgent = want && ip + kSegAhead > st.hi;
            if (ballot64(urgent) == 0ull) break;
#ifdef CJ_SEG_TIMING
            const unsigned long long t0 = __builtin_readcyclecounter();
#endif
            seg_refill(st, want, wave_ring, plan);
#ifdef CJ_SEG_TIMING
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            t_ref += __builtin_readcyclecounter() - t0; n_rounds++;
#endif
        }
        if (ballot64(!done) == 0ull) break;
        bool go = !done && !fin;
        bool emit = false;
        uint2 slot = make_uint2(0u, 0u);
        // ---- rare events, one uniform branch: the sentinel behind a lane's last record; a lane that has crossed a boundary stands
        //      on a token: is it the candidate of the segment it is in now? ----
        if (ballot64(!done && (fin || (!lead && ip >= next_b))) != 0ull) {
            if (!done && fin) { emit = true; slot = make_uint2(0u, r & 0xffffu); done = true; }
            else if (go && !lead && ip >= next_b) {
                while (tb + 1u < k && ip >= next_b + seg) { tb += 1u; next_b += seg; }      // (a long literal run may cross several)
                const uint32_t cm = s_cand[g0 + tb];
                if (cm == kCandPending) { go = false; emit = true; slot = make_uint2(0u, r & 0xffffu); cnt += 1u; }   // that lane is still in its lead-in (tiny segments): an empty record, wait a step
                else if (cm == ip) { link = tb; done = true; go = false; emit = true; slot = make_uint2(0u, r & 0xffffu); }      // the region's sentinel
                else { tb += 1u; next_b = tb < k ? next_b + seg : 0xFFFFFFFFu; }
            }
        }
#ifdef CJ_SEG_TIMING
        const unsigned long long t1 = __builtin_readcyclecounter();
#endif
        const Elem e = seg_elem<kCodec>(st, ip, iend, go);
#ifdef CJ_SEG_TIMING
        asm volatile("" :: "v"(e.next), "v"(e.lit), "v"(e.mlen), "v"(e.offset));
        t_el += __builtin_readcyclecounter() - t1;
#endif
        if (ballot64(go && lead) != 0ull) {
            if (go && lead) {
                // lead-in: only the position matters.  Malformed here = this was no token: try the next byte.
                uint32_t nx = e.ok && !e.last ? e.next : ip + 1u;
                nx = nx <= ip ? ip + 1u : nx;
                ip = nx;
                if (ip >= my_b) {
                    lead = false;
                    const uint32_t cand = ip < iend ? ip : kCandNone;
                    s_cand[wave * 64u + lane] = cand;
                    if (cand == kCandNone) done = true;
                    it_base = (it + 1u) & ~3u; pad = (it + 1u) & 3u;                // the first slot is put in the next iteration
                }
                go = false;
            }
        }
        if (go) {
            const uint32_t op2 = r + e.lit;
            const bool has_match = kCodec == CJ_CODEC_LZ4_BLOCK ? !e.last : e.mlen != 0u;
            const uint32_t off16 = e.offset & 0xffffu;
            // (Snappy copy-4: no offset above 65 535 is valid in a 64 KiB chunk)
            const bool bad_now = !e.ok || (has_match && (e.offset == 0u || e.offset > 0xffffu)) || it - it_base + 6u > region_slots
                                 || (!e.last && e.next >= iend);                  // (the grammars end a stream with `last`)
            const int32_t reach = has_match ? (int32_t)off16 - (int32_t)op2 : need;
            need = reach > need ? reach : need;
            // LZ4: cap - op >= lit + 12 and cap - op2 >= mlen + 5 for every sequence with a match; both bounds grow along the chain
            if (kCodec == CJ_CODEC_LZ4_BLOCK && has_match) lim = op2 + (e.mlen + 5u > 12u ? e.mlen + 5u : 12u);
            near += has_match && off16 < 4096u ? 1u : 0u;
            emit = true;
            slot = make_uint2((e.lit_at - mis) | (e.lit << 16), (r & 0xffffu) | ((has_match ? off16 : 0u) << 16));
            cnt += 1u;
            r = op2 + (has_match ? e.mlen : 0u);
            ip = e.next;
            saw_last = e.last;
            fin = e.last;
            if (bad_now) { bad = true; done = true; }
        }
        // ---- the slot of this iteration ----
        const uint32_t q = it & 3u;                                            // (uniform)
        if (q == 0u) p0 = emit ? slot : p0;
        else if (q == 1u) p1 = emit ? slot : p1;
        else if (q == 2u) p2 = emit ? slot : p2;
        else p3 = emit ? slot : p3;
        grp = grp || emit;
        if (q == 3u) {
            if (grp) {
                uint4* g = reinterpret_cast<uint4*>(region + (it - 3u - it_base));
                g[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
                g[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
            }
            grp = false;
        }
        it += 1u;
    }
#ifdef CJ_SEG_TIMING
    t_all = __builtin_readcyclecounter() - t_all;
    if (lane == 0 && (blockIdx.x == 5 || blockIdx.x == gridDim.x / 2) && wave == 0)
        printf("seg timing block %u: cycles %llu refill %llu elem %llu rounds %u steps %u\n", blockIdx.x, t_all, t_ref, t_el, n_rounds, it);
#endif
    if (grp) {                                                                 // the group that was not completed (slots behind the sentinel: never read)
        uint4* g = reinterpret_cast<uint4*>(region + ((it & ~3u) - it_base));
        g[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
        g[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
    }

    // ---- epilogue: which lanes does the chain run through, what lies in front of each ----
    volatile uint32_t* my_live = s_live + wave * 64u + lane;
    bool live = walk && j == 0u;
    *my_live = live ? 1u : 0u;
    for (uint32_t round = 1; round < k; round++) {               // (lanes of one wavefront: in order, no barrier)
        if (live && link != kLinkNone) s_live[g0 + link] = 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        live = *my_live != 0u;
    }
    const uint32_t v_cnt = live ? cnt : 0u, v_out = live ? r : 0u, v_near = live ? near : 0u;
    uint32_t s_cnt = v_cnt, s_out = v_out, s_near = v_near;      // inclusive scans over the chunk's k lanes
    for (uint32_t d = 1; d < k; d <<= 1) {
        const uint32_t t0 = (uint32_t)__shfl_up((int)s_cnt, d, 64), t1 = (uint32_t)__shfl_up((int)s_out, d, 64), t2 = (uint32_t)__shfl_up((int)s_near, d, 64);
        if (j >= d) { s_cnt += t0; s_out += t1; s_near += t2; }
    }
    const uint32_t first = s_cnt - v_cnt, opb = s_out - v_out;
    const uint32_t last_lane = (lane & ~(k - 1u)) + k - 1u;
    const uint32_t nseq = (uint32_t)__shfl((int)s_cnt, (int)last_lane, 64), total = (uint32_t)__shfl((int)s_out, (int)last_lane, 64),
                   near_all = (uint32_t)__shfl((int)s_near, (int)last_lane, 64);
    // the checks that needed absolute positions; a live lane that neither linked nor saw the end ran into something malformed
    bool lane_ok = true;
    if (live) {
        lane_ok = !bad && (link != kLinkNone || saw_last) && need <= (int32_t)opb;
        if constexpr (kCodec == CJ_CODEC_LZ4_BLOCK) {
            if (cnt > (saw_last ? 1u : 0u)) lane_ok = lane_ok && (uint64_t)opb + lim <= cap;      // its last sequence with a match
            if (saw_last) lane_ok = lane_ok && (uint64_t)opb + r <= cap;
        }
    }
    const uint64_t okm = ballot64(lane_ok), lastm = ballot64(live && saw_last);
    const uint64_t gmask = (k == 64u ? ~0ull : ((1ull << k) - 1ull)) << (lane & ~(k - 1u));
    bool chunk_ok = walk && (okm & gmask) == gmask && __popcll(lastm & gmask) == 1;
    if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) chunk_ok = chunk_ok && total == cap;
    else chunk_ok = chunk_ok && total <= cap;
    if (exists) {
        if (walk) {
            if (chunk_ok && nseq >= kLdsMinSeq && nseq <= kSyncStride * kSyncEvery) {
                res = (int64_t)total;
                pm.nseq = nseq; pm.in_skip = skip;
                SegMeta* sm = segmeta + c;
                sm->first[j] = j == 0u ? near_all : first;
                sm->opb[j] = opb | (pad << 20);
            } else { res = 0; pm.nseq = 0u; pm.in_skip = kRouteWave; }          // the wavefront kernel decodes it or names its error
        }
        if (j == 0u) { a.result[c] = res; meta[c] = pm; }
    }
}

size_t seg_recs_bytes(size_t n_chunks) { return n_chunks * (size_t)kRecPitch * sizeof(uint2); }
size_t seg_meta_bytes(size_t n_chunks) { return n_chunks * sizeof(SegMeta); }

void launch_seg_parse(const BatchArgs& a, int codec, uint32_t klog, void* recs, void* segmeta, void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    const uint32_t per_block = (64u * kSegWaves) >> klog;         // chunks per block
    const dim3 grid((a.n_chunks + per_block - 1u) / per_block), block(64u * kSegWaves);
    if (codec == CJ_CODEC_SNAPPY_RAW)
        hipLaunchKernelGGL((seg_parse_kernel<CJ_CODEC_SNAPPY_RAW>), grid, block, 0, s, a, klog, (uint2*)recs, (SegMeta*)segmeta, (ParseMeta*)meta);
    else
        hipLaunchKernelGGL((seg_parse_kernel<CJ_CODEC_LZ4_BLOCK>), grid, block, 0, s, a, klog, (uint2*)recs, (SegMeta*)segmeta, (ParseMeta*)meta);
}

}  // namespace cj
