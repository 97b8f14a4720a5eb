// lz4_decode_lanes.hip — LZ4 *block* decoding with ONE LANE per independent chunk, and the parse-only
// variant that feeds the workgroup-per-chunk LDS decoder.
//
// Same contract and accept/reject rules as lz4_decode.hip (reference call sites
// /root/reference/src/lz4.rs:88,90,164,168 -> LZ4_decompress_safe), different mapping: the
// wave-per-chunk kernel is bound by the serial token chain of each chunk (one dependent memory round
// trip per copy, ~3.7k cycles per sequence measured).  Here every lane walks its own chunk, so a
// wavefront retires 64 sequences per step; the price is uncoalesced 16 B accesses.
//   lz4_decode_lanes_kernel : full decode, 16 B "wild" copies with exact tails
//   lz4_parse_kernel        : walk + validate only (reads nothing but the compressed stream), emits
//                             (ip, op) sync points every 8 sequences + the decoded size; chunks the LDS
//                             decoder cannot take (capacity > 64 KiB, > 16 384 sequences) are routed to the wavefront kernel.
#include "lz4_lane_walk.hpp"
#include "lane_stream.hpp"
#include <cstdlib>

namespace cj {

__global__ __launch_bounds__(64) void lz4_decode_lanes_kernel(BatchArgs a) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= a.n_chunks) return;
    const uint8_t* in = a.in_base + a.in_off[c];
    uint64_t n64 = a.in_len[c];
    uint8_t* out = a.out_base + a.out_off[c];
    uint64_t cap64 = a.out_cap[c];
    const int64_t status = lz4_block_prologue(a.flags, in, n64, cap64);
    if (status != 0) { a.result[c] = status; return; }
    const uint32_t cap = (uint32_t)cap64, iend = (uint32_t)n64;
    if (cap == 0) { a.result[c] = (iend == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT; return; }
    if (iend == 0) { a.result[c] = CJ_E_CORRUPT; return; }
    a.result[c] = lz4_lane_walk<true>(in, iend, out, cap, nullptr, 0, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// lz4_parse_kernel — lane-per-chunk walk + validation, reading the compressed stream through a per-lane
// 256-byte line cache in LDS.  A wave-load with 64 unrelated addresses costs ~2.3k cycles here (64 separate
// line requests, measured), and a lane touches each 128 B line ~16 times; so instead the wavefront refills
// the caches cooperatively — 8 lanes fetch one lane's next 128 B line with aligned 16 B loads, 8 lines per
// load instruction — and the per-sequence reads become LDS reads (lane_stream.hpp).
// ---------------------------------------------------------------------------------------------------
#ifndef CJ_PARSE_AHEAD
#define CJ_PARSE_AHEAD 32u
#endif
constexpr uint32_t kParseAhead = CJ_PARSE_AHEAD;   // cached bytes a lane must have ahead of its position before a step (measured, 100 k chunks:
                                                   // 16: 2.66 ms, 24: 2.59, 32: 2.585, 40: 2.62, 48: 2.71, 64: 2.90, 96: 3.67, 128: 5.0 — refill rounds are what costs)
__global__ __launch_bounds__(64 * kParseWaves) void lz4_parse_kernel(BatchArgs a, uint2* sync, ParseMeta* meta) {
    __shared__ __attribute__((aligned(16))) uint8_t rings[kParseWaves * 64 * kRingStride];
    const uint32_t c = blockIdx.x * (64u * kParseWaves) + threadIdx.x;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t wave_ring = (uint32_t)(uintptr_t)rings + wave * 64u * kRingStride;
    const bool exists = c < a.n_chunks;

    // ---- per-lane setup (same prologue / special cases as the other mappings) ----
    const uint8_t* in0 = nullptr; const uint8_t* in = nullptr;
    uint64_t n64 = 0, cap64 = 0;
    ParseMeta pm = {0u, 0u};
    int64_t r = 0;
    bool done = true;
    const bool linked = (a.flags & kFlagLinkedFrame) != 0u;
    const uint32_t win = lds_window(a.flags);                // the decoder's window for this batch
    uint32_t hist = 0;
    if (exists) {
        in0 = in = a.in_base + a.in_off[c];
        n64 = a.in_len[c];
        cap64 = a.out_cap[c];
        if (linked && a.hist != nullptr) hist = a.hist[c];
        if (linked && (n64 >> 63)) {                       // stored block: nothing to parse
            n64 &= 0x7FFFFFFFFFFFFFFFull;
            r = n64 <= cap64 ? (int64_t)n64 : (int64_t)CJ_E_CORRUPT;
            pm.in_skip = kRouteStored;
        } else
        r = lz4_block_prologue(a.flags, in, n64, cap64);
        if (r == 0 && pm.in_skip == 0u) {
            const uint32_t cap0 = (uint32_t)cap64, iend0 = (uint32_t)n64;
            if (cap0 == 0) r = (iend0 == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT;
            else if (iend0 == 0) r = CJ_E_CORRUPT;
            else if (cap0 > win || iend0 > win - 32u) { r = 0; pm.in_skip = kRouteWave; }   // too big for LDS: wave kernel decodes + validates
            else done = false;
        }
    }
    const uint32_t cap = (uint32_t)cap64;
    LaneStream st;
    const uint32_t mis = done ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(in) & 127u);
    st.base = done ? (CJ_REFILL_TOUCH ? a.in_base : nullptr) : in - mis;
    st.lo = 0; st.hi = 0;
#if CJ_REFILL_TOUCH
    st.touch = 0;
#endif
    st.end = done ? 0u : mis + (uint32_t)n64;
    st.ring = wave_ring + lane * kRingStride;
    const uint32_t iend = st.end;
    const RefillPlan plan = refill_plan(st);
    uint2* csync = sync + (size_t)c * kSyncPitch;

    uint32_t ip = mis, op = 0, nseq = 0;
    SyncBatch sb = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // ---- wave-convergent walk: refill rounds, then one sequence per active lane ----
    while (ballot64(!done) != 0ull) {
        if (!done && ip >= st.hi) st.lo = st.hi = ip & ~127u;      // jumped past the window (long literal run): re-anchor
        // A refill round costs one HBM round trip for the whole wave, so it is triggered only when some lane is about
        // to run dry (< 48 cached bytes ahead) and then tops up EVERY lane that has room; lanes drift apart, but a
        // round every few sequences serves them all.
        for (;;) {
            const bool want = !done && st.hi < iend && (st.hi - st.lo < kRingBytes || ip >= st.lo + 128u);
            const bool urgent = want && ip + kParseAhead > st.hi;
            if (ballot64(urgent) == 0ull) break;
            refill_round(st, want, wave_ring, plan);
        }
        // A trip takes at most eight consecutive sequences of a lane, so at most ONE of them starts a sync group: it is only noted where
        // it occurs (three selects) and handed to the sync batch once, at the end of the trip — put() is a branch with a dozen moves and
        // the group's stores behind it, and some lane of the wave is at a multiple of 8 in nearly every step.
        bool sp_hit = false;
        uint32_t sp_ip = 0, sp_op = 0, sp_slot = 0;
        if (!done) {
            // one sequence; mirrors lz4_lane_walk<false> with reads through the line cache
            sp_hit = (nseq % kSyncEvery) == 0u;
            sp_ip = ip - mis; sp_op = op; sp_slot = nseq / kSyncEvery;
            nseq += 1;
        }
        // ---- the common case as straight-line code: token and offset both in the line cache, length extensions of at most one
        //      byte, not near the end of the block, match valid.  Nothing is committed unless all of that holds; every other
        //      sequence (and every error) takes the general walk below from the same state.  The general walk alone is ~235
        //      instructions per step, half of them scalar mask bookkeeping for its ~30 conditional blocks, and the kernel is
        //      bound by exactly that ----
        // One dependent LDS round trip per sequence: where the NEXT token sits follows from the current token alone (the extension bytes
        // that would move it further are exactly the ones this path refuses), so the offset field and the next token are read together.
        struct FastSeq { bool ok; uint32_t ip3, op3, t4n; };
        // the trip's bounds: [ip, ip + 4) cached and inside the block; the offset field cached and at least 8 bytes before the block's end
        // (rem_in >= lit + 8 implies every bound the general walk checks while it reads one-byte extensions); the output margins
        const uint32_t win_end = st.hi < iend ? st.hi : iend;
        const int32_t ip2_max = (int32_t)win_end - 4 < (int32_t)iend - 8 ? (int32_t)win_end - 4 : (int32_t)iend - 8;      // (signed: a short block makes it negative)
        const bool ip_low_ok = ip >= st.lo;                       // (the position only moves forward during the trip)
        const auto fast_seq = [&](uint32_t t4) __attribute__((always_inline)) -> FastSeq {
            const uint32_t token = t4 & 0xffu, e1 = (t4 >> 8) & 0xffu;
            const bool x1 = (token >> 4) == 15u;
            const uint32_t lit = (token >> 4) + (x1 ? e1 : 0u);
            const uint32_t ip2 = ip + (x1 ? 2u : 1u) + lit;
            const uint32_t mc = token & 15u;
            const bool x2 = mc == 15u;
            const uint32_t ip3 = ip2 + (x2 ? 3u : 2u);
            const LaneStream::Pair rq = st.ring32x2_request(ip2, ip3);
            // (bitwise: as a chain of && this compiles to nested branches, four or five per sequence)
            const uint32_t op2 = op + lit;
            const bool ok_early = ip_low_ok & ((int32_t)(ip + 4u) <= (int32_t)win_end) & ((int32_t)ip2 <= ip2_max)
                                  & !(x1 & (e1 == 255u)) & ((int32_t)(op2 + 12u) <= (int32_t)cap);
            uint32_t o4, t4n;
            st.ring32x2_arrive(rq, ip2, ip3, o4, t4n);
            const uint32_t offset = o4 & 0xffffu, e2 = (o4 >> 16) & 0xffu;
            const uint32_t op3 = op2 + mc + (x2 ? e2 : 0u) + 4u;
            const bool ok = ok_early & !(x2 & (e2 == 255u)) & ((int32_t)(op3 + 5u) <= (int32_t)cap)
                            & (offset != 0u) & (offset <= op2 + hist);
            return FastSeq{ok, ip3, op3, t4n};
        };
        bool fast_ok = false;
        uint32_t t4_next = 0;
        if (!done) {
            const FastSeq f = fast_seq(st.ring32(ip));
            fast_ok = f.ok; t4_next = f.t4n;
            if (f.ok) { ip = f.ip3; op = f.op3; }
        }
#ifndef CJ_PARSE_SINGLE
        // ---- MORE sequences in the same trip where the first went the straight way and the next one does too (token and offset
        //      still inside the cached window): the trip's fixed cost — the refill test, the loop's ballots, the branch into the
        //      general walk — is spent once for all of them.  Nothing is committed (not even the sync point) unless it holds.
        //      Up to 1 / 2 / 3 / 5 / 7 / 10 / 15 more per trip: 736 / 740 / 747 / 750 / 757 / 742 / 709 GB/s (none: 706); a longer
        //      lookahead (48, 64 bytes) only adds refill rounds. ----
#ifndef CJ_PARSE_EXTRA
#define CJ_PARSE_EXTRA 7
#endif
        static_assert(CJ_PARSE_EXTRA + 1 <= (int)kSyncEvery, "a trip must not cross two sync groups: the deferred put holds one");
        bool more = fast_ok;                          // (fast_ok itself still says whether the FIRST sequence needs the general walk below)
#pragma unroll
        for (int rep = 0; rep < CJ_PARSE_EXTRA; rep++) {
            if (ballot64(more) == 0ull) break;
            if (more) {
                const FastSeq f = fast_seq(t4_next);
                if (f.ok) {
                    const bool hit = (nseq % kSyncEvery) == 0u;
                    sp_ip = hit ? ip - mis : sp_ip; sp_op = hit ? op : sp_op; sp_slot = hit ? nseq / kSyncEvery : sp_slot;
                    sp_hit = sp_hit | hit;
                    nseq += 1;
                    ip = f.ip3; op = f.op3;
                }
                more = f.ok; t4_next = f.t4n;
            }
        }
#endif
        if (!done && !fast_ok) {
            bool bad = false, last = false;
            const uint32_t t4 = st.ld32(ip);
            const uint32_t token = t4 & 0xffu;
            ip += 1;
            uint32_t lit = token >> 4;                  // iend <= kLdsInMax here: at most 255 * 65504 + 15, no 64-bit needed
            if (lit == 15u) {
                if (ip + 15u >= iend) bad = true;
                else {
                    uint32_t b = (t4 >> 8) & 0xffu;
                    ip += 1; lit += b;
                    if (ip + 15u > iend) bad = true;
                    while (!bad && b == 255u) {
                        b = st.ld8(ip);
                        ip += 1; lit += b;
                        if (ip + 15u > iend) bad = true;
                    }
                }
            }
            if (!bad) {
                const uint32_t rem_out = cap - op, rem_in = iend - ip;
                if (rem_out < lit + 12u || rem_in < lit + 8u) {
                    if (rem_in != lit || rem_out < lit) bad = true;
                    else { op += lit; last = true; }
                } else {
                    ip += lit; op += lit;
                    const uint32_t o4 = st.ld32(ip);
                    const uint32_t offset = o4 & 0xffffu;
                    ip += 2;
                    uint32_t mlen = token & 15u;
                    if (mlen == 15u) {
                        uint32_t b = (o4 >> 16) & 0xffu;
                        ip += 1; mlen += b;
                        if (ip + 4u > iend) bad = true;
                        while (!bad && b == 255u) {
                            b = st.ld8(ip);
                            ip += 1; mlen += b;
                            if (ip + 4u > iend) bad = true;
                        }
                    }
                    mlen += 4u;
                    if (!bad) {
                        if (offset == 0u || offset > op + hist) bad = true;
                        else if (cap - op < mlen + 5u) bad = true;
                        else op += mlen;
                    }
                }
            }
            if (bad) { r = CJ_E_CORRUPT; done = true; }
            else if (last) {
                r = (int64_t)op;
                done = true;
                if (r > 0) {
                    if (nseq > lds_window_max_seq(win) || (nseq < lds_window_min_seq(win) && !linked)) pm.in_skip = kRouteWave;
                    else { pm.nseq = nseq; pm.in_skip = (uint32_t)(in - in0); }
                }
            }
        }
        if (sp_hit) sb.put(csync, sp_slot, make_uint2(sp_ip, sp_op));
    }
    if (exists) {
        sb.flush(csync, (nseq + kSyncEvery - 1u) / kSyncEvery);
        a.result[c] = r;
        meta[c] = pm;
    }
}

void launch_lz4_decode_lanes(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + 63u) / 64u), block(64);
    hipLaunchKernelGGL(lz4_decode_lanes_kernel, grid, block, 0, s, a);
}

void launch_lz4_parse(const BatchArgs& a, void* sync, void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    const uint32_t per_block = 64u * kParseWaves;
    dim3 grid((a.n_chunks + per_block - 1u) / per_block), block(per_block);
    hipLaunchKernelGGL(lz4_parse_kernel, grid, block, 0, s, a, (uint2*)sync, (ParseMeta*)meta);
}

}  // namespace cj
