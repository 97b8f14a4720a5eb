// cj_enc2.hpp — the round-based LZ77 matcher of the LZ4-block and Snappy-raw encoders (gfx950; one workgroup of two wavefronts per chunk,
// one wavefront per sub-piece of a split large buffer).
//
// What a round does (tests/hostsim/enc2_model.c states the same thing as scalar C; the kernels emit exactly its bytes):
//   probe    kR consecutive positions — lane l owns the FOUR CONSECUTIVE positions 4 l .. 4 l + 3 of every group of 256, so two
//            dwords per lane yield the four position dwords with three v_alignbyte — against the 8192 x u16 hash table, in BLOCKS of
//            128 positions (half a wavefront): a block reads its slots and then stores its own positions into them before the next block
//            reads (the DS queue of a wavefront executes in order; the second wavefront starts behind the first one's stores), so the
//            table a position sees is at most 128 positions stale.  Then candidate dword, compare.  Straight-line code.
//   heads    a verified position whose left neighbour is verified with the SAME offset lies inside its neighbour's match: only
//            the first position of such a run (its head) is a candidate.  On match-heavy data half of all positions verify and
//            one in fifteen is a head; the heads are compacted into consecutive lanes (ranks from v_mbcnt, a 256-byte LDS list).
//   extend   ONE pass over the compacted heads measures every candidate forwards (32 bytes per lane and round trip; the last few
//            long ones of a window by sixteen lanes each, 256 bytes per round trip) and backwards (16 bytes) — the previous
//            matcher ran this code once per 64 positions with a lane or two active.
//   select   greedy, in position order: the first head whose interval still has four bytes past the end of the previous match wins
//            (a head the previous match ran over still offers its tail).  The rule has one solution, so it is ITERATED in parallel —
//            every head re-decided against the prefix maximum of the selected ends (DPP scan) until the mask reproduces itself —
//            instead of walked; a window that has not settled after kSelPasses takes the serial walk (`cur` through SGPRs).
//   queue    selected sequences are appended to a 64-entry queue in LDS and emitted lane-parallel when it is full — one emission
//            pass per ~64 sequences instead of one per round.
//   (until round 5 a round probed the table as it was when the round began and inserted at its end, only the positions outside the emitted
//    matches: on repetitive data that cost 4-9 % of the ratio — html 4.15 against liblz4's 4.58 — and a coverage bitmap, a parity scan and a
//    fourth meeting of the two wavefronts per round.  Every position entering the table at once: html 4.48, the corpus 1.850 -> 1.898.)
// Long backward extensions are finished by the whole wavefront after the selection, literal runs of 256 bytes and more are copied
// by the whole wavefront inside the flush; only a literal run of 64 KiB and more (an input above 64 KiB) sends its window down a serial path.
#pragma once
#include "cj_match.hpp"

#if defined(__HIPCC__)
// v_writelane_b32: clang has no builtin for it; a declaration with the intrinsic's name as its assembler name is lowered to the
// intrinsic (the compiler then also places the hazard no-ops a hand-written instruction would need)
extern "C" __device__ int cj_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
#endif

namespace cj {
#if defined(__HIPCC__)
namespace enc2 {

constexpr uint32_t kQueueCap = 64u;
constexpr uint32_t kFwdTrips = 8u;                  // forward measurement in the lanes: 32 bytes per round trip, 4 + 256 bytes at most
constexpr uint32_t kLeaderMin = 4u;
constexpr uint32_t kLeaders = 6u;                     // long heads finished one at a time by the whole wavefront, each killing the long heads it outlives
constexpr uint32_t kGroupHeads = 4u;                // ... until at most this many heads of the window still match: those are finished four at a time, sixteen lanes each
#ifndef CJ_SEL_PASSES
#define CJ_SEL_PASSES 16
#endif
constexpr uint32_t kSelPasses = CJ_SEL_PASSES;                // parallel selection passes before a window falls back to the serial walk
constexpr uint32_t kLaneLit = 256u;                 // a lane copies its sequence's literals itself below this; longer runs are copied by the whole wavefront after the lanes' pass
constexpr uint32_t kMaxLit = 65536u;                // a queue entry holds the literal count in 16 bits; a longer run (inputs above 64 KiB) takes the serial path

// the wave mask of a condition straight from the compare (HIP's __ballot goes through an integer: v_cndmask + v_cmp per call)
__device__ __forceinline__ uint64_t bal(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Input and output are addressed as (uniform base in SGPRs) + (32-bit offset in a VGPR): global_load/store ... v_off, s[base:base+1].
// The compiler does that only for a pointer it KNOWS to be uniform and global: both halves through v_readfirstlane, and the
// address space spelled out (a pointer rebuilt from integers is a flat pointer otherwise: flat_load + 64-bit VALU address arithmetic).
#define CJ_GAS __attribute__((address_space(1)))
typedef const CJ_GAS uint8_t* gcptr;
typedef CJ_GAS uint8_t* gptr;
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint32_t v2raw __attribute__((ext_vector_type(2)));
typedef uint32_t v4raw __attribute__((ext_vector_type(4)));
typedef v2raw v2_unaligned __attribute__((aligned(1)));
typedef v4raw v4_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ gcptr uniform_gptr(const uint8_t* p) {
    const uint64_t v = (uint64_t)p;
    return (gcptr)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}
__device__ __forceinline__ uint32_t g8(gcptr b, uint32_t off) { return b[off]; }
__device__ __forceinline__ uint32_t g32(gcptr b, uint32_t off) { return *(const CJ_GAS u32_unaligned*)(b + off); }
__device__ __forceinline__ uint2 g64(gcptr b, uint32_t off) { const v2raw v = *(const CJ_GAS v2_unaligned*)(b + off); return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint4 g128(gcptr b, uint32_t off) { const v4raw v = *(const CJ_GAS v4_unaligned*)(b + off); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void s8(gptr b, uint32_t off, uint32_t v) { b[off] = (uint8_t)v; }
__device__ __forceinline__ void s32(gptr b, uint32_t off, uint32_t v) { *(CJ_GAS u32_unaligned*)(b + off) = v; }
__device__ __forceinline__ void s64(gptr b, uint32_t off, uint2 v) { v2raw r; r.x = v.x; r.y = v.y; *(CJ_GAS v2_unaligned*)(b + off) = r; }
__device__ __forceinline__ void s128(gptr b, uint32_t off, uint4 v) { v4raw r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; *(CJ_GAS v4_unaligned*)(b + off) = r; }

// v_ffbl_b32 / v_ffbh_u32 as the hardware defines them: ~0 for a zero operand (the C builtins are undefined there, and a
// select around them costs a compare and a v_cndmask per dword)
__device__ __forceinline__ uint32_t ffbl_raw(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t ffbh_raw(uint32_t x) { uint32_t r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// index of the first differing byte of two 16-byte blocks (16: none), branch-free: OR-ing the dword's bit offset into ~0 keeps
// ~0, so an unsigned minimum picks the first dword that differs
__device__ __forceinline__ uint32_t first_diff(const uint4& x, const uint4& y) {
    const uint32_t b0 = ffbl_raw(x.x ^ y.x), b1 = ffbl_raw(x.y ^ y.y) | 32u, b2 = ffbl_raw(x.z ^ y.z) | 64u, b3 = ffbl_raw(x.w ^ y.w) | 96u;
    return umin(umin(umin(b0, b1), umin(b2, b3)) >> 3, 16u);
}
// equal bytes counted from the END of two 16-byte blocks (16: all)
__device__ __forceinline__ uint32_t last_same(const uint4& x, const uint4& y) {
    const uint32_t b3 = ffbh_raw(x.w ^ y.w), b2 = ffbh_raw(x.z ^ y.z) | 32u, b1 = ffbh_raw(x.y ^ y.y) | 64u, b0 = ffbh_raw(x.x ^ y.x) | 96u;
    return umin(umin(umin(b0, b1), umin(b2, b3)) >> 3, 16u);
}

// exact n-byte copy by ONE lane, n < 256 (a queued literal run): 16-byte blocks, then the remainder as two overlapping pieces
// of the largest power of two that fits (the second one ends exactly at n) — never a byte beyond [0, n) on either side
__device__ __forceinline__ void lane_copy(gptr out, uint32_t o, gcptr in, uint32_t i, uint32_t n) {
    if (n >= 16u) {
        for (uint32_t k = 0; k + 16u <= n; k += 16u) s128(out, o + k, g128(in, i + k));
        s128(out, o + n - 16u, g128(in, i + n - 16u));
    } else if (n >= 8u) {
        const uint2 a = g64(in, i), b = g64(in, i + n - 8u);
        s64(out, o, a); s64(out, o + n - 8u, b);
    } else if (n >= 4u) {
        const uint32_t a = g32(in, i), b = g32(in, i + n - 4u);
        s32(out, o, a); s32(out, o + n - 4u, b);
    } else if (n > 0u) {
        const uint32_t a = g8(in, i), b = g8(in, i + (n >> 1)), c = g8(in, i + n - 1u);
        s8(out, o, a); s8(out, o + (n >> 1), b); s8(out, o + n - 1u, c);
    }
}

// The state of one chunk's walk.  Fmt supplies the stream format:
//   Fmt::last_start(n), Fmt::limit(n)          last position a match may start at / must end by
//   Fmt::seq_size(lit, code, off)               encoded bytes of one queued sequence (code = mlen - 4, any; lit < 65536)
//   Fmt::emit_lane(in, out, o, lit0, lit, code, off) -> where its literals go   one lane writes one queued sequence at output offset o
//                                               (the literal bytes themselves only below kLaneLit)
//   Fmt::emit_wave(in, out, op, lit0, lit, off, mlen) -> new op   the whole wavefront writes one sequence of any size
// kW = wavefronts per chunk (1 or 2).  With two, a round is 512 positions and wavefront w owns its group w of 256: wavefront 0 probes and
// inserts its two blocks, then wavefront 1 its two (one meeting in between); both measure their heads side by side, then select in position
// order — wavefront 0's heads, then wavefront 1's (`cur`, the queue count and the output position travel through LDS).  A wavefront issues
// one instruction per ~5 cycles whatever it holds (tools/issue_rate_probe.hip) and a CU's LDS holds nine tables: two wavefronts per
// table are how the CU gets more instruction streams.  tests/hostsim/enc2_model.c with R = 512 states exactly what this computes.
#ifdef CJ_ENC_PROFILE
// phase cycle counters of the matcher (debug builds: tools/exp_r05_encprofile.sh): [0] rounds, [1] probe, [2] measure, [3] select + push,
// [4] table reads + inserts (inside 1), [5] flushes (inside 3), [6] waiting for the other wavefront, [7] windows
__device__ unsigned long long g_enc_prof[16];
__device__ __forceinline__ uint64_t prof_now() { uint64_t t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
// (accumulated in registers, written once per chunk: an atomic per phase would sit in the next phase's vmcnt wait)
#define CJ_PROF(i, expr) do { const uint64_t t_ = prof_now(); expr; prof_acc[i] += prof_now() - t_; } while (0)
#define CJ_PROF_COUNT(i, v) do { prof_acc[i] += (v); } while (0)
#else
#define CJ_PROF(i, expr) do { expr; } while (0)
#define CJ_PROF_COUNT(i, v) do { } while (0)
#endif

template <class Fmt, int kW>
struct Walk {
    static constexpr uint32_t kRound = 256u * kW;
    // LDS scratch in dwords: heads (64 per wavefront) · queue (3 per entry) · shared scalars
    static constexpr uint32_t kHeadsAt = 0u, kQueueAt = 64u * kW, kSharedAt = kQueueAt + 3u * kQueueCap, kWords = kSharedAt + 8u;

    gcptr in;               // position 0 (start of the piece), uniform
    gptr out;               // uniform
    uint32_t n;             // end of this chunk's range
    uint32_t last_start, limit;
    uint32_t* scr;          // kWords dwords of LDS
    HashTab ht;
    uint32_t op;            // output position after the last EMITTED sequence (queued ones are not counted yet)
    uint32_t q_n;           // queued sequences
    uint32_t wv;            // this wavefront's index in the chunk's workgroup (0 when kW == 1)
#ifdef CJ_ENC_PROFILE
    uint64_t prof_acc[12] = {};      // [8] selection passes, [9] windows that fell back to the serial walk, [10] heads
#endif

    // both wavefronts of a chunk meet: LDS writes before it are visible behind it
    __device__ __forceinline__ void meet() const {
        if constexpr (kW > 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }

    __device__ __forceinline__ void flush() {
        if (q_n == 0u) return;
        const uint32_t lane = lane_id();
        const bool on = lane < q_n;
        const uint32_t lit0 = scr[kQueueAt + 3u * lane], pk = scr[kQueueAt + 3u * lane + 1u], code = scr[kQueueAt + 3u * lane + 2u];
        const uint32_t off = pk & 0xffffu, lit = pk >> 16;
        uint32_t total;
        const uint32_t before = wave_excl_add(on ? Fmt::seq_size(lit, code, off) : 0u, total);
        uint32_t lit_at = 0;
        if (on) lit_at = Fmt::emit_lane(in, out, op + before, lit0, lit, code, off);        // (copies its literals itself below kLaneLit)
        for (uint64_t lm = bal(on && lit >= kLaneLit); lm != 0ull; lm &= lm - 1ull) {      // long literal runs: by the whole wavefront
            const uint32_t i = ctz64(lm);
            wave_copy((uint8_t*)out + rdlane(lit_at, i), (const uint8_t*)in + rdlane(lit0, i), rdlane(lit, i));
        }
        op = uni(op + total);
        q_n = 0u;
    }

    struct Heads {            // one wavefront's verified run heads of a round
        uint32_t dd[4];       // candidate distances of the lane's four positions (0: none)
        bool pre[4];          // has a candidate and is not a follower (same distance as its left neighbour)
        bool hd[4];           // is a verified head
        uint64_t hm[4];
        uint32_t total;
    };
    struct Meas {             // one window of up to 64 heads, measured: lane i = head i
        uint32_t mw, P, d, E, BS;
        bool back_more;           // 16 bytes before the head matched and more literals may be pending: the selection decides whether it matters
    };

    // the table part of the probe: the 256 positions gpos + 4 lane + k read their slots and enter them, block by block (lanes 0-31, then
    // lanes 32-63: a wavefront's DS operations execute in order, so the second block's reads see the first block's stores); within a
    // block one store instruction per k, the highest lane winning a contested slot — the model's order
    __device__ __forceinline__ void probe_table(uint32_t gpos, uint32_t round_last, uint32_t D0, uint32_t D1, uint32_t (&v)[4], Heads& h) {
        const uint32_t lane = lane_id();
        uint32_t hs[4], t[4] = {0u, 0u, 0u, 0u};
        v[0] = D0;
        v[1] = __builtin_amdgcn_alignbyte(D1, D0, 1);
        v[2] = __builtin_amdgcn_alignbyte(D1, D0, 2);
        v[3] = __builtin_amdgcn_alignbyte(D1, D0, 3);
#pragma unroll
        for (int k = 0; k < 4; k++) hs[k] = hash_slot(v[k]);
        const uint32_t p0 = gpos + 4u * lane;
#pragma unroll
        for (int k = 0; k < 4; k++) h.dd[k] = 0u;
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            const bool mine = (lane >> 5) == (uint32_t)blk;
            if (mine) {
#pragma unroll
                for (int k = 0; k < 4; k++) t[k] = ht.get(hs[k]);
            }
            // distance to the slot's position, modulo the 64 KiB lap of the 16-bit table; 0 = no candidate
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t p = p0 + k;
                const uint32_t d = (p - t[k]) & 0xffffu;
                if (mine) h.dd[k] = (d != 0u && d <= p && p <= round_last) ? d : 0u;
            }
            // A FOLLOWER — same candidate distance as its left neighbour — lies inside that neighbour's match if it is one: it is neither
            // verified (probe_verify) nor inserted.  (The DPP move runs with every lane: lane 32's left neighbour belongs to block 0.)
            const uint32_t left0 = dpp_from<kDppWaveShr1>(0u, h.dd[3]);            // lane l - 1's last position; lane 0 of a group starts afresh
            if (mine) {
                h.pre[0] = h.dd[0] != 0u && h.dd[0] != left0;
                h.pre[1] = h.dd[1] != 0u && h.dd[1] != h.dd[0];
                h.pre[2] = h.dd[2] != 0u && h.dd[2] != h.dd[1];
                h.pre[3] = h.dd[3] != 0u && h.dd[3] != h.dd[2];
#pragma unroll
                for (int k = 0; k < 4; k++) if (p0 + k <= round_last && (h.dd[k] == 0u || h.pre[k])) ht.set(hs[k], p0 + k);
            }
            ht.settle();
        }
    }

    // the rest of the probe: candidate dwords, heads
    __device__ __forceinline__ void probe_verify(uint32_t gpos, const uint32_t (&v)[4], Heads& h) {
        const uint32_t lane = lane_id();
        // Only the FIRST position of a run of equal distances (h.pre) is verified: its followers lie inside its match if it is one, and a
        // candidate dword is a scattered access — the vector memory path takes ~1.5 cycles per lane for those (64 lanes: ~94 cycles per
        // instruction and CU, tools/issue_rate_probe.hip).  On match-heavy data half of all positions are followers.
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w[k] = ~v[k];
            if (h.pre[k]) w[k] = g32(in, gpos + 4u * lane + k - h.dd[k]);
        }
        h.total = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            h.hd[k] = h.pre[k] && w[k] == v[k];
            h.hm[k] = bal(h.hd[k]);
            h.total += (uint32_t)__builtin_popcountll(h.hm[k]);
        }
    }

    // heads w0 .. w0 + 63 of this wavefront into consecutive lanes, measured forwards (4 + 64 bytes at most) and backwards (16)
    __device__ __forceinline__ void measure(const Heads& h, uint32_t gpos, uint32_t w0, uint32_t cur, Meas& m) {
        const uint32_t lane = lane_id();
        uint32_t* heads = scr + kHeadsAt + 64u * wv;
        {   // ranks: heads of lower lanes, of this lane's lower positions
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) r += bits_below_lane(h.hm[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (h.hd[k] && r - w0 < 64u) heads[r - w0] = (h.dd[k] << 16) | (4u * lane + k);
                r += h.hd[k] ? 1u : 0u;
            }
        }
        m.mw = umin(h.total - w0, 64u);
        const bool is_head = lane < m.mw;
        const uint32_t hv = is_head ? heads[lane] : (1u << 16);
        const uint32_t P = gpos + (hv & 0xffffu), d = hv >> 16, C = P - d;
        // ONE round trip carries the first two forward blocks and the backward block of every head
        const uint32_t a = P + 4u;
        uint32_t fwd = 0, back = 0;
        bool more = is_head, back_more = false;
        {
            const bool blk0 = is_head && a + 16u <= n, blk1 = is_head && a + 32u <= n;
            const uint32_t room = P - cur;                       // P >= cur for a round's first window; later windows may start behind cur (BS is not used then)
            const uint32_t blim = umin(room, C);
            const bool bk_on = is_head && blim > 0u, bk_blk = bk_on && C >= 16u;
            uint4 x0 = make_uint4(0, 0, 0, 0), y0 = make_uint4(0, 0, 0, 1), x1 = x0, y1 = y0, bx = x0, by = y0;
            if (blk0) { x0 = g128(in, a); y0 = g128(in, a - d); }
            if (blk1) { x1 = g128(in, a + 16u); y1 = g128(in, a + 16u - d); }
            if (bk_blk) { bx = g128(in, P - 16u); by = g128(in, C - 16u); }
            if (blk0) {
                const uint32_t e0 = first_diff(x0, y0), e1 = first_diff(x1, y1);
                fwd = e0 == 16u && blk1 ? 16u + e1 : e0;       // (no second block this close to the end: the loop below goes on from 16)
                more = fwd == (blk1 ? 32u : 16u);
            }
            if (bk_blk) {
                const uint32_t sm = last_same(bx, by);
                back = umin(sm, blim);
                back_more = sm == 16u && blim > 16u;
            } else if (bk_on) {                                  // candidate within the first 16 bytes of the piece
                while (back < blim && g8(in, P - 1u - back) == g8(in, C - 1u - back)) back += 1u;
            }
        }
        // LEADERS (round 6).  Real data is full of long matches whose heads lie INSIDE each other — a run of a thousand zeros is fifty
        // heads of this window, every one of them a match to the end of the run — and a head whose match ends no later than an earlier
        // head's is never selected (if that one is selected `cur` has passed it; if not, `cur` was already within four bytes of its end).
        // So the first long head is finished by the whole wavefront (1 KiB per round trip), and every later long head that starts before
        // its end E is asked ONE byte: does its own candidate still match AT E?  If not it ends at or before E: dead, without ever being
        // measured (the groups below took four of them at a time, 256 bytes per head and trip: mr, kppkn, geo, html spent 12-24 k of a
        // round's 30-37 k cycles there, profiles/r06/experiments e02).  Exact: only heads that can never be selected are dropped.
        uint32_t E_lead = 0u;                                    // finished leaders' ends (their lanes only)
        bool led = false;
        for (uint32_t ld = 0; ld < kLeaders; ld++) {
            const uint64_t mm = bal(is_head && more);
            if ((uint32_t)__builtin_popcountll(mm) < 2u) break;
            const uint32_t i = ctz64(mm);
            const uint32_t j0 = rdlane(a + fwd, i), di = rdlane(d, i);
            // ... only where it pays: at least kLeaderMin other long heads start inside the 36 bytes of this one that are already known to
            // match (a run, a repeated record).  Text has long matches too, but side by side: a leader's serial round trips would only
            // delay the trips below, which take all of them at once (alice29 7.6 k -> 8.8 k cycles of a round with unconditional leaders).
            if ((uint32_t)__builtin_popcountll(bal(is_head && more && lane != i && P < j0)) < kLeaderMin) break;
            uint32_t cnt = 0;
            for (;;) {                                           // the whole wavefront, 16 bytes per lane
                const uint32_t j = j0 + cnt + 16u * lane;
                uint32_t e = 0;
                if (j < limit) {
                    if (j + 16u <= n) e = first_diff(g128(in, j), g128(in, j - di));
                    else while (e < 16u && j + e < n && g8(in, j + e) == g8(in, j + e - di)) e += 1u;
                    e = umin(e, limit - j);
                }
                const uint64_t full = bal(e == 16u);
                if (full == ~0ull) { cnt += 1024u; continue; }
                const uint32_t first = ctz64(~full);
                cnt += 16u * first + rdlane(e, first);
                break;
            }
            const uint32_t Ei = umin(j0 + cnt, limit);           // (uniform; the first round trip may already have run past the limit)
            if (lane == i) { E_lead = Ei; led = true; more = false; }
            // the later long heads that start before Ei: one byte at Ei (none there when Ei is the limit of an input that ends with it)
            const bool ask = is_head && more && P < Ei;
            if (bal(ask) != 0ull) {
                bool alive = false;
                if (ask && Ei < n) alive = g8(in, Ei) == g8(in, Ei - d);
                if (ask && !alive) { more = false; E_lead = 0u; led = true; }          // dead: E = 0 below
            }
        }
        // the rest: matches beyond 4 + 32 bytes two blocks per round trip, the last 16 bytes of the input byte by byte
        // (four blocks per trip — 64 bytes, 88 registers — measured no faster on any corpus file: r06 e05)
        for (uint32_t it = 1; it < kFwdTrips; it++) {
            if ((uint32_t)__builtin_popcountll(bal(more)) <= kGroupHeads) break;      // few enough for the groups below (or none)
            const bool b0 = more && a + fwd + 16u <= n, b1 = more && a + fwd + 32u <= n;
            uint4 x0 = make_uint4(0, 0, 0, 0), y0 = make_uint4(0, 0, 0, 1), x1 = x0, y1 = y0;
            if (b0) { x0 = g128(in, a + fwd); y0 = g128(in, a + fwd - d); }
            if (b1) { x1 = g128(in, a + fwd + 16u); y1 = g128(in, a + fwd + 16u - d); }
            if (more && !b0) {
                while (a + fwd < limit && g8(in, a + fwd) == g8(in, a + fwd - d)) fwd += 1u;
                more = false;
            }
            if (b0) {
                const uint32_t e0 = first_diff(x0, y0), e1 = first_diff(x1, y1);
                const uint32_t step = e0 == 16u && b1 ? 16u + e1 : e0;
                fwd += step;
                more = step == (b1 ? 32u : 16u);
            }
        }
        if (a + fwd >= limit) { fwd = limit - a; more = false; }      // (a <= limit: P <= last_start)
        uint32_t E = is_head ? a + fwd : 0u;                      // E = 0: never selected
        if (led) E = E_lead;                                      // a finished leader's end, or 0 for a head a leader has outlived
        // longer matches (repeated records, runs): FOUR heads at a time, sixteen lanes each — 256 bytes per head and round trip.  (One
        // head at a time by the whole wavefront made every long match of a window a round trip of its own: geo.protodata, xml and
        // mr spent half of a round there, r05 e27.)
        for (uint64_t lm = bal(is_head && more); lm != 0ull; ) {
            uint32_t idx[4]; uint32_t nb = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) { idx[g] = lm != 0ull ? ctz64(lm) : 0u; nb += lm != 0ull ? 1u : 0u; lm &= lm - 1ull; }
            const uint32_t grp = lane >> 4, sub = lane & 15u;
            const uint32_t hi = grp == 0u ? idx[0] : grp == 1u ? idx[1] : grp == 2u ? idx[2] : idx[3];
            const uint32_t Eh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hi << 2), (int)E);
            const uint32_t dh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hi << 2), (int)d);
            bool act = grp < nb;
            uint32_t cnt = 0;
            for (;;) {
                uint32_t e = 0;
                const uint32_t j = Eh + cnt + 16u * sub;
                if (act && j < limit) {
                    if (j + 16u <= n) e = first_diff(g128(in, j), g128(in, j - dh));
                    else while (e < 16u && j + e < n && g8(in, j + e) == g8(in, j + e - dh)) e += 1u;
                    e = umin(e, limit - j);
                }
                const uint64_t full = bal(act && e == 16u);
                const uint32_t bits = (uint32_t)(full >> (lane & 48u)) & 0xffffu;
                const uint32_t first = ffbl_raw(~bits) & 15u;                                  // (all sixteen full: not used)
                const uint32_t ef = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 48u) + first) << 2), (int)e);
                if (act) {
                    if (bits == 0xffffu) cnt += 256u;
                    else { cnt += 16u * first + ef; act = false; }
                }
                if (bal(act) == 0ull) break;
            }
            const uint32_t Ex = Eh + cnt;
#pragma unroll
            for (int g = 0; g < 4; g++)
                if ((uint32_t)g < nb) E = (uint32_t)cj_llvm_writelane((int)rdlane(Ex, 16u * g), (int)idx[g], (int)E);
        }
        m.P = P; m.d = d;
        m.E = E;
        m.BS = P - back;
        m.back_more = back_more;
    }

    // the greedy walk over one measured window, its sequences into the queue
    __device__ __forceinline__ void select(const Meas& m, uint32_t& cur) {
        const uint32_t lane = lane_id();
        const uint32_t P = m.P, d = m.d, E = m.E, BS = m.BS;
        const uint32_t cur0 = cur;
        uint64_t sel = 0ull;
        uint32_t PE = 0u;
        bool slow = false;
        {
            // The greedy rule — head i is selected iff E_i >= cur_i + 4, cur_i = the end of the last selected head before it — has ONE
            // solution, and every set that reproduces itself under the rule IS it (head i's decision follows from the decisions before
            // it).  So the selection is iterated in parallel instead of walked: start from "every head", take the prefix maximum of
            // the selected ends (DPP scan), re-decide all heads at once, until nothing changes — two or three passes of ~15
            // instructions where the serial walk cost ~400 cycles per match (VALU -> SGPR -> SALU -> VALU round trips and three taken
            // branches per match: 3.5 k of a round's 6.7 k cycles, tools/exp_r05_encprofile.sh).  Each pass fixes at least one more
            // head from the front; a window that has not settled after kSelPasses takes the walk.
            bool s_me = E != 0u;
            uint64_t smask = bal(s_me);
            bool settled = false;
            uint32_t cp = 0, top = cur;
            for (uint32_t pass = 0; pass < kSelPasses; pass++) {
                cp = wave_excl_max(s_me ? E : 0u, cur, top);      // the end of the last selected head below this lane (or cur)
                s_me = E >= cp + 4u && cp <= last_start;
                const uint64_t nm = bal(s_me);
                CJ_PROF_COUNT(8, 1);
                if (nm == smask) { settled = true; break; }
                smask = nm;
            }
            if (settled) { sel = smask; PE = cp; cur = top; }
            else {
                CJ_PROF_COUNT(9, 1);
                for (;;) {                                       // the walk: the chain carries `cur` only
                    if (cur > last_start) break;
                    const uint64_t mm = bal(E >= cur + 4u);
                    if (mm == 0ull) break;
                    const uint32_t first = ctz64(mm);
                    PE = (uint32_t)cj_llvm_writelane((int)cur, (int)first, (int)PE);
                    sel |= 1ull << first;
                    cur = rdlane(E, first);
                }
            }
            const bool selected = settled ? s_me : ((sel >> lane) & 1ull) != 0ull;
            // a selected head whose 16 measured bytes backwards all matched, with more than 16 literals pending in front of it: the
            // whole wavefront finishes its backward extension (rare: the start of the match moves, nobody else's decision does)
            uint32_t BSx = BS;
            for (uint64_t lm = bal(selected && m.back_more && P >= PE && P - PE > 16u); lm != 0ull; lm &= lm - 1ull) {
                const uint32_t i = ctz64(lm);
                const uint32_t Pi = rdlane(P, i), Ci = Pi - rdlane(d, i), room = Pi - rdlane(PE, i);
                const uint32_t bk = 16u + wave_extend_back((const uint8_t*)in, Pi - 16u, Ci - 16u, room - 16u);
                BSx = (uint32_t)cj_llvm_writelane((int)(Pi - bk), (int)i, (int)BSx);
            }
            const uint32_t s = umax(BSx, PE);
            const uint32_t lit = s - PE, code = E - s - 4u;
            const bool needs_wave = selected && lit >= kMaxLit;
            slow = bal(needs_wave) != 0ull;
            if (!slow) {
                const uint32_t ns = (uint32_t)__builtin_popcountll(sel);
                if (q_n + ns > kQueueCap) CJ_PROF(5, flush());
                if (selected) {
                    const uint32_t slot = kQueueAt + 3u * (q_n + bits_below_lane(sel));
                    scr[slot] = PE;
                    scr[slot + 1u] = d | (lit << 16);
                    scr[slot + 2u] = code;
                }
                q_n = uni(q_n + ns);
            }
        }
        if (slow) {
            // serial cooperative path: same decisions, extensions finished by the whole wavefront, every selected sequence of this
            // window emitted by the whole wavefront
            cur = cur0;
            for (uint32_t i = 0; i < m.mw; i++) {
                if (cur > last_start) break;
                const uint32_t Pi = rdlane(P, i), di = rdlane(d, i);
                const uint32_t Ei = rdlane(E, i);
                if (Ei < cur + 4u) continue;
                uint32_t s = cur;
                if (Pi >= cur) {
                    const uint32_t room = Pi - cur;
                    uint32_t bk = umin(Pi - rdlane(BS, i), room);
                    if (rdlane(m.back_more ? 1u : 0u, i) != 0u && bk == 16u && room > 16u) bk += wave_extend_back((const uint8_t*)in, Pi - 16u, Pi - di - 16u, room - 16u);
                    s = Pi - bk;
                }
                flush();
                op = Fmt::emit_wave(in, out, op, cur, s - cur, di, Ei - s);
                cur = Ei;
            }
        }
    }

    // one round over [pos, pos + span); cur = end of the last selected match on entry and exit (the same in every wavefront of the chunk)
    __device__ __forceinline__ void round(uint32_t pos, uint32_t span, uint32_t D0, uint32_t D1, uint32_t& cur_io) {
        const uint32_t lane = lane_id();
        // wave-uniform state is TOLD to be uniform (v_readfirstlane): the compiler cannot see it through the chunk bookkeeping, and a
        // walk it believes divergent becomes an exec-masked loop with `cur` in a VGPR (measured: twice the scalar instructions)
        uint32_t cur = uni(cur_io);
        pos = uni(pos); span = uni(span);
        const uint32_t gpos = pos + 256u * wv;
        const uint32_t round_last = umin(last_start, pos + span - 1u);
        Heads h;
        uint32_t v[4];
        CJ_PROF_COUNT(0, 1);
        // the table, in position order: wavefront 0's two blocks, then wavefront 1's.  (Wavefront 1's stores of the PREVIOUS round lie
        // before that round's turns in its program, and both wavefronts have met twice since: no meeting at the start of a round.)
        if constexpr (kW > 1) { if (wv != 0u) CJ_PROF(6, meet()); }
        CJ_PROF(4, probe_table(gpos, round_last, D0, D1, v, h));
        if constexpr (kW > 1) { if (wv == 0u) meet(); }
        CJ_PROF(1, probe_verify(gpos, v, h));
        // every wavefront measures its first window at once; then the turns: wavefront 0 selects (all its windows), then wavefront 1
        // (measuring a wavefront's first TWO windows ahead of the turns — text and tables have two to four — takes 96 registers and
        //  gives the corpus +2 %, synthetic data -2 %: r05 e32, not kept)
        Meas m;
        for (uint32_t w0 = 0; w0 < h.total; w0 += 64u) {
            CJ_PROF(2, measure(h, gpos, w0, cur, m));
            CJ_PROF_COUNT(7, 1);
            CJ_PROF_COUNT(10, m.mw);
            if constexpr (kW > 1) {
                if (w0 == 0u && wv != 0u) { CJ_PROF(6, meet()); cur = uni(scr[kSharedAt]); q_n = uni(scr[kSharedAt + 1u]); op = uni(scr[kSharedAt + 2u]); }
            }
            CJ_PROF(3, select(m, cur));
        }
        if constexpr (kW > 1) {
            if (h.total == 0u && wv != 0u) { meet(); cur = uni(scr[kSharedAt]); q_n = uni(scr[kSharedAt + 1u]); op = uni(scr[kSharedAt + 2u]); }
        }
        if constexpr (kW > 1) {
            if (lane == 0u) { scr[kSharedAt] = cur; scr[kSharedAt + 1u] = q_n; scr[kSharedAt + 2u] = op; }
            if (wv == 0u) meet();
            meet();                                              // both turns are over
            cur = uni(scr[kSharedAt]); q_n = uni(scr[kSharedAt + 1u]); op = uni(scr[kSharedAt + 2u]);
        }
        cur_io = cur;
    }

    // the whole range [q0, n): rounds, then the queue; returns the end of the last match (the final literals start there)
    __device__ __forceinline__ uint32_t run(uint32_t q0) {
        const uint32_t lane = lane_id();
        q0 = uni(q0); n = uni(n); last_start = uni(last_start); limit = uni(limit); op = uni(op); wv = uni(wv);
        uint32_t pos = q0, cur = q0;
        uint32_t span = q0 == 0u ? 64u : kRound;      // short first rounds while the table is empty (a sub-piece's table is pre-indexed)
        uint32_t D0 = 0, D1 = 0, own_pos = ~0u;
        meet();                                               // both wavefronts have cleared their half of the table
        while (pos <= last_start) {
            if (own_pos != pos) {
                const uint32_t b = pos + 256u * wv + 4u * lane;
                D0 = D1 = 0u;
                if (b <= last_start) { D0 = g32(in, b); D1 = g32(in, b + 4u); }
            }
            // the next round's dwords travel while this round runs
            const uint32_t round_end = pos + span;
            uint32_t N0 = 0, N1 = 0;
            {
                const uint32_t b = round_end + 256u * wv + 4u * lane;
                if (b <= last_start) { N0 = g32(in, b); N1 = g32(in, b + 4u); }
            }
            round(pos, span, D0, D1, cur);
            D0 = N0; D1 = N1;
            own_pos = round_end;
            span = span * 2u < kRound ? span * 2u : kRound;
            cur = uni(cur);
            pos = cur > round_end ? cur : round_end;
        }
        if (wv == 0u) flush();
#ifdef CJ_ENC_PROFILE
        if (lane == 0u) for (int i = 0; i < 12; i++) atomicAdd(&g_enc_prof[i], (unsigned long long)prof_acc[i]);
#endif
        return cur;
    }
};

}  // namespace enc2
#endif
}  // namespace cj
