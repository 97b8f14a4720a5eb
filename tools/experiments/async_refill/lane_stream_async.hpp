// lane_stream_async.hpp — the lane-per-chunk parse kernels' view of their streams, refilled ASYNCHRONOUSLY.
//
// lane_stream.hpp's rings are refilled by a wave-convergent round that WAITS for its loads: one HBM round trip (~3 400 cycles with
// everything around it) every fifth step — a third of the time of a kernel whose 1 563 wavefronts have nothing else to run
// (profiles/r04/experiments s03).  Here the lines arrive by LDS-DMA (global_load_lds_dwordx4: no destination registers, no ds_write,
// nobody waits): a lane that has left a 128-byte line behind requests the next one as soon as it has worked 32 bytes into its last
// cached line, and the request is COMMITTED (hi += 128) by the s_waitcnt of a later service call, several steps on, when the bytes
// have long landed.
//
// An LDS-DMA instruction writes lane l's 16 bytes to M0 + 16 l with a wave-uniform M0, so the cooperative fetch — lanes 8u .. 8u + 7
// of load r fetch the eight pieces of target lane 8r + u's line — lands target t's line at block r + 128 u: the lines live in two
// SLOTS of 8 KiB per wavefront, slot s, target t at s * 8192 + (t / 8) * 1024 + (t % 8) * 128.  Which of its two slots a target's next
// line goes to depends on where that lane stands, so every load is issued as two masked instructions, one per slot (16 per round).
#pragma once
#include "lz4_lane_walk.hpp"

namespace cj {

constexpr uint32_t kAsWaveBytes = 2u * 8192u;              // two slots of 64 lines per wavefront

struct LaneStreamA {
    const uint8_t* base;    // 128 B aligned address at or below the first stream byte
    uint32_t lo, hi;        // committed window [lo, hi) in offsets from base; lo a multiple of 128, hi - lo in {0, 128, 256}
    uint32_t pend;          // 128 when the line [hi, hi + 128) has been requested and not yet committed
    uint32_t end;           // offset of the end of the stream
    uint32_t b0, b1;        // LDS byte addresses of this lane's line at lo / at lo + 128

    __device__ __forceinline__ uint32_t addr(uint32_t p) const { return (((p - lo) & 128u) ? b1 : b0) + (p & 127u); }
    __device__ __forceinline__ bool in_window(uint32_t p) const { return p >= lo && p + 4u <= hi && p + 4u <= end; }
    // the 4 bytes at p, read from the lane's lines whether or not they are cached (the value only means something if in_window(p))
    __device__ __forceinline__ uint32_t ring32(uint32_t p) const {
        const uint32_t a0 = addr(p & ~3u), a1 = addr((p + 4u) & ~3u);
        uint32_t w0, w1;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0), "v"(a1) : "memory");
        return __builtin_amdgcn_alignbyte(w1, w0, p & 3u);
    }
    __device__ __forceinline__ uint32_t ld32(uint32_t p) const {        // 4 bytes at offset p (little endian), anywhere in the stream
        if (in_window(p)) return ring32(p);
        const uint32_t v = ld_le_tail(base, p, end);
        __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0) only, here (not on the common path behind the branch)
        return v;
    }
    __device__ __forceinline__ uint32_t ld8(uint32_t p) const { return ld32(p) & 0xffu; }
};

// what a lane needs to know about the 8 lanes it fetches for (lane t = 8 r + lane / 8 in load r): their stream base and end never change
struct AsPlan { uint32_t blo[8], bhi[8], end[8]; };
__device__ __forceinline__ AsPlan as_plan(const LaneStreamA& st) {
    AsPlan p;
    const uint32_t lane = lane_id();
    const uint32_t blo = (uint32_t)(uintptr_t)st.base, bhi = (uint32_t)((uintptr_t)st.base >> 32);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int t = 8 * r + (int)(lane >> 3);
        p.blo[r] = (uint32_t)__shfl((int)blo, t); p.bhi[r] = (uint32_t)__shfl((int)bhi, t); p.end[r] = (uint32_t)__shfl((int)st.end, t);
    }
    return p;
}

// the lanes of `mask` load 16 bytes from g to the LDS address lds_base + 16 * lane (lds_base wave-uniform)
__device__ __forceinline__ void as_glds16(uint64_t mask, const uint8_t* g, uint32_t lds_base) {
    uint64_t sv; uint32_t keep;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b32 %1, m0\n\t"
                 "s_mov_b64 exec, %2\n\t"
                 "s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(sv), "=&s"(keep) : "s"(mask), "v"(g), "s"(lds_base) : "memory");
}

__device__ __forceinline__ void as_init(LaneStreamA& st, uint32_t wave_buf, uint32_t ip) {
    const uint32_t lane = lane_id();
    const uint32_t tb = wave_buf + (lane >> 3) * 1024u + (lane & 7u) * 128u;
    st.lo = st.hi = ip & ~127u; st.pend = 0u;
    st.b0 = tb; st.b1 = tb + 8192u;
}

// Called once per step, before the step: keeps every live lane supplied.  `outstanding` (wave-uniform) = LDS-DMA loads in flight.
__device__ __forceinline__ void as_service(LaneStreamA& st, bool alive, uint32_t ip, uint32_t ahead, uint32_t wave_buf, const AsPlan& plan, uint32_t& outstanding) {
    const uint32_t lane = lane_id();
    // left the line at lo behind (and stands in the next one, cached or requested): it goes, the other slot's line is the first now
    if (alive && ip >= st.lo + 128u && ip < st.hi + st.pend) { st.lo += 128u; const uint32_t t = st.b0; st.b0 = st.b1; st.b1 = t; }
    const bool jumped = alive && ip >= st.hi + st.pend;                 // behind everything cached or requested (start, long literal run)
    bool urgent = alive && st.hi < st.end && ip + ahead > st.hi;        // about to run out of committed bytes
    if (ballot64(jumped || (urgent && st.pend != 0u)) != 0ull) {
        if (outstanding != 0u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); outstanding = 0u; }
        st.hi += st.pend; st.pend = 0u;                                 // everything requested has landed
        if (alive && ip >= st.hi) {                                     // re-anchor: nothing of this lane is in flight now
            const uint32_t tb = wave_buf + (lane >> 3) * 1024u + (lane & 7u) * 128u;
            st.lo = st.hi = ip & ~127u; st.b0 = tb; st.b1 = tb + 8192u;
        }
        urgent = alive && st.hi < st.end && ip + ahead > st.hi;
    }
    const bool room = alive && st.hi + st.pend < st.end && st.hi + st.pend - st.lo < 256u;
#ifndef CJ_AS_HUNGRY
#define CJ_AS_HUNGRY 96u
#endif
    const bool hungry = room && st.pend == 0u && st.hi < ip + CJ_AS_HUNGRY;      // 32 bytes into its last cached line
    if (ballot64(urgent || hungry) == 0ull) return;
    {   // a round: every lane that has room requests its next line
        const uint32_t rq = st.hi + st.pend;                            // the line to request (a multiple of 128)
        const uint32_t slot_addr = ((rq - st.lo) & 128u) ? st.b1 : st.b0;
        const uint32_t par = ((slot_addr - wave_buf) >> 13) & 1u;                    // which of the wavefront's two slots
        const uint32_t mine = rq | (room ? 1u : 0u) | (par << 1);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int t = 8 * r + (int)(lane >> 3);
            const uint32_t th = (uint32_t)__shfl((int)mine, t);
            const uint32_t off = (th & ~127u) + 16u * (lane & 7u);
            const bool ok = (th & 1u) && off < plan.end[r];
            const uint8_t* src = reinterpret_cast<const uint8_t*>(((uint64_t)plan.bhi[r] << 32) | plan.blo[r]) + off;      // 16 B aligned, never crosses into a page past the stream
            const uint64_t m0 = ballot64(ok && !(th & 2u)), m1 = ballot64(ok && (th & 2u));
            if (m0 != 0ull) { as_glds16(m0, src, wave_buf + (uint32_t)r * 1024u); outstanding += 1u; }
            if (m1 != 0ull) { as_glds16(m1, src, wave_buf + 8192u + (uint32_t)r * 1024u); outstanding += 1u; }
        }
        if (room) st.pend = 128u;
    }
    // a lane without a committed byte at its position cannot step: its line (just requested) is waited for at once
    if (ballot64(alive && st.hi < st.end && ip + ahead > st.hi && st.pend != 0u && ip + 4u > st.hi) != 0ull) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); outstanding = 0u;
        st.hi += st.pend; st.pend = 0u;
    }
}

}  // namespace cj
