// lane_stream.hpp — per-lane 336-byte line cache in LDS for the lane-per-chunk PARSE kernels (LZ4 and Snappy).
// A wave-load with 64 unrelated addresses costs ~2.3k cycles here (64 separate line requests, measured), and a lane
// touches each 128 B line ~16 times; so instead the wavefront refills the caches cooperatively — 8 lanes fetch one
// lane's next 128 B line with aligned 16 B loads, 8 lines per load instruction — and the per-sequence reads become
// LDS reads.  Lane rings are 352 B apart (16 B aligned: a fetched piece is one ds_write_b128).
#pragma once
#include "lz4_lane_walk.hpp"

namespace cj {

#ifndef CJ_PARSE_WAVES
#define CJ_PARSE_WAVES 1
#endif
#ifndef CJ_REFILL_TOUCH
#define CJ_REFILL_TOUCH 1
#endif
constexpr uint32_t kParseWaves = CJ_PARSE_WAVES;                    // waves per block (ring storage 64 x 352 B = 22 KiB static LDS: seven blocks per CU)
// Ring size.  A refill round costs ~4.3 k cycles whatever it brings (profiles/r04/experiments x07: memory latency and ring writes), and with
// 256-byte rings every trip needs one (a lane that runs dry sits inside its last line: room for ONE more).  336 bytes take TWO lines per
// round — half the rounds — and are the most that keeps every wavefront of a 100 k-chunk batch resident (1 563 wavefronts = 7 per CU:
// 64 x 352 B = 22 KiB of LDS each).  Not a power of two: a stream offset maps to ring offset (p - cyc) [+ R if negative].
constexpr uint32_t kRingBytes = 336;
constexpr uint32_t kRingStride = kRingBytes + 16;      // + the ring's first 16 bytes again behind its end (reads of 8 / 12 bytes from any dword never wrap)
constexpr uint32_t kRingLines = 2;                     // lines a lane can take in one round

struct LaneStream {
    const uint8_t* base;    // 128 B aligned address at or below the first stream byte
    uint32_t lo, hi;        // cached window [lo, hi) in offsets from base; hi a multiple of 128, hi - lo <= kRingBytes
    uint32_t cyc;           // stream offset that sits at ring offset 0 in the current lap: hi - cyc in [0, kRingBytes), a multiple of 16
    uint32_t end;           // offset of the end of the stream
    uint32_t ring;          // LDS byte offset of this lane's ring
#if CJ_REFILL_TOUCH
    uint32_t touch, touch2; // a word of each line the lane will ask for next, requested when the current ones arrived (see refill_round)
#endif

    __device__ __forceinline__ void anchor(uint32_t p) { lo = hi = cyc = p & ~127u; }
    // LDS address of the dword that holds stream offset p (any p inside the window; outside it the address is some harmless place)
    __device__ __forceinline__ uint32_t addr(uint32_t p) const {
        const int32_t o = (int32_t)((p & ~3u) - cyc);
        return ring + (uint32_t)(o + ((o >> 31) & (int32_t)kRingBytes));
    }
    // lines this lane can take now: what fits behind the position it still needs (everything before ip is dead), what the stream has left
    __device__ __forceinline__ uint32_t lines_wanted(uint32_t ip, bool done) const {
        if (done || hi >= end) return 0u;
        const uint32_t room = (ip + kRingBytes - hi) >> 7, left = (end - hi + 127u) >> 7;
        const uint32_t n = room < left ? room : left;
        return n < kRingLines ? n : kRingLines;
    }
    __device__ __forceinline__ uint32_t ld32(uint32_t p) const {        // 4 bytes at offset p (little endian)
        if (p >= lo && p + 4u <= hi && p + 4u <= end) return ring32(p);
        // outside the window (long literal run, stream tail): a global read.  It waits for its own data here, inside the branch:
        // otherwise the compiler puts a vmcnt(0) wait on the common path after the branch, where it also waits for the sync point
        // store of the step (vmcnt counts stores) — a store round trip every eighth step for nothing.
        const uint32_t v = ld_le_tail(base, p, end);
        __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0) only
        return v;
    }
    __device__ __forceinline__ uint32_t ld8(uint32_t p) const { return ld32(p) & 0xffu; }
    // the straight-line form for the parse kernels' common case: is [p, p + 4) cached, and the 4 bytes at p read from the ring
    // whether or not it is (the value only means something if in_window(p))
    __device__ __forceinline__ bool in_window(uint32_t p) const { return p >= lo && p + 4u <= hi && p + 4u <= end; }
    __device__ __forceinline__ uint32_t ring32(uint32_t p) const {
        uint64_t w;
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr(p)) : "memory");
        return __builtin_amdgcn_alignbyte((uint32_t)(w >> 32), (uint32_t)w, p & 3u);
    }
    // Two positions under one wait (the parse kernels: a sequence's offset field and the NEXT sequence's token — where that token sits
    // follows from the current token alone, so a sequence costs one dependent LDS round trip instead of two), as request + arrival: what
    // does not depend on the bytes (window tests, output margins) is written between the two and runs during the round trip.  (As plain
    // LDS loads the compiler issued the two reads a block apart, each with its own wait.)
    struct Pair { uint64_t w, x; };
    __device__ __forceinline__ Pair ring32x2_request(uint32_t p, uint32_t q) const {
        Pair r;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %3 offset1:1"
                     : "=&v"(r.w), "=&v"(r.x) : "v"(addr(p)), "v"(addr(q)) : "memory");
        return r;
    }
    __device__ __forceinline__ void ring32x2_arrive(Pair r, uint32_t p, uint32_t q, uint32_t& vp, uint32_t& vq) const {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.w), "+v"(r.x) :: "memory");
        vp = __builtin_amdgcn_alignbyte((uint32_t)(r.w >> 32), (uint32_t)r.w, p & 3u);
        vq = __builtin_amdgcn_alignbyte((uint32_t)(r.x >> 32), (uint32_t)r.x, q & 3u);
    }
    // 8 bytes at p (three dwords of the ring; the Snappy parse: a copy element and the tag bytes of the record behind it)
    struct Trio { uint64_t w; uint32_t w2; };
    __device__ __forceinline__ Trio ring64_request(uint32_t p) const {
        Trio r;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8"
                     : "=&v"(r.w), "=&v"(r.w2) : "v"(addr(p)) : "memory");
        return r;
    }
    __device__ __forceinline__ void ring64_arrive(Trio r, uint32_t p, uint32_t& lo8, uint32_t& hi8) const {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.w), "+v"(r.w2) :: "memory");
        lo8 = __builtin_amdgcn_alignbyte((uint32_t)(r.w >> 32), (uint32_t)r.w, p & 3u);
        hi8 = __builtin_amdgcn_alignbyte(r.w2, (uint32_t)(r.w >> 32), p & 3u);
    }
};

// One wave-convergent refill round: every lane takes the lines it has room for — `lines` of them, 0 .. kRingLines — (8 lanes per line,
// aligned 16 B loads, 8 lines per load instruction) and the data is written to the lanes' rings.
// (Measured alternative: issuing in one round and committing in the next with a 512 B ring halves occupancy —
// 33 KiB of LDS per wave — and ran slower: 8.6 ms vs 3.6 ms for 100 k chunks.)
// What a lane needs to know about the 8 lanes it fetches for (lane t = 8 r + lane / 8 in step r): their stream base and
// end never change, so they are exchanged once per kernel instead of once per round (5 -> 1 ds_bpermute per step).
struct RefillPlan { uint32_t blo[8], bhi[8], end[8]; };
__device__ __forceinline__ RefillPlan refill_plan(const LaneStream& st) {
    RefillPlan p;
    const uint32_t lane = lane_id();
    const uint32_t blo = (uint32_t)(uintptr_t)st.base, bhi = (uint32_t)((uintptr_t)st.base >> 32);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int t = 8 * r + (int)(lane >> 3);
        p.blo[r] = (uint32_t)__shfl((int)blo, t); p.bhi[r] = (uint32_t)__shfl((int)bhi, t); p.end[r] = (uint32_t)__shfl((int)st.end, t);
    }
    return p;
}

__device__ __forceinline__ void refill_round(LaneStream& st, uint32_t lines, uint32_t wave_ring, const RefillPlan& plan) {
    const uint32_t lane = lane_id(), piece = lane & 7u;
#if CJ_REFILL_TOUCH
    asm volatile("" :: "v"(st.touch), "v"(st.touch2));                 // the previous round's touches are accounted for here, not earlier
#endif
    // hi is a multiple of 128: its low bits carry the lane's line count and (hi - cyc) / 16, the ring offset its next line goes to
    const uint32_t mine = st.hi | (((st.hi - st.cyc) >> 4) << 2) | lines;
    uint32_t th[8];
#pragma unroll
    for (int r = 0; r < 8; r++) th[r] = (uint32_t)__shfl((int)mine, 8 * r + (int)(lane >> 3));
    uint4 v[8 * kRingLines];
    uint32_t dsta[8 * kRingLines];
#pragma unroll
    for (int k = 0; k < (int)kRingLines; k++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int t = 8 * r + (int)(lane >> 3), i = 8 * k + r;
            const uint32_t n = th[r] & 3u, woff = ((th[r] >> 2) & 31u) << 4;
            const uint32_t off = (th[r] & ~127u) + 128u * (uint32_t)k + 16u * piece;
            dsta[i] = 0xffffffffu;
            if (n > (uint32_t)k && off < plan.end[r]) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(((uint64_t)plan.bhi[r] << 32) | plan.blo[r]) + off;
                v[i] = *reinterpret_cast<const uint4*>(src);           // 16 B aligned, never crosses into a page past the stream
                uint32_t ro = woff + 128u * (uint32_t)k + 16u * piece;
                ro -= ro >= kRingBytes ? kRingBytes : 0u;
                dsta[i] = wave_ring + (uint32_t)t * kRingStride + ro;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8 * (int)kRingLines; i++) {
        if (dsta[i] != 0xffffffffu) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 q = {v[i].x, v[i].y, v[i].z, v[i].w};
            asm volatile("ds_write_b128 %0, %1" :: "v"(dsta[i]), "v"(q) : "memory");
            if (dsta[i] == wave_ring + (uint32_t)(8 * (i & 7) + (int)(lane >> 3)) * kRingStride)       // the ring's first piece: again behind the ring's end
                asm volatile("ds_write_b128 %0, %1" :: "v"(dsta[i] + kRingBytes), "v"(q) : "memory");
        }
    }
    if (lines) {
        st.hi += 128u * lines;
        if (st.hi - st.cyc >= kRingBytes) st.cyc += kRingBytes;
        if (st.hi - st.lo > kRingBytes) st.lo = st.hi - kRingBytes;
    }
#if CJ_REFILL_TOUCH
    {   // one word of each line this lane asks for next (its round is a trip or two away): that round finds them in the L2
        typedef const uint32_t __attribute__((address_space(1)))* GlobalWord;     // (a flat load would also count against the LDS reads' lgkmcnt)
        const uint32_t last = st.end ? (st.end - 1u) & ~127u : 0u, n1 = st.hi, n2 = st.hi + 128u;
        st.touch = *(GlobalWord)(uintptr_t)(st.base + (n1 < last ? n1 : last));
        st.touch2 = *(GlobalWord)(uintptr_t)(st.base + (n2 < last ? n2 : last));
    }
#endif
}

}  // namespace cj
