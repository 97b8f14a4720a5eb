// lane_stream.hpp — per-lane 256-byte line cache in LDS for the lane-per-chunk PARSE kernels (LZ4 and Snappy).
// A wave-load with 64 unrelated addresses costs ~2.3k cycles here (64 separate line requests, measured), and a lane
// touches each 128 B line ~16 times; so instead the wavefront refills the caches cooperatively — 8 lanes fetch one
// lane's next 128 B line with aligned 16 B loads, 8 lines per load instruction — and the per-sequence reads become
// LDS reads.  Lane rings are 272 B apart (16 B aligned: a fetched piece is one ds_write_b128).
#pragma once
#include "lz4_lane_walk.hpp"

namespace cj {

#ifndef CJ_PARSE_WAVES
#define CJ_PARSE_WAVES 2
#endif
#ifndef CJ_REFILL_TOUCH
#define CJ_REFILL_TOUCH 1
#endif
constexpr uint32_t kParseWaves = CJ_PARSE_WAVES;                    // waves per block (ring storage 2 x 64 x 272 B = 34 KiB static LDS)
constexpr uint32_t kRingBytes = 256;                   // two 128 B lines per lane
constexpr uint32_t kRingStride = kRingBytes + 16;      // 16 B aligned rings (one ds_write_b128 per fetched piece; the first 8 of the 16 spare bytes mirror the ring's first 8); the lanes read at unrelated
                                                       // offsets anyway, so the exact skew between rings does not matter for bank conflicts

struct LaneStream {
    const uint8_t* base;    // 128 B aligned address at or below the first stream byte
    uint32_t lo, hi;        // cached window [lo, hi) in offsets from base; multiples of 128, hi - lo <= kRingBytes
    uint32_t end;           // offset of the end of the stream
    uint32_t ring;          // LDS byte offset of this lane's ring
#if CJ_REFILL_TOUCH
    uint32_t touch;         // a word of the line the lane will ask for next, requested when the current one arrived (see refill_round)
#endif

    __device__ __forceinline__ uint32_t ld32(uint32_t p) const {        // 4 bytes at offset p (little endian)
        if (p >= lo && p + 4u <= hi && p + 4u <= end) {
            const uint32_t a0 = ring + (p & (kRingBytes - 4u)), a1 = ring + ((p + 4u) & (kRingBytes - 4u));
            uint32_t w0, w1;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(w0), "=&v"(w1) : "v"(a0), "v"(a1) : "memory");
            return __builtin_amdgcn_alignbyte(w1, w0, p & 3u);
        }
        // outside the window (long literal run, stream tail): a global read.  It waits for its own data here, inside the branch:
        // otherwise the compiler puts a vmcnt(0) wait on the common path after the branch, where it also waits for the sync point
        // store of the step (vmcnt counts stores) — a store round trip every eighth step for nothing.
        const uint32_t v = ld_le_tail(base, p, end);
        __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0) only
        return v;
    }
    __device__ __forceinline__ uint32_t ld8(uint32_t p) const { return ld32(p) & 0xffu; }
    // the straight-line form for the parse kernels' common case: is [p, p + 4) cached, and the 4 bytes at p read from the ring
    // whether or not it is (the address never leaves the lane's ring; the value only means something if in_window(p))
    __device__ __forceinline__ bool in_window(uint32_t p) const { return p >= lo && p + 4u <= hi && p + 4u <= end; }
    // (the ring's first 8 bytes are mirrored behind its end by refill_round: the 8 / 12 bytes from any dword of the ring are read without a wrap)
    __device__ __forceinline__ uint32_t ring32(uint32_t p) const {
        uint64_t w;
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(ring + (p & (kRingBytes - 4u))) : "memory");
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
                     : "=&v"(r.w), "=&v"(r.x) : "v"(ring + (p & (kRingBytes - 4u))), "v"(ring + (q & (kRingBytes - 4u))) : "memory");
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
                     : "=&v"(r.w), "=&v"(r.w2) : "v"(ring + (p & (kRingBytes - 4u))) : "memory");
        return r;
    }
    __device__ __forceinline__ void ring64_arrive(Trio r, uint32_t p, uint32_t& lo8, uint32_t& hi8) const {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.w), "+v"(r.w2) :: "memory");
        lo8 = __builtin_amdgcn_alignbyte((uint32_t)(r.w >> 32), (uint32_t)r.w, p & 3u);
        hi8 = __builtin_amdgcn_alignbyte(r.w2, (uint32_t)(r.w >> 32), p & 3u);
    }
};

// One wave-convergent refill round: every lane that has room fetches its next 128 B line (8 lanes per line,
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

__device__ __forceinline__ void refill_round(LaneStream& st, bool want, uint32_t wave_ring, const RefillPlan& plan) {
    const uint32_t lane = lane_id(), piece = lane & 7u;
#if CJ_REFILL_TOUCH
    asm volatile("" :: "v"(st.touch));                                 // the previous round's touch is accounted for here, not earlier
#endif
    const uint32_t mine = st.hi | (want ? 1u : 0u);                    // hi is a multiple of 128
    uint32_t th[8];
#pragma unroll
    for (int r = 0; r < 8; r++) th[r] = (uint32_t)__shfl((int)mine, 8 * r + (int)(lane >> 3));
    uint4 v[8];
    uint32_t dsta[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int t = 8 * r + (int)(lane >> 3);
        const uint32_t off = (th[r] & ~1u) + 16u * piece;
        v[r] = make_uint4(0, 0, 0, 0);
        dsta[r] = 0xffffffffu;
        if ((th[r] & 1u) && off < plan.end[r]) {
            // (a pointer rebuilt from integers is a FLAT pointer to the compiler: flat_load counts against lgkmcnt too, so every LDS wait
            //  of the walk would also wait for these — the address space is spelled out)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef const u32x4 __attribute__((address_space(1)))* GlobalVec;
            const u32x4 q = *(GlobalVec)(uintptr_t)((((uint64_t)plan.bhi[r] << 32) | plan.blo[r]) + off);      // 16 B aligned, never crosses into a page past the stream
            v[r] = make_uint4(q.x, q.y, q.z, q.w);
            dsta[r] = wave_ring + (uint32_t)t * kRingStride + (off & (kRingBytes - 1u));
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (dsta[r] != 0xffffffffu) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 q = {v[r].x, v[r].y, v[r].z, v[r].w};
            asm volatile("ds_write_b128 %0, %1" :: "v"(dsta[r]), "v"(q) : "memory");
            if (dsta[r] == wave_ring + (uint32_t)(8 * r + (int)(lane >> 3)) * kRingStride) {        // the ring's first piece: its first 8 bytes again behind the ring's end
                const uint64_t head = ((uint64_t)v[r].y << 32) | v[r].x;
                asm volatile("ds_write_b64 %0, %1" :: "v"(dsta[r] + kRingBytes), "v"(head) : "memory");
            }
        }
    }
    if (want) {
        if (st.hi - st.lo >= kRingBytes) st.lo += 128u;
        st.hi += 128u;
    }
#if CJ_REFILL_TOUCH
    {   // one word of the line this lane asks for next (its round is a trip or two away): that round finds the line in the L2
        const uint32_t last = st.end ? (st.end - 1u) & ~127u : 0u, nx = st.hi + 128u * (CJ_REFILL_TOUCH - 1u);
        typedef const uint32_t __attribute__((address_space(1)))* GlobalWord;     // (a flat load would also count against the LDS reads' lgkmcnt)
        st.touch = *(GlobalWord)(uintptr_t)(st.base + (nx < last ? nx : last));
    }
#endif
}


}  // namespace cj
