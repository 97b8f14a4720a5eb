// Experiment x05 (profiles/r04/experiments/README.md): two chunks per lane in the lane-per-chunk parse.  Not built into the product;
// it sat in cramjam_amd/csrc/lz4_decode_lanes.hip behind launch_lz4_parse (env CJ_PARSE_DUAL=1) and was bit-exact on the bench data.
// ---------------------------------------------------------------------------------------------------
// lz4_parse_dual_kernel — the same walk with TWO chunks per lane.  What a step of lz4_parse_kernel costs is the dependent chain
// token -> addresses -> LDS -> fields -> tests of one set of 64 chunks at ~10 cycles per dependent instruction, with 1.5 wavefronts on a
// SIMD that idles 85 % of the time (profiles/r04/experiments x04).  Here a wavefront walks 128 chunks as two independent sets A and B
// whose straight path is ONE branch-free block: both tokens decoded, all four ring reads requested, one wait, both validity tests, both
// commits as selects — the scheduler interleaves the two chains.  Refill rounds, the general walk and the sync batches run per set
// (the same code as the single kernel, as methods of Lz4Set).
// ---------------------------------------------------------------------------------------------------
struct Lz4Set {
    // the chunk
    const uint8_t* in0; const uint8_t* in;
    uint32_t c, cap, iend, mis, hist;
    bool exists, linked;
    uint2* csync;
    // the walk
    LaneStream st;
    RefillPlan plan;
    uint32_t ip, op, nseq;
    bool done;
    int64_t r;
    ParseMeta pm;
    SyncBatch sb;
    // the trip
    bool sp_hit, fast_ok, more, ip_low_ok;
    uint32_t sp_ip, sp_op, sp_slot, win_end, t4;
    int32_t ip2_max;

    __device__ __forceinline__ void setup(const BatchArgs& a, uint32_t chunk, uint2* sync, uint32_t set_ring) {
        c = chunk;
        exists = c < a.n_chunks;
        in0 = nullptr; in = nullptr;
        uint64_t n64 = 0, cap64 = 0;
        pm = ParseMeta{0u, 0u};
        r = 0; done = true; hist = 0;
        linked = (a.flags & kFlagLinkedFrame) != 0u;
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
                else if (cap0 > kLdsOutMax || iend0 > kLdsInMax) { r = 0; pm.in_skip = kRouteWave; }
                else done = false;
            }
        }
        cap = (uint32_t)cap64;
        mis = done ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(in) & 127u);
        st.base = done ? (CJ_REFILL_TOUCH ? a.in_base : nullptr) : in - mis;
        st.lo = 0; st.hi = 0;
#if CJ_REFILL_TOUCH
        st.touch = 0;
#endif
        st.end = done ? 0u : mis + (uint32_t)n64;
        st.ring = set_ring + lane_id() * kRingStride;
        iend = st.end;
        plan = refill_plan(st);
        csync = sync + (size_t)c * kSyncPitch;
        ip = mis; op = 0; nseq = 0;
        sb = SyncBatch{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        fast_ok = false; more = false; t4 = 0;
    }
    __device__ __forceinline__ void refill(uint32_t set_ring) {
        if (!done && ip >= st.hi) st.lo = st.hi = ip & ~127u;      // jumped past the window (long literal run): re-anchor
        for (;;) {
            const bool want = !done && st.hi < iend && (st.hi - st.lo < kRingBytes || ip >= st.lo + 128u);
            const bool urgent = want && ip + kParseAhead > st.hi;
            if (ballot64(urgent) == 0ull) break;
            refill_round(st, want, set_ring, plan);
        }
    }
    __device__ __forceinline__ void trip_begin() {
        sp_hit = false; sp_ip = 0; sp_op = 0; sp_slot = 0;
        if (!done) {
            sp_hit = (nseq % kSyncEvery) == 0u;
            sp_ip = ip - mis; sp_op = op; sp_slot = nseq / kSyncEvery;
            nseq += 1;
        }
        win_end = st.hi < iend ? st.hi : iend;
        ip2_max = (int32_t)win_end - 4 < (int32_t)iend - 8 ? (int32_t)win_end - 4 : (int32_t)iend - 8;
        ip_low_ok = ip >= st.lo;
    }
    // the general walk for the trip's first sequence (lz4_parse_kernel's, unchanged)
    __device__ __forceinline__ void general() {
        if (!done && !fast_ok) {
            bool bad = false, last = false;
            const uint32_t t4g = st.ld32(ip);
            const uint32_t token = t4g & 0xffu;
            ip += 1;
            uint32_t lit = token >> 4;
            if (lit == 15u) {
                if (ip + 15u >= iend) bad = true;
                else {
                    uint32_t b = (t4g >> 8) & 0xffu;
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
                    if ((nseq + kSyncEvery - 1u) / kSyncEvery > kSyncStride || (nseq < kLdsMinSeq && !linked)) pm.in_skip = kRouteWave;
                    else { pm.nseq = nseq; pm.in_skip = (uint32_t)(in - in0); }
                }
            }
        }
    }
    __device__ __forceinline__ void trip_end() { if (sp_hit) sb.put(csync, sp_slot, make_uint2(sp_ip, sp_op)); }
    __device__ __forceinline__ void finish(const BatchArgs& a, ParseMeta* meta) {
        if (exists) {
            sb.flush(csync, (nseq + kSyncEvery - 1u) / kSyncEvery);
            a.result[c] = r;
            meta[c] = pm;
        }
    }
    // ---- the straight path in three pieces, so that the caller can lay two sets side by side ----
    struct Tok { uint32_t lit, ip2, ip3, mc, op2; bool x2, early; };
    __device__ __forceinline__ Tok decode(bool act) const {
        Tok k;
        const uint32_t token = t4 & 0xffu, e1 = (t4 >> 8) & 0xffu;
        const bool x1 = (token >> 4) == 15u;
        k.lit = (token >> 4) + (x1 ? e1 : 0u);
        k.ip2 = ip + (x1 ? 2u : 1u) + k.lit;
        k.mc = token & 15u;
        k.x2 = k.mc == 15u;
        k.ip3 = k.ip2 + (k.x2 ? 3u : 2u);
        k.op2 = op + k.lit;
        k.early = act & ip_low_ok & ((int32_t)(ip + 4u) <= (int32_t)win_end) & ((int32_t)k.ip2 <= ip2_max)
                  & !(x1 & (e1 == 255u)) & ((int32_t)(k.op2 + 12u) <= (int32_t)cap);
        return k;
    }
    // extra: a sequence behind the trip's first one — counted and (maybe) noted as a sync point only if it holds
    __device__ __forceinline__ void commit(const Tok& k, uint32_t o4, uint32_t t4n, bool extra) {
        const uint32_t offset = o4 & 0xffffu, e2 = (o4 >> 16) & 0xffu;
        const uint32_t op3 = k.op2 + k.mc + (k.x2 ? e2 : 0u) + 4u;
        const bool ok = k.early & !(k.x2 & (e2 == 255u)) & ((int32_t)(op3 + 5u) <= (int32_t)cap) & (offset != 0u) & (offset <= k.op2 + hist);
        if (extra) {
            const bool hit = ok & ((nseq % kSyncEvery) == 0u);
            sp_ip = hit ? ip - mis : sp_ip; sp_op = hit ? op : sp_op; sp_slot = hit ? nseq / kSyncEvery : sp_slot;
            sp_hit = sp_hit | hit;
            nseq += ok ? 1u : 0u;
        } else fast_ok = ok;
        ip = ok ? k.ip3 : ip; op = ok ? op3 : op;
        more = ok; t4 = t4n;
    }
};

// one sequence of each set, side by side: both decodes, four ring reads, ONE wait, both commits — no branch in between
__device__ __forceinline__ void lz4_fast_pair(Lz4Set& A, Lz4Set& B, bool actA, bool actB, bool extra) {
    const Lz4Set::Tok ka = A.decode(actA), kb = B.decode(actB);
    LaneStream::Pair ra = A.st.ring32x2_request(ka.ip2, ka.ip3), rb = B.st.ring32x2_request(kb.ip2, kb.ip3);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra.w), "+v"(ra.x), "+v"(rb.w), "+v"(rb.x) :: "memory");
    const uint32_t o4a = __builtin_amdgcn_alignbyte((uint32_t)(ra.w >> 32), (uint32_t)ra.w, ka.ip2 & 3u);
    const uint32_t t4a = __builtin_amdgcn_alignbyte((uint32_t)(ra.x >> 32), (uint32_t)ra.x, ka.ip3 & 3u);
    const uint32_t o4b = __builtin_amdgcn_alignbyte((uint32_t)(rb.w >> 32), (uint32_t)rb.w, kb.ip2 & 3u);
    const uint32_t t4b = __builtin_amdgcn_alignbyte((uint32_t)(rb.x >> 32), (uint32_t)rb.x, kb.ip3 & 3u);
    A.commit(ka, o4a, t4a, extra);
    B.commit(kb, o4b, t4b, extra);
}

__global__ __launch_bounds__(64) void lz4_parse_dual_kernel(BatchArgs a, uint2* sync, ParseMeta* meta) {
    __shared__ __attribute__((aligned(16))) uint8_t rings[2 * 64 * kRingStride];
    const uint32_t lane = lane_id();
    const uint32_t ringA = (uint32_t)(uintptr_t)rings, ringB = ringA + 64u * kRingStride;
    Lz4Set A, B;
    A.setup(a, blockIdx.x * 128u + lane, sync, ringA);
    B.setup(a, blockIdx.x * 128u + 64u + lane, sync, ringB);
    while (ballot64(!A.done || !B.done) != 0ull) {
        A.refill(ringA);
        B.refill(ringB);
        A.trip_begin();
        B.trip_begin();
        {   // the trip's first tokens, both under one wait (a lane that is done reads its own ring, too: harmless)
            uint64_t wa, wb;
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(wa), "=&v"(wb) : "v"(A.st.ring + (A.ip & (kRingBytes - 4u))), "v"(B.st.ring + (B.ip & (kRingBytes - 4u))) : "memory");
            A.t4 = __builtin_amdgcn_alignbyte((uint32_t)(wa >> 32), (uint32_t)wa, A.ip & 3u);
            B.t4 = __builtin_amdgcn_alignbyte((uint32_t)(wb >> 32), (uint32_t)wb, B.ip & 3u);
        }
        lz4_fast_pair(A, B, !A.done, !B.done, false);
#pragma unroll
        for (int rep = 0; rep < CJ_PARSE_EXTRA; rep++) {
            if (ballot64(A.more || B.more) == 0ull) break;
            lz4_fast_pair(A, B, A.more, B.more, true);
        }
        A.general();
        B.general();
        A.trip_end();
        B.trip_end();
    }
    A.finish(a, meta);
    B.finish(a, meta);
}

