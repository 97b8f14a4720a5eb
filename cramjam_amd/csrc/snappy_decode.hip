// snappy_decode.hip — Snappy *raw* decoder for gfx950, one wavefront per independent chunk.
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/snappy.rs:57,106:
// libcramjam::snappy::raw::decompress -> snap 1.1.1 raw::Decoder::decompress.  Error conditions
// follow snap: Empty, Header (bad varint), TooBig (> u32::MAX), BufferTooSmall (preamble > cap),
// and one "corrupt" class for Literal/CopyRead/CopyWrite/Offset/HeaderMismatch.
//
// Same shape as lz4_decode.hip: tag grammar parsed wave-uniformly from the 512-byte register window,
// lanes move bytes.
#include "lz4_lane_walk.hpp"   // lane_copy / lane_match / ParseMeta route flags are codec independent
#include "lane_stream.hpp"
#include "snappy_records.hpp"

namespace cj {

// route: nullptr = decode every chunk; else only chunks the parse stage flagged kRouteWave
__global__ __launch_bounds__(kBlockThreads) void snappy_decode_kernel(BatchArgs a, const ParseMeta* route) {
    const uint32_t chunk = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (chunk >= a.n_chunks) return;
    if (route != nullptr && (route[chunk].in_skip & kRouteWave) == 0u) return;
    const uint8_t* in = a.in_base + a.in_off[chunk];
    const uint64_t n64 = a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    const uint64_t cap64 = a.out_cap[chunk];
    int64_t status = 0;

    if (n64 == 0) status = CJ_E_SNAPPY_EMPTY;
    else if (n64 > 0xFFFFFFF0ull) status = CJ_E_SNAPPY_CORRUPT;
    if (status != 0) { if (lane_id() == 0) a.result[chunk] = status; return; }

    InWindow w;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u);
    w.base = in - mis;
    w.iend = mis + (uint32_t)n64;
    w.anchor(mis);
    const uint32_t iend = w.iend;
    uint32_t ip = mis;

    // preamble: snap bytes::read_varu64 (up to 10 bytes), then the u32 range check
    uint64_t ulen = 0;
    {
        uint32_t shift = 0, i = 0;
        bool done = false;
        while (ip < iend && i < 10u) {
            uint32_t b = w.fetch32_any(ip) & 0xffu;
            ip += 1;
            if (b < 0x80u) {
                if (i == 9u && b > 1u) break;
                ulen |= (uint64_t)b << shift;
                done = true;
                break;
            }
            ulen |= (uint64_t)(b & 0x7fu) << shift;
            shift += 7; i += 1;
        }
        if (!done) status = CJ_E_SNAPPY_HEADER;
        else if (ulen > 0xFFFFFFFFull) status = CJ_E_SNAPPY_TOO_BIG;
        else if (ulen > cap64) status = CJ_E_SNAPPY_BUF_SMALL;
    }
    if (status != 0) { if (lane_id() == 0) a.result[chunk] = status; return; }

    const uint32_t dn = (uint32_t)ulen;
    uint32_t op = 0;
    bool bad = false;

    while (ip < iend) {
        w.ensure(ip);
        const uint32_t t4 = w.fetch32(ip);
        const uint32_t tag = t4 & 0xffu;
        ip += 1;
        const uint32_t kind = tag & 3u;
        if (kind == 0u) {
            uint64_t len = (tag >> 2) + 1u;
            if (len > 60u) {
                const uint32_t nb = (uint32_t)len - 60u;        // 1..4 length bytes
                if (iend - ip < nb) { bad = true; break; }
                uint32_t v = w.fetch32_any(ip);
                if (nb < 4u) v &= (1u << (8u * nb)) - 1u;
                ip += nb;
                len = (uint64_t)v + 1u;
            }
            if (len > (uint64_t)(iend - ip) || len > (uint64_t)(dn - op)) { bad = true; break; }
            wave_copy(out + op, w.base + ip, (uint32_t)len);
            ip += (uint32_t)len; op += (uint32_t)len;
            continue;
        }
        uint32_t len, offset;
        if (kind == 1u) {
            if (iend - ip < 1u) { bad = true; break; }
            len = 4u + ((tag >> 2) & 7u);
            offset = ((tag >> 5) << 8) | ((t4 >> 8) & 0xffu);
            ip += 1;
        } else if (kind == 2u) {
            if (iend - ip < 2u) { bad = true; break; }
            len = 1u + (tag >> 2);
            offset = (t4 >> 8) & 0xffffu;
            ip += 2;
        } else {
            if (iend - ip < 4u) { bad = true; break; }
            len = 1u + (tag >> 2);
            offset = w.fetch32_any(ip);
            ip += 4;
        }
        if (offset == 0u || offset > op) { bad = true; break; }
        if (len > dn - op) { bad = true; break; }
        wave_order();
        wave_match_copy(out + op, offset, len);
        wave_order();
        op += len;
    }
    if (!bad && op != dn) bad = true;
    if (lane_id() == 0) a.result[chunk] = bad ? (int64_t)CJ_E_SNAPPY_CORRUPT : (int64_t)dn;
}

// ---------------------------------------------------------------------------------------------------
// One LANE per chunk (large batches of short-element data; see lz4_decode_lanes.hip for the rationale).
// Same accept/reject rules as the wave kernel above (snap 1.1.1 raw::Decoder::decompress).  Copies are
// 16 B wild copies while 16 B of room remain before the DECODED length, exact bytes at the tail, so nothing
// past the decoded length is written.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t snappy_lane_walk(const uint8_t* in, uint64_t n64, uint8_t* out, uint64_t cap64) {
    if (n64 == 0) return CJ_E_SNAPPY_EMPTY;
    if (n64 > 0xFFFFFFF0ull) return CJ_E_SNAPPY_CORRUPT;
    const uint32_t iend = (uint32_t)n64;
    uint32_t ip = 0;
    uint64_t ulen = 0;
    {
        uint32_t shift = 0, i = 0;
        bool done = false;
        while (ip < iend && i < 10u) {
            const uint32_t b = in[ip];
            ip += 1;
            if (b < 0x80u) {
                if (i == 9u && b > 1u) break;
                ulen |= (uint64_t)b << shift;
                done = true;
                break;
            }
            ulen |= (uint64_t)(b & 0x7fu) << shift;
            shift += 7; i += 1;
        }
        if (!done) return CJ_E_SNAPPY_HEADER;
        if (ulen > 0xFFFFFFFFull) return CJ_E_SNAPPY_TOO_BIG;
        if (ulen > cap64) return CJ_E_SNAPPY_BUF_SMALL;
    }
    const uint32_t dn = (uint32_t)ulen;
    uint32_t op = 0;
    while (ip < iend) {
        const uint32_t t4 = ld_le_tail(in, ip, iend);
        const uint32_t tag = t4 & 0xffu;
        ip += 1;
        const uint32_t kind = tag & 3u;
        if (kind == 0u) {
            uint64_t len = (tag >> 2) + 1u;
            if (len > 60u) {
                const uint32_t nb = (uint32_t)len - 60u;
                if (iend - ip < nb) return CJ_E_SNAPPY_CORRUPT;
                uint32_t v = ld_le_tail(in, ip, iend);
                if (nb < 4u) v &= (1u << (8u * nb)) - 1u;
                ip += nb;
                len = (uint64_t)v + 1u;
            }
            if (len > (uint64_t)(iend - ip) || len > (uint64_t)(dn - op)) return CJ_E_SNAPPY_CORRUPT;
            lane_copy(out + op, in + ip, (uint32_t)len, dn - op, iend - ip);
            ip += (uint32_t)len; op += (uint32_t)len;
            continue;
        }
        uint32_t len, offset;
        if (kind == 1u) {
            if (iend - ip < 1u) return CJ_E_SNAPPY_CORRUPT;
            len = 4u + ((tag >> 2) & 7u);
            offset = ((tag >> 5) << 8) | ((t4 >> 8) & 0xffu);
            ip += 1;
        } else if (kind == 2u) {
            if (iend - ip < 2u) return CJ_E_SNAPPY_CORRUPT;
            len = 1u + (tag >> 2);
            offset = (t4 >> 8) & 0xffffu;
            ip += 2;
        } else {
            if (iend - ip < 4u) return CJ_E_SNAPPY_CORRUPT;
            len = 1u + (tag >> 2);
            offset = ld32u(in + ip);
            ip += 4;
        }
        if (offset == 0u || offset > op) return CJ_E_SNAPPY_CORRUPT;
        if (len > dn - op) return CJ_E_SNAPPY_CORRUPT;
        lane_match(out + op, offset, len, dn - op);
        op += len;
    }
    if (op != dn) return CJ_E_SNAPPY_CORRUPT;
    return (int64_t)dn;
}

__global__ __launch_bounds__(64) void snappy_decode_lanes_kernel(BatchArgs a) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= a.n_chunks) return;
    a.result[c] = snappy_lane_walk(a.in_base + a.in_off[c], a.in_len[c], a.out_base + a.out_off[c], a.out_cap[c]);
}

void launch_snappy_decode(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlockThreads);
    hipLaunchKernelGGL(snappy_decode_kernel, grid, block, 0, s, a, (const ParseMeta*)nullptr);
}


void launch_snappy_decode_lanes(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    hipLaunchKernelGGL(snappy_decode_lanes_kernel, dim3((a.n_chunks + 63u) / 64u), dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------
// snappy_parse_kernel — lane-per-chunk walk + validation through the LDS line cache (lane_stream.hpp), the Snappy
// counterpart of lz4_parse_kernel: emits the decoded size, the record count and an (ip, op) sync point every
// kSyncEvery records for the workgroup-per-chunk LDS decoder; chunks that decoder cannot take (capacity or input
// too large, too few / too many records) are flagged kRouteWave for the wave kernel.
// ---------------------------------------------------------------------------------------------------
#ifndef CJ_SN_PARSE_AHEAD
#define CJ_SN_PARSE_AHEAD 32u
#endif
constexpr uint32_t kSnParseAhead = CJ_SN_PARSE_AHEAD;   // cached bytes a lane must have ahead of its position before a step (24: 2.78 ms, 32: 2.79, 40: 2.83, 48: 2.89)
__global__ __launch_bounds__(64 * kParseWaves) void snappy_parse_kernel(BatchArgs a, uint2* sync, ParseMeta* meta) {
    __shared__ __attribute__((aligned(16))) uint8_t rings[kParseWaves * 64 * kRingStride];
    const uint32_t c = blockIdx.x * (64u * kParseWaves) + threadIdx.x;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t wave_ring = (uint32_t)(uintptr_t)rings + wave * 64u * kRingStride;
    const bool exists = c < a.n_chunks;

    const uint8_t* in = nullptr;
    uint64_t n64 = 0, cap64 = 0;
    ParseMeta pm = {0u, 0u};
    int64_t r = 0;
    bool done = true;
    uint32_t dn = 0, hdr = 0;
    const uint32_t win = lds_window(a.flags);                // the decoder's window for this batch
    if (exists) {
        in = a.in_base + a.in_off[c];
        n64 = a.in_len[c];
        cap64 = a.out_cap[c];
        if (n64 == 0) r = CJ_E_SNAPPY_EMPTY;
        else if (n64 > 0xFFFFFFF0ull) r = CJ_E_SNAPPY_CORRUPT;
        else {
            uint64_t ulen = 0;
            uint32_t shift = 0, i = 0;
            bool ok = false;
            while (hdr < (uint32_t)n64 && i < 10u) {
                const uint32_t b = in[hdr];
                hdr += 1;
                if (b < 0x80u) { if (!(i == 9u && b > 1u)) { ulen |= (uint64_t)b << shift; ok = true; } break; }
                ulen |= (uint64_t)(b & 0x7fu) << shift;
                shift += 7; i += 1;
            }
            if (!ok) r = CJ_E_SNAPPY_HEADER;
            else if (ulen > 0xFFFFFFFFull) r = CJ_E_SNAPPY_TOO_BIG;
            else if (ulen > cap64) r = CJ_E_SNAPPY_BUF_SMALL;
            else if (ulen > win || n64 - hdr > win - 32u || ulen == 0) {
                if (ulen == 0) r = (hdr == (uint32_t)n64) ? 0 : (int64_t)CJ_E_SNAPPY_CORRUPT;   // nothing to decode: trailing elements are errors
                else pm.in_skip = kRouteWave;              // too big for the LDS window: the wave kernel decodes + validates
            } else { dn = (uint32_t)ulen; done = false; }
        }
    }
    LaneStream st;
    const uint32_t mis = done ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(in + hdr) & 127u);
    st.base = done ? (CJ_REFILL_TOUCH ? a.in_base : nullptr) : in + hdr - mis;
    st.lo = 0; st.hi = 0;
#if CJ_REFILL_TOUCH
    st.touch = 0;
#endif
    st.end = done ? 0u : mis + (uint32_t)n64 - hdr;
    st.ring = wave_ring + lane * kRingStride;
    const uint32_t iend = st.end;
    const RefillPlan plan = refill_plan(st);
    uint2* csync = sync + (size_t)c * kSyncPitch;
    const auto rd = [&st](uint32_t p) { return st.ld32(p); };

    uint32_t ip = mis, op = 0, nrec = 0;
    SyncBatch sb = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    while (ballot64(!done) != 0ull) {
        if (!done && ip >= st.hi) st.lo = st.hi = ip & ~127u;      // jumped past the window (long literal): re-anchor
        for (;;) {
            const bool want = !done && st.hi < iend && (st.hi - st.lo < kRingBytes || ip >= st.lo + 128u);
            const bool urgent = want && ip + kSnParseAhead > st.hi;
            if (ballot64(urgent) == 0ull) break;
            refill_round(st, want, wave_ring, plan);
        }
        // (at most one of a trip's eight records starts a sync group: noted where it occurs, handed to the sync batch at the trip's end —
        //  lz4_parse_kernel)
        bool sp_hit = false;
        uint32_t sp_ip = 0, sp_op = 0, sp_slot = 0;
        if (!done) {
            sp_hit = (nrec % kSyncEvery) == 0u;
            sp_ip = ip - mis; sp_op = op; sp_slot = nrec / kSyncEvery;
            nrec += 1;
        }
        // ---- the common record as straight-line code (same idea as in lz4_parse_kernel): an optional literal with a one- or
        //      two-byte header followed by a 1- or 2-byte-offset copy, everything cached, valid and not the stream's last record.
        //      Nothing is committed unless all of that holds; every other record takes snappy_record_step from the same state ----
        // One dependent LDS round trip per record: the read at the copy element brings 8 bytes, and the record behind it starts 2 or 3
        // bytes further — its tag (and a literal's length byte) come with the copy element.
        struct FastRec { bool ok; uint32_t ip3, op3, t4n; };
        const uint32_t win_end = st.hi < iend ? st.hi : iend;            // [p, p + 4) cached and inside the stream: p + 4 <= win_end (and p >= lo: the position only moves forward)
        const bool ip_low_ok = ip >= st.lo;
        const auto fast_rec = [&](uint32_t t4) __attribute__((always_inline)) -> FastRec {
            const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
            const bool is_lit = (tag & 3u) == 0u;
            const uint32_t lhdr = is_lit ? (l6 == 60u ? 2u : 1u) : 0u;
            const uint32_t lit_len = is_lit ? (l6 == 60u ? ((t4 >> 8) & 0xffu) + 1u : l6 + 1u) : 0u;
            const uint32_t ip2 = ip + lhdr + lit_len;                        // the copy element
            const LaneStream::Trio rq = st.ring64_request(ip2);
            const uint32_t op2 = op + lit_len;
            // (bitwise: a chain of && compiles to nested branches)
            const bool ok_early = ip_low_ok & (ip + 4u <= win_end) & (ip2 + 4u <= win_end) & !(is_lit & (l6 > 60u)) & (lit_len <= dn - op);
            uint32_t c4, c8;
            st.ring64_arrive(rq, ip2, c4, c8);
            const uint32_t ctag = c4 & 0xffu, kind = ctag & 3u;
            const uint32_t clen = kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
            const uint32_t offset = kind == 1u ? ((ctag >> 5) << 8) | ((c4 >> 8) & 0xffu) : (c4 >> 8) & 0xffffu;
            const uint32_t ip3 = ip2 + (kind == 1u ? 2u : 3u);
            const uint32_t t4n = __builtin_amdgcn_alignbyte(c8, c4, kind == 1u ? 2u : 3u);
            const bool ok = ok_early & ((kind == 1u) | (kind == 2u)) & (offset != 0u) & (offset <= op2) & (clen <= dn - op2) & (ip3 < iend);
            return FastRec{ok, ip3, op2 + clen, t4n};
        };
        bool fast_ok = false;
        uint32_t t4_next = 0;
        if (!done) {
            const FastRec f = fast_rec(st.ring32(ip));
            fast_ok = f.ok; t4_next = f.t4n;
            if (f.ok) { ip = f.ip3; op = f.op3; }
        }
        // ---- more records in the same trip where the first went the straight way and the next does too (lz4_parse_kernel): the
        //      trip's fixed cost once for all of them; nothing is committed — not even the sync point — unless the record holds ----
#ifndef CJ_SN_PARSE_EXTRA
#define CJ_SN_PARSE_EXTRA 7
#endif
        static_assert(CJ_SN_PARSE_EXTRA + 1 <= (int)kSyncEvery, "a trip must not cross two sync groups: the deferred put holds one");
        bool more = fast_ok;
#pragma unroll
        for (int rep = 0; rep < CJ_SN_PARSE_EXTRA; rep++) {
            if (ballot64(more) == 0ull) break;
            if (more) {
                const FastRec f = fast_rec(t4_next);
                if (f.ok) {
                    const bool hit = (nrec % kSyncEvery) == 0u;
                    sp_ip = hit ? ip - mis : sp_ip; sp_op = hit ? op : sp_op; sp_slot = hit ? nrec / kSyncEvery : sp_slot;
                    sp_hit = sp_hit | hit;
                    nrec += 1;
                    ip = f.ip3; op = f.op3;
                }
                more = f.ok; t4_next = f.t4n;
            }
        }
        if (!done && !fast_ok) {
            SnRecord rec;
            if (snappy_record_step(rd, ip, op, iend, dn, rec) != 0) { r = CJ_E_SNAPPY_CORRUPT; done = true; }
            else if (ip >= iend) {
                done = true;
                if (op != dn) r = CJ_E_SNAPPY_CORRUPT;
                else {
                    r = (int64_t)dn;
                    if (nrec > lds_window_max_seq(win) || nrec < lds_window_min_seq(win)) pm.in_skip = kRouteWave;
                    else { pm.nseq = nrec; pm.in_skip = hdr; }
                }
            }
        }
        if (sp_hit) sb.put(csync, sp_slot, make_uint2(sp_ip, sp_op));
    }
    if (exists) {
        sb.flush(csync, (nrec + kSyncEvery - 1u) / kSyncEvery);
        a.result[c] = r;
        meta[c] = pm;
    }
}

void launch_snappy_parse(const BatchArgs& a, void* sync, void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    const uint32_t per_block = 64u * kParseWaves;
    hipLaunchKernelGGL(snappy_parse_kernel, dim3((a.n_chunks + per_block - 1u) / per_block), dim3(per_block), 0, s, a, (uint2*)sync, (ParseMeta*)meta);
}

void launch_snappy_decode_routed(const BatchArgs& a, const void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlockThreads);
    hipLaunchKernelGGL(snappy_decode_kernel, grid, block, 0, s, a, (const ParseMeta*)meta);
}

}  // namespace cj
