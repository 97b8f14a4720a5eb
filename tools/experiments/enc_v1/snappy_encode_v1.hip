// snappy_encode.hip — Snappy *raw* encoder for gfx950, one wavefront per independent chunk.
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/snappy.rs:75,97:
// libcramjam::snappy::raw::compress -> snap 1.1.1 raw::Encoder::compress.  Output = varint(len)
// preamble + literal / copy-1 / copy-2 elements (format_description.txt); copies longer than 64
// are split exactly like the CPU encoders do (64,64,...,[60],rest) so every element is canonical.
// The 64 KiB window of the match finder keeps every offset < 65536, so copy-4 is never emitted.
#include "cj_match.hpp"

namespace cj {

__device__ __forceinline__ uint32_t emit_snappy_literal(uint8_t* out, uint32_t op, const uint8_t* src, uint32_t len) {
    const uint32_t lane = lane_id();
    const uint32_t n1 = len - 1u;
    uint32_t hdr;
    if (n1 < 60u) {
        if (lane == 0) out[op] = (uint8_t)(n1 << 2);
        hdr = 1;
    } else {
        const uint32_t nb = n1 < 256u ? 1u : n1 < 65536u ? 2u : n1 < 16777216u ? 3u : 4u;
        if (lane == 0) out[op] = (uint8_t)((59u + nb) << 2);
        if (lane >= 1u && lane <= nb) out[op + lane] = (uint8_t)(n1 >> (8u * (lane - 1u)));
        hdr = 1u + nb;
    }
    wave_copy(out + op + hdr, src, len);
    return op + hdr + len;
}

__device__ __forceinline__ uint32_t emit_snappy_copy(uint8_t* out, uint32_t op, uint32_t off, uint32_t len) {
    const uint32_t lane = lane_id();
    // 64-byte pieces while len >= 68 (keeps >= 4 for the tail)
    const uint32_t n64 = len >= 68u ? (len - 4u) / 64u : 0u;
    for (uint32_t k = 0; k < n64; k += 64u) {
        const uint32_t i = k + lane;
        if (i < n64) {
            uint8_t* p = out + op + 3u * i;
            p[0] = (uint8_t)(2u | (63u << 2));
            p[1] = (uint8_t)off;
            p[2] = (uint8_t)(off >> 8);
        }
    }
    op += 3u * n64;
    len -= 64u * n64;
    if (len > 64u) {
        if (lane == 0) { out[op] = (uint8_t)(2u | (59u << 2)); out[op + 1] = (uint8_t)off; out[op + 2] = (uint8_t)(off >> 8); }
        op += 3; len -= 60u;
    }
    if (len < 12u && off < 2048u) {
        if (lane == 0) { out[op] = (uint8_t)(1u | ((len - 4u) << 2) | ((off >> 8) << 5)); out[op + 1] = (uint8_t)off; }
        op += 2;
    } else {
        if (lane == 0) { out[op] = (uint8_t)(2u | ((len - 1u) << 2)); out[op + 1] = (uint8_t)off; out[op + 2] = (uint8_t)(off >> 8); }
        op += 3;
    }
    return op;
}

constexpr uint32_t kSnCoopLit = 256;     // literal runs / copies at least this long are emitted by the whole wavefront
constexpr uint32_t kSnCoopMatch = 1024;

__device__ __forceinline__ uint32_t snappy_literal_size(uint32_t lit) {     // header + data bytes of a literal element (0 for none)
    if (lit == 0u) return 0u;
    const uint32_t n1 = lit - 1u;
    return lit + (n1 < 60u ? 1u : n1 < 256u ? 2u : n1 < 65536u ? 3u : n1 < 16777216u ? 4u : 5u);
}

__device__ __forceinline__ uint32_t snappy_copy_size(uint32_t off, uint32_t len) {   // mirrors emit_snappy_copy's splitting
    const uint32_t n64 = len >= 68u ? (len - 4u) / 64u : 0u;
    uint32_t bytes = 3u * n64;
    len -= 64u * n64;
    if (len > 64u) { bytes += 3u; len -= 60u; }
    return bytes + ((len < 12u && off < 2048u) ? 2u : 3u);
}

// Lane-parallel emission: lane k < q_n writes literal + copy elements of sequence k at its precomputed output position
__device__ __forceinline__ void snappy_emit_queue(const uint8_t* in, uint8_t* out, uint32_t q_n, uint32_t lit0, uint32_t lit,
                                                  uint32_t off, uint32_t len, uint32_t qop) {
    if (lane_id() >= q_n) return;
    uint8_t* o = out + qop;
    if (lit) {
        const uint32_t n1 = lit - 1u;
        if (n1 < 60u) *o++ = (uint8_t)(n1 << 2);
        else {
            const uint32_t nb = n1 < 256u ? 1u : n1 < 65536u ? 2u : n1 < 16777216u ? 3u : 4u;
            *o++ = (uint8_t)((59u + nb) << 2);
            for (uint32_t k = 0; k < nb; k++) *o++ = (uint8_t)(n1 >> (8u * k));
        }
        lane_copy_exact(o, in + lit0, lit);
        o += lit;
    }
    while (len >= 68u) { o[0] = (uint8_t)(2u | (63u << 2)); o[1] = (uint8_t)off; o[2] = (uint8_t)(off >> 8); o += 3; len -= 64u; }
    if (len > 64u) { o[0] = (uint8_t)(2u | (59u << 2)); o[1] = (uint8_t)off; o[2] = (uint8_t)(off >> 8); o += 3; len -= 60u; }
    if (len < 12u && off < 2048u) { o[0] = (uint8_t)(1u | ((len - 4u) << 2) | ((off >> 8) << 5)); o[1] = (uint8_t)off; }
    else { o[0] = (uint8_t)(2u | ((len - 1u) << 2)); o[1] = (uint8_t)off; o[2] = (uint8_t)(off >> 8); }
}

// kSplit: four wavefronts per 64 KiB piece, each with its own pre-indexed hash table — see lz4_encode.hip
template <bool kSplit, bool kGlobalTable>
__device__ __forceinline__ void snappy_encode_chunk(const BatchArgs& a, uint32_t chunk, const HashTab<kGlobalTable>& ht) {
    const uint64_t base_off = kSplit ? a.in_off[chunk & ~(split_per(a.flags) - 1u)] : a.in_off[chunk];      // the piece's first sub-piece
    const uint8_t* in = a.in_base + base_off;               // position 0 = start of the piece
    const uint32_t q0 = (uint32_t)(a.in_off[chunk] - base_off);      // this wave's range = [q0, n)
    const uint64_t n64 = q0 + a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    const uint64_t cap64 = a.out_cap[chunk];
    const uint32_t lane = lane_id();

    // snap: TooBig above u32::MAX (we also keep positions in 32 bits); BufferTooSmall below max_compress_len
    if (n64 > 0xFFFFFFFFull - 64u) { if (lane == 0) a.result[chunk] = CJ_E_SNAPPY_TOO_BIG; return; }
    const uint64_t need = 32u + (n64 - q0) + (n64 - q0) / 6u;
    if (need > 0xFFFFFFFFull) { if (lane == 0) a.result[chunk] = CJ_E_SNAPPY_TOO_BIG; return; }
    if (cap64 < need) { if (lane == 0) a.result[chunk] = CJ_E_SNAPPY_BUF_SMALL; return; }
    const uint32_t n = (uint32_t)n64;

    uint32_t op = 0;
    {   // varint preamble
        uint32_t v = n - q0;
        while (v >= 0x80u) { if (lane == 0) out[op] = (uint8_t)(v | 0x80u); op += 1; v >>= 7; }
        if (lane == 0) out[op] = (uint8_t)v;
        op += 1;
    }
    uint32_t anchor = q0;
    if (n - q0 >= 8u) {
        ht.clear();
        if constexpr (kSplit) ht.preindex(in, q0);
        ht.settle();
        const uint32_t last_start = n - 4u;
        uint32_t pos = q0;
        OwnDwords own;
        own.load(in, pos, last_start);
        // A round probes its positions against the table as it was when the round began, so what repeats INSIDE a round is
        // only found through older entries.  With an empty table (the start of a chunk; a sub-piece's table is pre-indexed)
        // the first rounds are short — 64, 128, 256 positions — so that a small or very repetitive input does not lose a
        // whole round's worth of matches (1 KiB of text: ratio 2.2 -> 3.7; 64 KiB chunks of the benchmark data 1.625 -> 1.629).
        uint32_t span = q0 == 0u ? 64u : kRoundPositions;
        while (pos <= last_start) {
            Round r;
            if (own.pos != pos) own.load(in, pos, last_start);      // a match ran past the expected start of this round
            const uint32_t round_last = last_start - pos < span ? last_start : pos + span - 1u;     // last position this round probes
            probe_round(in, ht, pos, round_last, n, anchor, r, own);
            const uint32_t round_end = pos + span;
            span = span * 2u < kRoundPositions ? span * 2u : kRoundPositions;
            own.load(in, round_end, last_start);                  // next round's dwords: in flight during selection and emission
            bool covered[kSub] = {};
            uint32_t q_n = 0, q_lit0 = 0, q_lit = 0, q_off = 0, q_mlen = 0, q_op = 0;
            // fast path: minimal serial walk, sequence fields for all candidates at once, ds_permute push into the queue
            bool fast_round = false;
            {
                Selection sl;
                select_walk(r, pos, anchor, op, sl,
                            [](uint32_t lit, uint32_t mc, uint32_t off) { return snappy_literal_size(lit) + snappy_copy_size(off, mc + 4u); },
                            [](uint32_t lit, uint32_t mc) { return lit >= kSnCoopLit || mc + 4u >= kSnCoopMatch; });
                if (!sl.coop && sl.count <= 64u) {
                    fast_round = true;
                    uint32_t base = 0;
#pragma unroll
                    for (int j = 0; j < kSub; j++) {
                        const uint32_t p = pos + 64u * j + lane, ext = r.ext[j];
                        const bool sel = ((sl.sel[j] >> lane) & 1ull) != 0ull;
                        const uint32_t room = p - sl.prev_end[j];
                        uint32_t bk = (ext >> 8) & 0x3fu;
                        bk = bk < room ? bk : room;
                        const uint32_t cnt = (uint32_t)__builtin_popcountll(sl.sel[j]);
                        const uint32_t below = bits_below_lane(sl.sel[j]);
                        const uint32_t dest = sel ? base + below : (base + cnt + (lane - below)) & 63u;
                        q_lit0 = queue_push(q_lit0, sl.prev_end[j], dest, base, cnt);
                        q_lit = queue_push(q_lit, room - bk, dest, base, cnt);
                        q_off = queue_push(q_off, p - r.cand[j], dest, base, cnt);
                        q_mlen = queue_push(q_mlen, 4u + (ext & 0xffu) + bk, dest, base, cnt);
                        q_op = queue_push(q_op, sl.out_pos[j], dest, base, cnt);
                        base += cnt;
                        covered[j] = sl.covered[j];
                    }
                    q_n = base;
                    op = sl.op;
                    anchor = sl.anchor;
                }
            }
            if (!fast_round) {
#pragma unroll
                for (int j = 0; j < kSub; j++) covered[j] = false;      // a round starts at pos >= anchor: nothing of it is covered yet
            }
#pragma unroll
            for (int j = 0; j < kSub; j++) {
                if (fast_round) break;
                const uint32_t pj = pos + 64u * j;
                uint64_t mask = r.mask[j];
                if (anchor > pj) mask = anchor - pj >= 64u ? 0ull : mask & (~0ull << (anchor - pj));
                while (mask) {
                    const uint32_t first = ctz64(mask);
                    uint32_t mpos = pj + first;
                    uint32_t mc = rdlane(r.cand[j], first), mlen;
                    finish_match(in, r.ext[j], first, anchor, n, mpos, mc, mlen);
                    const uint32_t lit = mpos - anchor, off = mpos - mc;
                    if (lit >= kSnCoopLit || mlen >= kSnCoopMatch) {
                        snappy_emit_queue(in, out, q_n, q_lit0, q_lit, q_off, q_mlen, q_op);
                        q_n = 0;
                        if (lit) op = emit_snappy_literal(out, op, in + anchor, lit);
                        op = emit_snappy_copy(out, op, off, mlen);
                    } else {
                        if (q_n == 64u) { snappy_emit_queue(in, out, q_n, q_lit0, q_lit, q_off, q_mlen, q_op); q_n = 0; }     // a round of 320 positions can select up to 80 matches of 4 bytes: the queue is flushed when its 64 lanes are full
                        if (lane == q_n) { q_lit0 = anchor; q_lit = lit; q_off = off; q_mlen = mlen; q_op = op; }
                        q_n += 1;
                        op += snappy_literal_size(lit) + snappy_copy_size(off, mlen);
                    }
                    anchor = mpos + mlen;
#pragma unroll
                    for (int jj = 0; jj < kSub; jj++) {
                        const uint32_t my = pos + 64u * jj + lane;
                        covered[jj] = covered[jj] || (my > mpos && my < anchor);
                    }
                    if (anchor >= pj + 64u) mask = 0;
                    else mask &= ~0ull << (anchor - pj);
                }
            }
            snappy_emit_queue(in, out, q_n, q_lit0, q_lit, q_off, q_mlen, q_op);
            insert_round(ht, pos, r.hslot, covered);
            pos = anchor > round_end ? anchor : round_end;
        }
    }
    if (anchor < n) op = emit_snappy_literal(out, op, in + anchor, n - anchor);
    if (lane == 0) a.result[chunk] = (int64_t)op;
}

template <bool kSplit>
__global__ __launch_bounds__(kSplit ? 256 : kEncThreads) void snappy_encode_kernel(BatchArgs a) {
    __shared__ uint16_t ht_all[kSplit ? 4 : kEncWaves][kHashSize];
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t chunk = uni(kSplit ? blockIdx.x * 4u + wave : blockIdx.x * kEncWaves + wave);
    if (chunk >= a.n_chunks) return;
    snappy_encode_chunk<kSplit, false>(a, chunk, HashTab<false>{ht_all[wave]});
}

struct SnappyEnc {
    template <bool kGlobalTable>
    static __device__ __forceinline__ void chunk(const BatchArgs& a, uint32_t c, const HashTab<kGlobalTable>& ht) { snappy_encode_chunk<false, kGlobalTable>(a, c, ht); }
};

void launch_snappy_encode(const BatchArgs& a, hipStream_t s, const EncFill* fill) {
    if (a.n_chunks == 0) return;
    if (fill && !(a.flags & kFlagSplitPieces)) { launch_encode_filled<SnappyEnc>(a, s, *fill); return; }
    dim3 grid((a.n_chunks + kEncWaves - 1) / kEncWaves), block(kEncThreads);
    if (a.flags & kFlagSplitPieces) {
        hipLaunchKernelGGL(snappy_encode_kernel<true>, dim3((a.n_chunks + 3u) / 4u), dim3(256), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(snappy_encode_kernel<false>, grid, block, 0, s, a);
}

}  // namespace cj
