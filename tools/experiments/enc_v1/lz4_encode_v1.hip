// lz4_encode.hip — LZ4 *block* encoder for gfx950, one wavefront per independent chunk.
//
// Replaces (on the GPU) what the reference reaches at /root/reference/src/lz4.rs:127,206:
// libcramjam::lz4::block::compress_into -> lz4 crate compress_to_buffer -> LZ4_compress_default,
// including the optional u32-LE uncompressed-size prefix (`store_size`, default on).
// Stream rules honoured (lz4_Block_format.md): min match 4, offset 1..65535, the last match starts
// at least 12 bytes before the end, the last 5 bytes are literals, inputs < 13 bytes are one literal
// run.  Every emitted block decodes with LZ4_decompress_safe at exact capacity (tests check this
// against the CPU oracle).
#include "cj_match.hpp"

namespace cj {

// writes the 255-run length extension for value v (v = len - 15): v/255 bytes of 255 then v%255
__device__ __forceinline__ uint32_t emit_len_ext(uint8_t* out, uint32_t op, uint32_t v) {
    const uint32_t full = v / 255u, rem = v - full * 255u, lane = lane_id();
    for (uint32_t k = 0; k < full; k += 64u)
        if (k + lane < full) out[op + k + lane] = 255u;
    if (lane == 0) out[op + full] = (uint8_t)rem;
    return op + full + 1u;
}

constexpr uint32_t kCoopLit = 256;       // literal runs / match codes at least this long are emitted by the whole wavefront
constexpr uint32_t kCoopMatch = 2048;

// Lane-parallel emission: lane k < q_n writes sequence k of the round (token, length bytes, literals, offset) at its
// precomputed output position.  All literal sources are loaded in one round trip instead of one per sequence.
__device__ __forceinline__ void lz4_emit_queue(const uint8_t* in, uint8_t* out, uint32_t q_n, uint32_t lit0, uint32_t lit,
                                               uint32_t off, uint32_t mcode, uint32_t qop) {
    if (lane_id() >= q_n) return;
    uint8_t* o = out + qop;
    *o++ = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
    if (lit >= 15u) {
        uint32_t v = lit - 15u;
        while (v >= 255u) { *o++ = 255u; v -= 255u; }
        *o++ = (uint8_t)v;
    }
    lane_copy_exact(o, in + lit0, lit);
    o += lit;
    o[0] = (uint8_t)off; o[1] = (uint8_t)(off >> 8);
    o += 2;
    if (mcode >= 15u) {
        uint32_t v = mcode - 15u;
        while (v >= 255u) { *o++ = 255u; v -= 255u; }
        *o++ = (uint8_t)v;
    }
}

// kSplit (large.hip, buffers of up to 32 MiB, where one wavefront per 64 KiB piece would leave most of the GPU idle and
// every call would take the 1.7 ms one wavefront needs for 64 KiB): every 64 KiB piece is cut into split_per(flags) = 16 or 4
// consecutive sub-pieces (4 KiB / 16 KiB), one wavefront each — the chunks of the batch ARE the sub-pieces, position 0 of a
// walk = the start of its piece.  Each wavefront has its own hash table and first indexes the data BEFORE its sub-piece
// (ht_preindex: a few µs), so it finds what a serial walk over the piece would (ratio 1.60 resp. 1.62 vs 1.63 on the benchmark
// data, 4.78 resp. 4.79 vs 4.88 on text); each writes its own stream, and the streams are stitched / concatenated like any
// other pieces.  64 KiB .. 4 MiB per call: 1.8-1.9 ms on one wavefront per piece, 0.6-0.7 ms with quarters, 0.30-0.44 ms with
// 4 KiB sub-pieces.  (Blocks of four wavefronts: four tables fill the 64 KiB of static LDS.)
template <bool kSplit, bool kGlobalTable>
__device__ __forceinline__ void lz4_encode_chunk(const BatchArgs& a, uint32_t chunk, const HashTab<kGlobalTable>& ht) {
    const uint64_t base_off = kSplit ? a.in_off[chunk & ~(split_per(a.flags) - 1u)] : a.in_off[chunk];      // the piece's first sub-piece
    const uint8_t* in = a.in_base + base_off;               // position 0 = start of the piece
    const uint32_t q0 = (uint32_t)(a.in_off[chunk] - base_off);      // this wave's range = [q0, n)
    const uint64_t n64 = q0 + a.in_len[chunk];
    uint8_t* out = a.out_base + a.out_off[chunk];
    uint64_t cap64 = a.out_cap[chunk];
    const uint32_t lane = lane_id();
    const bool prefix = (a.flags & CJ_FLAG_LZ4_SIZE_PREFIX) != 0;

    if (n64 > 0x7E000000ull) { if (lane == 0) a.result[chunk] = CJ_E_INPUT_TOO_LARGE; return; }
    const uint32_t n = (uint32_t)n64;
    // the engine only launches with capacity >= LZ4_compressBound(n) (+4); anything smaller is refused here
    const uint64_t need = (uint64_t)(n - q0) + (n - q0) / 255u + 16u + (prefix ? 4u : 0u);
    if (cap64 < need) { if (lane == 0) a.result[chunk] = CJ_E_COMPRESS_FAILED; return; }
    if (prefix) {
        if (lane < 4) out[lane] = (uint8_t)(n >> (8u * lane));
        out += 4;
    }

    uint32_t anchor = q0, op = 0;
    if (n - q0 >= 13u) {
        ht.clear();
        if constexpr (kSplit) ht.preindex(in, q0);
        ht.settle();
        const uint32_t last_start = n - 12u;    // a match may start here at the latest
        const uint32_t matchlimit = n - 5u;     // and must end here at the latest
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
            probe_round(in, ht, pos, round_last, matchlimit, anchor, r, own);
            const uint32_t round_end = pos + span;
            span = span * 2u < kRoundPositions ? span * 2u : kRoundPositions;
            own.load(in, round_end, last_start);                  // next round's dwords: in flight during selection and emission
            bool covered[kSub] = {};
            uint32_t q_n = 0, q_lit0 = 0, q_lit = 0, q_off = 0, q_mcode = 0, q_op = 0;
            // ---- fast path: minimal serial walk, then everything else for all candidates at once (cj_match.hpp) ----
            bool fast_round = false;
            {
                Selection sl;
                select_walk(r, pos, anchor, op, sl,
                            [](uint32_t lit, uint32_t mc, uint32_t) { return 1u + (lit >= 15u ? 1u + (lit - 15u) / 255u : 0u) + lit + 2u + (mc >= 15u ? 1u + (mc - 15u) / 255u : 0u); },
                            [](uint32_t lit, uint32_t mc) { return lit >= kCoopLit || mc >= kCoopMatch; });
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
                        // selected lanes -> queue lanes base + rank; the rest -> the other lanes, in order
                        const uint32_t dest = sel ? base + below : (base + cnt + (lane - below)) & 63u;
                        q_lit0 = queue_push(q_lit0, sl.prev_end[j], dest, base, cnt);
                        q_lit = queue_push(q_lit, room - bk, dest, base, cnt);
                        q_off = queue_push(q_off, p - r.cand[j], dest, base, cnt);
                        q_mcode = queue_push(q_mcode, (ext & 0xffu) + bk, dest, base, cnt);
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
                    finish_match(in, r.ext[j], first, anchor, matchlimit, mpos, mc, mlen);
                    const uint32_t lit = mpos - anchor, mcode = mlen - 4u, off = mpos - mc;
                    if (lit >= kCoopLit || mcode >= kCoopMatch) {
                        // long literal run / very long match: whole-wave emission right away (after the queue, to keep op order)
                        lz4_emit_queue(in, out, q_n, q_lit0, q_lit, q_off, q_mcode, q_op);
                        q_n = 0;
                        if (lane == 0) out[op] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
                        op += 1;
                        if (lit >= 15u) op = emit_len_ext(out, op, lit - 15u);
                        wave_copy(out + op, in + anchor, lit);
                        op += lit;
                        if (lane < 2) out[op + lane] = (uint8_t)(off >> (8u * lane));
                        op += 2;
                        if (mcode >= 15u) op = emit_len_ext(out, op, mcode - 15u);
                    } else {
                        if (q_n == 64u) { lz4_emit_queue(in, out, q_n, q_lit0, q_lit, q_off, q_mcode, q_op); q_n = 0; }     // a round of 320 positions can select up to 80 matches of 4 bytes: the queue is flushed when its 64 lanes are full
                        if (lane == q_n) { q_lit0 = anchor; q_lit = lit; q_off = off; q_mcode = mcode; q_op = op; }
                        q_n += 1;
                        op += 1u + (lit >= 15u ? 1u + (lit - 15u) / 255u : 0u) + lit + 2u + (mcode >= 15u ? 1u + (mcode - 15u) / 255u : 0u);
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
            lz4_emit_queue(in, out, q_n, q_lit0, q_lit, q_off, q_mcode, q_op);
            insert_round(ht, pos, r.hslot, covered);
            pos = anchor > round_end ? anchor : round_end;
        }
    }
    uint64_t tail_report = 0;
    {   // last literals
        const uint32_t lit = n - anchor;
        if (a.flags & kFlagReportTail) tail_report = (uint64_t)lit << 32;
        if (lane == 0) out[op] = (uint8_t)((lit < 15u ? lit : 15u) << 4);
        op += 1;
        if (lit >= 15u) op = emit_len_ext(out, op, lit - 15u);
        wave_copy(out + op, in + anchor, lit);
        op += lit;
    }
    if (lane == 0) a.result[chunk] = (int64_t)(((uint64_t)op + (prefix ? 4u : 0u)) | tail_report);
}

template <bool kSplit>
__global__ __launch_bounds__(kSplit ? 256 : kEncThreads) void lz4_encode_kernel(BatchArgs a) {
    __shared__ uint16_t ht_all[kSplit ? 4 : kEncWaves][kHashSize];
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t chunk = uni(kSplit ? blockIdx.x * 4u + wave : blockIdx.x * kEncWaves + wave);
    if (chunk >= a.n_chunks) return;
    lz4_encode_chunk<kSplit, false>(a, chunk, HashTab<false>{ht_all[wave]});
}

struct Lz4Enc {
    template <bool kGlobalTable>
    static __device__ __forceinline__ void chunk(const BatchArgs& a, uint32_t c, const HashTab<kGlobalTable>& ht) { lz4_encode_chunk<false, kGlobalTable>(a, c, ht); }
};

void launch_lz4_encode(const BatchArgs& a, hipStream_t s, const EncFill* fill) {
    if (a.n_chunks == 0) return;
    if (fill && !(a.flags & kFlagSplitPieces)) { launch_encode_filled<Lz4Enc>(a, s, *fill); return; }
    dim3 grid((a.n_chunks + kEncWaves - 1) / kEncWaves), block(kEncThreads);
    if (a.flags & kFlagSplitPieces) {
        hipLaunchKernelGGL(lz4_encode_kernel<true>, dim3((a.n_chunks + 3u) / 4u), dim3(256), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(lz4_encode_kernel<false>, grid, block, 0, s, a);
}

}  // namespace cj
