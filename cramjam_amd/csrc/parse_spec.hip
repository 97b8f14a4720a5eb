// parse_spec.hip — the parse stage for small and medium batches (LZ4 sequences and Snappy records): ONE WAVEFRONT PER CHUNK, 64 lanes parsing 64
// SEGMENTS of the same chunk at once.
//
// The token chain of an LZ4 block is serial: where sequence k+1 starts is only known after sequence k has been read,
// and the other two parse kernels walk it as such (a lane per chunk: 3.5 ms per 64 KiB chunk; a wave per chunk on the
// scalar unit: 0.9 ms).  But the "next sequence" function depends on the INPUT BYTES ONLY, and two walks that ever
// stand on the same position stay together for ever.  So:
//   1a  lane l starts at the guessed position l * SEG and walks to the end of its segment, marking every position it
//       visits in a bitmap (1 bit per input byte, atomic OR; only the owner marks inside a segment).  Lane 0 starts at the true
//       start; the others usually fall onto a real sequence boundary after a few hundred bytes.
//   1b  every lane keeps walking through the following segments until it steps on a marked position: from there on
//       its path IS the path of that segment's owner.  It records the merge position and the owner.
//   2   the true path is now a chain of path pieces: lane 0 from 0 to its merge position, the owner there up to its own
//       merge position, ... — at most 64 hops, followed with v_readlane.
//   3   the lanes on the chain re-walk their piece and count sequences and output bytes; a prefix sum over the lanes
//       gives every piece its first sequence index and output position.
//   4   they walk once more, now with the true (index, output position): the full LZ4_decompress_safe validation (same
//       rules as lz4_parse_kernel, which the stream-level verdict must equal) and the (ip, op) sync points.
// ≈240 per-lane steps instead of ≈2 700 serial ones (benchmark data: 1a up to 88 — a walk from a guessed position takes
// many tiny steps before it falls onto a real boundary —, 1b up to 31, 3 and 4 about 60 each): 0.14 ms per chunk instead of
// 0.9 ms (wave walk) / 3.5 ms (lane walk).  The chunk is staged in LDS first (64 KiB + 8 KiB bitmap per wavefront, two per
// CU), which bounds the throughput to ≈3.4 chunks/µs: the engine uses this kernel below 8 192 chunks, where it wins
// (1 chunk 0.18 ms, 512 chunks 0.25 ms, 4 096 chunks 1.6 ms; the lane walk takes 3.3 ms for any of these).
// Outputs are exactly those of the other parse kernels (result, ParseMeta, sync points), so the decode stage and every
// parity test are unchanged.
#include "lz4_lane_walk.hpp"
#include "parse_grammar.hpp"

namespace cj {

namespace {

constexpr uint32_t kSpecLdsIn = 65536;                   // staged compressed chunk (<= kLdsInMax + alignment slack)
constexpr uint32_t kSpecLdsBytes = kSpecLdsIn + 8192;    // + 1 bit per input byte

// 4 bytes at byte offset p of the staged chunk (LDS address a_in + p), any alignment; bytes past the staged data are
// whatever the LDS holds — callers never let such bytes decide anything (every length is bounds-checked against iend)
__device__ __forceinline__ uint32_t sp_ld32(uint32_t a_in, uint32_t p) {
    const uint32_t a = a_in + (p & ~3u);
    uint32_t w0, w1;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w0), "=&v"(w1) : "v"(a) : "memory");
    return __builtin_amdgcn_alignbyte(w1, w0, p & 3u);
}

__device__ __forceinline__ uint32_t wave_excl_scan_add(uint32_t v, uint32_t& total) {
    const uint32_t lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)x, d, 64);
        if (lane >= (uint32_t)d) x += t;
    }
    total = rdlane(x, 63);
    return x - v;
}

}  // namespace

// The segmented walk over the staged element stream in[0, iend) (global pointer `in`, any alignment).  cap = output
// capacity (LZ4) / decoded length (Snappy).  Returns the decoded size or -1 (malformed); nseq_out = number of sequences.
template <class G>
__device__ __forceinline__ int64_t spec_walk(const uint8_t* in, uint32_t iend, uint32_t cap, uint2* csync, uint8_t* smem, uint32_t& nseq_out) {
    const uint32_t lane = lane_id();
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(smem + kSpecLdsIn);
    const uint32_t a_bits = (uint32_t)(uintptr_t)s_bits;

    // ---- stage the chunk (16 B aligned loads), clear the bitmap ----
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);
    {
        const uint4* src = reinterpret_cast<const uint4*>(in - mis);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nvec = (mis + iend + 15u) >> 4;
        // 8 loads in flight per lane (one wave must not pay a round trip per KiB).  Named scalars and clamped indices —
        // lanes past the end repeat the last vector: with an array or a predicated access the compiler keeps the
        // values in scratch memory.
#define CJ_SP_LD(k) const uint32_t j##k = b0 + 64u * k##u + lane; const uint32_t x##k = j##k < nvec ? j##k : nvec - 1u; const uint4 v##k = src[x##k];
#define CJ_SP_ST(k) dst[x##k] = v##k;
        for (uint32_t b0 = 0; b0 < nvec; b0 += 64u * 8u) {
            CJ_SP_LD(0) CJ_SP_LD(1) CJ_SP_LD(2) CJ_SP_LD(3) CJ_SP_LD(4) CJ_SP_LD(5) CJ_SP_LD(6) CJ_SP_LD(7)
            CJ_SP_ST(0) CJ_SP_ST(1) CJ_SP_ST(2) CJ_SP_ST(3) CJ_SP_ST(4) CJ_SP_ST(5) CJ_SP_ST(6) CJ_SP_ST(7)
        }
#undef CJ_SP_LD
#undef CJ_SP_ST
        for (uint32_t i = lane; i < 2048u; i += 64u) s_bits[i] = 0u;
    }
    __syncthreads();
    const uint32_t a_in = (uint32_t)(uintptr_t)smem + mis;
    const auto rd = [a_in](uint32_t q) { return sp_ld32(a_in, q); };

    // ---- segments: SEG = 4 x (odd number) bytes, so that the lanes' start positions fall into 64 different LDS banks
    //      (a stride of 32 k bytes would put all lanes on at most 8 banks) ----
    uint32_t nl = (iend + 255u) / 256u;
    nl = nl > 64u ? 64u : nl;
    const uint32_t seg = (((iend + nl - 1u) / nl + 3u) & ~3u) | 4u;
    const bool active = lane < nl && lane * seg < iend;
    const uint32_t seg_end = (lane + 1u) * seg;

    // ---- 1a: own segment, marking ----
    uint32_t p = active ? lane * seg : kPosEnd;          // position, or kPosEnd / kPosErr once the walk is over
    while (ballot64(p < seg_end && p < iend) != 0ull) {
        if (p < seg_end && p < iend) {
            asm volatile("ds_or_b32 %0, %1" :: "v"(a_bits + 4u * (p >> 5)), "v"(1u << (p & 31u)) : "memory");
            Seq s;
            p = G::at(rd, p, iend, s) ? s.next : kPosErr;
        }
    }
    __syncthreads();                                      // all marks are in place

    // ---- 1b: walk on through the following segments until the path joins an owner's path ----
    uint32_t merge_pos = p;                               // kPosEnd / kPosErr when the walk ended without joining
    {
        bool going = active && p < iend;
        while (ballot64(going) != 0ull) {
            if (going) {
                uint32_t w;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a_bits + 4u * (p >> 5)) : "memory");
                if ((w >> (p & 31u)) & 1u) { merge_pos = p; going = false; }
                else {
                    Seq s;
                    p = G::at(rd, p, iend, s) ? s.next : kPosErr;
                    if (p >= iend) { merge_pos = p; going = false; }      // kPosEnd, kPosErr (or a position past the input: malformed)
                }
            }
        }
        if (active && merge_pos >= iend && merge_pos != kPosEnd) merge_pos = kPosErr;
    }

    // ---- 2: the true path = chain of pieces, starting with lane 0 at position 0 ----
    uint32_t entry = 0;                                   // per lane: where the true path enters this lane's piece
    uint64_t chain = 0ull;
    {
        uint32_t cur = 0;
        for (uint32_t hop = 0; hop < 64u; hop++) {
            chain |= 1ull << cur;
            const uint32_t m = rdlane(merge_pos, cur);
            if (m >= iend) break;                         // kPosEnd / kPosErr: the last piece
            const uint32_t nxt = m / seg;
            entry = lane == nxt ? m : entry;
            cur = nxt;
        }
    }
    const bool on_chain = ((chain >> lane) & 1ull) != 0ull;
    const uint32_t piece_end = merge_pos;                 // exclusive end of this lane's piece (or kPosEnd / kPosErr)

    // ---- 3: count sequences and output bytes of every piece ----
    uint32_t cnt = 0, outb = 0;
    {
        uint32_t q = on_chain ? entry : kPosEnd;
        while (ballot64(q < iend && q != piece_end) != 0ull) {
            if (q < iend && q != piece_end) {
                Seq s;
                if (G::at(rd, q, iend, s)) { cnt += 1; outb += s.lit + s.mlen; q = s.next; }
                else q = kPosErr;
            }
        }
    }
    uint32_t total_seq, total_out;
    const uint32_t base_idx = wave_excl_scan_add(cnt, total_seq);
    const uint32_t base_op = wave_excl_scan_add(outb, total_out);

    // ---- 4: walk the pieces with the true sequence index and output position: validation + sync points ----
    bool bad = false;
    uint32_t final_op = 0;                                // set by the lane that meets the last sequence
    bool saw_last = false;
    {
        uint32_t q = on_chain ? entry : kPosEnd, idx = base_idx, op = base_op;
        while (ballot64(q < iend && q != piece_end && !bad) != 0ull) {
            if (q < iend && q != piece_end && !bad) {
                if ((idx % kSyncEvery) == 0u) {
                    const uint32_t slot = idx / kSyncEvery;
                    if (slot < kSyncStride) csync[slot] = make_uint2(q, op);
                }
                Seq s;
                bool fin = false;
                if (!G::at(rd, q, iend, s) || !G::check(s, op, cap, fin)) bad = true;
                else if (fin) { final_op = op; saw_last = true; q = kPosEnd; }
                else { q = s.next; idx += 1; }
            }
        }
    }
    // the chain must end in a last sequence and nothing on it may be malformed
    const bool any_bad = ballot64(on_chain && (bad || piece_end == kPosErr)) != 0ull;
    const uint64_t last_lanes = ballot64(on_chain && saw_last);
    nseq_out = total_seq;
    if (any_bad || last_lanes == 0ull) return -1;
    const uint32_t op_end = rdlane(final_op, ctz64(last_lanes));
    if (!G::result_ok(op_end, cap)) return -1;
    return (int64_t)op_end;
}

__global__ __launch_bounds__(64) void lz4_parse_spec_kernel(BatchArgs a, uint2* sync, ParseMeta* meta) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t c = blockIdx.x;
    if (c >= a.n_chunks) return;
    if ((meta[c].in_skip & kRouteLane) != 0u) return;
    const uint32_t lane = lane_id();
    const uint8_t* in = a.in_base + a.in_off[c];
    const uint8_t* const in0 = in;
    uint64_t n64 = a.in_len[c], cap64 = a.out_cap[c];
    ParseMeta pm = {0u, 0u};
    int64_t r = lz4_block_prologue(a.flags, in, n64, cap64);
    bool walk = false;
    if (r == 0) {
        const uint32_t cap0 = (uint32_t)cap64, iend0 = (uint32_t)n64;
        if (cap0 == 0) r = (iend0 == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT;
        else if (iend0 == 0) r = CJ_E_CORRUPT;
        else if (cap0 > kLdsOutMax || iend0 > kLdsInMax) pm.in_skip = kRouteWave;      // too big for the LDS window
        else walk = true;
    }
    if (walk) {
        uint32_t nseq = 0;
        r = spec_walk<Lz4Grammar>(in, (uint32_t)n64, (uint32_t)cap64, sync + (size_t)c * kSyncPitch, smem, nseq);
        if (r < 0) r = CJ_E_CORRUPT;
        else if (r > 0) {
            if ((nseq + kSyncEvery - 1u) / kSyncEvery > kSyncStride || nseq < kLdsMinSeq) pm.in_skip = kRouteWave;
            else { pm.nseq = nseq; pm.in_skip = (uint32_t)(in - in0); }
        }
    }
    if (lane == 0) { a.result[c] = r; meta[c] = pm; }
}

__global__ __launch_bounds__(64) void snappy_parse_spec_kernel(BatchArgs a, uint2* sync, ParseMeta* meta) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t c = blockIdx.x;
    if (c >= a.n_chunks) return;
    if ((meta[c].in_skip & kRouteLane) != 0u) return;
    const uint32_t lane = lane_id();
    const uint8_t* in = a.in_base + a.in_off[c];
    const uint64_t n64 = a.in_len[c], cap64 = a.out_cap[c];
    ParseMeta pm = {0u, 0u};
    int64_t r = 0;
    uint32_t dn = 0, hdr = 0;
    bool walk = false;
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
        else if (ulen == 0) r = (hdr == (uint32_t)n64) ? 0 : (int64_t)CJ_E_SNAPPY_CORRUPT;
        else if (ulen > kLdsOutMax || n64 - hdr > kLdsInMax) pm.in_skip = kRouteWave;
        else if (hdr == (uint32_t)n64) r = CJ_E_SNAPPY_CORRUPT;                 // a length but no elements
        else { dn = (uint32_t)ulen; walk = true; }
    }
    if (walk) {
        uint32_t nrec = 0;
        r = spec_walk<SnappyGrammar>(in + hdr, (uint32_t)n64 - hdr, dn, sync + (size_t)c * kSyncPitch, smem, nrec);
        if (r < 0) r = CJ_E_SNAPPY_CORRUPT;
        else {
            if ((nrec + kSyncEvery - 1u) / kSyncEvery > kSyncStride || nrec < kLdsMinSeq) pm.in_skip = kRouteWave;
            else { pm.nseq = nrec; pm.in_skip = hdr; }
        }
    }
    if (lane == 0) { a.result[c] = r; meta[c] = pm; }
}

void launch_lz4_parse_spec(const BatchArgs& a, void* sync, void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_parse_spec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSpecLdsBytes);
    hipLaunchKernelGGL(lz4_parse_spec_kernel, dim3(a.n_chunks), dim3(64), kSpecLdsBytes, s, a, (uint2*)sync, (ParseMeta*)meta);
}


void launch_snappy_parse_spec(const BatchArgs& a, void* sync, void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(snappy_parse_spec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSpecLdsBytes);
    hipLaunchKernelGGL(snappy_parse_spec_kernel, dim3(a.n_chunks), dim3(64), kSpecLdsBytes, s, a, (uint2*)sync, (ParseMeta*)meta);
}

}  // namespace cj
