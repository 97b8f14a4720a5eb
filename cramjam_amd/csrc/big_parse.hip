// big_parse.hip — the parse stage for ONE LARGE stream (large.hip: decompress_block / decompress_raw on a single buffer
// of megabytes; reference call sites /root/reference/src/lz4.rs:143 decompress_block, /root/reference/src/snappy.rs:106
// decompress_raw).  The element chain of such a stream is serial; parse_spec.hip breaks that for one 64 KiB chunk
// inside one wavefront, this file does it for a stream of any length across the whole GPU:
//   K1  mark     one wavefront per PIECE of the input (16 KiB; 64 KiB for streams from 4 MiB, fewer serial steps in K2): 64 lanes walk 64 sub-segments from guessed positions,
//                marking what they visit (1a), then walk on until they join a later lane's path or leave the piece (1b).
//                Output: the piece's bitmap, and per lane where its path joins (merge) and where it leaves the piece (exit).
//   K1c          for each of the first 64 bytes of the piece (the true entry almost always lies there): where a walk from
//                that byte joins the marked paths, i.e. the entry of the NEXT piece for a true entry at that byte.
//   K2  thread   one wavefront threads the pieces in order through K1c's tables (64 pieces per round, tables in LDS, a
//                chain of dependent LDS reads); an entry beyond byte 64 walks the stream itself.  The only serial part.
//   K3  count    per piece, with the true entry known: the true path is stitched from the lanes' pieces and every lane
//                counts its sequences and output bytes.
//   K4  scan     exclusive prefix over the pieces (sequence index, output position).
//   K5  emit     per piece: the lanes walk their part again with the true (index, output position): validation with the
//                decoder's own rules and an ABSOLUTE sync point (ip, op) for every 8th sequence.
//   K6  slabs    one thread per 64 KiB slab of OUTPUT: which sync points / input bytes the slab's decoder needs.
// Results are those of a serial walk (tests/test_large_gpu.py checks the verdict, the size and the decoded bytes).
#include "lz4_lane_walk.hpp"
#include "parse_grammar.hpp"
#include "big_parse.hpp"

namespace cj {

namespace {

constexpr uint32_t kBpSlack = 1024;                       // staged bytes past the piece's end (sequences that cross it)
__host__ __device__ inline uint32_t bp_lds_in(uint32_t piece) { return piece + kBpSlack + 32u; }              // staged window
__host__ __device__ inline uint32_t bp_lds_all(uint32_t piece) { return bp_lds_in(piece) + piece / 8u + 64u * 4u; } // + bitmap + links

// the stream through a staged window: positions [lo, hi) come from LDS, anything else from global memory
struct WinReader {
    uint32_t a_win;          // LDS address of stream position lo
    uint32_t lo, hi;         // staged positions (hi exclusive; reads take up to 8 bytes from an aligned address)
    const uint8_t* g;        // global pointer to stream position 0
    __device__ __forceinline__ uint32_t operator()(uint32_t p) const {
        if (p >= lo && p + 8u <= hi) {
            const uint32_t a = a_win + (p - lo);
            const uint32_t al = a & ~3u;
            uint32_t w0, w1;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(w0), "=&v"(w1) : "v"(al) : "memory");
            return __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
        }
        return ld32u(g + p);                                // the input buffer is padded by 16 bytes
    }
};

struct GlobalReader {
    const uint8_t* g;
    __device__ __forceinline__ uint32_t operator()(uint32_t p) const { return ld32u(g + p); }
};

// stage stream positions [lo, hi) into smem (16 B aligned loads); returns the LDS address of position lo
__device__ __forceinline__ uint32_t bp_stage(const uint8_t* g, uint32_t lo, uint32_t hi, uint8_t* smem, uint32_t lane) {
    const uint8_t* src0 = g + lo;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src0) & 15u);
    const uint4* src = reinterpret_cast<const uint4*>(src0 - mis);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint32_t nvec = (mis + (hi - lo) + 15u) >> 4;
#define CJ_BP_LD(k) const uint32_t j##k = b0 + 64u * k##u + lane; const uint32_t x##k = j##k < nvec ? j##k : nvec - 1u; const uint4 v##k = src[x##k];
#define CJ_BP_ST(k) dst[x##k] = v##k;
    for (uint32_t b0 = 0; b0 < nvec; b0 += 64u * 8u) {
        CJ_BP_LD(0) CJ_BP_LD(1) CJ_BP_LD(2) CJ_BP_LD(3) CJ_BP_LD(4) CJ_BP_LD(5) CJ_BP_LD(6) CJ_BP_LD(7)
        CJ_BP_ST(0) CJ_BP_ST(1) CJ_BP_ST(2) CJ_BP_ST(3) CJ_BP_ST(4) CJ_BP_ST(5) CJ_BP_ST(6) CJ_BP_ST(7)
    }
#undef CJ_BP_LD
#undef CJ_BP_ST
    return (uint32_t)(uintptr_t)smem + mis;
}

__device__ __forceinline__ uint32_t bp_scan32(uint32_t v, uint32_t& total) {
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
__device__ __forceinline__ uint64_t bp_scan64(uint64_t v, uint64_t& total) {
    const uint32_t lane = lane_id();
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
        if (lane >= (uint32_t)d) x += ((uint64_t)hi << 32) | lo;
    }
    total = ((uint64_t)rdlane((uint32_t)(x >> 32), 63) << 32) | rdlane((uint32_t)x, 63);
    return x - v;
}

// ---------------------------------------------------------------------------------------------------------------------
// K1
template <class G>
__global__ __launch_bounds__(64) void big_mark_kernel(BigParse a0) {
    BigParse a = a0;
    uint32_t piece_index = blockIdx.x;
    if (a0.jobs != nullptr) { const uint2 pm = a0.piece_map[blockIdx.x]; a = a0.jobs[pm.x]; piece_index = pm.y; }
    const uint32_t P = a.piece, sub = P / 64u;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t p = piece_index, lane = lane_id();
    const uint32_t B = a.start + p * P;
    const uint32_t E = a.iend - B > P ? B + P : a.iend;         // piece = [B, E)
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(smem + bp_lds_in(P));
    uint32_t* s_link = s_bits + P / 32u;
    const uint32_t a_bits = (uint32_t)(uintptr_t)s_bits;
    const uint32_t whi = a.iend - B > P + kBpSlack ? B + P + kBpSlack : a.iend;
    WinReader rd;
    rd.a_win = bp_stage(a.in, B, whi, smem, lane);
    rd.lo = B; rd.hi = whi; rd.g = a.in;
    for (uint32_t i = lane; i < P / 32u; i += 64u) s_bits[i] = 0u;
    __syncthreads();

    // 1a: own sub-segment, marking
    const uint32_t s0 = B + lane * sub;
    const uint32_t s1 = s0 + sub < E ? s0 + sub : E;
    uint32_t pos = s0 < E ? s0 : kPosEnd;
    while (ballot64(pos < s1) != 0ull) {
        if (pos < s1) {
            const uint32_t r = pos - B;
            asm volatile("ds_or_b32 %0, %1" :: "v"(a_bits + 4u * (r >> 5)), "v"(1u << (r & 31u)) : "memory");
            Seq s;
            pos = walk_step<G>(rd, pos, a.iend, s, a.in) ? s.next : kPosErr;
        }
    }
    __syncthreads();
    // 1b: walk on until the path joins an owner's path or leaves the piece
    {
        bool going = pos < E;
        while (ballot64(going) != 0ull) {
            if (going) {
                const uint32_t r = pos - B;
                uint32_t w;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a_bits + 4u * (r >> 5)) : "memory");
                if ((w >> (r & 31u)) & 1u) going = false;
                else {
                    Seq s;
                    pos = walk_step<G>(rd, pos, a.iend, s, a.in) ? s.next : kPosErr;
                    if (pos >= E) going = false;
                }
            }
        }
    }
    const uint32_t merge = s0 < E ? pos : kPosErr;       // < E: joins the owner of that position; >= E: leaves the piece / kPosEnd / kPosErr
    // exit of every lane's path: follow the joins (a join always points to a LATER lane: 6 doubling rounds cover 64 lanes)
    uint32_t ex = merge;
    for (int round = 0; round < 6; round++) {
        s_link[lane] = ex;
        __syncthreads();
        if (ex < E) ex = s_link[(ex - B) / sub];
        __syncthreads();
    }
    a.merge[(size_t)p * 64u + lane] = merge;
    a.exitp[(size_t)p * 64u + lane] = ex;
    // 1c: the true entry of a piece almost always lies in its first 64 bytes (the overhang of the sequence that crosses the
    //     piece's start).  For each of them: where a walk from there joins the marked paths, and with it the entry of the
    //     NEXT piece — K2 then only chases these tables (next[p][e - B]) instead of walking the stream itself.
    s_link[lane] = ex;
    __syncthreads();
    {
        const uint32_t merge0 = rdlane(merge, 0);
        uint32_t q = B + lane < E ? B + lane : kPosErr;
        bool going = q < E;
        while (ballot64(going) != 0ull) {
            if (going) {
                const uint32_t r = q - B;
                uint32_t w;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a_bits + 4u * (r >> 5)) : "memory");
                if ((w >> (r & 31u)) & 1u) going = false;
                else {
                    Seq s;
                    q = walk_step<G>(rd, q, a.iend, s, a.in) ? s.next : kPosErr;
                    if (q >= E) going = false;
                }
            }
        }
        uint32_t fe = q, nxt = q;
        if (q < E) {
            const uint32_t o = (q - B) / sub;                 // the entry itself is in lane 0's sub-segment (sub >= 256)
            fe = o == 0u ? merge0 : q;
            nxt = s_link[o];
        }
        a.next_tab[(size_t)p * 64u + lane] = nxt;
        a.fe_tab[(size_t)p * 64u + lane] = fe;
    }
    uint32_t* gb = a.bits + (size_t)p * (P / 32u);
    for (uint32_t i = lane; i < P / 32u; i += 64u) gb[i] = s_bits[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// K2: the true entry of every piece.  entry[p] = (entry position | kPosEnd = the chain does not touch this piece,
// end of the entry lane's part of the chain).  One wavefront, 64 pieces per round: the lanes fetch the pieces' next-entry
// tables (K1, phase 1c) into LDS, lane 0's chain of 64 dependent LDS reads threads the entries through them, then the lanes
// write the entries out in parallel.  An entry beyond the first 64 bytes of its piece (a sequence longer than that crosses
// the piece's start) takes the walk through global memory, as does a piece the chain jumps over.
template <class G>
__global__ __launch_bounds__(64) void big_thread_kernel(BigParse a0) {
    const BigParse a = a0.jobs != nullptr ? a0.jobs[blockIdx.x] : a0;
    __shared__ uint32_t s_next[64 * 64];
    __shared__ uint32_t s_ent[64], s_fe[64];
    const uint32_t P = a.piece, sub = P / 64u, lane = lane_id();
    GlobalReader rd = {a.in};
    uint32_t e = a.start;                                   // uniform
    for (uint32_t p0 = 0; p0 < a.np; p0 += 64u) {
        const uint32_t cnt = a.np - p0 < 64u ? a.np - p0 : 64u;
        {   // 64 tables of 64 entries, coalesced: 16 B per lane and step
            const uint4* src = reinterpret_cast<const uint4*>(a.next_tab + (size_t)p0 * 64u);
            uint4* dst = reinterpret_cast<uint4*>(s_next);
            for (uint32_t i = lane; i < cnt * 16u; i += 64u) dst[i] = src[i];
        }
        __syncthreads();
        for (uint32_t i = 0; i < cnt; i++) {                // uniform control flow; the chain runs through LDS
            const uint32_t p = p0 + i;
            const uint32_t B = a.start + p * P;
            const uint32_t E = a.iend - B > P ? B + P : a.iend;
            uint32_t ent = kPosEnd, fe = kPosErr;           // fe = kPosErr: "take it from K1's table" (phase 3 below)
            if (e < E) {
                ent = e;
                if (e - B < 64u) e = s_next[i * 64u + (e - B)];
                else {                                       // rare: walk the stream itself
                    const uint32_t* gb = a.bits + (size_t)p * (P / 32u);
                    const uint32_t lx = (e - B) / sub;
                    uint32_t q = e;
                    while (q < E) {
                        const uint32_t r = q - B;
                        if ((gb[r >> 5] >> (r & 31u)) & 1u) break;
                        Seq s;
                        q = walk_step<G>(rd, q, a.iend, s, a.in) ? s.next : kPosErr;
                    }
                    if (q < E) {
                        const uint32_t o = (q - B) / sub;
                        fe = o == lx ? a.merge[(size_t)p * 64u + lx] : q;
                        e = a.exitp[(size_t)p * 64u + o];
                    } else { fe = q; e = q; }
                    if (fe == kPosErr) fe = kPosErr - 2u;    // (keep the marker value free; any value >= E means the same to K3 / K5)
                }
            }
            if (lane == 0) { s_ent[i] = ent; s_fe[i] = fe; }
        }
        __syncthreads();
        if (lane < cnt) {
            const uint32_t p = p0 + lane, B = a.start + p * P;
            uint32_t ent = s_ent[lane], fe = s_fe[lane];
            if (ent != kPosEnd && fe == kPosErr) fe = a.fe_tab[(size_t)p * 64u + (ent - B)];
            a.entry[p] = make_uint2(ent, ent == kPosEnd ? kPosEnd : fe);
        }
        __syncthreads();
    }
    if (lane == 0) a.status[0] = e == kPosEnd ? 0u : 1u;    // the chain must end with a last sequence exactly at the end of the input
}

// the chain inside one piece: per lane (entry, end); end < E = the next lane's entry, else the piece's exit / kPosEnd / kPosErr
__device__ __forceinline__ void bp_chain(const BigParse& a, uint32_t p, uint32_t B, uint32_t E, uint32_t lane, uint32_t& entry, uint32_t& end, bool& on) {
    const uint32_t sub = a.piece / 64u;
    const uint2 en = a.entry[p];
    const uint32_t merge = a.merge[(size_t)p * 64u + lane];
    entry = kPosEnd; end = kPosEnd; on = false;
    if (en.x == kPosEnd) return;
    uint32_t cur = (en.x - B) / sub, ent = en.x, fin = en.y;
    for (uint32_t hop = 0; hop < 64u; hop++) {
        if (lane == cur) { entry = ent; end = fin; on = true; }
        if (fin >= E) break;
        cur = (fin - B) / sub;
        ent = fin;
        fin = rdlane(merge, cur);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K3
template <class G>
__global__ __launch_bounds__(64) void big_count_kernel(BigParse a0) {
    BigParse a = a0;
    uint32_t piece_index = blockIdx.x;
    if (a0.jobs != nullptr) { const uint2 pm = a0.piece_map[blockIdx.x]; a = a0.jobs[pm.x]; piece_index = pm.y; }
    const uint32_t P = a.piece;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t p = piece_index, lane = lane_id();
    const uint32_t B = a.start + p * P;
    const uint32_t E = a.iend - B > P ? B + P : a.iend;
    const uint32_t whi = a.iend - B > P + kBpSlack ? B + P + kBpSlack : a.iend;
    uint32_t entry, end; bool on;
    bp_chain(a, p, B, E, lane, entry, end, on);
    if (a.entry[p].x == kPosEnd) {
        if (lane == 0) { a.totals[2 * (size_t)p] = 0ull; a.totals[2 * (size_t)p + 1] = 0ull; }
        return;
    }
    WinReader rd;
    rd.a_win = bp_stage(a.in, B, whi, smem, lane);
    rd.lo = B; rd.hi = whi; rd.g = a.in;
    __syncthreads();
    uint32_t cnt = 0; uint64_t outb = 0;
    uint32_t q = on ? entry : kPosEnd;
    while (ballot64(q < E && q != end) != 0ull) {
        if (q < E && q != end) {
            Seq s;
            if (walk_step<G>(rd, q, a.iend, s, a.in)) { cnt += 1; outb += (uint64_t)s.lit + s.mlen; q = s.next; }
            else q = kPosErr;
        }
    }
    uint32_t tc; uint64_t to;
    const uint32_t bc = bp_scan32(cnt, tc);
    const uint64_t bo = bp_scan64(outb, to);
    a.lane_idx[(size_t)p * 64u + lane] = bc;
    a.lane_op[(size_t)p * 64u + lane] = bo;
    if (lane == 0) { a.totals[2 * (size_t)p] = tc; a.totals[2 * (size_t)p + 1] = to; }
}

// K4: totals[2p], totals[2p+1] -> exclusive prefix; status[2..3] = total sequences (lo, hi), status[4..5] = total output
__global__ __launch_bounds__(64) void big_scan_kernel(BigParse a0) {
    const BigParse a = a0.jobs != nullptr ? a0.jobs[blockIdx.x] : a0;
    const uint32_t lane = lane_id();
    uint64_t run_c = 0, run_o = 0;
    for (uint32_t p0 = 0; p0 < a.np; p0 += 64u) {
        const uint32_t p = p0 + lane;
        const uint64_t c = p < a.np ? a.totals[2 * (size_t)p] : 0ull, o = p < a.np ? a.totals[2 * (size_t)p + 1] : 0ull;
        uint64_t tc, to;
        const uint64_t bc = bp_scan64(c, tc), bo = bp_scan64(o, to);
        if (p < a.np) { a.totals[2 * (size_t)p] = run_c + bc; a.totals[2 * (size_t)p + 1] = run_o + bo; }
        run_c += tc; run_o += to;
    }
    if (lane == 0) {
        a.status[2] = (uint32_t)run_c; a.status[3] = (uint32_t)(run_c >> 32);
        a.status[4] = (uint32_t)run_o; a.status[5] = (uint32_t)(run_o >> 32);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K5: validation + absolute sync points.  status[1] |= 1 on any violation; status[6..7] = decoded size (set by the lane
// that meets the last sequence)
template <class G>
__global__ __launch_bounds__(64) void big_emit_kernel(BigParse a0) {
    BigParse a = a0;
    uint32_t piece_index = blockIdx.x;
    if (a0.jobs != nullptr) { const uint2 pm = a0.piece_map[blockIdx.x]; a = a0.jobs[pm.x]; piece_index = pm.y; }
    const uint32_t P = a.piece;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t p = piece_index, lane = lane_id();
    const uint32_t B = a.start + p * P;
    const uint32_t E = a.iend - B > P ? B + P : a.iend;
    const uint32_t whi = a.iend - B > P + kBpSlack ? B + P + kBpSlack : a.iend;
    if (a.entry[p].x == kPosEnd) return;
    uint32_t entry, end; bool on;
    bp_chain(a, p, B, E, lane, entry, end, on);
    WinReader rd;
    rd.a_win = bp_stage(a.in, B, whi, smem, lane);
    rd.lo = B; rd.hi = whi; rd.g = a.in;
    __syncthreads();
    uint64_t idx = a.totals[2 * (size_t)p] + a.lane_idx[(size_t)p * 64u + lane];
    uint64_t op = a.totals[2 * (size_t)p + 1] + a.lane_op[(size_t)p * 64u + lane];
    bool bad = false;
    uint32_t q = on ? entry : kPosEnd;
    while (ballot64(q < E && q != end && !bad) != 0ull) {
        if (q < E && q != end && !bad) {
            if ((idx % kSyncEvery) == 0u) a.sync[idx / kSyncEvery] = make_uint2(q, (uint32_t)op);
            Seq s;
            bool fin = false;
            if (!walk_step<G>(rd, q, a.iend, s, a.in) || !G::check(s, op, a.cap, fin)) bad = true;
            else if (fin) {
                if (!G::result_ok(op, a.cap)) bad = true;
                else { a.status[6] = (uint32_t)op; a.status[7] = (uint32_t)(op >> 32); a.status[8] = 1u; }
                q = kPosEnd;
            } else { q = s.next; idx += 1; }
        }
    }
    if (bad || (on && end == kPosErr)) atomicOr(&a.status[1], 1u);
}

// ---------------------------------------------------------------------------------------------------------------------
// K6: slab s = output [s * 65536, min((s+1) * 65536, total)).  The decoder of a slab walks whole sync groups (8
// sequences): from the last sync point at or before the slab's first byte to the end of the group that holds its last
// byte; what lies outside the slab is clipped there.
__global__ __launch_bounds__(256) void big_slab_kernel(BigSlabs d0, const BigSlabs* jobs, const uint2* slab_job, uint32_t n_all) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_all) return;
    const BigSlabs d = jobs != nullptr ? jobs[slab_job[t].x] : d0;
    const uint32_t s = jobs != nullptr ? slab_job[t].y : t;
    if (s >= d.n_slabs) return;
    const uint64_t S = (uint64_t)s * 65536ull;
    const uint64_t Eo = S + 65536ull < d.total ? S + 65536ull : d.total;
    // last sync index with op <= x
    const auto last_le = [&](uint64_t x) {
        uint32_t lo = 0, hi = d.n_sync;                      // sync[0].op = 0 <= x always
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            if ((uint64_t)d.sync[mid].y <= x) lo = mid; else hi = mid;
        }
        return lo;
    };
    const uint32_t k0 = last_le(S), k1 = last_le(Eo - 1ull);
    const uint64_t seq_end = (uint64_t)(k1 + 1u) * kSyncEvery < d.n_seq ? (uint64_t)(k1 + 1u) * kSyncEvery : d.n_seq;
    const uint32_t nrec = (uint32_t)(seq_end - (uint64_t)k0 * kSyncEvery);
    const uint32_t in_lo = d.sync[k0].x;
    const uint32_t in_hi = k1 + 1u < d.n_sync ? d.sync[k1 + 1u].x : d.iend;
    d.in_off[s] = d.in_base_off + in_lo;
    d.in_len[s] = in_hi - in_lo;
    d.out_off[s] = d.out_base_off + S;
    d.out_cap[s] = S;                                       // (the decoder's slab mode: the slab's first output position in its stream)
    d.result[s] = (int64_t)(Eo - S);
    d.meta[s] = make_uint2(nrec, (s + 1u == d.n_slabs ? d.iend : in_hi + 64u) - in_lo);      // the stream's end relative to the slab's input; only the last slab needs the true one
    d.first[s] = make_uint2(d.sync_index_base + k0, in_lo);
    atomicMax(d.max_rec, nrec);
}

template <class G>
void run(const BigParse& a, hipStream_t s) {
    const uint32_t P = a.piece;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_mark_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_all(kBigPieceLarge));
    hipLaunchKernelGGL(big_mark_kernel<G>, dim3(a.np), dim3(64), bp_lds_all(P), s, a);
    hipLaunchKernelGGL(big_thread_kernel<G>, dim3(1), dim3(64), 0, s, a);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_count_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_in(kBigPieceLarge));
    hipLaunchKernelGGL(big_count_kernel<G>, dim3(a.np), dim3(64), bp_lds_in(P), s, a);
    hipLaunchKernelGGL(big_scan_kernel, dim3(1), dim3(64), 0, s, a);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_emit_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_in(kBigPieceLarge));
    hipLaunchKernelGGL(big_emit_kernel<G>, dim3(a.np), dim3(64), bp_lds_in(P), s, a);
}

}  // namespace

template <class G>
void run_many(const BigParse& hdr, uint32_t n_jobs, uint32_t n_pieces, uint32_t max_piece, hipStream_t s) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_mark_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_all(kBigPieceLarge));
    hipLaunchKernelGGL(big_mark_kernel<G>, dim3(n_pieces), dim3(64), bp_lds_all(max_piece), s, hdr);
    hipLaunchKernelGGL(big_thread_kernel<G>, dim3(n_jobs), dim3(64), 0, s, hdr);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_count_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_in(kBigPieceLarge));
    hipLaunchKernelGGL(big_count_kernel<G>, dim3(n_pieces), dim3(64), bp_lds_in(max_piece), s, hdr);
    hipLaunchKernelGGL(big_scan_kernel, dim3(n_jobs), dim3(64), 0, s, hdr);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_emit_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_in(kBigPieceLarge));
    hipLaunchKernelGGL(big_emit_kernel<G>, dim3(n_pieces), dim3(64), bp_lds_in(max_piece), s, hdr);
}

void launch_big_parse_many(const BigParse* jobs, uint32_t n_jobs, const uint2* piece_map, uint32_t n_pieces, uint32_t max_piece, int codec, hipStream_t s) {
    if (n_jobs == 0 || n_pieces == 0) return;
    BigParse hdr = {};
    hdr.jobs = jobs; hdr.piece_map = piece_map;
    if (codec == CJ_CODEC_SNAPPY_RAW) run_many<SnappyGrammar>(hdr, n_jobs, n_pieces, max_piece, s);
    else run_many<Lz4Grammar>(hdr, n_jobs, n_pieces, max_piece, s);
}

void launch_big_parse(const BigParse& a, int codec, hipStream_t s) {
    if (a.np == 0) return;
    if (codec == CJ_CODEC_SNAPPY_RAW) run<SnappyGrammar>(a, s);
    else run<Lz4Grammar>(a, s);
}

void launch_big_slabs(const BigSlabs& d, hipStream_t s) {
    if (d.n_slabs == 0) return;
    hipLaunchKernelGGL(big_slab_kernel, dim3((d.n_slabs + 255u) / 256u), dim3(256), 0, s, d, (const BigSlabs*)nullptr, (const uint2*)nullptr, d.n_slabs);
}

void launch_big_slabs_many(const BigSlabs* jobs, const uint2* slab_job, uint32_t n_slabs, hipStream_t s) {
    if (n_slabs == 0) return;
    hipLaunchKernelGGL(big_slab_kernel, dim3((n_slabs + 255u) / 256u), dim3(256), 0, s, BigSlabs{}, jobs, slab_job, n_slabs);
}

}  // namespace cj
