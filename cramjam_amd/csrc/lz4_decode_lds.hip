// lz4_decode_lds.hip — the WORKGROUP decoder: LZ4 *block* / Snappy *raw* decoding with the chunk's 64 KiB output window
// resident in LDS (160 KiB/CU on gfx950), two persistent workgroups of eight wavefronts per CU.
// Same results as the other mappings (reference call sites /root/reference/src/lz4.rs:88,90,164,168, src/snappy.rs:57,106).
//
// Why: with one wave/lane per chunk every match copy is a dependent read of the chunk's own earlier output somewhere in
// the last 64 KiB — an HBM round trip of a 128 B line for ~18 useful bytes.  Here the history never leaves the CU.
// Phases per chunk (cycles measured on the benchmark data with two workgroups per CU, profiles/r02):
//   S0   8.8 k  stage the compressed chunk in the still unused window (16 B/lane loads), clear the ready bitmap
//   P   (batches up to CJ_FUSED_MAX_CHUNKS chunks, lz4_decode_fused_kernel) the parse stage itself: 256 lanes walk 256 segments
//               of the staged chunk, the true path is stitched, validated and written as 16 B records (fused_parse below)
//   D1  12.5 k  (larger batches, after lz4_parse_kernel / snappy_parse_kernel) one thread per sync point re-walks 8 sequences
//               in LDS (aligned dword pairs + v_alignbyte) and writes their records to the workgroup's table
//   D1f         match forwarding for chunks of near matches (deep dependency chains), see below
//   D2   22 k   literals: one lane per sequence, global -> window, head / dword / tail stores, ready bits set
//   D3   52 k   matches: one lane per sequence, copied as soon as the ready bitmap covers its source bytes (the earliest
//               unresolved match is always ready: deadlock-free); one hand-scheduled poll step for the common shape
//   D4   2.3 k  stream the finished window to HBM (non-temporal 16 B stores)
// What bounds it (profiles/r02/experiments): the LDS PIPE.  A sparse copy costs the pipe a few cycles per INSTRUCTION whatever
// the number of active lanes (tools/lds_throughput_probe.hip: ds_read_b64 2.4, ds_write_b64 6.5, ds_or_b32 4.2 with 4 lanes;
// a misaligned access lanes + 1), the resolver issues ~14 of them per poll round for ~4 ready matches, and with two
// workgroups per CU in D3 the pipe is saturated — which is also why every dependent LDS read of the other phases takes
// ~700 cycles.  Shortening the poll loop's instruction stream (3x), deeper prefetch in D2 or keeping every match of a thread
// in flight changed nothing or lost; DESIGN.md §5.1 has the numbers.
// The slab mode (one large stream: chunk c = slab c of its output) and the linked mode (LZ4 frames with linked blocks: the
// previous block stays in a second window) share the body.
#include "lds_shared.hpp"
#include "big_chunks.hpp"

namespace cj {

// phase cycle counters (debug aid, enabled by CJ_FLAG_DEBUG_PROFILE): S0, D1, D2, D3, D4, blocks
// (accumulated in LDS by thread 0 and flushed once per chunk: an atomicAdd per mark sits in front of the next barrier's vmcnt(0)
//  and costs a global round trip, 6-8 k cycles under this kernel's load — round 2's marks measured themselves)
__device__ unsigned long long g_lds_phase_cycles[16];
#define CJ_PHASE_MARK(idx)                                                              \
    do {                                                                                \
        if (prof && tid == 0) {                                                         \
            unsigned long long now_ = __builtin_readcyclecounter();                     \
            s_prof[idx] += (uint32_t)(now_ - t_prev);                                   \
            t_prev = now_;                                                              \
        }                                                                               \
    } while (0)

// =====================================================================================================
// TWO workgroups per CU: only what must be in LDS stays there (output window + ready bitmap, 72.4 KiB).  The compressed
// bytes are staged in the still unused window for the token walk and read again from global memory (L2 hits) for the
// literal copies, and the record table lives in a per-workgroup global scratch slot that is written once and read twice
// with coalesced 16 B accesses (L2-resident: the grid is persistent, 2 slots per CU).  Two chunks per CU overlap each
// other's latency (one chunk per CU with everything in LDS — round 1's first version — ran at 385 GB/s against 508).
// Workgroups are persistent and pull chunk indices from a global counter.
// =====================================================================================================
#ifndef CJ_L2_THREADS
#define CJ_L2_THREADS 512
#endif
constexpr uint32_t kL2Threads = CJ_L2_THREADS;
constexpr uint32_t kL2OffBits = CJ_L2_WINDOW;
constexpr uint32_t kL2BitWords = kL2OffBits / 32u;    // ready bitmap: one bit per byte of the window
// LDS of a workgroup with a window of `win` bytes: window | ready bitmap | [0] fail flag, [8] current chunk, [64,128) dummy bytes, [128,384) dummy dwords |
// 128 bytes of phase counters and the next chunk's descriptors
constexpr uint32_t lds2_bytes(uint32_t win) { return win + win / 8u + 384u + 128u; }
constexpr uint32_t kL2Bytes = lds2_bytes(kL2OffBits);      // 74880 B: two workgroups fit one CU's LDS
#ifndef CJ_D2_LONG
#define CJ_D2_LONG 256u
#endif
constexpr uint32_t kD2LongRun = CJ_D2_LONG;      // literal runs at least this long are placed by the whole wavefront (512 / 256 / 128: x-ray 249.8 / 262.4 / 262.5 GB/s, whole corpus 212.5 / 213.8 / 213.6)
#ifndef CJ_D3_MASK64
#define CJ_D3_MASK64 1
#endif
#ifndef CJ_LZ4_D1_FAST
#define CJ_LZ4_D1_FAST 1
#endif
#ifndef CJ_SN_D1_FAST
#define CJ_SN_D1_FAST 1
#endif
#ifndef CJ_DENSE_LANES
#define CJ_DENSE_LANES 24u
#endif
// the record table of a workgroup (16-byte units) by window: the 8-byte records of a chunk + sentinel, then the 16-byte records the forwarding phase appends (at most two per forwarded record)
constexpr uint32_t lds2_tab_records_plain(uint32_t win) { return (lds_window_max_seq(win) + 1u) / 2u + 64u + 2u * (6144u * (win / 1024u) / 64u) + 64u; }
constexpr uint32_t lds2_tab_records(uint32_t win) {      // ... or the lists of the parse inside the decoder (lds_shared.hpp: fl_slot_units), whichever is larger
    return win >= 65536u ? 4u * kSyncStride * kSyncEvery
                         : (lds2_tab_records_plain(win) > fl_slot_units(win, win / 128u) ? lds2_tab_records_plain(win) : fl_slot_units(win, win / 128u));
}
static_assert(4u * lds2_tab_records(32768u) <= 2u * lds2_tab_records(65536u) && 8u * lds2_tab_records(16384u) <= 2u * lds2_tab_records(65536u),
              "the small windows' workgroups (four / eight per CU) fit the slots of the two 64 KiB ones");
constexpr uint32_t kL2TabRecords = 4u * kSyncStride * kSyncEvery;   // records of a chunk (the parse kernel routes chunks with more than 16 384 sequences elsewhere) + the extra literal copies of D1f (at most two per record, D1f takes at most 6 144 records)


// kLinked (LZ4 frames with linked blocks, frame.hip): the workgroup takes a whole FRAME (frames[f] = first block index,
// block count) and walks its blocks in order; the LDS holds TWO 64 KiB windows, the block being decoded and the previous
// block (every non-last block of such a frame decodes to exactly 64 KiB — the host checks that before it launches this
// path), so a match that reaches back past the start of its block reads final bytes from the other window.
constexpr uint32_t kL2LinkedBytes = 2u * 65536u + 8192u + 384u + 128u; // 139 776 B: one workgroup per CU

// two workgroups of eight wavefronts per CU = four wavefronts per SIMD: the register allocator must stay within 128 VGPRs
// (without the attribute it sees only the 512-thread bound and may take more, which silently halves the residency)
#ifndef CJ_L2_WAVES_PER_EU
#define CJ_L2_WAVES_PER_EU 4
#endif
#define CJ_L2_ATTR __attribute__((amdgpu_waves_per_eu(CJ_L2_WAVES_PER_EU, CJ_L2_WAVES_PER_EU)))
// kSlab (large.hip: ONE large stream): chunk c is the 64 KiB slab c of the stream's OUTPUT.  Its records are cut from the
// stream's absolute sync points (frames[c] = {first sync index, stream position of that sync point}; n_frames = stream
// length): sequences that begin before the slab or end after it are clipped, and the part of a match whose source lies
// before the slab ("cross") is copied from the finished output of the earlier slabs in global memory once slab c-1 has
// published its completion flag.  Slabs are claimed in order, so the slab a workgroup waits for is always running.
#ifndef CJ_SLAB_POLL_MASK
#define CJ_SLAB_POLL_MASK 1u
#endif
#ifndef CJ_FWD_ROUNDS
#define CJ_FWD_ROUNDS 16u
#endif
#ifdef CJ_SLAB_TRACE
__device__ unsigned long long g_slab_trace[8192 * 8];
#define CJ_TRACE(slot) do { if (c < 8192u && lane == 0) g_slab_trace[c * 8u + (slot)] = wall_clock64(); } while (0)
#define CJ_TRACE_T0(slot) do { if (c < 8192u && tid == 0) g_slab_trace[c * 8u + (slot)] = wall_clock64(); } while (0)
#else
#define CJ_TRACE(slot) do {} while (0)
#define CJ_TRACE_T0(slot) do {} while (0)
#endif
__device__ unsigned long long g_fwd_chunks = 0ull;            // test hook (batches with CJ_FLAG_DEBUG_PROFILE): chunks / slabs that went through D1f
#ifndef CJ_FWD_ROUNDS_BATCH
#define CJ_FWD_ROUNDS_BATCH 2u
#endif
#ifndef CJ_FWD_NEAR
#define CJ_FWD_NEAR 4096
#endif
#ifndef CJ_FWD_SHARE_NUM
#define CJ_FWD_SHARE_NUM 2u                    // forwarding where more than 1 / 2 of the matches are near
#endif
constexpr uint32_t kFwdMaxRounds = CJ_FWD_ROUNDS, kFwdBatchRounds = CJ_FWD_ROUNDS_BATCH, kFwdNear = CJ_FWD_NEAR;
struct SlabArgs { uint32_t* done; uint4* cross; uint32_t tab_stride, cross_stride, rel; uint32_t* defer; uint32_t defer_stride; uint32_t prev = 1u; };      // prev: slab c waits for slab c - prev
#ifndef CJ_SLAB_PATIENCE
#define CJ_SLAB_PATIENCE 64u
#endif
constexpr uint32_t kSlabPatience = CJ_SLAB_PATIENCE;
// rel = 1 (LZ4 frames with linked blocks, frame.hip): chunk c is block c of the frame — its sync points are its own (ip, op
// relative to the block), the stream length is the block's, a STORED block is copied; history = the blocks before it.

// kRecFeed (with kSlab; chunks of 64 KiB .. 256 KiB in a device batch, big_chunks.hpp): the work items are the 64 KiB slabs of the
// listed chunks' output in SLAB-MAJOR order (item w = slab w / cap of listed chunk w % cap: when slab s of a chunk is claimed,
// its slab s - 1 finished thousands of claims ago, so the completion flags never make anybody wait), and the chunk's sequences are
// records in memory (big_parse_kernel): D1 is one thread per record — find its region, add the region's position base, clip
// to the slab exactly like the walking D1 of the large-stream slabs — instead of one thread per eight sequences walking tokens.
struct FeedArgs { const uint4* recs; const BigMeta* bigmeta; uint32_t cap; };

// kWinT / kThreadsT: the window and the workgroup of an instantiation.  The chains-in-flight sweep of round 6 (profiles/r06/experiments
// h01-h04): at 16 wavefronts per CU, four workgroups of four on 32 KiB windows decode 32 KiB pieces 32 % faster than two of eight, eight
// of two on 16 KiB windows 16 KiB pieces 87 % faster — what a CU's LDS and registers hold in flight is what bounds this decoder.
template <int kCodec, bool kLinked, bool kSlab, bool kFused = false, bool kRecFeed = false, uint32_t kWinT = CJ_L2_WINDOW, uint32_t kThreadsT = CJ_L2_THREADS>
__device__ __forceinline__ void lds2_body(const BatchArgs& a, const uint2* sync, const ParseMeta* meta, uint4* tabs, uint32_t* counter,
                                          const uint2* frames, uint32_t n_frames, const SlabArgs& sl, const FeedArgs& fd = FeedArgs{nullptr, nullptr, 0u}) {
    static_assert(!kRecFeed || (kSlab && !kLinked && !kFused), "records are fed to the slab mode");
    static_assert(!kLinked || kWinT == 65536u, "the two-window mode holds 64 KiB blocks");
    // (these shadow the file's defaults: everything below is written in terms of them)
    constexpr uint32_t kL2Threads = kThreadsT, kL2OffBits = kWinT, kL2BitWords = kWinT / 32u, kL2Bytes = lds2_bytes(kWinT);
    constexpr uint32_t kFwdMaxRecords = 6144u * (kWinT / 1024u) / 64u;      // D1f: 10 bytes of index per record in the window
    constexpr uint32_t kFwdNear = (uint32_t)CJ_FWD_NEAR * (kWinT / 1024u) / 64u;      // ... for chunks of mostly NEAR matches: a sixteenth of the window (16 KiB chunks have no others below 4 096)
    constexpr uint32_t kWinInMax = kWinT - 32u;
    constexpr uint32_t kMaxSeqT = lds_window_max_seq(kWinT);                // records of a chunk (the parse kernels route chunks with more elsewhere)                             // compressed bytes staged in the window (<= 15 B misalignment + 15 B round-up)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t kOffBits = kLinked ? 131072u : kL2OffBits, kOffVars = kOffBits + (kLinked ? 8192u : kL2OffBits / 8u);
    constexpr uint32_t kBitWords = kLinked ? 2048u : kL2BitWords;
    uint8_t* s_out = smem;
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(smem + kOffBits);
    uint32_t* s_fail = reinterpret_cast<uint32_t*>(smem + kOffVars);
    uint32_t* s_chunk = reinterpret_cast<uint32_t*>(smem + kOffVars + 8u);
    uint32_t a_out = (uint32_t)(uintptr_t)s_out;
    uint32_t a_prev = a_out;                                 // kLinked: LDS address of the previous block's window
    const Dummies dm = {(uint32_t)(uintptr_t)(smem + kOffVars + 64u) + (threadIdx.x & 63u),
                        (uint32_t)(uintptr_t)(smem + kOffVars + 128u) + 4u * (threadIdx.x & 63u)};
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint4* table = tabs + (size_t)blockIdx.x * (kSlab ? sl.tab_stride : lds2_tab_records(kWinT));
    // Batches of independent chunks keep 8-BYTE records: { lit_src | lit << 16, lit_start | offset << 16 } — the match
    // destination is lit_start + lit and the match length is the NEXT record's lit_start minus that (mod 2^16: a chunk of exactly
    // 64 KiB ends at position 0x10000), a sentinel { 0, U } follows the last record.  Half the table traffic of the 16-byte
    // { lit_src, lit, dst, offset | length << 16 } records, which the slab and linked modes keep (their records are clipped, split
    // and moved out of position order).  Records appended by the forwarding phase (literal copies, out of position order)
    // keep the 16-byte form in their own region of the slot.
    constexpr bool kCompact = !kSlab && !kLinked;
    constexpr uint32_t kExtraBase = (kMaxSeqT + 1u) / 2u + 64u;      // first 16-byte slot behind the 8-byte records (+ sentinel)
    uint2* table2 = reinterpret_cast<uint2*>(table);
    const auto rec_store = [&](uint32_t i, uint32_t lit_src, uint32_t lit, uint32_t dst, uint32_t w) {      // a record in position order
        if constexpr (kCompact) table2[i] = make_uint2(lit_src | (lit << 16), ((dst - lit) & 0xffffu) | ((w & 0xffffu) << 16));
        else table[i] = make_uint4(lit_src, lit, dst, w);
    };
    const auto rec_load = [&](uint32_t i, uint32_t n_main) -> uint4 {                   // the 16-byte view of record i (i >= n_main: appended records)
        if constexpr (kCompact) {
            if (i >= n_main) return table[kExtraBase + (i - n_main)];
            const uint4 t = ld16u(reinterpret_cast<const uint8_t*>(table2 + i));
            const uint32_t lit = t.x >> 16, dst = (t.y & 0xffffu) + lit, off = t.y >> 16;
            const uint32_t m = off ? ((t.w & 0xffffu) - dst) & 0xffffu : 0u;
            return make_uint4(t.x & 0xffffu, lit, dst, off | (m << 16));
        } else return table[i];
    };
    // The same in two steps for the batch loops of D2 and D3: the stored form is REQUESTED a batch ahead (no branch around the load:
    // an index past the end reads the last record) and only looked at when its batch starts — written as one rec_load the decoding
    // sits behind the load and the wave waits for the round trip on the spot (that was 2 x ~5 round trips per chunk on the chain).
    const auto rec_fetch = [&](uint32_t i, uint32_t n_main, uint32_t n_all) -> uint4 {
        const uint32_t ic = i < n_all ? i : (n_all ? n_all - 1u : 0u);
        if constexpr (kCompact) return ld16u(reinterpret_cast<const uint8_t*>(table) + (ic < n_main ? ic * 8u : (kExtraBase + (ic - n_main)) * 16u));
        else return table[ic];
    };
    const auto rec_view = [&](const uint4& t, uint32_t i, uint32_t n_main, uint32_t n_all) -> uint4 {
        if (i >= n_all) return make_uint4(0, 0, 0, 0);
        if constexpr (kCompact) {
            if (i >= n_main) return t;
            const uint32_t lit = t.x >> 16, dst = (t.y & 0xffffu) + lit, off = t.y >> 16;
            const uint32_t m = off ? ((t.w & 0xffffu) - dst) & 0xffffu : 0u;
            return make_uint4(t.x & 0xffffu, lit, dst, off | (m << 16));
        } else return t;
    };
    const auto rec_set_offset = [&](uint32_t i, const uint4& r, uint32_t off) {          // forwarding: another source (0 = the match is gone)
        if constexpr (kCompact) reinterpret_cast<uint32_t*>(table2 + i)[1] = ((r.z - r.y) & 0xffffu) | (off << 16);
        else table[i] = make_uint4(r.x, r.y, r.z, off ? (off | (r.w & 0xffff0000u)) : 0u);
    };
    const auto rec_append = [&](uint32_t k, uint32_t n_main, const uint4& r) {           // forwarding / slab clipping: records out of position order
        if constexpr (kCompact) table[kExtraBase + k] = r;
        else table[n_main + k] = r;
    };
    uint32_t* s_ncross = reinterpret_cast<uint32_t*>(smem + kOffVars + 16u);     // kSlab: entries in the cross list / extra records
    uint32_t* s_nextra = reinterpret_cast<uint32_t*>(smem + kOffVars + 20u);
    uint32_t* s_fwd = reinterpret_cast<uint32_t*>(smem + kOffVars + 24u);          // D1f: rounds in which a record moved
    uint32_t* s_small = reinterpret_cast<uint32_t*>(smem + kOffVars + 28u);        // D1: matches with an offset below kFwdNear
    volatile uint32_t* s_prevok = reinterpret_cast<volatile uint32_t*>(smem + kOffVars + 32u);      // kSlab: a wave has seen the previous slab's flag and fenced
    const bool prof = (a.flags & CJ_FLAG_DEBUG_PROFILE) != 0;
    uint32_t* s_prof = reinterpret_cast<uint32_t*>(smem + kOffVars + 384u);
    if (prof && threadIdx.x < 16u) s_prof[threadIdx.x] = 0u;
    unsigned long long t_prev = prof ? __builtin_readcyclecounter() : 0ull;
    uint32_t fr_first = 0, fr_n = 0, fr_k = 0;               // kLinked: current frame and position in it
    // batches: thread 0 claims the NEXT chunk right after the current one is known, so the atomic's round trip runs under the
    // descriptor loads instead of in front of them (slabs are claimed when they are started: their order matters)
    uint32_t next_c = 0;
    // (the FIRST chunk of a workgroup is its own index: with every workgroup of a launch asking the counter at once — 512 of them, 2 048 on
    //  16 KiB windows — the atomics queue up at one L2 address, ~40 cycles each, in front of the first chunk: S0 of 4 096 x 16 KiB chunks
    //  63 k cycles, 12 k without; the counter hands out the chunks from gridDim.x on: profiles/r06/experiments f07)
    if constexpr (!kLinked && !kSlab) next_c = blockIdx.x;
    const uint32_t claim_n = gridDim.x >= 64u * kClaimCounters ? kClaimCounters : (gridDim.x >= 64u ? gridDim.x >> 6 : 1u);      // counters of this launch: one per 64 workgroups
    uint32_t claim_k = blockIdx.x % claim_n;                  // the counter this workgroup asks (its own until that has run dry)

    if constexpr (!kSlab && !kLinked) __builtin_amdgcn_s_setprio(2);
    for (;;) {
        uint32_t c;
        if constexpr (kLinked) {
            if (fr_k == fr_n) {                              // next frame
                if (tid == 0) *s_chunk = atomicAdd(counter, 1u);
                __syncthreads();
                const uint32_t f = *s_chunk;
                __syncthreads();
                if (f >= n_frames) break;
                fr_first = frames[f].x; fr_n = frames[f].y; fr_k = 0;
                if (fr_n == 0u) continue;
            }
            c = fr_first + fr_k;
            s_out = smem + ((fr_k & 1u) ? 65536u : 0u);
            a_out = (uint32_t)(uintptr_t)s_out;
            a_prev = (uint32_t)(uintptr_t)(smem + ((fr_k & 1u) ? 0u : 65536u));
            fr_k += 1;
            if (tid == 0) *s_fail = 0u;
            for (uint32_t i = tid; i < kBitWords; i += kL2Threads) s_bits[i] = 0u;
            __syncthreads();                                 // also: the previous block's D4 has finished reading its window
        } else {
            if (tid == 0) {
                uint32_t cc = kSlab ? atomicAdd(counter, 1u) : next_c;
                if constexpr (!kSlab) {
                    // this workgroup's counter has run dry: the other counters' chunks (only at a batch's end: up to claim_n - 1 round trips, once or twice per workgroup)
                    for (uint32_t t = 1; cc >= a.n_chunks && t < claim_n && a.n_chunks > gridDim.x; t++) {       // (a batch of at most gridDim.x chunks has nothing behind the first ones)
                        claim_k = claim_k + 1u == claim_n ? 0u : claim_k + 1u;
                        cc = gridDim.x + claim_n * atomicAdd(counter + claim_k * kClaimStride, 1u) + claim_k;
                    }
                }
                *s_chunk = cc; *s_fail = 0u; *s_ncross = 0u; *s_nextra = 0u; *s_small = 0u; *s_prevok = 0u;
            }
            for (uint32_t i = tid; i < kBitWords; i += kL2Threads) s_bits[i] = 0u;
            __syncthreads();
            c = *s_chunk;
            __syncthreads();                                 // everyone has read s_chunk before thread 0 can overwrite it
            if (c >= a.n_chunks) break;
            // (... and one counter per 64 workgroups of the launch — 8 / 16 / 32 on 64 / 32 / 16 KiB windows, 256 bytes apart —, counter k handing
            //  out every n-th chunk behind the first gridDim.x: a launch's first claims, which wait in front of the next barrier, meet 63 others
            //  at their address instead of 2 047.  A workgroup whose counter has run dry takes the other counters' chunks: above.)
            if constexpr (!kSlab) { if (tid == 0) next_c = gridDim.x + claim_n * atomicAdd(counter + claim_k * kClaimStride, 1u) + claim_k; }
        }
        ParseMeta pm = {1u, 0u};                            // kFused: nothing has looked at the chunk yet
        uint64_t d_in_off, d_in_len, d_out_off, d_result;
        if constexpr (!kFused) pm = meta[c];
        // the chunk's descriptors in ONE round trip: left to itself the compiler waits for pm.nseq (the early-out below) before it
        // even requests the others, and the chunk's bytes are a third dependent round trip behind those
        d_in_off = a.in_off[c]; d_in_len = a.in_len[c]; d_out_off = a.out_off[c];
        d_result = kFused ? a.out_cap[c] : (uint64_t)a.result[c];      // kFused: the capacity (the parse computes the size)
        asm volatile("" :: "v"(pm.nseq), "v"(pm.in_skip), "v"((uint32_t)d_in_off), "v"((uint32_t)d_in_len), "v"((uint32_t)d_out_off), "v"((uint32_t)d_result));
        uint32_t f_cap = 0;                                  // kFused: output capacity (LZ4) / announced length (Snappy) handed to the parse
        if constexpr (kFused) {
            // the prologue of the parse kernels: size prefix / length preamble, the special cases, what this decoder cannot hold.
            // Everything that is not a plain chunk goes to the wavefront-per-chunk kernel (it names every error exactly).
            ParseMeta* meta_w = const_cast<ParseMeta*>(meta);
            const uint8_t* in0 = a.in_base + d_in_off;
            uint64_t n64 = d_in_len, cap64 = d_result;
            bool route = false;
            uint32_t skip = 0;
            if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) {
                uint64_t ulen = 0;
                uint32_t shift = 0, i = 0, hdr = 0;
                bool ok = false;
                if (n64 == 0 || n64 > 0xFFFFFFF0ull) route = true;
                else {
                    const uint32_t h0 = ld32u(in0), h1 = n64 > 4 ? ld32u(in0 + 4) : 0u;     // (the second word only where the chunk has it: reads stay in the chunk's granules, cramjam_hip.h)
                    while (hdr < (uint32_t)n64 && i < 5u) {
                        const uint32_t bb = (hdr < 4u ? h0 >> (8u * hdr) : h1 >> (8u * (hdr - 4u))) & 0xffu;
                        hdr += 1;
                        if (bb < 0x80u) { ulen |= (uint64_t)bb << shift; ok = true; break; }
                        ulen |= (uint64_t)(bb & 0x7fu) << shift;
                        shift += 7; i += 1;
                    }
                    if (!ok || ulen > 0xFFFFFFFFull || ulen > cap64 || ulen == 0 || ulen > kWinT || n64 - hdr > kWinInMax || hdr == (uint32_t)n64) route = true;
                    skip = hdr; cap64 = ulen;
                }
            } else {
                const uint8_t* inp = in0;
                if (lz4_block_prologue(a.flags, inp, n64, cap64) != 0) route = true;
                else {
                    skip = (uint32_t)(inp - in0);
                    if (cap64 == 0 || n64 == 0 || cap64 > kWinT || n64 > kWinInMax) route = true;
                }
            }
            if (route) { if (tid == 0) meta_w[c] = ParseMeta{0u, kRouteWave}; continue; }
            pm.in_skip = skip;
            f_cap = (uint32_t)cap64;
        }
        // kSlab: does this chunk have a predecessor whose completion it may have to wait for?  (large streams: out_cap[c] holds the
        // slab's first output position IN ITS STREAM — several streams may share one launch; linked frames: block 0 has none)
        const bool has_prev = kSlab && (sl.rel ? c > 0u : a.out_cap[c] != 0ull);
        if constexpr (kLinked) {
            if (pm.in_skip & kRouteStored) {                 // stored block: its bytes ARE the window (history for the next block)
                const uint32_t len = (uint32_t)a.result[c];
                const uint8_t* src = a.in_base + a.in_off[c];
                uint8_t* dsto = a.out_base + a.out_off[c];
                for (uint32_t i = tid * 16u; i < len; i += kL2Threads * 16u) {
                    if (i + 16u <= len) {
                        const uint4 v = ld16u(src + i);
                        *reinterpret_cast<uint4*>(s_out + i) = v;
                        st16u_nt(dsto + i, v);
                    } else {
                        for (uint32_t q = i; q < len; q++) { const uint8_t b = src[q]; s_out[q] = b; dsto[q] = b; }
                    }
                }
                continue;                                    // the barrier at the top of the next block orders these LDS writes
            }
        }
        // kSlab: done[c] means "chunks 0..c are complete" (a Snappy copy may reach back over many slabs, and a slab without
        // cross matches never waited for its predecessor): the flag is set after done[c-1] has been seen.  Every wave's
        // stores are out (release fence + barrier) before thread 0 stores the flag.
        bool wt_tail = false;                                // kSlab: the chunk ends with plain byte stores (size not a multiple of 16)
        const auto publish = [&]() {
            if constexpr (kSlab) {
                // the chunk's bytes were stored write-through (sc0 sc1): acknowledged = in memory.  Every wave waits for its stores,
                // barrier, relaxed flag store.  No L2 write-back (buffer_wbl2: several µs with 64 KiB freshly written) — except
                // after plain byte stores (a chunk tail), where ONE agent-scope release + an explicit s_waitcnt (the compiler may
                // drop the fence's own) precede the flag
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                CJ_TRACE_T0(4);                                            // stores acknowledged
                if (tid == 0) {
                    if (has_prev) while (__hip_atomic_load(&sl.done[c - sl.prev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
                    if (wt_tail) {                           // a chunk whose size is not a multiple of 16 ended with plain byte stores
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __hip_atomic_store(&sl.done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    CJ_TRACE_T0(5);                                        // flag stored
                }
            }
        };
        if constexpr (kSlab) {
            if (pm.in_skip & kRouteStored) {                 // linked frame: a stored block is its own output
                const uint32_t len = (uint32_t)a.result[c];
                const uint8_t* src = a.in_base + a.in_off[c];
                uint8_t* dsto = a.out_base + a.out_off[c];
                for (uint32_t i = tid * 16u; i < len; i += kL2Threads * 16u) {
                    if (i + 16u <= len) st16u_wt(dsto + i, ld16u(src + i));
                    else for (uint32_t q = i; q < len; q++) dsto[q] = src[q];
                }
                wt_tail = (len & 15u) != 0u;
                publish();
                continue;
            }
        }
        if (pm.nseq == 0u) { publish(); continue; }          // error, empty, or routed to another kernel
        uint32_t nseq = pm.nseq;                             // kFused: set by the parse below
        uint32_t U = (uint32_t)d_result;                     // decoded size, 1..65536
        const bool slab_meta = kSlab && !sl.rel;            // large streams: meta[c].in_skip = the stream's end relative to this slab's input
        const uint32_t in_skip = slab_meta ? 0u : pm.in_skip;
        const uint8_t* in = a.in_base + d_in_off + in_skip;
        const uint32_t iend = (uint32_t)d_in_len - in_skip;
        const uint8_t* in_al = in - (reinterpret_cast<uintptr_t>(in) & 3u);
        const uint8_t* last_dw = in_al + ((((uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u)) + iend - 1u) & ~3u);
        // offset (relative to in) up to which reads are safe: the end of the 16 B granule holding the last input byte
        const uint32_t safe_end = ((((uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u)) + iend + 15u) & ~15u) - (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);
        uint8_t* out = a.out_base + d_out_off;
        const uint2* csync = kRecFeed ? nullptr : kSlab ? sync + frames[c].x : sync + (size_t)c * kSyncPitch;
        const uint32_t nsp = (nseq + kSyncEvery - 1u) / kSyncEvery;
        if constexpr (kRecFeed) {
            // Nothing stages this mode's input, so D2's literal loads would come from HBM one dependent round trip per batch (~5 k cycles
            // under this kernel's load; the 64 KiB path's S0 leaves the chunk's bytes in L2 as a side effect).  The item's meta names the
            // 128-byte lines its literals lie in (big_items_kernel): every thread touches one by LDS-DMA into the dummy dwords — no
            // register, nobody waits — while D1 runs.
            const uint32_t lo_line = pm.in_skip & 0xfffu, n_lines = (pm.in_skip >> 12) & 0x3ffu;
            const uint32_t dummy = (uint32_t)(uintptr_t)(smem + kOffVars + 128u);
            const uint8_t* p0 = in + 128u * lo_line;
            p0 -= reinterpret_cast<uintptr_t>(p0) & 127u;
            const uint8_t* p_last = in + iend - 1u;
            for (uint32_t l = tid; l < n_lines; l += kL2Threads) {
                const uint8_t* g = p0 + 128u * l;
                if (g <= p_last) {
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(g), "s"(dummy) : "memory");
                }
            }
        }
        const bool staged = !kRecFeed && (!kSlab || iend <= kWinInMax);   // kSlab: a slab inside a long literal run may span more input than the window holds; kRecFeed: nothing walks the bytes
        // ---- S0: stage the compressed chunk in the (still unused) output window so that D1's dependent token
        //      reads are LDS reads; D2 re-reads the literal bytes from global memory (L2 hits) because it overwrites
        //      the window while other lanes still need their sources ----
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);
        // this thread's first sync point (D1) is requested together with the chunk's bytes: one round trip instead of two
        uint2 p_first = make_uint2(0u, 0u);
        if constexpr (!kFused && !kRecFeed) p_first = csync[tid < nsp ? tid : 0u];
        {
            const uint4* src = reinterpret_cast<const uint4*>(in - mis);
            uint4* dst = reinterpret_cast<uint4*>(s_out);
            const uint32_t nvec = staged ? (mis + iend + 15u) >> 4 : 0u;
            // five loads in flight per thread (a 40 KiB chunk is 5 x 512 vectors): written as one load and one store per
            // iteration the compiler waits for every load before the next one — five dependent round trips per chunk.  The
            // first group is straight-line code (a loop header would make the compiler wait for the sync point requested
            // above before it issues the group's loads); longer chunks continue in the loop.
            const auto group = [&](uint32_t i0) {
                const uint32_t last = nvec - 1u;
                const uint32_t i1 = i0 + kL2Threads, i2 = i0 + 2u * kL2Threads, i3 = i0 + 3u * kL2Threads, i4 = i0 + 4u * kL2Threads;
                const uint4 v0 = src[i0 < last ? i0 : last], v1 = src[i1 < last ? i1 : last], v2 = src[i2 < last ? i2 : last],
                            v3 = src[i3 < last ? i3 : last], v4 = src[i4 < last ? i4 : last];
                if (i0 < nvec) dst[i0] = v0;
                if (i1 < nvec) dst[i1] = v1;
                if (i2 < nvec) dst[i2] = v2;
                if (i3 < nvec) dst[i3] = v3;
                if (i4 < nvec) dst[i4] = v4;
            };
            if (nvec > 0u) group(tid);
            for (uint32_t i0 = tid + 5u * kL2Threads; i0 < nvec; i0 += 5u * kL2Threads) group(i0);
        }
        __syncthreads();
        CJ_PHASE_MARK(0);

        // ---- D1: expand sync points into sequence records (LDS -> global table) ----
        const uint32_t a_in = a_out + mis;
        if constexpr (kFused) {
            using G = typename std::conditional<kCodec == CJ_CODEC_SNAPPY_RAW, SnappyGrammar, Lz4Grammar>::type;
            const bool ok = fused_parse<G, kL2Threads, kWinT>(a_in, iend, f_cap, s_bits, reinterpret_cast<uint32_t*>(smem + kL2Bytes), table2, s_small, nseq, U, prof ? s_prof : nullptr);
            ParseMeta* meta_w = const_cast<ParseMeta*>(meta);
            if (!ok) { if (tid == 0) meta_w[c] = ParseMeta{0u, kRouteWave}; continue; }      // (uniform)
            if (tid == 0) { meta_w[c] = ParseMeta{0u, 0u}; a.result[c] = (int64_t)U; table2[nseq] = make_uint2(0u, U & 0xffffu); }
            for (uint32_t i = tid; i < kBitWords; i += kL2Threads) s_bits[i] = 0u;                // the walk's marks: a bitmap again
            __syncthreads();
            CJ_PHASE_MARK(1);
        }
        uint32_t nrec_all = nseq;                            // kSlab: + the internal remainders of cross matches (extra records)
        if constexpr (kFused) {
            // (the records are in the table already)
        } else
        if constexpr (kSlab) {
            using G = typename std::conditional<kCodec == CJ_CODEC_SNAPPY_RAW, SnappyGrammar, Lz4Grammar>::type;
            const uint64_t S = a.out_off[c];
            const int64_t op_bias = sl.rel ? 0 : (int64_t)a.out_cap[c];   // the stream's sync points count output from the stream's start; S is the slab's place in the output BUFFER
            uint4* cross = sl.cross + (size_t)blockIdx.x * sl.cross_stride;
            // one sequence (literal source, lengths, offset, output position relative to the slab — maybe negative) -> its record,
            // clipped to the slab; the part of its match whose source lies before the slab -> the cross list.  Returns "near match".
            const auto slab_emit = [&](uint32_t sq, uint32_t src, uint32_t lit, uint32_t mlen, uint32_t off, int64_t op) -> uint32_t {
                    uint32_t near = 0;
                    int64_t o0 = op;
                    if (o0 < 0) { const uint32_t cut = (uint64_t)(-o0) < lit ? (uint32_t)(-o0) : lit; src += cut; lit -= cut; o0 += cut; }
                    int64_t d0 = o0 + lit;
                    if (d0 < 0) { const uint32_t cut = (uint64_t)(-d0) < mlen ? (uint32_t)(-d0) : mlen; mlen -= cut; d0 += cut; }
                    if (d0 < 0) { d0 = 0; o0 = 0; }                      // entirely before the slab (lit = mlen = 0 now)
                    if (o0 < 0) o0 = 0;
                    if (o0 >= (int64_t)U) { lit = 0; mlen = 0; d0 = U; }
                    else {
                        if (o0 + lit > (int64_t)U) { lit = U - (uint32_t)o0; mlen = 0; }
                        d0 = o0 + lit;
                        if (d0 + mlen > (int64_t)U) mlen = U - (uint32_t)d0;
                    }
                    const uint32_t dst = (uint32_t)d0;
                    uint4 rec = make_uint4(src, lit, dst, 0u);
                    if (mlen > 0u) {
                        if (off > dst) {                                  // source starts before the slab
                            const uint32_t back = off - dst, n1 = mlen < back ? mlen : back;
                            const uint64_t src_abs = S + dst - off;
                            const uint32_t k = atomicAdd(s_ncross, 1u);
                            cross[k] = make_uint4((uint32_t)src_abs, (uint32_t)(src_abs >> 32), dst, n1);
                            if (mlen == n1 && off <= 0xffffu) rec.w = off | (n1 << 16);      // (off > dst marks it as a cross copy: D3 skips it, D1f forwards into it; Snappy's 4-byte offsets stay in the cross list only)
                            if (mlen > n1) {
                                // the rest of the match repeats bytes from the start of THIS slab: an ordinary match at dst + n1.
                                // It keeps the sequence's place in the record order (D3's progress argument: a record waits
                                // only for records before it), so the sequence's literals move to an extra record instead —
                                // literals depend on nothing and may sit anywhere.
                                if (lit > 0u) table[nseq + atomicAdd(s_nextra, 1u)] = rec;
                                rec = make_uint4(0x80000000u | (lit + n1), 0u, dst + n1, off | ((mlen - n1) << 16));     // x: bytes before it that no record describes (D1f)
                            }
                        } else rec.w = off | (mlen << 16);
                        near += off < kFwdNear ? 1u : 0u;
                    }
                    table[sq] = rec;
                    return near;
            };
            if constexpr (kRecFeed) {
                // the slab's records: from the one that holds its first output byte to the one that holds the next slab's
                const uint32_t bi = c % fd.cap, slab = c / fd.cap;
                const BigMeta* bm = fd.bigmeta + bi;
                const uint32_t R0 = bm->slab_first[slab];
                // per region: first record, slot adjustment, position base — in LDS behind the decoder's own (the regions are found by
                // a binary search: a lane without records shares its first index with the next lane that has some)
                uint32_t* s_feed = reinterpret_cast<uint32_t*>(smem + kL2Bytes);
                if (tid < kBigLanes) {
                    const uint32_t f = tid ? bm->first[tid] : 0u, o = bm->opb[tid];
                    s_feed[tid] = f; s_feed[kBigLanes + tid] = tid * kBigRegion + (o >> 28) - f; s_feed[2u * kBigLanes + tid] = o & 0x0fffffffu;
                }
                __syncthreads();
                const uint4* crecs = fd.recs + (size_t)bi * kBigRecPitch;
                uint32_t near = 0;
                for (uint32_t i = tid; i < nseq; i += kL2Threads) {
                    const uint32_t gi = R0 + i;
                    uint32_t t = 0;
#pragma unroll
                    for (uint32_t step = kBigLanes / 2u; step != 0u; step >>= 1) t += gi >= s_feed[t + step] ? step : 0u;
                    const uint4 r = crecs[gi + s_feed[kBigLanes + t]];
                    const uint32_t mlen = (r.w >> 16) | ((r.x >> 24) << 16);
                    near += slab_emit(i, r.x & 0x00ffffffu, r.y, mlen, r.w & 0xffffu, (int64_t)(uint64_t)(r.z + s_feed[2u * kBigLanes + t]) - op_bias);
                }
                if (near) atomicAdd(s_small, near);
            } else {
            const uint32_t in_lo = frames[c].y;
            const uint32_t s_iend = sl.rel ? iend : pm.in_skip;           // the stream's end, relative to this slab's input
            const auto rd = [&](uint32_t p) { return staged ? lds_ld32a(a_in + p) : ld32u(in + p); };
            for (uint32_t sp = tid; sp < nsp; sp += kL2Threads) {
                const uint2 p = sp == tid ? p_first : csync[sp];
                uint32_t ip = p.x - in_lo;
                int64_t op = (int64_t)(uint64_t)p.y - op_bias;          // may be negative: the group starts before the slab
                uint32_t sq = sp * kSyncEvery, near = 0;
                for (uint32_t j = 0; j < kSyncEvery && sq < nseq; j++, sq++) {
                    if (op >= (int64_t)U) { table[sq] = make_uint4(0u, 0u, U, 0u); continue; }     // the rest of the group lies past the slab
                    Seq q;
                    (void)G::at(rd, ip, s_iend, q, in);                      // the parse stage accepted this stream
                    near += slab_emit(sq, q.lit_at, q.lit, q.mlen, q.offset, op);
                    ip = q.next;
                    op += (int64_t)q.lit + q.mlen;
                }
                if (near) atomicAdd(s_small, near);
            }
            }
            __syncthreads();
            nrec_all = nseq + *s_nextra;
        } else
        if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) {
            // Snappy: a record = optional literal element + optional copy element (snappy_records.hpp)
            const auto rd = [a_in](uint32_t p) { return lds_ld32a(a_in + p); };
            for (uint32_t sp = tid; sp < nsp; sp += kL2Threads) {
                const uint2 p = sp == tid ? p_first : csync[sp];
                uint32_t ip = p.x, op = p.y;
                uint32_t s = sp * kSyncEvery;
                uint32_t near = 0;
#if CJ_SN_D1_FAST
                // The common record — an optional literal with a one-byte header, then a copy with a 1- or 2-byte offset — as straight-line
                // code with ONE dependent LDS read: the copy element is read as 8 bytes, and the tag of the record behind it (2 or 3 bytes
                // further) comes with it (snappy_parse_kernel's fast_rec without its checks: the parse accepted this stream).  Every other
                // shape takes snappy_record_step from the same state.
                uint32_t t4 = rd(ip);
                for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                    const uint32_t tag = t4 & 0xffu, l6 = tag >> 2;
                    const bool is_lit = (tag & 3u) == 0u;
                    const uint32_t lhdr = is_lit ? 1u : 0u, lit_len = is_lit ? l6 + 1u : 0u;
                    const uint32_t ip2 = ip + lhdr + lit_len;
                    const uint2 c8 = lds_ld64a(a_in + ip2);
                    const uint32_t ctag = c8.x & 0xffu, kind = ctag & 3u;
                    const bool fast = !(is_lit & (l6 >= 60u)) & ((kind == 1u) | (kind == 2u)) & (ip2 < iend);
                    SnRecord rec;
                    if (fast) {
                        const uint32_t clen = kind == 1u ? 4u + ((ctag >> 2) & 7u) : 1u + (ctag >> 2);
                        const uint32_t offset = kind == 1u ? ((ctag >> 5) << 8) | ((c8.x >> 8) & 0xffu) : (c8.x >> 8) & 0xffffu;
                        rec.lit_src = is_lit ? ip + 1u : 0u; rec.lit_len = lit_len; rec.dst = op + lit_len; rec.w = offset | (clen << 16);
                        ip = ip2 + (kind == 1u ? 2u : 3u); op = rec.dst + clen;
                        t4 = __builtin_amdgcn_alignbyte(c8.y, c8.x, kind == 1u ? 2u : 3u);
                    } else {
                        (void)snappy_record_step(rd, ip, op, iend, U, rec);     // the parse kernel accepted this stream
                        t4 = rd(ip);
                    }
                    rec_store(s, rec.lit_src, rec.lit_len, rec.dst, rec.w);
                    near += (rec.w != 0u && (rec.w & 0xffffu) < kFwdNear) ? 1u : 0u;
                }
#else
                for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                    SnRecord rec;
                    (void)snappy_record_step(rd, ip, op, iend, U, rec);     // the parse kernel accepted this stream
                    rec_store(s, rec.lit_src, rec.lit_len, rec.dst, rec.w);
                    near += (rec.w != 0u && (rec.w & 0xffffu) < kFwdNear) ? 1u : 0u;
                }
#endif
                if (near) atomicAdd(s_small, near);
            }
        } else
        for (uint32_t sp = tid; sp < nsp; sp += kL2Threads) {
            const uint2 p = sp == tid ? p_first : csync[sp];
            uint32_t ip = p.x, op = p.y;
            uint32_t s = sp * kSyncEvery, near = 0;
#if CJ_LZ4_D1_FAST
            // ONE dependent LDS read per sequence: the offset field is read as 8 bytes, and the token behind it (2 or 3 bytes further) and
            // that token's first length byte come with it — what lz4_parse_kernel does since r04 x03, and the Snappy loop above since g04.
            // Only a length of 270 or more (a second length byte) reads on byte by byte.
            uint32_t t4 = lds_ld32a(a_in + ip);                     // token + 3 following bytes (may over-read: harmless)
            for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                const uint32_t token = t4 & 0xffu;
                ip += 1;
                uint32_t lit = token >> 4;
                if (lit == 15u) {
                    uint32_t b = (t4 >> 8) & 0xffu;
                    ip += 1; lit += b;
                    while (b == 255u) { b = lds_ld8(a_in + ip); ip += 1; lit += b; }
                }
                const uint32_t lit_src = ip;
                ip += lit; op += lit;
                uint32_t w = 0, mlen = 0;
                if (s + 1u < nseq) {
                    const uint2 o8 = lds_ld64a(a_in + ip);
                    const uint32_t offset = o8.x & 0xffffu;
                    mlen = token & 15u;
                    const bool ext = mlen == 15u;
                    const uint32_t b0 = (o8.x >> 16) & 0xffu;
                    ip += ext ? 3u : 2u;
                    t4 = __builtin_amdgcn_alignbyte(o8.y, o8.x, ext ? 3u : 2u);
                    if (ext) {
                        mlen += b0;
                        if (b0 == 255u) {
                            uint32_t b = b0;
                            while (b == 255u) { b = lds_ld8(a_in + ip); ip += 1; mlen += b; }
                            t4 = lds_ld32a(a_in + ip);
                        }
                    }
                    mlen += 4u;
                    w = offset | (mlen << 16);
                    near += offset < kFwdNear ? 1u : 0u;
                }
                rec_store(s, lit_src, lit, op, w);
                op += mlen;
            }
#else
            for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                const uint32_t t4 = lds_ld32a(a_in + ip);           // token + 3 following bytes (may over-read: harmless)
                const uint32_t token = t4 & 0xffu;
                ip += 1;
                uint32_t lit = token >> 4;
                if (lit == 15u) {
                    uint32_t b = (t4 >> 8) & 0xffu;
                    ip += 1; lit += b;
                    while (b == 255u) { b = lds_ld8(a_in + ip); ip += 1; lit += b; }
                }
                const uint32_t lit_src = ip;
                ip += lit; op += lit;
                uint32_t w = 0, mlen = 0;
                if (s + 1u < nseq) {
                    const uint32_t o4 = lds_ld32a(a_in + ip);
                    const uint32_t offset = o4 & 0xffffu;
                    ip += 2;
                    mlen = token & 15u;
                    if (mlen == 15u) {
                        uint32_t b = (o4 >> 16) & 0xffu;
                        ip += 1; mlen += b;
                        while (b == 255u) { b = lds_ld8(a_in + ip); ip += 1; mlen += b; }
                    }
                    mlen += 4u;
                    w = offset | (mlen << 16);
                    near += offset < kFwdNear ? 1u : 0u;
                }
                rec_store(s, lit_src, lit, op, w);
                op += mlen;
            }
#endif
            if (near) atomicAdd(s_small, near);
        }
        if constexpr (kCompact && !kFused) { if (tid == 0) table2[nseq] = make_uint2(0u, U & 0xffffu); }      // sentinel: where the last record's match ends
        __syncthreads();
        CJ_PHASE_MARK(1);
#ifndef CJ_NO_FORWARD
        // ---- D1f: MATCH FORWARDING.  D3 resolves matches as a dependency DAG and pays its latency per LEVEL; real data
        //      (text, logs, records) is deep: a phrase is copied from its previous occurrence, which was copied from the one
        //      before ... (thousands of levels in 64 KiB).  But if the source range of match A lies entirely inside the
        //      destination of an earlier non-overlapping match B, A can copy from B's SOURCE instead (offset += B's offset),
        //      and if it lies inside a literal run, A is a literal copy from the input and depends on nothing.  Iterated
        //      (each round reads the other records' current offsets: pointer doubling), chains collapse: depth 7 048 -> 67 on
        //      the "bottles" text, 562 -> 44 on log lines, 40 -> 23 on the benchmark data, where a third of the matches
        //      become literal copies.  kSlab: a match forwarded into a copy from an earlier slab becomes such a copy itself.
        //      The record index lives in the output window and the bitmap (both free between D1 and D2), 8 bytes per record:
        //      start | dst << 16 and the state word (current offset / input position + flags); the records tile the window
        //      (start[i+1] = end of record i), so a match's length is start[i+1] - dst[i]; plus, per 16 bytes of output,
        //      the last record that starts at or before them.
        if constexpr (!kLinked) {
            // mostly near matches: the chains are deep, forwarding pays (it costs ~25 k cycles + 10 k per round).  kSlab: always
            // when the slab waits for bytes of earlier slabs — the forwarding runs before that wait, what it removes from
            // the dependency depth comes off the serial chain through the slabs
            // (batches: only chunks that fill more than half of the window — in a small chunk every match is near, and its chains are short)
            if (nseq <= kFwdMaxRecords && staged && ((*s_small * CJ_FWD_SHARE_NUM > nseq && (kSlab || U > 8u * kFwdNear)) || (kSlab && *s_ncross > 0u))) {
                uint32_t* f_w0 = reinterpret_cast<uint32_t*>(s_out);
                uint32_t* f_st = f_w0 + kFwdMaxRecords;
                uint16_t* f_ls = reinterpret_cast<uint16_t*>(f_st + kFwdMaxRecords);      // literal source of every record (16 bits: the chunk is staged)
                uint16_t* f_blk = reinterpret_cast<uint16_t*>(s_bits);
                constexpr uint32_t kLit = 0x80000000u, kHole = 0x40000000u, kStop = 0x20000000u, kSplit = 0x10000000u, kVal = 0x0fffffffu;
                for (uint32_t i = tid; i < nseq; i += kL2Threads) {
                    const uint4 r = rec_load(i, nseq);
                    const bool hole = r.y == 0u && (r.x & 0x80000000u) != 0u;      // kSlab remainder record: x = bytes before it that no record describes
                    const uint32_t start = r.z - r.y - (hole ? (r.x & 0x7fffffffu) : 0u);
                    const uint32_t off = r.w & 0xffffu, m = r.w >> 16;
                    f_w0[i] = (start < 65535u ? start : 65535u) | ((r.z < 65535u ? r.z : 65535u) << 16);
                    f_ls[i] = (uint16_t)r.x;
                    // not forwardable: no match, self-overlapping, a cross copy already, the last record
                    f_st[i] = off | (hole ? kHole : 0u) | ((m == 0u || off < m || off > r.z || i + 1u == nseq) ? kStop : 0u);
                }
                if (tid == 0) { *s_fwd = 0u; if (prof) atomicAdd(&g_fwd_chunks, 1ull); }      // (counted under CJ_FLAG_DEBUG_PROFILE: the thread's next load would wait for the atomic)
                __syncthreads();
                for (uint32_t i = tid; i < nseq; i += kL2Threads) {          // blocks whose first byte lies in [start_i, start_{i+1})
                    const uint32_t b0 = ((f_w0[i] & 0xffffu) + 15u) >> 4;
                    const uint32_t b1 = i + 1u < nseq ? ((f_w0[i + 1u] & 0xffffu) + 15u) >> 4 : kL2OffBits / 16u;
                    for (uint32_t b = i == 0u ? 0u : b0; b < b1; b++) f_blk[b] = (uint16_t)i;
                }
                __syncthreads();
                // (batches: two rounds — corpus64k LZ4 / Snappy GB/s by cap: none 172 / 238, 2: 188 / 234, 4: 186 / 229, 8 and 16: 180 / 226;
                //  slab mode keeps sixteen: there the depth it removes comes off the serial chain through the slabs)
                const uint32_t max_rounds = kSlab ? kFwdMaxRounds : kFwdBatchRounds;
                for (uint32_t round = 0; round < max_rounds; round++) {
                    uint32_t changed = 0;
                    for (uint32_t i = tid; i + 1u < nseq; i += kL2Threads) {
                        const uint32_t st = f_st[i];
                        if (st & (kLit | kStop)) continue;
                        const uint32_t dst = f_w0[i] >> 16, nstart = f_w0[i + 1u] & 0xffffu;
                        if (nstart >= 65535u) { f_st[i] = st | kStop; continue; }  // (the clamp hides whether the match ends at 65535 or 65536)
                        const uint32_t m = nstart - dst, off = st & kVal;
                        const uint32_t sp = dst - off;                             // current source position
                        uint32_t r = f_blk[sp >> 4];
                        while (r + 1u < nseq && (f_w0[r + 1u] & 0xffffu) <= sp) r++;
                        if (r > i || r + 1u >= nseq) { f_st[i] = st | kStop; continue; }
                        const uint32_t w0 = f_w0[r], bst = f_st[r];
                        const uint32_t bstart = w0 & 0xffffu, bdst = w0 >> 16, bend = f_w0[r + 1u] & 0xffffu;
                        if (sp >= bstart && sp + m <= bdst) {                      // inside B's literal run (r == i: the record's own)
                            if (bst & kHole) { f_st[i] = st | kStop; continue; }   // ... which is not one
                            f_st[i] = kLit | (st & kHole) | (((uint32_t)f_ls[r] + (sp - bstart)) & kVal); changed = 1;
                        } else if (sp >= bdst && sp + m <= bend && bend > bdst) {  // inside B's match
                            const uint32_t boff = bst & kVal, bm = bend - bdst;
                            if (bst & kLit) { f_st[i] = kLit | (st & kHole) | ((boff + (sp - bdst)) & kVal); changed = 1; }
                            else if (boff >= bm && off + boff <= kVal) {           // (B a cross copy: A becomes one too — and is final)
                                f_st[i] = (st & kHole) | (off + boff) | (off + boff > dst ? kStop : 0u); changed = 1;
                            }
                            else f_st[i] = st | kStop;                             // B repeats itself: A stays
                        } else f_st[i] = st | kStop;                               // straddles two records: stays
                    }
                    if (changed) atomicOr(s_fwd, 1u << (round & 31u));
                    __syncthreads();
                    if (((*s_fwd >> (round & 31u)) & 1u) == 0u) break;          // uniform: nothing moved in this round
                }
                // Straddlers: a match whose source spans the boundary between two records cannot be forwarded as a whole, and the
                // chains that remain after the rounds run through such matches.  If each of its two parts, taken alone, ends in a
                // literal run or in a copy from an earlier slab (one step, with the final states of the rounds), the match is
                // replaced by those two dependency-free copies.
                for (uint32_t i = tid; i + 2u < nseq; i += kL2Threads) {
                    const uint32_t st = f_st[i];
                    if ((st & kLit) || !(st & kStop)) continue;
                    const uint32_t dst = f_w0[i] >> 16, nstart = f_w0[i + 1u] & 0xffffu, off = st & kVal;
                    if (nstart >= 65535u || nstart <= dst) continue;
                    const uint32_t m = nstart - dst;
                    if (off < m || off > dst) continue;                            // self-overlapping / a cross copy
                    const uint32_t sp = dst - off;
                    uint32_t r = f_blk[sp >> 4];
                    while (r + 1u < nseq && (f_w0[r + 1u] & 0xffffu) <= sp) r++;
                    if (r + 2u >= nseq || r + 1u > i) continue;
                    const uint32_t cut = f_w0[r + 1u] & 0xffffu;                   // the boundary inside the source range
                    if (sp + m <= cut || sp + m > (f_w0[r + 2u] & 0xffffu)) continue;     // not a two-record straddler
                    uint32_t kind[2], val[2];                                      // per part: 1 = literal copy (input position), 2 = cross copy (distance)
                    bool ok = true;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t q = r + (uint32_t)h, p0 = h ? cut : sp, p1 = h ? sp + m : cut;
                        const uint32_t qs = f_w0[q] & 0xffffu, qd = f_w0[q] >> 16, qe = f_w0[q + 1u] & 0xffffu, qst = f_st[q];
                        kind[h] = 0; val[h] = 0;
                        if (p0 >= qs && p1 <= qd && !(qst & kHole)) { kind[h] = 1; val[h] = (uint32_t)f_ls[q] + (p0 - qs); }
                        else if (p0 >= qd && p1 <= qe && !(qst & kSplit)) {
                            const uint32_t qv = qst & kVal;
                            if (qst & kLit) { kind[h] = 1; val[h] = qv + (p0 - qd); }
                            else if (kSlab && qv > qd && qv >= qe - qd) { kind[h] = 2; val[h] = qv; }
                        }
                        ok = ok && kind[h] != 0u;
                    }
                    if (!ok) continue;
                    const uint32_t len[2] = {cut - sp, sp + m - cut};
                    uint32_t d = dst;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if (kind[h] == 1u) rec_append(atomicAdd(s_nextra, 1u), nseq, make_uint4(val[h], len[h], d + len[h], 0u));
                        else if constexpr (kSlab) {
                            const uint64_t src_abs = a.out_off[c] + (h ? cut : sp) - val[h];
                            (sl.cross + (size_t)blockIdx.x * sl.cross_stride)[atomicAdd(s_ncross, 1u)] = make_uint4((uint32_t)src_abs, (uint32_t)(src_abs >> 32), d, len[h]);
                        }
                        d += len[h];
                    }
                    f_st[i] = st | kSplit;
                }
                __syncthreads();
                for (uint32_t i = tid; i + 1u < nseq; i += kL2Threads) {
                    const uint32_t st = f_st[i], v = st & kVal;
                    const uint4 r = rec_load(i, nseq);
                    const uint32_t m = r.w >> 16;
                    if (m == 0u) continue;
                    if (st & kSplit) { rec_set_offset(i, r, 0u); continue; }
                    if (st & kLit) {
                        rec_append(atomicAdd(s_nextra, 1u), nseq, make_uint4(v, m, r.z + m, 0u));      // a literal copy of m bytes ending at dst + m
                        rec_set_offset(i, r, 0u);
                    } else if (v != (r.w & 0xffffu)) {
                        if constexpr (kSlab) {
                            if (v > r.z) {                                        // forwarded into an earlier slab: a cross copy of its own
                                const uint64_t src_abs = a.out_off[c] + r.z - v;
                                (sl.cross + (size_t)blockIdx.x * sl.cross_stride)[atomicAdd(s_ncross, 1u)] = make_uint4((uint32_t)src_abs, (uint32_t)(src_abs >> 32), r.z, m);
                                rec_set_offset(i, r, 0u);
                                continue;
                            }
                        }
                        rec_set_offset(i, r, v);
                    }
                }
                __syncthreads();                                               // the index is dead: the bitmap is a bitmap again
                for (uint32_t i = tid; i < kBitWords; i += kL2Threads) s_bits[i] = 0u;
                __syncthreads();
                nrec_all = nseq + *s_nextra;
            }
        }
#endif

        // ---- D2: literals, one lane per sequence -> LDS window ----
        // what is left of a run after its first bytes were taken from LDS / the whole run (old path): global -> window
        const auto place_from_global = [&](uint32_t n, uint32_t src, uint32_t dst) {
            uint64_t lm = ballot64(n >= kD2LongRun);
            while (lm) {
                const uint32_t l = ctz64(lm);
                lm &= lm - 1ull;
                const uint32_t ln = rdlane(n, l), ls = rdlane(src, l), ld = rdlane(dst, l);
                wave_copy_to_lds(a_out + ld, in + ls, ln);
                wave_bits_set(s_bits, ld, ld + ln);
                if (lane == l) n = 0;
            }
            while (ballot64(n > 0u)) {                         // <=64 bytes per pass
                const uint32_t step = n < 64u ? n : 64u;
                const uint32_t tier = wave_tier(step, step > 0u);
                if (step > 0u) {
                    const uint8_t* g = in + src;
                    if (src + tier <= safe_end) {                       // the vector loads stay inside the chunk's last granule
                        if (tier <= 16u) lds_store_tier<16>(gl_ld_exact<16>(g), a_out + dst, 0u, step, dm);
                        else if (tier <= 32u) lds_store_tier<32>(gl_ld_exact<32>(g), a_out + dst, 0u, step, dm);
                        else lds_store_tier<64>(gl_ld_exact<64>(g), a_out + dst, 0u, step, dm);
                    } else {                                            // last few sequences of the chunk: clamped dwords
                        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(g) & 3u);
                        const uint8_t* ga = g - sh;
                        if (tier <= 16u) lds_store_tier<16>(gl_ld_aligned<6>(ga, last_dw), a_out + dst, sh, step, dm);
                        else if (tier <= 32u) lds_store_tier<32>(gl_ld_aligned<10>(ga, last_dw), a_out + dst, sh, step, dm);
                        else lds_store_tier<64>(gl_ld_aligned<18>(ga, last_dw), a_out + dst, sh, step, dm);
                    }
                    bits_set(s_bits, dst, dst + step);
                    n -= step; src += step; dst += step;
                }
            }
        };
        // (the next batch's records are requested before the current batch is processed: a coalesced table read is a
        //  full global round trip and a wave owns only ~5 batches)
        const uint32_t i0 = wave * 64u + lane;
        const auto fetch_batch = [&](uint32_t b) { return rec_fetch(b < nrec_all ? b + lane : i0, nseq, nrec_all); };   // past the wave's last batch: its FIRST again — D3 starts on it
        uint4 raw_nx = rec_fetch(i0, nseq, nrec_all);
        // kRecFeed (round 6): the same pipeline with exact stores — the slab's literals come from HBM (nothing staged the input), and a
        // batch that requests its bytes only when it is placed is a global round trip per batch on the wave's chain (D2 43.8 k of a
        // slab's 102 k cycles for 6.5 batches per wavefront, profiles/r06/experiments b01).
        constexpr bool kPipeD2 = kCompact || kRecFeed;
        if constexpr (kPipeD2) {
            // A sequence whose match this lane copies in D3 OWNS the bytes behind its literals (lds_store_own): up to 32 literal
            // bytes per sequence go the lean way, and their 32 source bytes are requested one batch ahead (two loads per record,
            // no branch around them: the wait for the batch that is placed must not cover the requests of the next one).
            struct OwnLit { uint32_t n, src, dst, nl; uint32_t v[8]; };
            const uint8_t* lit_base = iend >= 32u ? in : reinterpret_cast<const uint8_t*>(table);      // 32 readable bytes for the lanes that load nothing
            const auto own_issue = [&](const uint4& rec, OwnLit& L) {
                L.n = rec.y; L.src = rec.x; L.dst = rec.z - rec.y;
                // (slab records: the bytes behind the literals may belong to a cross copy that another wavefront has already made — exact stores there)
                const bool own = (!kCompact || ((rec.w & 0xffffu) != 0u && (rec.w >> 16) >= 4u)) && L.src + 32u <= safe_end && L.n < kD2LongRun;
                L.nl = own ? (L.n < 32u ? L.n : 32u) : 0u;
                const uint8_t* g = lit_base + (L.nl ? L.src : 0u);
                uint4 q0, q1;
                __builtin_memcpy(&q0, g, 16);
                __builtin_memcpy(&q1, g + 16, 16);
                L.v[0] = q0.x; L.v[1] = q0.y; L.v[2] = q0.z; L.v[3] = q0.w; L.v[4] = q1.x; L.v[5] = q1.y; L.v[6] = q1.z; L.v[7] = q1.w;
            };
            const auto own_place = [&](OwnLit& L) {
                if constexpr (kCompact) {
                    if (ballot64(L.nl > 16u)) lds_store_own<32>(L.v, a_out + L.dst, L.nl);
                    else lds_store_own<16>(L.v, a_out + L.dst, L.nl);
                } else if (ballot64(L.nl > 16u)) {
                    DW<10> r;
#pragma unroll
                    for (int i = 0; i < 8; i++) r.w[i] = L.v[i];
                    r.w[8] = 0u; r.w[9] = 0u;
                    lds_store_tier<32>(r, a_out + L.dst, 0u, L.nl, dm);
                } else {
                    DW<6> r;
#pragma unroll
                    for (int i = 0; i < 4; i++) r.w[i] = L.v[i];
                    r.w[4] = 0u; r.w[5] = 0u;
                    lds_store_tier<16>(r, a_out + L.dst, 0u, L.nl, dm);
                }
                bits_set32(s_bits, L.dst, L.nl);
                if (ballot64(L.n > L.nl)) place_from_global(L.n - L.nl, L.src + L.nl, L.dst + L.nl);
            };
            OwnLit A, B;                                     // (two named buffers, the loop unrolled by two: a copy would wait for the loads)
            own_issue(rec_view(raw_nx, i0, nseq, nrec_all), A);
            raw_nx = fetch_batch(wave * 64u + kL2Threads);
            for (uint32_t base = wave * 64u; base < nrec_all; base += 2u * kL2Threads) {
                own_issue(rec_view(raw_nx, base + kL2Threads + lane, nseq, nrec_all), B);
                raw_nx = fetch_batch(base + 2u * kL2Threads);
                own_place(A);
                if (base + kL2Threads >= nrec_all) break;
                own_issue(rec_view(raw_nx, base + 2u * kL2Threads + lane, nseq, nrec_all), A);
                raw_nx = fetch_batch(base + 3u * kL2Threads);
                own_place(B);
            }
        } else
        for (uint32_t base = wave * 64u; base < nrec_all; base += kL2Threads) {
            const uint4 rec = rec_view(raw_nx, base + lane, nseq, nrec_all);
            raw_nx = fetch_batch(base + kL2Threads);
            place_from_global(rec.y, rec.x, rec.z - rec.y);
        }
        // kSlab — D2b: the parts of matches whose source lies before this slab come from the finished output of the earlier
        // slabs in global memory.  Each wave copies its share of the cross list as soon as it sees slab c-1's flag: from
        // inside D3's wait loop (non-blocking poll while none of its matches is ready) or, blocking, after its last batch —
        // so everything that does not depend on earlier slabs is resolved while the predecessor is still running.
        bool cross_done = true, prev_seen = false;
        uint32_t ncross = 0;
        if constexpr (kSlab) { ncross = *s_ncross; cross_done = ncross <= wave * 64u; }      // complete since the barrier after D1
        const auto try_cross = [&](bool block) {
            if constexpr (kSlab) {
                // one relaxed poll -> ONE agent-scope acquire per workgroup (it invalidates the CU's L1, which all its waves share,
                // and costs µs): the wave that sees the flag first fences and tells the others through LDS
                if (*s_prevok == 0u) {
                    uint32_t f = 0;
                    for (;;) {
                        if (lane == 0) f = __hip_atomic_load(&sl.done[c - sl.prev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        f = rdlane(f, 0);
                        if (f != 0u || !block || *s_prevok != 0u) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (f == 0u && *s_prevok == 0u) return;
                    if (*s_prevok == 0u) {
                        CJ_TRACE(0);                                       // flag seen
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (lane == 0) *s_prevok = 1u;
                        CJ_TRACE(1);                                       // acquire done
                    }
                }
                const uint4* cross = sl.cross + (size_t)blockIdx.x * sl.cross_stride;
                for (uint32_t base = wave * 64u; base < ncross; base += kL2Threads) {
                    uint4 e = make_uint4(0, 0, 0, 0);
                    if (base + lane < ncross) e = cross[base + lane];
                    const uint8_t* g = a.out_base + (((uint64_t)e.y << 32) | e.x);
                    uint32_t n = e.w, dst = e.z;
                    uint64_t lm = ballot64(n >= kLongRun);
                    while (lm) {
                        const uint32_t l = ctz64(lm);
                        lm &= lm - 1ull;
                        const uint32_t ln = rdlane(n, l), ld = rdlane(dst, l);
                        const uint8_t* lg = reinterpret_cast<const uint8_t*>(((uint64_t)rdlane((uint32_t)((uintptr_t)g >> 32), l) << 32) | rdlane((uint32_t)(uintptr_t)g, l));
                        wave_copy_to_lds(a_out + ld, lg, ln);
                        wave_bits_set(s_bits, ld, ld + ln);
                        if (lane == l) n = 0;
                    }
                    while (ballot64(n > 0u)) {                 // <=16 bytes per pass (register budget: this sits inside D3's loop; 32-byte
                        const uint32_t step = n < 16u ? n : 16u;   // passes measured no faster); the output buffer is padded, over-reads are harmless
                        if (step > 0u) {
                            lds_store_tier<16>(gl_ld_exact<16>(g), a_out + dst, 0u, step, dm);
                            bits_set(s_bits, dst, dst + step);
                            n -= step; g += step; dst += step;
                        }
                    }
                }
                cross_done = true;
                if (wave == 0u) CJ_TRACE(2);                                   // wave 0's cross share copied
            }
        };
        CJ_PHASE_MARK(2);
        // Batches: the wavefronts that stage, expand and place literals (one serial instruction stream each, ~15 cycles per instruction
        // with four wavefronts on a SIMD) issue ahead of the wavefronts that poll in D3 — the polls fill the gaps that remain.
        // profiles/r04/experiments p01: D1 11.6 k -> 10.6 k, D2 23.9 k -> 21.6 k, D3 52.9 k -> 54.5 k cycles per chunk, 617.6 -> 627.4 GB/s
        // (the opposite assignment: 611 GB/s).
        if constexpr (!kSlab && !kLinked) __builtin_amdgcn_s_setprio(0);

        // ---- D3: matches (same resolver as variant 1).  No barrier after D2: readiness is exact per byte through the
        //      bitmap, so a wave starts on its matches while other waves are still placing literals ----
        // kSlab: a wave must not sit on a batch whose lanes wait for bytes of an EARLIER slab while its later batches could be
        // resolved: as long as the predecessor's flag has not been seen (or something is already deferred), a batch that
        // made no progress for kPatience polls is abandoned — its pending lanes go to the wave's deferred list (global
        // memory, record indices in increasing order) and are finished in a second pass after the cross copy.  Progress:
        // pass 1 always terminates; in pass 2 the lowest unfinished record of the slab is always in the batch its wave is
        // standing on (the lists are sorted and a record waits only for records before it).
        uint32_t* dlist = nullptr;
        uint32_t ndef = 0, dpos = 0;
        bool pass2 = false;
        if constexpr (kSlab) dlist = sl.defer + (size_t)blockIdx.x * sl.defer_stride + (size_t)wave * (sl.defer_stride / (kL2Threads / 64u));
        // ---- D3, batches of independent chunks: the poll step as ONE hand-scheduled block.  The resolver is bound by how long a
        //      trip round the poll loop takes (the dependency chain is 21-33 levels deep on the benchmark data and every level
        //      costs a producer's publish + a consumer's poll + its copy), and the loop below this one compiles to ~100
        //      instructions per trip, most of them mask bookkeeping.  Here a trip is: bitmap words of the waiting lanes (exec =
        //      waiting), ready test (two v_bfi + v_or + v_cmpx), and for the ready lanes the copy as first + last 8 (4) bytes at
        //      their exact addresses — few lanes are active, and gfx950 charges a misaligned DS access per ACTIVE lane —, the
        //      publish (two ds_or) and the mask update: ~12 instructions when nothing is ready, ~35 with copies.
        //      Lanes outside the fast shape (longer than 32 bytes, self-overlapping, 1-3 bytes) keep the general path.
        if constexpr (!kSlab && !kLinked) {
            // (one batch per wave at a time.  Two — the stuck batch stays in its slot while the other slot runs through the wave's following
            //  batches — was measured: D3 56 k -> 65 k cycles per chunk on the benchmark data, 266 k -> 299 k on the corpus; a trip that polls
            //  two slots takes twice as long, and the trip time is what a level of the dependency chain costs.  profiles/r04/experiments d01)
            struct D3Slot {
                uint32_t pa, pm0, pm1, qa, qm0, qm1, as0, as1, as3, ad0, ad1, ad3;     // fast lanes: poll word + mask, publish word + mask, copy plan
                uint32_t dst, off, m;                                                  // the other lanes
                bool spend;
                uint64_t mp, mA, mB, mC;                                               // fast lanes still waiting; by copy shape
                bool any_slow, live;
            };
            const auto d3_fill = [&](D3Slot& S, const uint4& rec) __attribute__((always_inline)) {
                const uint32_t dst = rec.z, off = rec.w & 0xffffu, m = rec.w >> 16;
                const uint32_t src = dst - off;
                const uint32_t need = off < m ? off : m;
                const bool pending = m > 0u;
                const bool fast = pending && m >= 4u && m <= 32u && off >= m;
                S.pa = 0; S.pm0 = 0; S.pm1 = 0; S.qa = 0; S.qm0 = 0; S.qm1 = 0;
                if (fast) {
#if CJ_D3_MASK64
                    // (need, m in 4 .. 32: the run of bits as one 64-bit shift — two instructions per mask instead of nine)
                    const uint64_t pmask = (~0ull >> (64u - need)) << (src & 31u), qmask = (~0ull >> (64u - m)) << (dst & 31u);
                    S.pa = (uint32_t)(uintptr_t)(s_bits + (src >> 5));
                    S.pm0 = (uint32_t)pmask; S.pm1 = (uint32_t)(pmask >> 32);
                    S.qa = (uint32_t)(uintptr_t)(s_bits + (dst >> 5));
                    S.qm0 = (uint32_t)qmask; S.qm1 = (uint32_t)(qmask >> 32);
#else
                    const uint32_t sh = src & 31u, e = sh + need;
                    S.pa = (uint32_t)(uintptr_t)(s_bits + (src >> 5));
                    S.pm0 = (e >= 32u ? ~0u : ((1u << e) - 1u)) & (~0u << sh);
                    S.pm1 = e > 32u ? ((1u << (e - 32u)) - 1u) : 0u;
                    const uint32_t dh = dst & 31u, de = dh + m;
                    S.qa = (uint32_t)(uintptr_t)(s_bits + (dst >> 5));
                    S.qm0 = (de >= 32u ? ~0u : ((1u << de) - 1u)) & (~0u << dh);
                    S.qm1 = de > 32u ? ((1u << (de - 32u)) - 1u) : 0u;
#endif
                }
                // copy plan: pieces at [0] and [m - 8] (m >= 8; 8 bytes each), or [0] and [m - 4] (m < 8; 4 bytes each); m > 16: also [8], [m - 16]
                S.as0 = a_out + src; S.ad0 = a_out + dst;
                const uint32_t o1 = m >= 8u ? m - 8u : m - 4u, o3 = m > 16u ? m - 16u : 0u;
                S.as1 = S.as0 + o1; S.ad1 = S.ad0 + o1; S.as3 = S.as0 + o3; S.ad3 = S.ad0 + o3;
                S.dst = dst; S.off = off; S.m = m;
                S.mp = ballot64(fast);
                S.mA = ballot64(fast && m > 16u); S.mB = ballot64(fast && m >= 8u); S.mC = ballot64(fast && m < 8u);
                S.spend = pending && !fast;
                S.any_slow = ballot64(S.spend) != 0ull;
                S.live = S.mp != 0ull || S.any_slow;
            };
            uint32_t spins = 0;
            // The FIRST look at a batch finds most of its 64 matches ready (two thirds of a chunk's matches are copied in trips with 16
            // lanes or more), and the pipe charges a misaligned access by its active lanes — 8 x (lanes + 1) cycles for the sparse copy
            // plan below.  With kDenseLanes lanes or more the copy is done with aligned accesses instead (lds_copy_dense); later trips
            // find a handful of lanes and keep the exact-address plan.  (threshold 16 / 24 / 32 / 40 lanes: 689.4 / 689.3 / 689.1 / 685.1 GB/s against 668
            // without it; the same test inside the asm loop — every trip — gave 681 and cost the corpus 1.5 %.)
            const auto d3_first = [&](D3Slot& S) __attribute__((always_inline)) {
                const bool f = ((S.mp >> lane) & 1ull) != 0ull;
                bool ready = false;
                if (f) { const uint2 w = lds_ld64(S.pa); ready = (w.x & S.pm0) == S.pm0 && (w.y & S.pm1) == S.pm1; }
                const uint64_t rm = ballot64(ready);
                if ((uint32_t)__builtin_popcountll(rm) >= CJ_DENSE_LANES) {
                    lds_copy_dense(S.as0, S.ad0, S.m, ready && S.m >= 8u);
                    if (ready && S.m < 8u) lds_copy_sparse(S.ad0, S.as0, S.m);
                    if (ready) asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4" :: "v"(S.qa), "v"(S.qm0), "v"(S.qm1) : "memory");
                    S.mp &= ~rm;
                    S.live = S.mp != 0ull || S.any_slow;
                }
            };
            const auto d3_step = [&](D3Slot& S) __attribute__((always_inline)) {
                if (S.mp != 0ull) {
                    // The poll LOOP as one block: it leaves only when no fast lane waits any more or after `lim` trips that found
                    // nothing (1 while the batch has lanes of the general path, which must be looked after between trips).
                    // Fixed registers: v[112:113] the bitmap words, v114/v115 scratch, v[116:123] the four 8-byte pieces — a piece's low
                    // half is what a 4..7-byte match stores, and inline asm cannot name half of an operand.
                    //   every ready lane: pieces [0] and [m - 8] (m >= 8) or [m - 4] (m < 8; the 8-byte read runs 4 bytes past the
                    //   source, harmless); m > 16: also [8] and [m - 16]
                    uint64_t sv, sr;
                    uint32_t ic;
                    const uint32_t lim = S.any_slow ? 1u : 1024u;          // (2 / 4 / 8 trips between two looks at the general-path lanes: corpus 177.4 -> 176.8 / 175.9 / 173.2 GB/s)
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b32 %[ic], 0\n"
                        "0:\n\t"
                        "s_mov_b64 exec, %[mp]\n\t"
                        "ds_read2_b32 v[112:113], %[pa] offset1:1\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_bfi_b32 v114, v112, 0, %[pm0]\n\t"              // pm0 & ~w0: needed bits that are not set yet
                        "v_bfi_b32 v115, v113, 0, %[pm1]\n\t"
                        "v_or_b32 v114, v114, v115\n\t"
                        "v_cmpx_eq_u32 0, v114\n\t"                        // exec = ready lanes
                        "s_cbranch_execz 1f\n\t"
                        "s_mov_b64 %[sr], exec\n\t"
                        "ds_read_b64 v[116:117], %[as0]\n\t"
                        "ds_read_b64 v[118:119], %[as1]\n\t"
                        "s_and_b64 exec, %[sr], %[mA]\n\t"                   // 17..32 bytes: the two middle pieces
                        "ds_read_b64 v[120:121], %[as0] offset:8\n\t"
                        "ds_read_b64 v[122:123], %[as3]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b64 %[ad0], v[120:121] offset:8\n\t"
                        "ds_write_b64 %[ad3], v[122:123]\n\t"
                        "s_and_b64 exec, %[sr], %[mB]\n\t"                   // 8..32 bytes: first and last 8
                        "ds_write_b64 %[ad0], v[116:117]\n\t"
                        "ds_write_b64 %[ad1], v[118:119]\n\t"
                        "s_and_b64 exec, %[sr], %[mC]\n\t"                   // 4..7 bytes: first and last 4
                        "ds_write_b32 %[ad0], v116\n\t"
                        "ds_write_b32 %[ad1], v118\n\t"
                        "s_mov_b64 exec, %[sr]\n\t"
                        "ds_or_b32 %[qa], %[qm0]\n\t"                        // publish: behind the copy's writes in the wave's DS queue
                        "ds_or_b32 %[qa], %[qm1] offset:4\n\t"
                        "s_andn2_b64 %[mp], %[mp], %[sr]\n\t"
                        "s_cbranch_scc1 0b\n\t"                              // lanes left: poll again
                        "s_branch 2f\n"
                        "1:\n\t"
                        "s_add_u32 %[ic], %[ic], 1\n\t"
                        "s_cmp_lt_u32 %[ic], %[lim]\n\t"
                        "s_cbranch_scc1 0b\n"
                        "2:\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [mp] "+s"(S.mp), [sv] "=&s"(sv), [sr] "=&s"(sr), [ic] "=&s"(ic)
                        : [pa] "v"(S.pa), [pm0] "v"(S.pm0), [pm1] "v"(S.pm1), [qa] "v"(S.qa), [qm0] "v"(S.qm0), [qm1] "v"(S.qm1),
                          [as0] "v"(S.as0), [as1] "v"(S.as1), [as3] "v"(S.as3), [ad0] "v"(S.ad0), [ad1] "v"(S.ad1), [ad3] "v"(S.ad3),
                          [mA] "s"(S.mA), [mB] "s"(S.mB), [mC] "s"(S.mC), [lim] "s"(lim)
                        : "memory", "vcc", "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
                    spins += ic;
                }
                if (S.any_slow) {
                    // (the values pass through an empty asm: left alone, the compiler hoists everything the copy routines below derive
                    //  from them — ~200 instructions of store addresses and tail selectors — out of the poll loop into the batch
                    //  setup, where EVERY batch pays for them, with or without such a lane)
                    uint32_t dst = S.dst, off = S.off, m = S.m, as0 = S.as0, ad0 = S.ad0;
                    asm volatile("" : "+v"(dst), "+v"(off), "+v"(m), "+v"(as0), "+v"(ad0));
                    const uint32_t src = dst - off, need = off < m ? off : m;
                    bool sready = false;
                    if (S.spend) sready = bits_ready(s_bits, src, src + need);
                    if (sready && m < kLongRun) {
                        lds_copy_serial(as0, ad0, off, m);
                        bits_set(s_bits, dst, dst + m);
                        S.spend = false;
                    }
                    uint64_t longm = ballot64(sready && m >= kLongRun);
                    while (longm) {
                        const uint32_t l = ctz64(longm);
                        longm &= longm - 1ull;
                        const uint32_t lmm = rdlane(m, l), lo = rdlane(off, l), ld = rdlane(dst, l);
                        const uint8_t* sb = s_out + (ld - lo);
                        if (lo == 1u || lo == 2u || lo == 4u) {
                            const uint32_t h = (0u - ld) & 15u, hh = h < lmm ? h : lmm;
                            if (lane < hh) s_out[ld + lane] = sb[lane % lo];
                            uint32_t wv = 0;
#pragma unroll
                            for (uint32_t i = 0; i < 4u; i++) wv |= (uint32_t)sb[(hh + i) % lo] << (8u * i);
                            const uint32_t nv = (lmm - hh) >> 4;
                            uint4* dv = reinterpret_cast<uint4*>(s_out + ld + hh);
                            for (uint32_t q = lane; q < nv; q += 64u) dv[q] = make_uint4(wv, wv, wv, wv);
                            const uint32_t t0 = hh + (nv << 4);
                            if (t0 + lane < lmm) s_out[ld + t0 + lane] = sb[(t0 + lane) % lo];
                        } else {
                            uint32_t rr = lane, step = 64u;
                            if (lo <= 64u) { rr = lane % lo; step = 64u % lo; }
                            for (uint32_t k = lane; k < lmm; k += 64u) {
                                s_out[ld + k] = sb[lo >= lmm ? k : rr];
                                rr += step;
                                if (rr >= lo) rr -= lo;
                            }
                        }
                        wave_bits_set(s_bits, ld, ld + lmm);
                        if (lane == l) S.spend = false;
                    }
                    S.any_slow = ballot64(S.spend) != 0ull;
                }
                S.live = S.mp != 0ull || S.any_slow;
            };
            D3Slot A;
            for (uint32_t base = wave * 64u; base < nrec_all; base += kL2Threads) {
                d3_fill(A, rec_view(raw_nx, base + lane, nseq, nrec_all));
                spins = 0;
                if (A.mp != 0ull) d3_first(A);
                if (base + kL2Threads < nrec_all) raw_nx = rec_fetch(base + kL2Threads + lane, nseq, nrec_all);
                while (A.live) {
                    d3_step(A);
                    if (++spins > kSpinLimit) { *s_fail = 1u; break; }
                }
            }
        } else
        // raw_nx = the wave's first batch (requested by D2's last iteration, or by D2's prologue if the wave has no batch)
        for (uint32_t base = wave * 64u;; base += kL2Threads) {
            uint4 rec;
            uint32_t ridx = base + lane;
            if constexpr (kSlab) {
                if (!pass2 && base >= nrec_all) { pass2 = true; if (!cross_done) try_cross(true); }
                if (pass2) {
                    if (dpos >= ndef) break;
                    rec = make_uint4(0, 0, 0, 0);
                    if (dpos + lane < ndef) { ridx = dlist[dpos + lane]; rec = table[ridx]; }
                    dpos += 64u;
                } else {
                    rec = rec_view(raw_nx, base + lane, nseq, nrec_all);
                    if (base + kL2Threads < nrec_all) raw_nx = rec_fetch(base + kL2Threads + lane, nseq, nrec_all);
                }
            } else {
                if (base >= nrec_all) break;
                rec = rec_view(raw_nx, base + lane, nseq, nrec_all);
                if (base + kL2Threads < nrec_all) raw_nx = rec_fetch(base + kL2Threads + lane, nseq, nrec_all);
            }
            const uint32_t dst = rec.z, off = rec.w & 0xffffu, m = rec.w >> 16;
            const uint32_t src = dst - off;                   // kLinked: "negative" (wraps) when the source starts in the previous block
            const uint32_t need = off < m ? off : m;
            bool pending = m > 0u && !(kSlab && off > dst);      // kSlab: off > dst = copied from earlier slabs (cross list)
            // kLinked: cross = source starts in the previous window; cross_full = it also ends there (final bytes, no polling)
            const bool cross = kLinked && off > dst;
            const bool cross_full = cross && off - dst >= m;
            const uint32_t asrc = cross ? a_prev + 65536u - (off - dst) : a_out + src;      // LDS address of the first source byte
            const bool fast = pending && m <= 32u && off >= m && (!cross || cross_full);
            uint32_t pa = 0, pm0 = 0, pm1 = 0, qa = 0, qm0 = 0, qm1 = 0;
            if (fast) {
                const uint32_t sh = src & 31u, e = sh + need;
                pa = (uint32_t)(uintptr_t)(s_bits + (cross ? 0u : (src >> 5)));
                pm0 = cross ? 0u : (e >= 32u ? ~0u : ((1u << e) - 1u)) & (~0u << sh);
                pm1 = cross ? 0u : (e > 32u ? ((1u << (e - 32u)) - 1u) : 0u);
                const uint32_t dh = dst & 31u, de = dh + m;
                qa = (uint32_t)(uintptr_t)(s_bits + (dst >> 5));
                qm0 = (de >= 32u ? ~0u : ((1u << de) - 1u)) & (~0u << dh);
                qm1 = de > 32u ? ((1u << (de - 32u)) - 1u) : 0u;
            }
            const bool any_slow = ballot64(pending && !fast) != 0ull;
            uint32_t spins = 0, idle = 0, stall = 0;
            uint64_t prev_mask = 0ull;
            while (ballot64(pending) != 0ull) {
                bool ready = false;
                if (pending && fast) {
                    const uint2 w = lds_ld64(pa);
                    ready = ((w.x & pm0) == pm0) && ((w.y & pm1) == pm1);
                }
                const uint64_t rm = ballot64(ready);
                if (rm != 0ull) {
                    if (ready) {
                        lds_copy_sparse(a_out + dst, asrc, m);
                        asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4" :: "v"(qa), "v"(qm0), "v"(qm1) : "memory");
                        pending = false;
                    }
                }
                if (any_slow) {
                    bool sready = false;
                    if (pending && !fast) {
                        if (!cross) sready = bits_ready(s_bits, src, src + need);
                        else {                                   // bytes in the previous window are final; the part in this window must be ready
                            const uint32_t own = off - dst >= need ? 0u : need - (off - dst);
                            sready = own == 0u || bits_ready(s_bits, 0u, own);
                        }
                    }
                    // a match that straddles the block start (a handful per block at most) = its first `back` bytes from the
                    // previous window + an ordinary, possibly self-overlapping, match whose source starts at window offset 0:
                    // the whole wavefront copies both parts
                    uint64_t strad = ballot64(sready && cross && !cross_full);
                    while (strad) {
                        const uint32_t l = ctz64(strad);
                        strad &= strad - 1ull;
                        const uint32_t lmm = rdlane(m, l), lo = rdlane(off, l), ld = rdlane(dst, l);
                        const uint32_t back = lo - ld, rem = lmm - back, d2 = ld + back;
                        for (uint32_t q = lane; q < back; q += 64u) lds_st8(a_out + ld + q, lds_ld8(a_prev + 65536u - back + q));
                        uint32_t rr = lane, step = 64u;
                        if (lo <= 64u) { rr = lane % lo; step = 64u % lo; }
                        for (uint32_t q = lane; q < rem; q += 64u) {           // periodic read of the lo bytes that start the window
                            s_out[d2 + q] = s_out[lo >= rem ? q : rr];
                            rr += step;
                            if (rr >= lo) rr -= lo;
                        }
                        wave_bits_set(s_bits, ld, ld + lmm);
                        if (lane == l) pending = false;
                    }
                    if (sready && cross && !cross_full) {
                        // done above
                    } else if (sready && m < kLongRun) {
                        lds_copy_serial(asrc, a_out + dst, off, m);
                        bits_set(s_bits, dst, dst + m);
                        pending = false;
                    }
                    uint64_t longm = ballot64(sready && m >= kLongRun && !(cross && !cross_full));
                    while (longm) {
                        const uint32_t l = ctz64(longm);
                        longm &= longm - 1ull;
                        const uint32_t lmm = rdlane(m, l), lo = rdlane(off, l), ld = rdlane(dst, l);
                        const uint8_t* sb = (kLinked && lo > ld) ? smem + (a_prev - (uint32_t)(uintptr_t)smem) + 65536u - (lo - ld) : s_out + (ld - lo);
                        if (lo == 1u || lo == 2u || lo == 4u) {
                            // run of a 1/2/4-byte pattern (zero fill, padding): every 16-byte aligned vector of the run holds the
                            // same bytes, so the body is written 16 bytes per lane instead of one
                            const uint32_t h = (0u - ld) & 15u, hh = h < lmm ? h : lmm;
                            if (lane < hh) s_out[ld + lane] = sb[lane % lo];
                            uint32_t w = 0;
#pragma unroll
                            for (uint32_t i = 0; i < 4u; i++) w |= (uint32_t)sb[(hh + i) % lo] << (8u * i);
                            const uint32_t nv = (lmm - hh) >> 4;
                            uint4* dv = reinterpret_cast<uint4*>(s_out + ld + hh);
                            for (uint32_t q = lane; q < nv; q += 64u) dv[q] = make_uint4(w, w, w, w);
                            const uint32_t t0 = hh + (nv << 4);
                            if (t0 + lane < lmm) s_out[ld + t0 + lane] = sb[(t0 + lane) % lo];
                        } else {
                        uint32_t rr = lane, step = 64u;
                        if (lo <= 64u) { rr = lane % lo; step = 64u % lo; }
                        for (uint32_t k = lane; k < lmm; k += 64u) {
                            s_out[ld + k] = sb[lo >= lmm ? k : rr];
                            rr += step;
                            if (rr >= lo) rr -= lo;
                        }
                        }
                        wave_bits_set(s_bits, ld, ld + lmm);
                        if (lane == l) pending = false;
                    }
                }
                if constexpr (kSlab) {
                    if (!cross_done) {                             // waiting on an earlier slab is not a stall of this one
                        spins = 0;
                        if (rm == 0ull && (++idle & CJ_SLAB_POLL_MASK) == 0u) try_cross(false);
                    }
                    if (!pass2 && (!cross_done || ndef > 0u)) {
                        const uint64_t pmask = ballot64(pending);
                        if (pmask != prev_mask) { prev_mask = pmask; stall = 0; }
                        else if (++stall > kSlabPatience) {        // abandon: the pending lanes are finished in pass 2
                            if (pending) dlist[ndef + (uint32_t)__popcll(pmask & ((1ull << lane) - 1ull))] = ridx;
                            ndef += (uint32_t)__popcll(pmask);
                            pending = false;
                        }
                    }
                }
                if (++spins > kSpinLimit) {
                    if constexpr (kSlab) {                         // a long wait is legitimate while the previous slab is still running
                        if (has_prev && !prev_seen) {
                            uint32_t f = 0;
                            if (lane == 0) f = __hip_atomic_load(&sl.done[c - sl.prev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            prev_seen = rdlane(f, 0) != 0u;
                            spins = 0;
                            continue;
                        }
                    }
                    *s_fail = 1u; break;
                }
            }
        }
        __syncthreads();
        CJ_TRACE_T0(3);                                         // D3 done
        CJ_PHASE_MARK(3);
        if constexpr (!kSlab && !kLinked) __builtin_amdgcn_s_setprio(2);
        // ---- D4: stream the window out (16 B per lane), exact tail ----
        {
            const uint32_t nvec = U >> 4;
            const uint4* src = reinterpret_cast<const uint4*>(s_out);
            if constexpr (kSlab) {
                for (uint32_t i = tid; i < nvec; i += kL2Threads) st16u_wt(out + 16u * i, src[i]);   // another workgroup waits for these bytes: write-through
            } else
            for (uint32_t i = tid; i < nvec; i += kL2Threads) st16u_nt(out + 16u * i, src[i]);   // streamed out, never re-read: keep L2 for the record tables
            for (uint32_t i = (nvec << 4) + tid; i < U; i += kL2Threads) out[i] = s_out[i];
            if constexpr (kSlab) wt_tail = (U & 15u) != 0u;
        }
        if (tid == 0 && *s_fail) a.result[c] = CJ_E_CORRUPT;    // cannot happen for a stream the parse kernel accepted
        publish();
        if (prof) {
            __syncthreads(); CJ_PHASE_MARK(4);
            if (tid < 16u) { const uint32_t v = tid == 5u ? 1u : s_prof[tid]; if (v) atomicAdd(&g_lds_phase_cycles[tid], (unsigned long long)v); s_prof[tid] = 0u; }
        }
    }
}

template <int kCodec, bool kLinked = false, uint32_t kWin = CJ_L2_WINDOW, uint32_t kThreads = CJ_L2_THREADS>
__global__ __launch_bounds__(kThreads) CJ_L2_ATTR void lz4_decode_lds2_kernel(BatchArgs a, const uint2* sync, const ParseMeta* meta,
                                                                     uint4* tabs, uint32_t* counter,
                                                                     const uint2* frames, uint32_t n_frames) {
    lds2_body<kCodec, kLinked, false, false, false, kWin, kThreads>(a, sync, meta, tabs, counter, frames, n_frames, SlabArgs{nullptr, nullptr, 0u, 0u, 0u, nullptr, 0u});
}

// the slab mode carries the cross-list copy inside D3's loop: capped at 128 VGPRs so that two workgroups still share a CU
template <int kCodec>
__global__ __launch_bounds__(kL2Threads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lz4_decode_slabs_kernel(
        BatchArgs a, const uint2* sync, const ParseMeta* meta, uint4* tabs, uint32_t* counter, const uint2* first, uint32_t stream_len, SlabArgs sl) {
    lds2_body<kCodec, false, true>(a, sync, meta, tabs, counter, first, stream_len, sl);
}

// the slabs of the big chunks of a device batch, records fed by big_parse_kernel (kRecFeed)
template <int kCodec>
__global__ __launch_bounds__(kBigSlabThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lz4_decode_bigslabs_kernel(
        BatchArgs a, const ParseMeta* meta, uint4* tabs, uint32_t* counter, SlabArgs sl, FeedArgs fd) {
    lds2_body<kCodec, false, true, false, true, kBigSlabBytes, kBigSlabThreads>(a, nullptr, meta, tabs, counter, nullptr, 0u, sl, fd);
}

// parse + decode in one kernel (batches of independent chunks).  meta: written here (kRouteWave for the chunks left to the wave kernel)
template <int kCodec, uint32_t kWin = CJ_L2_WINDOW, uint32_t kThreads = CJ_L2_THREADS>
__global__ __launch_bounds__(kThreads) CJ_L2_ATTR void lz4_decode_fused_kernel(BatchArgs a, ParseMeta* meta, uint4* tabs, uint32_t* counter) {
    lds2_body<kCodec, false, false, true, false, kWin, kThreads>(a, nullptr, meta, tabs, counter, nullptr, 0u, SlabArgs{nullptr, nullptr, 0u, 0u, 0u, nullptr, 0u});
}

template <int kCodec, uint32_t kWin, uint32_t kThreads>
static void launch_fused_window(const BatchArgs& a, void* meta, void* tabs, uint32_t* counter, uint32_t grid, hipStream_t s) {
    constexpr uint32_t bytes = lds2_bytes(kWin) + fused_aux_bytes(kThreads);
    static_assert((kWin >= 65536u ? 2u : kWin >= 32768u ? 4u : 8u) * bytes <= 163840u, "workgroups per CU");
    const auto k = lz4_decode_fused_kernel<kCodec, kWin, kThreads>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), bytes, s, a, (ParseMeta*)meta, (uint4*)tabs, counter);
}

// win: the window of this batch (64 KiB, or 32 / 16 KiB for batches of small chunks: four workgroups of four wavefronts / eight of two per CU,
// the parse on the workgroup's 256 / 128 lanes)
void launch_lz4_decode_fused(const BatchArgs& a, void* meta, void* tabs, uint32_t* counter, uint32_t grid, hipStream_t s, int codec, uint32_t win) {
    if (a.n_chunks == 0) return;
    const bool sn = codec == CJ_CODEC_SNAPPY_RAW;
    if (win >= 65536u) { if (sn) launch_fused_window<CJ_CODEC_SNAPPY_RAW, CJ_L2_WINDOW, CJ_L2_THREADS>(a, meta, tabs, counter, grid, s); else launch_fused_window<CJ_CODEC_LZ4_BLOCK, CJ_L2_WINDOW, CJ_L2_THREADS>(a, meta, tabs, counter, grid, s); }
    else if (win >= 32768u) { if (sn) launch_fused_window<CJ_CODEC_SNAPPY_RAW, 32768u, 256u>(a, meta, tabs, counter, grid, s); else launch_fused_window<CJ_CODEC_LZ4_BLOCK, 32768u, 256u>(a, meta, tabs, counter, grid, s); }
    else { if (sn) launch_fused_window<CJ_CODEC_SNAPPY_RAW, 16384u, 128u>(a, meta, tabs, counter, grid, s); else launch_fused_window<CJ_CODEC_LZ4_BLOCK, 16384u, 128u>(a, meta, tabs, counter, grid, s); }
}

// items: the slab work items' descriptors (a.n_chunks = kBigSlabs * cap of them), meta: their ParseMeta (nseq = records of the slab, 0 = nothing to do)
void launch_lz4_decode_big_slabs(const BatchArgs& items, const void* meta, const void* recs, const void* bigmeta, uint32_t cap, void* tabs, uint32_t* counter,
                                 uint32_t* done, void* cross, uint32_t tab_stride, uint32_t cross_stride, uint32_t grid, hipStream_t s, int codec) {
    if (items.n_chunks == 0) return;
    const uint32_t defer_stride = tab_stride + 8u * 64u;
    const SlabArgs sl = {done, (uint4*)cross, tab_stride, cross_stride, 0u,
                         reinterpret_cast<uint32_t*>((uint4*)cross + (size_t)grid * cross_stride), defer_stride, cap};
    const FeedArgs fd = {(const uint4*)recs, (const BigMeta*)bigmeta, cap};
    constexpr uint32_t bytes = lds2_bytes(kBigSlabBytes) + 3u * kBigLanes * 4u;          // + the regions' table (kRecFeed)
    static_assert(kBigSlabWgsPerCu * bytes <= 163840u, "workgroups per CU");
    if (codec == CJ_CODEC_SNAPPY_RAW) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_bigslabs_kernel<CJ_CODEC_SNAPPY_RAW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL((lz4_decode_bigslabs_kernel<CJ_CODEC_SNAPPY_RAW>), dim3(grid), dim3(kBigSlabThreads), bytes, s, items, (const ParseMeta*)meta, (uint4*)tabs, counter, sl, fd);
        return;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_bigslabs_kernel<CJ_CODEC_LZ4_BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL((lz4_decode_bigslabs_kernel<CJ_CODEC_LZ4_BLOCK>), dim3(grid), dim3(kBigSlabThreads), bytes, s, items, (const ParseMeta*)meta, (uint4*)tabs, counter, sl, fd);
}

size_t lz4_lds2_tab_bytes(uint32_t grid, uint32_t win) { return (size_t)grid * lds2_tab_records(win) * sizeof(uint4); }
// workgroups per CU by window: what 160 KiB of LDS hold, at 16 wavefronts of 128 registers per CU either way
uint32_t lz4_lds2_wgs_per_cu(uint32_t win) { return win >= 65536u ? 2u : win >= 32768u ? 4u : 8u; }

template <int kCodec, uint32_t kWin, uint32_t kThreads>
static void launch_lds2_window(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter, uint32_t grid, hipStream_t s) {
#ifndef CJ_L2_LDS_PAD
#define CJ_L2_LDS_PAD 0u                  // (tuning variants: unused LDS per workgroup, to hold the workgroups per CU below what the window alone allows)
#endif
    constexpr uint32_t bytes = lds2_bytes(kWin) + CJ_L2_LDS_PAD;
    const auto k = lz4_decode_lds2_kernel<kCodec, false, kWin, kThreads>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), bytes, s, a, (const uint2*)sync, (const ParseMeta*)meta, (uint4*)tabs, counter, (const uint2*)nullptr, 0u);
}

// win: the window of this batch (lds_window(flags): 64 KiB, or 32 / 16 KiB for batches of small chunks — four workgroups of four
// wavefronts / eight of two per CU; profiles/r06/experiments h01-h04)
void launch_lz4_decode_lds2(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter,
                            uint32_t grid, hipStream_t s, int codec, uint32_t win) {
    if (a.n_chunks == 0) return;
    const bool sn = codec == CJ_CODEC_SNAPPY_RAW;
    if (win >= 65536u) { if (sn) launch_lds2_window<CJ_CODEC_SNAPPY_RAW, CJ_L2_WINDOW, CJ_L2_THREADS>(a, sync, meta, tabs, counter, grid, s); else launch_lds2_window<CJ_CODEC_LZ4_BLOCK, CJ_L2_WINDOW, CJ_L2_THREADS>(a, sync, meta, tabs, counter, grid, s); }
    else if (win >= 32768u) { if (sn) launch_lds2_window<CJ_CODEC_SNAPPY_RAW, 32768u, 256u>(a, sync, meta, tabs, counter, grid, s); else launch_lds2_window<CJ_CODEC_LZ4_BLOCK, 32768u, 256u>(a, sync, meta, tabs, counter, grid, s); }
    else { if (sn) launch_lds2_window<CJ_CODEC_SNAPPY_RAW, 16384u, 128u>(a, sync, meta, tabs, counter, grid, s); else launch_lds2_window<CJ_CODEC_LZ4_BLOCK, 16384u, 128u>(a, sync, meta, tabs, counter, grid, s); }
}

void launch_lz4_decode_lds2_linked(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter,
                                   const void* frames, uint32_t n_frames, uint32_t grid, hipStream_t s) {
    if (a.n_chunks == 0 || n_frames == 0) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_lds2_kernel<CJ_CODEC_LZ4_BLOCK, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL2LinkedBytes);
    hipLaunchKernelGGL((lz4_decode_lds2_kernel<CJ_CODEC_LZ4_BLOCK, true>), dim3(grid), dim3(kL2Threads), kL2LinkedBytes, s, a,
                       (const uint2*)sync, (const ParseMeta*)meta, (uint4*)tabs, counter, (const uint2*)frames, n_frames);
}

// large.hip: the slabs of one large stream.  tabs: grid * tab_stride records; cross: grid * cross_stride entries; done: one
// zeroed word per slab
void launch_lz4_decode_lds2_slabs(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter,
                                  const void* first, uint32_t stream_len, uint32_t* done, void* cross, uint32_t tab_stride,
                                  uint32_t cross_stride, uint32_t grid, hipStream_t s, int codec, bool rel) {
    if (a.n_chunks == 0) return;
    // the deferred lists (one per wave: at most its share of the records + a batch) follow the cross lists
    const uint32_t defer_stride = tab_stride + 8u * 64u;
    const SlabArgs sl = {done, (uint4*)cross, tab_stride, cross_stride, rel ? 1u : 0u,
                         reinterpret_cast<uint32_t*>((uint4*)cross + (size_t)grid * cross_stride), defer_stride};
    if (codec == CJ_CODEC_SNAPPY_RAW) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_slabs_kernel<CJ_CODEC_SNAPPY_RAW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL2Bytes);
        hipLaunchKernelGGL((lz4_decode_slabs_kernel<CJ_CODEC_SNAPPY_RAW>), dim3(grid), dim3(kL2Threads), kL2Bytes, s, a,
                           (const uint2*)sync, (const ParseMeta*)meta, (uint4*)tabs, counter, (const uint2*)first, stream_len, sl);
        return;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_slabs_kernel<CJ_CODEC_LZ4_BLOCK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL2Bytes);
    hipLaunchKernelGGL((lz4_decode_slabs_kernel<CJ_CODEC_LZ4_BLOCK>), dim3(grid), dim3(kL2Threads), kL2Bytes, s, a,
                       (const uint2*)sync, (const ParseMeta*)meta, (uint4*)tabs, counter, (const uint2*)first, stream_len, sl);
}


}  // namespace cj
#ifdef CJ_SLAB_TRACE
extern "C" CJ_API int cj_debug_slab_trace(unsigned long long* out, int n_slabs) {      // (variant builds only: tests/perf/slab_chain_trace.py)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cj::g_slab_trace), (size_t)n_slabs * 64) == hipSuccess ? 0 : -1;
}
#endif
extern "C" long long cj_debug_forwarded_chunks(int reset) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(cj::g_fwd_chunks), 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(cj::g_fwd_chunks), &z, 8) != hipSuccess) return -1; }
    return (long long)v;
}
extern "C" int cj_debug_fused_parse_paths(unsigned long long* out3, int reset) {      // chunks of the one-kernel path: P3 / P4 from the lists, walked P4, walked P3 + P4
    unsigned long long v[4] = {0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(cj::g_fused_paths), 32) != hipSuccess) return -1;
    out3[0] = v[0]; out3[1] = v[1]; out3[2] = v[2];
    if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cj::g_fused_paths), z, 32) != hipSuccess) return -1; }
    return 0;
}
extern "C" int cj_debug_lds_phase_cycles(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(cj::g_lds_phase_cycles), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cj::g_lds_phase_cycles), z, 128) != hipSuccess) return -1; }
    return 0;
}
namespace cj {
size_t lz4_lds_scratch_sync_bytes(size_t n_chunks) { return n_chunks * (size_t)kSyncPitch * sizeof(uint2); }
size_t lz4_lds_scratch_meta_bytes(size_t n_chunks) { return n_chunks * sizeof(ParseMeta); }

}  // namespace cj
