// lz4_decode_lvl.hip — the LEVEL-ORDERED workgroup decoder for batches of independent chunks (LZ4 *block* / Snappy *raw*):
// the chunk's 64 KiB output window in LDS, two persistent workgroups of eight wavefronts per CU, and every match copied in a
// DENSE pass of its dependency level instead of being polled for.  Same results as the other mappings (reference call sites
// /root/reference/src/lz4.rs:88,90,164,168, src/snappy.rs:57,106).
//
// Why (profiles/r02/experiments): the bitmap resolver of lz4_decode_lds.hip finds ~4 of 64 matches ready per poll
// and pays ~14 LDS-pipe cycles per match; it saturates the CU's LDS pipe, which then also stretches every dependent LDS read of
// the other phases to ~700 cycles.  A dense copy of 64 matches costs ~90 pipe cycles (tools/lds_mskor_probe.hip: ds_read2_b32 14,
// ds_write_b32 7, ds_mskor_b32 12 with 64 lanes at random addresses), a level costs ~150 cycles of latency (read -> write ->
// s_barrier), and nothing polls.  What it needs is every match's LEVEL first:
//   S0   stage the compressed chunk in the still unused window (16 B per lane)
//   D1   (after lz4_parse_kernel / snappy_parse_kernel) one thread per sync point re-walks 8 sequences in LDS and writes 8-byte
//        records {lit_src | lit << 16, start | offset << 16} to the workgroup's table in global memory (L2)     | or P: fused_parse
//   X    the POSITION INDEX in the window (free until D2): per 16 bytes of output one word of record-start / match-start bits
//        and a halfword "records that start before this granule | its first byte lies in a match" -> which record holds byte p,
//        and is p in that record's match part, in two LDS reads and a popcount
//   L    levels: the source bytes of match r touch the records [qa, qb] (two index lookups), level[r] = 1 + max level[qa..qb];
//        the eight wavefronts race through the batches of 64 records with "unknown" markers (the lowest unresolved record can
//        always be resolved: deadlock-free), counting the matches per level
//   K    exclusive scan of the counts; every match takes a slot of its level (one returning LDS atomic) and stores its
//        descriptor {dst | offset << 16, length | level << 16} there: the matches SORTED BY LEVEL, in global memory (L2)
//   D2   literals: one lane per record, global -> window, whole aligned dwords with byte-masked edges (ds_mskor_b32)
//   D3   matches, level by level: a level of more than 64 matches is copied by all wavefronts, one lane per match, and closed
//        with a barrier; a run of small levels (<= 64 matches together) is copied by wavefront 0 alone, level after level with
//        NO barrier (the DS operations of one wavefront execute in order); descriptors are prefetched one segment ahead
//   D4   stream the window out (non-temporal 16 B stores)
// A copy reads aligned dwords (ds_read2_b32), shifts them onto the destination's dword grid (v_alignbyte) and stores whole
// dwords; the first and the last dword of a run are byte-masked atomic stores (ds_mskor_b32: MEM = MEM & ~mask | data), so
// neighbouring runs written by other lanes of the same level never lose bytes.  Self-overlapping matches are a chain of
// non-overlapping copies whose distance doubles (tests/test_level_decoder_model.py models every phase against the oracle).
// Chunks this decoder cannot take (more than 1023 levels) are handed to the wavefront-per-chunk kernel like the parse stage's
// other leftovers.
#include "lvl_shared.hpp"
#include <type_traits>

namespace cj {

constexpr uint32_t kLvThreads = 512;
constexpr uint32_t kLvMaxRec = kSyncStride * kSyncEvery;       // records per chunk (the parse stage routes longer chunks elsewhere)
// LDS map (bytes from the start of the dynamic segment).  The window starts at 16: a copy may read up to 3 bytes in front of it.
constexpr uint32_t kLvOffWin = 16;
constexpr uint32_t kLvOffLvl = kLvOffWin;                       // scratch inside the window until D2: u16 level[16384]
constexpr uint32_t kLvOffA = kLvOffWin + 32768;                 //   u32 A[4096]: record-start bits | match-start bits << 16 per 16 B of output
constexpr uint32_t kLvOffC = kLvOffWin + 49152;                 //   u16 C[4096]: records starting before the granule | first byte in a match << 15
constexpr uint32_t kLvOffAux = kLvOffWin + 65536;
constexpr uint32_t kLvOffHist = kLvOffAux;                      // u32 hist / slot[1026]  (fused: the parse's 8 KiB of marks overlay hist + lstart)
constexpr uint32_t kLvOffStart = kLvOffHist + 4112;             // u16 lstart[1026]: first sorted entry of every level
constexpr uint32_t kLvOffVars = kLvOffAux + 8192;               // 64 B of variables, 64 dummy bytes, 256 B of dummy dwords
constexpr uint32_t kLvOffFused = kLvOffVars + 384 + 64;         // fused_parse's 6 KiB (behind the 64 bytes of phase counters)
constexpr uint32_t kLvBytes = kLvOffFused;                      // 74 128 B
constexpr uint32_t kLvBytesFused = kLvOffFused + kFusedAux;     // 80 272 B: two workgroups per CU (163 840 B)
static_assert(2u * kLvBytesFused <= 163840u, "two workgroups per CU");
static_assert(kLvOffStart + 2u * (kLvMaxLevel + 3u) <= kLvOffVars, "lstart fits in front of the variables");
constexpr uint32_t kLvSlotBytes = 24u * kLvMaxRec + 1024u;      // per workgroup in the table scratch: 8-byte records, then the two sorted lists
constexpr uint32_t kLvSortedOff = (kLvMaxRec + 64u) * 8u;
static_assert(kLvSortedOff + 2u * kLvMaxRec * 8u <= kLvSlotBytes, "records + sorted descriptors fit the slot");

__device__ unsigned long long g_lvl_phase_cycles[16];           // S0, D1/P, X, L, K, D2, D3, D4, chunks, levels, barriers of D3 (flag 0x1000)
#define CJ_LV_MARK(idx)                                                                 \
    do {                                                                                \
        if (prof && tid == 0) {                                                         \
            unsigned long long now_ = __builtin_readcyclecounter();                     \
            s_prof[idx] += (uint32_t)(now_ - t_prev);                                   \
            t_prev = now_;                                                              \
        }                                                                               \
    } while (0)


template <int kCodec, bool kFused>
__device__ __forceinline__ void lvl_body(const BatchArgs& a, const uint2* sync, ParseMeta* meta, uint8_t* tabs, uint32_t* counter) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_out = smem + kLvOffWin;
    const uint32_t a_smem = (uint32_t)(uintptr_t)smem;
    const uint32_t a_out = a_smem + kLvOffWin;
    const uint32_t a_lvl = a_smem + kLvOffLvl, a_A = a_smem + kLvOffA, a_C = a_smem + kLvOffC;
    const uint32_t a_hist = a_smem + kLvOffHist, a_start = a_smem + kLvOffStart;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + kLvOffHist);
    uint16_t* s_start = reinterpret_cast<uint16_t*>(smem + kLvOffStart);
    uint32_t* s_var = reinterpret_cast<uint32_t*>(smem + kLvOffVars);         // [0] chunk, [1] max level, [2] overflow, [3] near (unused), [4] matches
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t dummy_w = a_smem + kLvOffVars + 128u + 4u * lane;
    const Dummies dm = {a_smem + kLvOffVars + 64u + lane, dummy_w};
    uint8_t* slot = tabs + (size_t)blockIdx.x * kLvSlotBytes;
    uint2* table2 = reinterpret_cast<uint2*>(slot);
    uint2* sorted = reinterpret_cast<uint2*>(slot + kLvSortedOff);
    uint2* sorted_s = sorted + kLvMaxRec;
    const bool prof = (a.flags & 0x1000u) != 0;
    uint32_t* s_prof = reinterpret_cast<uint32_t*>(smem + kLvOffVars + 384u);      // phase counters, flushed once per chunk (a mark's atomicAdd would sit in front of the next barrier's vmcnt(0))
    if (prof && threadIdx.x < 16u) s_prof[threadIdx.x] = 0u;
    unsigned long long t_prev = prof ? __builtin_readcyclecounter() : 0ull;
    uint32_t next_c = 0;
    if (tid == 0) next_c = atomicAdd(counter, 1u);

    for (;;) {
        if (tid == 0) { s_var[0] = next_c; s_var[1] = 0u; s_var[2] = 0u; s_var[3] = 0u; }
        __syncthreads();                                     // also: the previous chunk's D4 has read the window
        const uint32_t c = s_var[0];
        __syncthreads();
        if (c >= a.n_chunks) break;
        if (tid == 0) next_c = atomicAdd(counter, 1u);
        ParseMeta pm = {1u, 0u};
        if constexpr (!kFused) pm = meta[c];
        const uint64_t d_in_off = a.in_off[c], d_in_len = a.in_len[c], d_out_off = a.out_off[c];
        const uint64_t d_result = kFused ? a.out_cap[c] : (uint64_t)a.result[c];
        asm volatile("" :: "v"(pm.nseq), "v"(pm.in_skip), "v"((uint32_t)d_in_off), "v"((uint32_t)d_in_len), "v"((uint32_t)d_out_off), "v"((uint32_t)d_result));
        uint32_t f_cap = 0;
        if constexpr (kFused) {
            // the prologue of the parse kernels (size prefix / length preamble, special cases); everything that is not a plain chunk
            // goes to the wavefront-per-chunk kernel, which names every error exactly
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
                    const uint32_t h0 = ld32u(in0), h1 = n64 > 4 ? ld32u(in0 + 4) : 0u;
                    while (hdr < (uint32_t)n64 && i < 5u) {
                        const uint32_t bb = (hdr < 4u ? h0 >> (8u * hdr) : h1 >> (8u * (hdr - 4u))) & 0xffu;
                        hdr += 1;
                        if (bb < 0x80u) { ulen |= (uint64_t)bb << shift; ok = true; break; }
                        ulen |= (uint64_t)(bb & 0x7fu) << shift;
                        shift += 7; i += 1;
                    }
                    if (!ok || ulen > 0xFFFFFFFFull || ulen > cap64 || ulen == 0 || ulen > kLdsOutMax || n64 - hdr > kLdsInMax || hdr == (uint32_t)n64) route = true;
                    skip = hdr; cap64 = ulen;
                }
            } else {
                const uint8_t* inp = in0;
                if (lz4_block_prologue(a.flags, inp, n64, cap64) != 0) route = true;
                else {
                    skip = (uint32_t)(inp - in0);
                    if (cap64 == 0 || n64 == 0 || cap64 > kLdsOutMax || n64 > kLdsInMax) route = true;
                }
            }
            if (route) { if (tid == 0) meta[c] = ParseMeta{0u, kRouteWave}; continue; }
            pm.in_skip = skip;
            f_cap = (uint32_t)cap64;
        }
        if (pm.nseq == 0u) continue;                         // error, empty, or routed to another kernel
        uint32_t nseq = pm.nseq;
        uint32_t U = (uint32_t)d_result;                     // decoded size, 1..65536 (kFused: set by the parse)
        const uint8_t* in = a.in_base + d_in_off + pm.in_skip;
        const uint32_t iend = (uint32_t)d_in_len - pm.in_skip;
        const uint8_t* in_al = in - (reinterpret_cast<uintptr_t>(in) & 3u);
        const uint8_t* last_dw = in_al + ((((uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u)) + iend - 1u) & ~3u);
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);
        const uint32_t safe_end = ((mis + iend + 15u) & ~15u) - mis;    // reads stay inside the 16 B granule of the last input byte
        uint8_t* out = a.out_base + d_out_off;
        const uint2* csync = sync + (size_t)c * kSyncPitch;
        const uint32_t nsp = (nseq + kSyncEvery - 1u) / kSyncEvery;

        // ---- S0: stage the compressed chunk in the still unused window ----
        uint2 p_first = make_uint2(0u, 0u);
        if constexpr (!kFused) p_first = csync[tid < nsp ? tid : 0u];
        {
            const uint4* src = reinterpret_cast<const uint4*>(in - mis);
            uint4* dst = reinterpret_cast<uint4*>(s_out);
            const uint32_t nvec = (mis + iend + 15u) >> 4;
            const auto group = [&](uint32_t i0) {
                const uint32_t last = nvec - 1u;
                const uint32_t i1 = i0 + kLvThreads, i2 = i0 + 2u * kLvThreads, i3 = i0 + 3u * kLvThreads, i4 = i0 + 4u * kLvThreads;
                const uint4 v0 = src[i0 < last ? i0 : last], v1 = src[i1 < last ? i1 : last], v2 = src[i2 < last ? i2 : last],
                            v3 = src[i3 < last ? i3 : last], v4 = src[i4 < last ? i4 : last];
                if (i0 < nvec) dst[i0] = v0;
                if (i1 < nvec) dst[i1] = v1;
                if (i2 < nvec) dst[i2] = v2;
                if (i3 < nvec) dst[i3] = v3;
                if (i4 < nvec) dst[i4] = v4;
            };
            if (nvec > 0u) group(tid);
            for (uint32_t i0 = tid + 5u * kLvThreads; i0 < nvec; i0 += 5u * kLvThreads) group(i0);
        }
        if constexpr (kFused) { for (uint32_t i = tid; i < 2048u; i += kLvThreads) s_hist[i] = 0u; }      // the parse's marks (8 KiB over hist + lstart)
        __syncthreads();
        CJ_LV_MARK(0);

        // ---- D1 / P: the chunk's records -> table2 ----
        const uint32_t a_in = a_out + mis;
        if constexpr (kFused) {
            using G = typename std::conditional<kCodec == CJ_CODEC_SNAPPY_RAW, SnappyGrammar, Lz4Grammar>::type;
            const bool ok = fused_parse<G, kLvThreads>(a_in, iend, f_cap, s_hist, reinterpret_cast<uint32_t*>(smem + kLvOffFused), table2, s_var + 3, nseq, U);
            if (!ok) { if (tid == 0) meta[c] = ParseMeta{0u, kRouteWave}; continue; }      // (uniform)
            if (tid == 0) { meta[c] = ParseMeta{0u, 0u}; a.result[c] = (int64_t)U; table2[nseq] = make_uint2(0u, U & 0xffffu); }
        } else if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) {
            const auto rd = [a_in](uint32_t p) { return lds_ld32a(a_in + p); };
            for (uint32_t sp = tid; sp < nsp; sp += kLvThreads) {
                const uint2 p = sp == tid ? p_first : csync[sp];
                uint32_t ip = p.x, op = p.y;
                uint32_t s = sp * kSyncEvery;
                for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                    SnRecord rec;
                    (void)snappy_record_step(rd, ip, op, iend, U, rec);     // the parse kernel accepted this stream
                    table2[s] = make_uint2(rec.lit_src | (rec.lit_len << 16), ((rec.dst - rec.lit_len) & 0xffffu) | ((rec.w & 0xffffu) << 16));
                }
            }
            if (tid == 0) table2[nseq] = make_uint2(0u, U & 0xffffu);
        } else {
            for (uint32_t sp = tid; sp < nsp; sp += kLvThreads) {
                const uint2 p = sp == tid ? p_first : csync[sp];
                uint32_t ip = p.x, op = p.y;
                uint32_t s = sp * kSyncEvery;
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
                    ip += lit;
                    uint32_t offset = 0, mlen = 0;
                    if (s + 1u < nseq) {
                        const uint32_t o4 = lds_ld32a(a_in + ip);
                        offset = o4 & 0xffffu;
                        ip += 2;
                        mlen = token & 15u;
                        if (mlen == 15u) {
                            uint32_t b = (o4 >> 16) & 0xffu;
                            ip += 1; mlen += b;
                            while (b == 255u) { b = lds_ld8(a_in + ip); ip += 1; mlen += b; }
                        }
                        mlen += 4u;
                    }
                    table2[s] = make_uint2(lit_src | (lit << 16), (op & 0xffffu) | (offset << 16));
                    op += lit + mlen;
                }
            }
            if (tid == 0) table2[nseq] = make_uint2(0u, U & 0xffffu);      // sentinel: where the last record's match ends
        }
        __syncthreads();                                     // the table is complete (and the staged bytes are dead)
        CJ_LV_MARK(1);

        // ---- the chunk's records in REGISTERS: wavefront w owns the batches w, w + 8, ... of 64 records; up to 8 batches (chunks of up to
        //      4 096 sequences) are loaded ONCE here and serve X, L, K and D2; longer chunks reload their groups of 8 batches per phase.
        //      (A dependent global round trip costs 3-4 k cycles under load: l01 paid one per batch and phase.) ----
        const uint32_t nbat = (nseq + 63u) >> 6, ng = (nbat + 63u) >> 6;
        const bool one = ng == 1u;
        uint32_t RX[8], RY[8], RN[8];                       // lit_src | lit << 16, start | offset << 16, start of the next record (16 bits)
        const auto load_group = [&](uint32_t g) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t r = (wave + 8u * (8u * g + (uint32_t)j)) * 64u + lane;
                uint4 t = make_uint4(0u, 0u, 0u, 0u);
                if (r < nseq) t = ld16u(reinterpret_cast<const uint8_t*>(table2 + r));
                RX[j] = t.x; RY[j] = t.y; RN[j] = t.w & 0xffffu;
            }
        };
#define CJ_LV_FOR_BATCHES(...)                                                                    \
        for (uint32_t g_ = 0; g_ < ng; g_++) {                                                    \
            if (!one) load_group(g_);                                                             \
            _Pragma("unroll") for (int j = 0; j < 8; j++) {                                       \
                const uint32_t rbase = (wave + 8u * (8u * g_ + (uint32_t)j)) * 64u;               \
                if (rbase < nseq) { const uint32_t r = rbase + lane; const bool valid = r < nseq; \
                    const uint32_t lit = RX[j] >> 16, start = RY[j] & 0xffffu, off = RY[j] >> 16, dst = start + lit;       \
                    const uint32_t m = off ? (RN[j] - dst) & 0xffffu : 0u;                        \
                    __VA_ARGS__ } } }
        if (one) load_group(0u);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CJ_LV_MARK(11); }      // 11: the records' round trip (global, L2)

        // ---- X: the position index.  Record r owns the granules whose first byte lies in (start of r, start of r + 1] ----
        {
            uint4* z = reinterpret_cast<uint4*>(smem + kLvOffA);
            for (uint32_t i = tid; i < 1024u; i += kLvThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);           // A: 16 KiB
            uint4* lv = reinterpret_cast<uint4*>(smem + kLvOffLvl);
            for (uint32_t i = tid; i < (nseq + 7u) / 8u + 1u; i += kLvThreads) lv[i] = make_uint4(~0u, ~0u, ~0u, ~0u);   // every level unknown
            for (uint32_t i = tid; i < kLvMaxLevel + 3u; i += kLvThreads) s_hist[i] = 0u;
            if (tid == 0) lds_st16(a_C, 0u);
        }
        __syncthreads();
        CJ_LV_MARK(12);                                      // 12: X's clears + barrier
        CJ_LV_FOR_BATCHES({
            if (valid) {
                const uint32_t nstart = dst + m;                          // the start of r + 1 (not folded to 16 bits)
                asm volatile("ds_or_b32 %0, %1" :: "v"(a_A + 4u * (start >> 4)), "v"(1u << (start & 15u)) : "memory");
                if (m) asm volatile("ds_or_b32 %0, %1" :: "v"(a_A + 4u * (dst >> 4)), "v"(0x10000u << (dst & 15u)) : "memory");
                for (uint32_t g = (start >> 4) + 1u; g <= (nstart >> 4); g++)
                    lds_st16(a_C + 2u * g, (r + 1u) | ((m && 16u * g >= dst && 16u * g < nstart) ? 0x8000u : 0u));
            }
        })
        __syncthreads();
        CJ_LV_MARK(2);

        // ---- L: levels ----
        {
            uint32_t my_max = 0;
            CJ_LV_FOR_BATCHES({
                bool pend = valid && m > 0u;
                if (valid && m == 0u) lds_st16(a_lvl + 2u * r, 0u);
                int32_t qa = 0, qb = -1;
                if (pend) {                                               // both index lookups behind ONE wait
                    const uint32_t s0 = dst - off, e1 = s0 + (off < m ? off : m) - 1u;
                    uint32_t a0, c0, a1, c1;
                    asm volatile("ds_read_b32 %0, %4\n\tds_read_u16 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_u16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(a0), "=&v"(c0), "=&v"(a1), "=&v"(c1)
                                 : "v"(a_A + 4u * (s0 >> 4)), "v"(a_C + 2u * (s0 >> 4)), "v"(a_A + 4u * (e1 >> 4)), "v"(a_C + 2u * (e1 >> 4)) : "memory");
                    const uint32_t k0 = (2u << (s0 & 15u)) - 1u, k1 = (2u << (e1 & 15u)) - 1u;
                    const uint32_t s1 = a1 & k1, d1 = (a1 >> 16) & k1;
                    qa = (int32_t)((c0 & 0x7fffu) + (uint32_t)__popc(a0 & k0) - 1u);
                    const int32_t q1 = (int32_t)((c1 & 0x7fffu) + (uint32_t)__popc(s1) - 1u);
                    const bool inm = (s1 | d1) == 0u ? (c1 >> 15) != 0u : (d1 != 0u && __clz((int)d1) <= __clz((int)s1));
                    qb = inm ? q1 : q1 - 1;
                    qb = qb < (int32_t)r - 1 ? qb : (int32_t)r - 1;
                }
                const int32_t b4 = qa & ~1;
                const bool shortr = qb - b4 <= 3;                         // the whole range in one aligned pair of dwords (the common case)
                uint32_t spins = 0;
                while (ballot64(pend) != 0ull) {
                    if (pend) {
                        uint32_t acc = 0;
                        bool fail = false;
                        if (shortr) {
                            if (qa <= qb) {
                                uint64_t e;
                                asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a_lvl + 2u * (uint32_t)b4) : "memory");
_Pragma("unroll")
                                for (int q = 0; q < 4; q++) {
                                    const uint32_t v = (uint32_t)(e >> (16 * q)) & 0xffffu;
                                    if (b4 + q >= qa && b4 + q <= qb) { fail = fail || v == kLvUnknown; acc = v > acc ? v : acc; }
                                }
                            }
                        } else {
                            for (int32_t cur = qa; cur <= qb && !fail;) {
                                const int32_t c4 = cur & ~1;
                                uint64_t e;
                                asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a_lvl + 2u * (uint32_t)c4) : "memory");
_Pragma("unroll")
                                for (int q = 0; q < 4; q++) {
                                    const uint32_t v = (uint32_t)(e >> (16 * q)) & 0xffffu;
                                    if (c4 + q >= cur && c4 + q <= qb) { fail = fail || v == kLvUnknown; acc = v > acc ? v : acc; }
                                }
                                cur = c4 + 4;
                            }
                        }
                        if (!fail) {
                            uint32_t L = acc + 1u;
                            if (L > kLvMaxLevel) { L = kLvMaxLevel + 1u; s_var[2] = 1u; }
                            lds_st16(a_lvl + 2u * r, L);
                            asm volatile("ds_add_u32 %0, %1" :: "v"(a_hist + 4u * L), "v"(1u) : "memory");
                            my_max = L > my_max ? L : my_max;
                            pend = false;
                        }
                    }
                    if (++spins > kSpinLimit) { s_var[2] = 1u; break; }
#ifndef CJ_LV_SPIN_SLEEP
#define CJ_LV_SPIN_SLEEP 1
#endif
                    if (spins > 1u) __builtin_amdgcn_s_sleep(CJ_LV_SPIN_SLEEP);           // producers first: a spinning wavefront takes issue slots and LDS cycles
                }
            })
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)my_max, d, 64); my_max = o > my_max ? o : my_max; }
            if (lane == 0 && my_max) atomicMax(&s_var[1], my_max);
        }
        __syncthreads();
        CJ_LV_MARK(3);
        const uint32_t nlev = s_var[1];
        if (s_var[2] != 0u) {                                // deeper than this decoder's level table: the wavefront kernel takes the chunk
            if (tid == 0) meta[c] = ParseMeta{0u, kRouteWave};
            continue;                                        // (uniform; the barrier at the top of the loop orders the LDS reuse)
        }

        // ---- K: slots per level, descriptors sorted by level into TWO lists: the levels of more than 64 matches (copied by all
        //      wavefronts, one lane per match) and the small ones (copied by wavefront 0 alone).  hist[L] becomes the level's next free
        //      slot (bit 31: small list) and ends up as its END; lstart[L] keeps its start (bit 15: small list) ----
        if (wave == 0) {
            uint32_t cnt[16], sumb = 0, sums = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                cnt[j] = s_hist[16u * lane + (uint32_t)j];
                if (cnt[j] > 64u) sumb += cnt[j]; else sums += cnt[j];
            }
            uint32_t totb, tots;
            uint32_t runb = wave_excl_scan_add32(sumb, totb), runs = wave_excl_scan_add32(sums, tots);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t L = 16u * lane + (uint32_t)j;
                const bool small = cnt[j] <= 64u;
                s_hist[L] = small ? (runs | 0x80000000u) : runb;
                s_start[L] = (uint16_t)(small ? (runs | 0x8000u) : runb);
                if (small) runs += cnt[j]; else runb += cnt[j];
            }
            if (lane == 0) { s_var[4] = totb; s_var[5] = tots; }
        }
        __syncthreads();
        CJ_LV_FOR_BATCHES({
            if (valid && m) {
                const uint32_t L = lds_ld16(a_lvl + 2u * r);
                uint32_t sl;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(sl) : "v"(a_hist + 4u * L), "v"(1u) : "memory");
                uint2* lst = (sl >> 31) ? sorted_s : sorted;
                lst[sl & 0x7fffffffu] = make_uint2(dst | (off << 16), m | (L << 16));
            }
        })
        __syncthreads();                                     // the sorted lists are complete; the window's scratch is dead
        CJ_LV_MARK(4);
        // D3's descriptors, requested NOW and used after D2 (a global round trip is ~6 k cycles here: none of them may sit between two
        // levels).  Thread t holds the entries 512 k + t of the large list for eight k, lane l of wavefront 0 the entries 64 k + l of
        // the small list for four k; longer lists are refilled block by block, eight / four blocks ahead.
        const uint32_t nbig = s_var[4], nsmall = s_var[5];
        uint2 EB[8], ES[4];
#pragma unroll
        for (int k = 0; k < 8; k++) { EB[k] = make_uint2(0u, 0u); if (512u * (uint32_t)k + tid < nbig) EB[k] = sorted[512u * (uint32_t)k + tid]; }
#pragma unroll
        for (int k = 0; k < 4; k++) { ES[k] = make_uint2(0u, 0u); if (wave == 0u && 64u * (uint32_t)k + lane < nsmall) ES[k] = sorted_s[64u * (uint32_t)k + lane]; }

        // ---- D2: literals, one lane per record: global -> window (the next batch's records are requested before the current batch
        //      is processed) ----
        {
            uint4 rec_nx = make_uint4(0, 0, 0, 0);
            if (wave * 64u + lane < nseq) rec_nx = ld16u(reinterpret_cast<const uint8_t*>(table2 + wave * 64u + lane));
            for (uint32_t base = wave * 64u; base < nseq; base += kLvThreads) {
                const uint4 t = rec_nx;
                rec_nx = make_uint4(0, 0, 0, 0);
                if (base + kLvThreads + lane < nseq) rec_nx = ld16u(reinterpret_cast<const uint8_t*>(table2 + base + kLvThreads + lane));
                uint32_t n = t.x >> 16, src = t.x & 0xffffu, dst = t.y & 0xffffu;
                uint64_t lm = ballot64(n >= kLongRun);
                while (lm) {
                    const uint32_t l = ctz64(lm);
                    lm &= lm - 1ull;
                    wave_copy_to_lds(a_out + rdlane(dst, l), in + rdlane(src, l), rdlane(n, l));
                    if (lane == l) n = 0;
                }
                // the chunk's last bytes (a vector load would leave the 16-byte granule of the last input byte): byte by byte
                if (ballot64(n > 0u && src + n + 32u > safe_end) != 0ull) {
                    if (n > 0u && src + n + 32u > safe_end) {
                        for (uint32_t k = 0; k < n; k++) lds_st8(a_out + dst + k, in[src + k]);
                        n = 0u;
                    }
                }
                while (ballot64(n > 0u) != 0ull) {
                    const uint32_t hb = dst & 3u;
                    const bool wide = ballot64(n + hb > 16u) != 0ull;             // (wave-uniform) 32 bytes of the grid per pass instead of 16
                    const uint32_t room = (wide ? 32u : 16u) - hb;
                    const uint32_t step = n < room ? n : room;
                    if (step > 0u) {
                        const uint8_t* g = in + src - hb;                         // the source, shifted onto the destination's dword grid
                        if (wide) {                                               // (the first record starts at destination 0: src >= hb)
                            const uint4 x = ld16u(g), y = ld16u(g + 16);
                            const uint32_t v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
                            lds_store_grid<8>(v, (a_out + dst) & ~3u, hb, step, dummy_w);
                        } else {
                            const uint4 x = ld16u(g);
                            const uint32_t v[4] = {x.x, x.y, x.z, x.w};
                            lds_store_grid<4>(v, (a_out + dst) & ~3u, hb, step, dummy_w);
                        }
                        n -= step; src += step; dst += step;
                    }
                }
            }
        }
        __syncthreads();
        CJ_LV_MARK(5);

#ifdef CJ_LV_D3_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        // ---- D3: matches, level by level.  A wavefront spends instructions only on the levels it takes part in: start / end / kind of
        //      64 levels sit in the lanes of two registers, the large levels are one bit mask, and what is left per level for a wavefront
        //      without work is a bit test and the barrier.  Small levels between two large ones are copied by wavefront 0 alone, one
        //      after the other WITHOUT barriers (the DS operations of a wavefront execute in order). ----
        {
            const uint32_t w64 = wave * 64u;
            uint32_t nbar = 0, kb = 0, ks = 0;                  // EB[0] = block kb of the large list, ES[0] = block ks of the small list
            bool dirty = false;                                  // wavefront 0 has copied small levels since the last barrier
            for (uint32_t Lb = 1; Lb <= nlev; Lb += 64u) {
                const uint32_t cnt64 = nlev - Lb + 1u < 64u ? nlev - Lb + 1u : 64u;
                uint32_t st = 0, en = 0;                                                     // lane i: level Lb + i
                if (lane < cnt64) {
                    asm volatile("ds_read_u16 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(st), "=&v"(en)
                                 : "v"(a_start + 2u * (Lb + lane)), "v"(a_hist + 4u * (Lb + lane)) : "memory");
                }
                const uint64_t mm = ballot64(lane < cnt64 && (st >> 15) == 0u);              // the large levels
                st &= 0x7fffu; en &= 0x7fffffffu;
                uint32_t i = 0;
                while (i < cnt64) {
                    if ((mm >> i) & 1ull) {
                        uint32_t ls = rdlane(st, uni(i));
                        const uint32_t le = rdlane(en, uni(i));
                        if (dirty) { __syncthreads(); nbar += 1; dirty = false; }
                        CJ_LV_MARK(6);                                           // 6: D3's walk over the levels
                        while (ls < le) {
                            while (ls >= 512u * (kb + 1u)) {                                 // (uniform) next block of the large list
#pragma unroll
                                for (int k = 0; k < 7; k++) EB[k] = EB[k + 1];
                                kb += 1u;
                                EB[7] = make_uint2(0u, 0u);
                                if (512u * (kb + 7u) + tid < nbig) EB[7] = sorted[512u * (kb + 7u) + tid];
                            }
                            const uint32_t part = le < 512u * (kb + 1u) ? le : 512u * (kb + 1u);
                            const uint32_t e = 512u * kb + tid;
                            if (512u * kb + w64 < part && 512u * kb + w64 + 64u > ls)        // this wavefront holds descriptors of [ls, part)
                                lvl_match_copy_dense(e >= ls && e < part, a_out, s_out, EB[0].x & 0xffffu, EB[0].x >> 16, EB[0].y & 0xffffu, dummy_w);
                            ls = part;
                        }
                        CJ_LV_MARK(13);                                          // 13: wavefront 0's copies of the large levels
                        __syncthreads(); nbar += 1;
                        CJ_LV_MARK(14);                                          // 14: ... its wait at their barriers
                        i += 1u;
                    } else {
                        const uint64_t rest = mm >> i;
                        const uint32_t j = rest ? i + ctz64(rest) : cnt64;             // the small levels i .. j - 1
                        if (wave == 0u) {
                            for (uint32_t k2 = i; k2 < j; k2++) {
                                uint32_t ls = rdlane(st, uni(k2));
                                const uint32_t le = rdlane(en, uni(k2));
                                while (ls < le) {
                                    while (ls >= 64u * (ks + 1u)) {
#pragma unroll
                                        for (int k = 0; k < 3; k++) ES[k] = ES[k + 1];
                                        ks += 1u;
                                        ES[3] = make_uint2(0u, 0u);
                                        if (64u * (ks + 3u) + lane < nsmall) ES[3] = sorted_s[64u * (ks + 3u) + lane];
                                    }
                                    const uint32_t part = le < 64u * (ks + 1u) ? le : 64u * (ks + 1u);
                                    const uint32_t e = 64u * ks + lane;
                                    lvl_match_copy_sparse(e >= ls && e < part, a_out, s_out, ES[0].x & 0xffffu, ES[0].x >> 16, ES[0].y & 0xffffu, dummy_w);
                                    ls = part;
                                }
                            }
                        }
                        dirty = true;
                        CJ_LV_MARK(15);                                          // 15: the runs of small levels (wavefront 0 copies)
                        i = j;
                    }
                }
            }
            if (dirty || nlev == 0u) { __syncthreads(); nbar += 1; }
            if (prof && tid == 0) { s_prof[9] += nlev; s_prof[10] += nbar; }
        }
#undef CJ_LV_FOR_BATCHES
#ifdef CJ_LV_D3_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        CJ_LV_MARK(6);

        // ---- D4: stream the window out (16 B per lane), exact tail ----
        {
            const uint32_t nvec = U >> 4;
            const uint4* src = reinterpret_cast<const uint4*>(s_out);
            for (uint32_t i = tid; i < nvec; i += kLvThreads) st16u_nt(out + 16u * i, src[i]);
            for (uint32_t i = (nvec << 4) + tid; i < U; i += kLvThreads) out[i] = s_out[i];
        }
        if (prof) {
            __syncthreads(); CJ_LV_MARK(7);
            if (tid < 16u) { const uint32_t v = tid == 8u ? 1u : s_prof[tid]; if (v) atomicAdd(&g_lvl_phase_cycles[tid], (unsigned long long)v); s_prof[tid] = 0u; }
        }
    }
}

template <int kCodec>
__global__ __launch_bounds__(kLvThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lz4_decode_lvl_kernel(BatchArgs a, const uint2* sync, ParseMeta* meta,
                                                                                                            uint8_t* tabs, uint32_t* counter) {
    lvl_body<kCodec, false>(a, sync, meta, tabs, counter);
}
template <int kCodec>
__global__ __launch_bounds__(kLvThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lz4_decode_lvl_fused_kernel(BatchArgs a, ParseMeta* meta, uint8_t* tabs,
                                                                                                                  uint32_t* counter) {
    lvl_body<kCodec, true>(a, nullptr, meta, tabs, counter);
}

size_t lz4_lvl_tab_bytes(uint32_t grid) { return (size_t)grid * kLvSlotBytes; }

void launch_lz4_decode_lvl(const BatchArgs& a, const void* sync, void* meta, void* tabs, uint32_t* counter, uint32_t grid, hipStream_t s, int codec, bool fused) {
    if (a.n_chunks == 0) return;
    const uint32_t bytes = fused ? kLvBytesFused : kLvBytes;
#define CJ_LV_LAUNCH(KERNEL, ...)                                                                                          \
    do {                                                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(kLvThreads), bytes, s, __VA_ARGS__);                                   \
    } while (0)
    if (fused) {
        if (codec == CJ_CODEC_SNAPPY_RAW) CJ_LV_LAUNCH((lz4_decode_lvl_fused_kernel<CJ_CODEC_SNAPPY_RAW>), a, (ParseMeta*)meta, (uint8_t*)tabs, counter);
        else CJ_LV_LAUNCH((lz4_decode_lvl_fused_kernel<CJ_CODEC_LZ4_BLOCK>), a, (ParseMeta*)meta, (uint8_t*)tabs, counter);
    } else {
        if (codec == CJ_CODEC_SNAPPY_RAW) CJ_LV_LAUNCH((lz4_decode_lvl_kernel<CJ_CODEC_SNAPPY_RAW>), a, (const uint2*)sync, (ParseMeta*)meta, (uint8_t*)tabs, counter);
        else CJ_LV_LAUNCH((lz4_decode_lvl_kernel<CJ_CODEC_LZ4_BLOCK>), a, (const uint2*)sync, (ParseMeta*)meta, (uint8_t*)tabs, counter);
    }
#undef CJ_LV_LAUNCH
}

}  // namespace cj

extern "C" CJ_API int cj_debug_lvl_phase_cycles(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(cj::g_lvl_phase_cycles), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cj::g_lvl_phase_cycles), z, 128) != hipSuccess) return -1; }
    return 0;
}
