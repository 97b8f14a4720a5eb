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
constexpr uint32_t kLvOffFused = kLvOffVars + 384;              // fused_parse's 6 KiB
constexpr uint32_t kLvBytes = kLvOffFused;                      // 74 128 B
constexpr uint32_t kLvBytesFused = kLvOffFused + kFusedAux;     // 80 272 B: two workgroups per CU (163 840 B)
static_assert(2u * kLvBytesFused <= 163840u, "two workgroups per CU");
static_assert(kLvOffStart + 2u * (kLvMaxLevel + 3u) <= kLvOffVars, "lstart fits in front of the variables");
constexpr uint32_t kLvSlotBytes = 3u * kLvMaxRec * 16u / 2u;    // per workgroup in the table scratch: 8-byte records, then the sorted descriptors
constexpr uint32_t kLvSortedOff = (kLvMaxRec + 64u) * 8u;
static_assert(kLvSortedOff + kLvMaxRec * 8u <= kLvSlotBytes, "records + sorted descriptors fit the slot");

__device__ unsigned long long g_lvl_phase_cycles[16];           // S0, D1/P, X, L, K, D2, D3, D4, chunks, levels, barriers of D3 (flag 0x1000)
#define CJ_LV_MARK(idx)                                                                 \
    do {                                                                                \
        if (prof && tid == 0) {                                                         \
            unsigned long long now_ = __builtin_readcyclecounter();                     \
            atomicAdd(&g_lvl_phase_cycles[idx], now_ - t_prev);                         \
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
    const bool prof = (a.flags & 0x1000u) != 0;
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

        // ---- X: the position index ----
        {
            uint4* z = reinterpret_cast<uint4*>(smem + kLvOffA);
            for (uint32_t i = tid; i < 1024u; i += kLvThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);           // A: 16 KiB
            uint4* lv = reinterpret_cast<uint4*>(smem + kLvOffLvl);
            for (uint32_t i = tid; i < (nseq + 7u) / 8u; i += kLvThreads) lv[i] = make_uint4(~0u, ~0u, ~0u, ~0u);   // every level unknown
            for (uint32_t i = tid; i < kLvMaxLevel + 3u; i += kLvThreads) s_hist[i] = 0u;
        }
        __syncthreads();
        const uint32_t g_last = (U - 1u) >> 4;
        for (uint32_t r = tid; r < nseq; r += kLvThreads) {
            const uint4 t = ld16u(reinterpret_cast<const uint8_t*>(table2 + r));                     // record r and the start of r + 1
            const uint2 pr = r ? table2[r - 1u] : make_uint2(0u, 0u);
            const uint32_t lit = t.x >> 16, start = t.y & 0xffffu, off = t.y >> 16, dst = start + lit;
            const uint32_t m = off ? ((t.w & 0xffffu) - dst) & 0xffffu : 0u;
            asm volatile("ds_or_b32 %0, %1" :: "v"(a_A + 4u * (start >> 4)), "v"(1u << (start & 15u)) : "memory");
            if (m) asm volatile("ds_or_b32 %0, %1" :: "v"(a_A + 4u * (dst >> 4)), "v"(0x10000u << (dst & 15u)) : "memory");
            // granules whose first byte lies in (start of r - 1, start of r]: the first record that starts at or after them is r
            const uint32_t pstart = pr.y & 0xffffu, pdst = pstart + (pr.x >> 16), pm_ = pr.y >> 16;
            const uint32_t gr = start >> 4;
            for (uint32_t g = r ? (pstart >> 4) + 1u : 0u; g <= gr; g++)
                lds_st16(a_C + 2u * g, r | ((r && pm_ && 16u * g >= pdst && 16u * g < start) ? 0x8000u : 0u));
            if (r + 1u == nseq)                               // behind the last record's start
                for (uint32_t g = gr + 1u; g <= g_last; g++) lds_st16(a_C + 2u * g, nseq | ((m && 16u * g >= dst) ? 0x8000u : 0u));
        }
        __syncthreads();
        CJ_LV_MARK(2);

        // ---- L: levels ----
        {
            uint32_t my_max = 0;
            for (uint32_t base = wave * 64u; base < nseq; base += kLvThreads) {
                const uint32_t r = base + lane;
                const bool valid = r < nseq;
                uint4 t = make_uint4(0u, 0u, 0u, 0u);
                if (valid) t = ld16u(reinterpret_cast<const uint8_t*>(table2 + r));
                const uint32_t lit = t.x >> 16, start = t.y & 0xffffu, off = t.y >> 16, dst = start + lit;
                const uint32_t m = off ? ((t.w & 0xffffu) - dst) & 0xffffu : 0u;
                bool pend = valid && m > 0u;
                if (valid && m == 0u) lds_st16(a_lvl + 2u * r, 0u);
                int32_t qa = 0, qb = -1;
                if (pend) {
                    const uint32_t s0 = dst - off, need = off < m ? off : m;
                    uint32_t q0, q1; bool i0, i1;
                    lvl_lookup(a_A, a_C, s0, q0, i0);
                    lvl_lookup(a_A, a_C, s0 + need - 1u, q1, i1);
                    qa = (int32_t)q0;
                    qb = i1 ? (int32_t)q1 : (int32_t)q1 - 1;
                    qb = qb < (int32_t)r - 1 ? qb : (int32_t)r - 1;
                }
                uint32_t spins = 0;
                while (ballot64(pend) != 0ull) {
                    if (pend) {
                        uint32_t acc = 0;
                        bool fail = false;
                        for (int32_t cur = qa; cur <= qb && !fail;) {
                            const int32_t b4 = cur & ~1;
                            uint64_t e;
                            asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(a_lvl + 2u * (uint32_t)b4) : "memory");
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t v = (uint32_t)(e >> (16 * j)) & 0xffffu;
                                if (b4 + j >= cur && b4 + j <= qb) { fail = fail || v == kLvUnknown; acc = v > acc ? v : acc; }
                            }
                            cur = b4 + 4;
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
                }
            }
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

        // ---- K: slots per level, descriptors sorted by level ----
        if (wave == 0) {
            uint32_t cnt[16], sum = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) { cnt[j] = s_hist[16u * lane + (uint32_t)j]; sum += cnt[j]; }
            uint32_t total;
            uint32_t run = wave_excl_scan_add32(sum, total);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t L = 16u * lane + (uint32_t)j;
                s_hist[L] = run;
                s_start[L] = (uint16_t)run;
                run += cnt[j];
            }
            if (lane == 0) { s_var[4] = total; s_start[kLvMaxLevel + 1u] = (uint16_t)total; s_start[kLvMaxLevel + 2u] = (uint16_t)total; }
        }
        __syncthreads();
        for (uint32_t r = tid; r < nseq; r += kLvThreads) {
            const uint32_t L = lds_ld16(a_lvl + 2u * r);
            if (L) {
                const uint4 t = ld16u(reinterpret_cast<const uint8_t*>(table2 + r));
                const uint32_t lit = t.x >> 16, start = t.y & 0xffffu, off = t.y >> 16, dst = start + lit;
                const uint32_t m = ((t.w & 0xffffu) - dst) & 0xffffu;
                uint32_t sl;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(sl) : "v"(a_hist + 4u * L), "v"(1u) : "memory");
                sorted[sl] = make_uint2(dst | (off << 16), m | (L << 16));
            }
        }
        __syncthreads();                                     // the sorted list is complete; the window's scratch is dead
        CJ_LV_MARK(4);
        const uint32_t nm = s_var[4];

        // ---- D2: literals, one lane per record: global -> window ----
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
                while (ballot64(n > 0u) != 0ull) {
                    const uint32_t hb = dst & 3u;
                    const bool wide = ballot64(n + hb > 16u) != 0ull;             // (wave-uniform) 32 bytes of the grid per pass instead of 16
                    const uint32_t room = (wide ? 32u : 16u) - hb;
                    const uint32_t step = n < room ? n : room;
                    if (step > 0u) {
                        const uint8_t* g = in + src - hb;                         // the source, shifted onto the destination's dword grid
                        const bool inside = src >= hb && src - hb + (wide ? 32u : 16u) <= safe_end;
                        if (wide) {
                            uint32_t v[8];
                            if (inside) {
                                const uint4 x = ld16u(g), y = ld16u(g + 16);
                                v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
                            } else {                                              // the chunk's first / last bytes: aligned dwords, clamped
                                const uint8_t* g0 = in + src;
                                const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(g0) & 3u);
                                const DW<10> w = gl_ld_aligned<10>(g0 - sh, last_dw);
                                uint32_t u[9];
#pragma unroll
                                for (int k = 0; k < 9; k++) u[k] = __builtin_amdgcn_alignbyte(w.w[k + 1], w.w[k], sh);      // u[k] = source bytes 4k ..
                                v[0] = u[0] << (8u * hb);
#pragma unroll
                                for (int k = 1; k < 8; k++) v[k] = hb ? __builtin_amdgcn_alignbyte(u[k], u[k - 1], 4u - hb) : u[k];
                            }
                            lds_store_grid<8>(v, (a_out + dst) & ~3u, hb, step, dummy_w);
                        } else {
                            uint32_t v[4];
                            if (inside) { const uint4 x = ld16u(g); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
                            else {
                                const uint8_t* g0 = in + src;
                                const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(g0) & 3u);
                                const DW<6> w = gl_ld_aligned<6>(g0 - sh, last_dw);
                                uint32_t u[5];
#pragma unroll
                                for (int k = 0; k < 5; k++) u[k] = __builtin_amdgcn_alignbyte(w.w[k + 1], w.w[k], sh);
                                v[0] = u[0] << (8u * hb);
#pragma unroll
                                for (int k = 1; k < 4; k++) v[k] = hb ? __builtin_amdgcn_alignbyte(u[k], u[k - 1], 4u - hb) : u[k];
                            }
                            lds_store_grid<4>(v, (a_out + dst) & ~3u, hb, step, dummy_w);
                        }
                        n -= step; src += step; dst += step;
                    }
                }
            }
        }
        __syncthreads();
        CJ_LV_MARK(5);

        // ---- D3: matches, level by level ----
        {
            uint32_t pos = 0, L = 1, nbar = 0;
            uint2 ent = make_uint2(0u, 0u);
            if (tid < nm) ent = sorted[tid];
            while (L <= nlev) {
                // lane i: the end of level L + i
                uint32_t li = L + 1u + lane;
                li = li > nlev + 1u ? nlev + 1u : li;
                const uint32_t le = li > nlev ? nm : lds_ld16(a_start + 2u * li);
                const uint32_t e0 = rdlane(le, 0);
                const bool multi = e0 - pos > 64u;
                uint32_t seg_end = e0, Lnext = L + 1u;
                if (!multi) {
                    const uint64_t okm = ballot64(le - pos <= 64u && L + lane <= nlev);
                    const uint32_t cnt = (uint32_t)__popcll(okm);                  // (the ends are monotonic: the low `cnt` lanes)
                    seg_end = rdlane(le, uni(cnt - 1u));
                    Lnext = L + cnt;
                }
                bool next_multi = false;
                if (Lnext <= nlev) {
                    const uint32_t k = Lnext - L;                                  // lane k holds the end of level Lnext
                    const uint32_t en = k < 64u ? rdlane(le, uni(k)) : (Lnext + 1u > nlev ? nm : (uint32_t)s_start[Lnext + 1u]);
                    next_multi = en - seg_end > 64u;
                }
                uint2 ent_nx = make_uint2(0u, 0u);
                if (seg_end + tid < nm) ent_nx = sorted[seg_end + tid];           // the next segment's descriptors, one segment ahead
                if (multi) {
                    lvl_match_copy(pos + tid < seg_end, a_out, s_out, ent.x & 0xffffu, ent.x >> 16, ent.y & 0xffffu, dummy_w);
                    for (uint32_t b = pos + kLvThreads; b < seg_end; b += kLvThreads) {
                        uint2 e2 = make_uint2(0u, 0u);
                        if (b + tid < seg_end) e2 = sorted[b + tid];
                        lvl_match_copy(b + tid < seg_end, a_out, s_out, e2.x & 0xffffu, e2.x >> 16, e2.y & 0xffffu, dummy_w);
                    }
                } else if (wave == 0u) {
                    const uint32_t nseg = seg_end - pos, lev = ent.y >> 16;
                    for (uint32_t Lc = L; Lc < Lnext; Lc++)
                        lvl_match_copy(lane < nseg && lev == Lc, a_out, s_out, ent.x & 0xffffu, ent.x >> 16, ent.y & 0xffffu, dummy_w);
                }
                if (multi || next_multi || Lnext > nlev) { __syncthreads(); nbar += 1; }
                pos = seg_end; L = Lnext; ent = ent_nx;
            }
            if (prof && tid == 0) { atomicAdd(&g_lvl_phase_cycles[9], (unsigned long long)nlev); atomicAdd(&g_lvl_phase_cycles[10], (unsigned long long)nbar); }
        }
        CJ_LV_MARK(6);

        // ---- D4: stream the window out (16 B per lane), exact tail ----
        {
            const uint32_t nvec = U >> 4;
            const uint4* src = reinterpret_cast<const uint4*>(s_out);
            for (uint32_t i = tid; i < nvec; i += kLvThreads) st16u_nt(out + 16u * i, src[i]);
            for (uint32_t i = (nvec << 4) + tid; i < U; i += kLvThreads) out[i] = s_out[i];
        }
        if (prof) { __syncthreads(); CJ_LV_MARK(7); if (tid == 0) atomicAdd(&g_lvl_phase_cycles[8], 1ull); }
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
