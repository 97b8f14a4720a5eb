// lz4_decode_lvl1.hip — the level-ordered workgroup decoder with EVERYTHING IN LDS: one persistent workgroup of sixteen wavefronts
// per CU owns all 160 KiB — the 64 KiB output window, the staged compressed chunk in a buffer of its own, and 32 KiB for the level
// table and the matches sorted by level.  Between "chunk staged" and "window streamed out" no phase touches global memory.
// Same results as the other mappings (reference call sites /root/reference/src/lz4.rs:88,90,164,168, src/snappy.rs:57,106).
//
// Why (profiles/r03/experiments): with the LDS pipe no longer saturated, the first level-ordered decoder (lz4_decode_lvl.hip: records
// and sorted descriptors in a global table, two workgroups per CU) was bound by GLOBAL round trips — 3-4 k cycles each once the CU's
// memory pipe also carries two chunks' 64 KiB of output stores: ~40 of them per chunk.  An LDS round trip is ~110 cycles
// (tools/lds_mskor_probe.hip), so here the records, the position index, the levels and the sorted descriptors all live in LDS:
//   S0   the compressed chunk -> the staging buffer (the next chunk's bytes are requested while this one is decoded)
//   D1   one thread per sync point re-walks 8 sequences and writes 8-byte records into the (still unused) window
//   X    position index (lvl_shared.hpp: lvl_lookup) in the window
//   L    levels with "unknown" markers, sixteen wavefronts racing through the batches of 64 records
//   K    scan of the level counts; every match takes a slot of its level: {dst, offset, length} sorted by level in the 32 KiB
//   D2   every thread takes its records' literal runs into registers, barrier (the window's scratch is dead), literals staging ->
//        window, LDS -> LDS
//   D3   matches level by level (dense copy per level + barrier; runs of small levels by wavefront 0 alone without barriers)
//   D4   stream the window out
// Takes the chunks of up to kL1MaxRec sequences; what it leaves (more sequences: real text has ~10 k per 64 KiB) keeps its
// ParseMeta and is decoded by lz4_decode_lvl.hip's kernel, which runs after it.
#include "lvl_shared.hpp"
#include <type_traits>

namespace cj {

constexpr uint32_t kL1Threads = 1024;
constexpr uint32_t kL1MaxRec = 4032;                            // records per chunk that fit the window scratch next to the index
// LDS map (bytes from the start of the dynamic segment; all of the CU's 160 KiB)
constexpr uint32_t kL1OffWin = 16;                              // a copy may read up to 3 bytes in front of the window
constexpr uint32_t kL1OffRec = kL1OffWin;                       // scratch inside the window until D2: uint2 rec[4033]
constexpr uint32_t kL1OffLvl = kL1OffWin + 32320;               //   u16 level[4032] (+ over-read)
constexpr uint32_t kL1OffA = kL1OffWin + 40448;                 //   u32 A[4096]
constexpr uint32_t kL1OffC = kL1OffWin + 56832;                 //   u16 C[4097]
static_assert(kL1OffRec + (kL1MaxRec + 1u) * 8u <= kL1OffLvl && kL1OffLvl + kL1MaxRec * 2u + 16u <= kL1OffA && kL1OffA + 16384u <= kL1OffC &&
              kL1OffC + 4097u * 2u <= kL1OffWin + 65536u, "window scratch");
constexpr uint32_t kL1OffStage = kL1OffWin + 65536;             // the staged compressed chunk (<= 65504 + 15 bytes)
constexpr uint32_t kL1OffAux = kL1OffStage + 65536;
constexpr uint32_t kL1OffHist = kL1OffAux;                      // u32 hist / slot[1026]
constexpr uint32_t kL1OffStart = kL1OffHist + 4112;             // u16 lstart[1026]
constexpr uint32_t kL1OffVars = kL1OffStart + 2064;             // 64 B of variables, 256 B of dummy dwords
constexpr uint32_t kL1OffSortD = kL1OffVars + 320;              // u16 dst[4032], offset[4032], length[4032]: the matches sorted by level
constexpr uint32_t kL1OffSortO = kL1OffSortD + 2u * kL1MaxRec;
constexpr uint32_t kL1OffSortM = kL1OffSortO + 2u * kL1MaxRec;
constexpr uint32_t kL1Bytes = kL1OffSortM + 2u * kL1MaxRec + 16u;
static_assert(kL1Bytes <= 163840u, "one workgroup per CU");

__device__ unsigned long long g_lvl1_phase_cycles[16];          // S0, D1, X, L, K, D2, D3, D4, chunks, levels, barriers of D3 (flag 0x1000)
#define CJ_L1_MARK(idx)                                                                 \
    do {                                                                                \
        if (prof && tid == 0) {                                                         \
            unsigned long long now_ = __builtin_readcyclecounter();                     \
            atomicAdd(&g_lvl1_phase_cycles[idx], now_ - t_prev);                        \
            t_prev = now_;                                                              \
        }                                                                               \
    } while (0)

template <int kCodec>
__device__ __forceinline__ void lvl1_body(const BatchArgs& a, const uint2* sync, ParseMeta* meta, uint32_t* counter) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_out = smem + kL1OffWin;
    const uint32_t a_smem = (uint32_t)(uintptr_t)smem;
    const uint32_t a_out = a_smem + kL1OffWin, a_stage = a_smem + kL1OffStage;
    const uint32_t a_lvl = a_smem + kL1OffLvl, a_A = a_smem + kL1OffA, a_C = a_smem + kL1OffC;
    const uint32_t a_hist = a_smem + kL1OffHist, a_start = a_smem + kL1OffStart;
    const uint32_t a_sd = a_smem + kL1OffSortD, a_so = a_smem + kL1OffSortO, a_sm = a_smem + kL1OffSortM;
    uint2* s_rec = reinterpret_cast<uint2*>(smem + kL1OffRec);
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + kL1OffHist);
    uint16_t* s_start = reinterpret_cast<uint16_t*>(smem + kL1OffStart);
    uint32_t* s_var = reinterpret_cast<uint32_t*>(smem + kL1OffVars);         // [0] chunk, [1] max level, [2] overflow, [4] matches
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t dummy_w = a_smem + kL1OffVars + 64u + 4u * lane;
    const bool prof = (a.flags & 0x1000u) != 0;
    unsigned long long t_prev = prof ? __builtin_readcyclecounter() : 0ull;
    uint32_t next_c = 0;
    if (tid == 0) next_c = atomicAdd(counter, 1u);

    for (;;) {
        if (tid == 0) { s_var[0] = next_c; s_var[1] = 0u; s_var[2] = 0u; }
        __syncthreads();                                     // also: the previous chunk's D4 has read the window
        const uint32_t c = s_var[0];
        __syncthreads();
        if (c >= a.n_chunks) break;
        if (tid == 0) next_c = atomicAdd(counter, 1u);
        const ParseMeta pm = meta[c];
        const uint64_t d_in_off = a.in_off[c], d_in_len = a.in_len[c], d_out_off = a.out_off[c];
        const uint64_t d_result = (uint64_t)a.result[c];
        asm volatile("" :: "v"(pm.nseq), "v"(pm.in_skip), "v"((uint32_t)d_in_off), "v"((uint32_t)d_in_len), "v"((uint32_t)d_out_off), "v"((uint32_t)d_result));
        if (pm.nseq == 0u || pm.nseq > kL1MaxRec) continue;  // error / empty / routed elsewhere, or left to the table decoder (uniform)
        const uint32_t nseq = pm.nseq;
        const uint32_t U = (uint32_t)d_result;               // decoded size, 1..65536
        const uint8_t* in = a.in_base + d_in_off + pm.in_skip;
        const uint32_t iend = (uint32_t)d_in_len - pm.in_skip;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);
        uint8_t* out = a.out_base + d_out_off;
        const uint2* csync = sync + (size_t)c * kSyncPitch;
        const uint32_t nsp = (nseq + kSyncEvery - 1u) / kSyncEvery;

        // ---- S0: stage the compressed chunk ----
        const uint2 p_first = csync[tid < nsp ? tid : 0u];
        {
            const uint4* src = reinterpret_cast<const uint4*>(in - mis);
            uint4* dst = reinterpret_cast<uint4*>(smem + kL1OffStage);
            const uint32_t nvec = (mis + iend + 15u) >> 4, last = nvec - 1u;
            const uint32_t i0 = tid, i1 = tid + kL1Threads, i2 = tid + 2u * kL1Threads, i3 = tid + 3u * kL1Threads;
            const uint4 v0 = src[i0 < last ? i0 : last], v1 = src[i1 < last ? i1 : last], v2 = src[i2 < last ? i2 : last], v3 = src[i3 < last ? i3 : last];
            if (i0 < nvec) dst[i0] = v0;
            if (i1 < nvec) dst[i1] = v1;
            if (i2 < nvec) dst[i2] = v2;
            if (i3 < nvec) dst[i3] = v3;
        }
        // the window's scratch: index bits, every level unknown, no match counted yet
        {
            uint4* z = reinterpret_cast<uint4*>(smem + kL1OffA);
            if (tid < 1024u) z[tid] = make_uint4(0u, 0u, 0u, 0u);                                             // A: 16 KiB
            uint4* lv = reinterpret_cast<uint4*>(smem + kL1OffLvl);
            for (uint32_t i = tid; i < (nseq + 7u) / 8u + 1u; i += kL1Threads) lv[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (uint32_t i = tid; i < kLvMaxLevel + 3u; i += kL1Threads) s_hist[i] = 0u;
        }
        __syncthreads();
        CJ_L1_MARK(0);

        // ---- D1: expand the sync points into records (staging buffer -> window scratch) ----
        const uint32_t a_in = a_stage + mis;
        if (tid < nsp) {
            uint32_t ip = p_first.x, op = p_first.y;
            uint32_t s = tid * kSyncEvery;
            if constexpr (kCodec == CJ_CODEC_SNAPPY_RAW) {
                const auto rd = [a_in](uint32_t p) { return lds_ld32a(a_in + p); };
                for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                    SnRecord rec;
                    (void)snappy_record_step(rd, ip, op, iend, U, rec);     // the parse kernel accepted this stream
                    s_rec[s] = make_uint2(rec.lit_src | (rec.lit_len << 16), ((rec.dst - rec.lit_len) & 0xffffu) | ((rec.w & 0xffffu) << 16));
                }
            } else {
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
                    s_rec[s] = make_uint2(lit_src | (lit << 16), (op & 0xffffu) | (offset << 16));
                    op += lit + mlen;
                }
            }
        }
        if (tid == 0) s_rec[nseq] = make_uint2(0u, U & 0xffffu);          // sentinel: where the last record's match ends
        __syncthreads();
        CJ_L1_MARK(1);

        // ---- X: the position index.  Record r owns the granules whose first byte lies in (start of r, start of r + 1] ----
        if (tid == 0) lds_st16(a_C, 0u);
        for (uint32_t r = tid; r < nseq; r += kL1Threads) {
            const uint2 t = s_rec[r];
            const uint32_t nstart0 = s_rec[r + 1u].y & 0xffffu;
            const uint32_t lit = t.x >> 16, start = t.y & 0xffffu, off = t.y >> 16, dst = start + lit;
            const uint32_t m = off ? (nstart0 - dst) & 0xffffu : 0u;
            const uint32_t nstart = r + 1u == nseq ? U : dst + m;          // (the start of r + 1, not folded to 16 bits)
            asm volatile("ds_or_b32 %0, %1" :: "v"(a_A + 4u * (start >> 4)), "v"(1u << (start & 15u)) : "memory");
            if (m) asm volatile("ds_or_b32 %0, %1" :: "v"(a_A + 4u * (dst >> 4)), "v"(0x10000u << (dst & 15u)) : "memory");
            for (uint32_t g = (start >> 4) + 1u; g <= (nstart >> 4); g++)
                lds_st16(a_C + 2u * g, (r + 1u) | ((m && 16u * g >= dst && 16u * g < nstart) ? 0x8000u : 0u));
        }
        __syncthreads();
        CJ_L1_MARK(2);

        // ---- L: levels ----
        {
            uint32_t my_max = 0;
            for (uint32_t base = wave * 64u; base < nseq; base += kL1Threads) {
                const uint32_t r = base + lane;
                const bool valid = r < nseq;
                uint2 t = make_uint2(0u, 0u);
                uint32_t nstart0 = 0;
                if (valid) { t = s_rec[r]; nstart0 = s_rec[r + 1u].y & 0xffffu; }
                const uint32_t lit = t.x >> 16, start = t.y & 0xffffu, off = t.y >> 16, dst = start + lit;
                const uint32_t m = off ? (nstart0 - dst) & 0xffffu : 0u;
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
#ifdef CJ_L1_X_NOL
                if (pend) { lds_st16(a_lvl + 2u * r, 1u); asm volatile("ds_add_u32 %0, %1" :: "v"(a_hist + 4u), "v"(1u) : "memory"); my_max = 1; pend = false; }
#endif
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
        CJ_L1_MARK(3);
        const uint32_t nlev = s_var[1];
        if (s_var[2] != 0u) {                                // deeper than the level table: the wavefront kernel takes the chunk
            if (tid == 0) meta[c] = ParseMeta{0u, kRouteWave};
            continue;
        }

        // ---- K: slots per level, the matches sorted by level ----
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
            if (lane == 0) s_var[4] = total;
        }
        __syncthreads();
        // every thread: its records' literal runs into registers (the records are dead after this pass) + the slot of every match
        uint32_t l_src[4], l_dst[4], l_len[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = tid + (uint32_t)i * kL1Threads;
            l_src[i] = 0u; l_dst[i] = 0u; l_len[i] = 0u;
            if (r < nseq) {
                const uint2 t = s_rec[r];
                const uint32_t nstart0 = s_rec[r + 1u].y & 0xffffu;
                const uint32_t lit = t.x >> 16, start = t.y & 0xffffu, off = t.y >> 16, dst = start + lit;
                l_src[i] = t.x & 0xffffu; l_dst[i] = start; l_len[i] = lit;
                const uint32_t L = lds_ld16(a_lvl + 2u * r);
                if (L) {
                    const uint32_t m = (nstart0 - dst) & 0xffffu;
                    uint32_t sl;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(sl) : "v"(a_hist + 4u * L), "v"(1u) : "memory");
                    lds_st16(a_sd + 2u * sl, dst);
                    lds_st16(a_so + 2u * sl, off);
                    lds_st16(a_sm + 2u * sl, m);
                }
            }
        }
        __syncthreads();                                     // the sorted list is complete; the window's scratch is dead
        CJ_L1_MARK(4);
        const uint32_t nm = s_var[4];

        // ---- D2: literals, staging buffer -> window ----
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if ((uint32_t)i * kL1Threads < nseq)             // (uniform)
                lvl_run_copy(l_len[i] > 0u, a_out + l_dst[i], a_in + l_src[i], l_len[i], dummy_w);
        }
        __syncthreads();
        CJ_L1_MARK(5);

        // ---- D3: matches, level by level ----
#ifndef CJ_L1_X_NOD3
        {
            const auto entry = [&](uint32_t e, uint32_t& dst, uint32_t& off, uint32_t& m) {
                asm volatile("ds_read_u16 %0, %3\n\tds_read_u16 %1, %4\n\tds_read_u16 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(dst), "=&v"(off), "=&v"(m) : "v"(a_sd + 2u * e), "v"(a_so + 2u * e), "v"(a_sm + 2u * e) : "memory");
            };
            uint32_t pos = 0, L = 1, nbar = 0;
            uint32_t e_dst = 0, e_off = 0, e_m = 0;
            if (tid < nm) entry(tid, e_dst, e_off, e_m);
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
                uint32_t n_dst = 0, n_off = 0, n_m = 0;
                if (seg_end + tid < nm) entry(seg_end + tid, n_dst, n_off, n_m);          // the next segment's descriptors
#ifndef CJ_L1_X_NOCOPY
                if (multi) {
                    lvl_match_copy(pos + tid < seg_end, a_out, s_out, e_dst, e_off, e_m, dummy_w);
                    for (uint32_t b = pos + kL1Threads; b < seg_end; b += kL1Threads) {
                        uint32_t d2 = 0, o2 = 0, m2 = 0;
                        if (b + tid < seg_end) entry(b + tid, d2, o2, m2);
                        lvl_match_copy(b + tid < seg_end, a_out, s_out, d2, o2, m2, dummy_w);
                    }
                } else if (wave == 0u) {
                    // lane i holds descriptor pos + i: level Lc owns the lanes [start of Lc - pos, end of Lc - pos)
                    uint32_t lo_l = 0;
                    for (uint32_t Lc = L; Lc < Lnext; Lc++) {
                        const uint32_t hi_l = rdlane(le, uni(Lc - L)) - pos;
                        lvl_match_copy(lane >= lo_l && lane < hi_l, a_out, s_out, e_dst, e_off, e_m, dummy_w);
                        lo_l = hi_l;
                    }
                }
#endif
                if (multi || next_multi || Lnext > nlev) { __syncthreads(); nbar += 1; }
                pos = seg_end; L = Lnext; e_dst = n_dst; e_off = n_off; e_m = n_m;
            }
            if (prof && tid == 0) { atomicAdd(&g_lvl1_phase_cycles[9], (unsigned long long)nlev); atomicAdd(&g_lvl1_phase_cycles[10], (unsigned long long)nbar); }
        }
#endif
        CJ_L1_MARK(6);

        // ---- D4: stream the window out (16 B per lane), exact tail ----
        {
            const uint32_t nvec = U >> 4;
            const uint4* src = reinterpret_cast<const uint4*>(s_out);
            for (uint32_t i = tid; i < nvec; i += kL1Threads) st16u_nt(out + 16u * i, src[i]);
            for (uint32_t i = (nvec << 4) + tid; i < U; i += kL1Threads) out[i] = s_out[i];
        }
        if (tid == 0) meta[c] = ParseMeta{0u, 0u};           // done: the table decoder that runs after this kernel skips the chunk
        if (prof) { __syncthreads(); CJ_L1_MARK(7); if (tid == 0) atomicAdd(&g_lvl1_phase_cycles[8], 1ull); }
    }
}

template <int kCodec>
__global__ __launch_bounds__(kL1Threads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lz4_decode_lvl1_kernel(BatchArgs a, const uint2* sync, ParseMeta* meta,
                                                                                                             uint32_t* counter) {
    lvl1_body<kCodec>(a, sync, meta, counter);
}

void launch_lz4_decode_lvl1(const BatchArgs& a, const void* sync, void* meta, uint32_t* counter, uint32_t grid, hipStream_t s, int codec) {
    if (a.n_chunks == 0) return;
    if (codec == CJ_CODEC_SNAPPY_RAW) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_lvl1_kernel<CJ_CODEC_SNAPPY_RAW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL1Bytes);
        hipLaunchKernelGGL((lz4_decode_lvl1_kernel<CJ_CODEC_SNAPPY_RAW>), dim3(grid), dim3(kL1Threads), kL1Bytes, s, a, (const uint2*)sync, (ParseMeta*)meta, counter);
        return;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_lvl1_kernel<CJ_CODEC_LZ4_BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL1Bytes);
    hipLaunchKernelGGL((lz4_decode_lvl1_kernel<CJ_CODEC_LZ4_BLOCK>), dim3(grid), dim3(kL1Threads), kL1Bytes, s, a, (const uint2*)sync, (ParseMeta*)meta, counter);
}

}  // namespace cj

extern "C" CJ_API int cj_debug_lvl1_phase_cycles(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(cj::g_lvl1_phase_cycles), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cj::g_lvl1_phase_cycles), z, 128) != hipSuccess) return -1; }
    return 0;
}
