// lz4_decode_lds.hip — LZ4 *block* decoder, one WORKGROUP (16 wavefronts) per chunk, with the whole
// 64 KiB output window, the compressed chunk and a sequence table resident in LDS (160 KiB/CU on gfx950).
//
// Runs after lz4_parse_kernel (lz4_decode_lanes.hip), which validated the stream, computed the decoded
// size and left an (ip, op) sync point every 8 sequences.  Same results as the other two mappings
// (reference call sites /root/reference/src/lz4.rs:88,90,164,168).
//
// Why: with one wave/lane per chunk every match copy is a dependent read of the chunk's own earlier
// output somewhere in the last 64 KiB — an HBM round trip of a 128 B line for ~18 useful bytes.  Here
// the history never leaves the CU: HBM sees exactly the algorithmic bytes (compressed chunk in with
// 16 B/lane coalesced loads, 64 KiB out with 16 B/lane coalesced stores), and the serial token chain
// is broken by the sync points: phases per chunk
//   S0 stage compressed bytes into LDS, clear the ready bitmap
//   D1 one thread per sync point re-walks 8 sequences in LDS and writes 16 B sequence records
//   D2 one lane per sequence copies its literals LDS->LDS and marks them ready (bitmap: 1 bit/byte)
//   D3 one lane per sequence resolves its match as soon as the bitmap says its source bytes are final;
//      the earliest unresolved match is always ready, so the spin is deadlock-free (and bounded anyway);
//      long copies (> 64 B) are done cooperatively by the whole wavefront
//   D4 stream the finished window to HBM
#include "lz4_lane_walk.hpp"

namespace cj {

constexpr uint32_t kLdsThreads = 1024;
constexpr uint32_t kLdsWaves = kLdsThreads / 64;
constexpr uint32_t kOffOut = 0;
constexpr uint32_t kOffBits = 65536;                 // 2048 x u32: one ready bit per output byte
constexpr uint32_t kOffIn = kOffBits + 8192;         // compressed bytes, then the record table
constexpr uint32_t kLdsBytes = 163840;               // all 160 KiB, one dynamic region (no static LDS: keeps the base 16 B aligned)
constexpr uint32_t kOffVars = kLdsBytes - 16;        // [0] = spin-limit failure flag
constexpr uint32_t kInTableBytes = kOffVars - kOffIn;
constexpr uint32_t kShortMax = 64;                   // copies up to this length are done by one lane
constexpr uint32_t kSpinLimit = 1u << 18;

// record: x = literal source index in s_in, y = literal length, z = match destination (= op after literals),
//         w = offset | match length << 16  (match length 0 on the final, literal-only sequence)

__device__ __forceinline__ void bits_set(uint32_t* bits, uint32_t lo, uint32_t hi) {     // [lo, hi), hi > lo
    uint32_t w0 = lo >> 5, w1 = (hi - 1u) >> 5;
    for (uint32_t w = w0; w <= w1; w++) {
        uint32_t m = ~0u;
        if (w == w0) m &= ~0u << (lo & 31u);
        if (w == w1) m &= ~0u >> (31u - ((hi - 1u) & 31u));
        __hip_atomic_fetch_or(&bits[w], m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__device__ __forceinline__ bool bits_ready(uint32_t* bits, uint32_t lo, uint32_t hi) {   // [lo, hi), hi > lo
    uint32_t w0 = lo >> 5, w1 = (hi - 1u) >> 5;
    for (uint32_t w = w0; w <= w1; w++) {
        uint32_t m = ~0u;
        if (w == w0) m &= ~0u << (lo & 31u);
        if (w == w1) m &= ~0u >> (31u - ((hi - 1u) & 31u));
        uint32_t v = __hip_atomic_load(&bits[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((v & m) != m) return false;
    }
    return true;
}

// whole-wave versions for long ranges (all lanes call with the same lo/hi)
__device__ __forceinline__ void wave_bits_set(uint32_t* bits, uint32_t lo, uint32_t hi) {
    const uint32_t w0 = lo >> 5, w1 = (hi - 1u) >> 5;
    for (uint32_t w = w0 + lane_id(); w <= w1; w += 64u) {
        uint32_t m = ~0u;
        if (w == w0) m &= ~0u << (lo & 31u);
        if (w == w1) m &= ~0u >> (31u - ((hi - 1u) & 31u));
        __hip_atomic_fetch_or(&bits[w], m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ __launch_bounds__(kLdsThreads) void lz4_decode_lds_kernel(BatchArgs a, const uint2* sync, const ParseMeta* meta) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_out = smem + kOffOut;
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(smem + kOffBits);
    uint8_t* s_in = smem + kOffIn;

    const uint32_t c = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const ParseMeta pm = meta[c];
    if (pm.nseq == 0u) return;                           // error, empty, or already decoded by the parse kernel
    const uint32_t nseq = pm.nseq;
    const uint32_t U = (uint32_t)a.result[c];            // decoded size, 1..65536
    const uint8_t* in = a.in_base + a.in_off[c] + pm.in_skip;
    const uint32_t iend = (uint32_t)a.in_len[c] - pm.in_skip;
    uint8_t* out = a.out_base + a.out_off[c];

    // ---- S0: stage the compressed chunk (16 B aligned loads), clear the bitmap ----
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);
    {
        const uint4* src = reinterpret_cast<const uint4*>(in - mis);
        uint4* dst = reinterpret_cast<uint4*>(s_in);
        const uint32_t nvec = (mis + iend + 15u) >> 4;
        for (uint32_t i = tid; i < nvec; i += kLdsThreads) dst[i] = src[i];
        for (uint32_t i = tid; i < 2048u; i += kLdsThreads) s_bits[i] = 0u;
    }
    const uint32_t tb_off = (mis + iend + 15u) & ~15u;
    uint4* table = reinterpret_cast<uint4*>(s_in + tb_off);
    const uint32_t tcap = (kInTableBytes - tb_off) >> 4;                 // records that fit
    const uint32_t sp_per_slab = tcap / kSyncEvery;                      // >= 1 by construction (iend <= kLdsInMax)
    const uint32_t nsp = (nseq + kSyncEvery - 1u) / kSyncEvery;
    const uint2* csync = sync + (size_t)c * kSyncStride;
    uint32_t* s_fail = reinterpret_cast<uint32_t*>(smem + kOffVars);
    if (tid == 0) *s_fail = 0u;
    __syncthreads();

    for (uint32_t sp0 = 0; sp0 < nsp; sp0 += sp_per_slab) {
        const uint32_t sp1 = min(nsp, sp0 + sp_per_slab);
        const uint32_t seq0 = sp0 * kSyncEvery;
        const uint32_t nrec = min(nseq, sp1 * kSyncEvery) - seq0;

        // ---- D1: expand sync points into sequence records ----
        for (uint32_t sp = sp0 + tid; sp < sp1; sp += kLdsThreads) {
            const uint2 p = csync[sp];
            uint32_t ip = p.x + mis, op = p.y;
            uint32_t s = sp * kSyncEvery;
            for (uint32_t j = 0; j < kSyncEvery && s < nseq; j++, s++) {
                const uint32_t token = s_in[ip++];
                uint32_t lit = token >> 4;
                if (lit == 15u) { uint32_t b; do { b = s_in[ip++]; lit += b; } while (b == 255u); }
                const uint32_t lit_src = ip;
                ip += lit; op += lit;
                uint32_t w = 0;
                uint32_t mlen = 0;
                if (s + 1u < nseq) {
                    const uint32_t offset = (uint32_t)s_in[ip] | ((uint32_t)s_in[ip + 1] << 8);
                    ip += 2;
                    mlen = token & 15u;
                    if (mlen == 15u) { uint32_t b; do { b = s_in[ip++]; mlen += b; } while (b == 255u); }
                    mlen += 4u;
                    w = offset | (mlen << 16);
                }
                table[s - seq0] = make_uint4(lit_src, lit, op, w);
                op += mlen;
            }
        }
        __syncthreads();

        // ---- D2: literals, one lane per sequence ----
        for (uint32_t base = wave * 64u; base < nrec; base += kLdsThreads) {
            const uint32_t r = base + lane;
            uint4 rec = make_uint4(0, 0, 0, 0);
            if (r < nrec) rec = table[r];
            const uint32_t n = rec.y, src = rec.x, dst = rec.z - rec.y;
            if (n > 0u && n <= kShortMax) {
                uint32_t k = 0;
                for (; k + 8u <= n; k += 8u) {
                    uint8_t t[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) t[q] = s_in[src + k + q];
#pragma unroll
                    for (int q = 0; q < 8; q++) s_out[dst + k + q] = t[q];
                }
                for (; k < n; k++) s_out[dst + k] = s_in[src + k];
                bits_set(s_bits, dst, dst + n);
            }
            uint64_t longm = ballot64(n > kShortMax);
            while (longm) {
                const uint32_t l = ctz64(longm);
                longm &= longm - 1u;
                const uint32_t ln = rdlane(n, l), ls = rdlane(src, l), ld = rdlane(dst, l);
                for (uint32_t k = lane; k < ln; k += 64u) s_out[ld + k] = s_in[ls + k];
                wave_bits_set(s_bits, ld, ld + ln);
            }
        }
        __syncthreads();

        // ---- D3: matches, one lane per sequence, dependency-exact through the ready bitmap ----
        for (uint32_t base = wave * 64u; base < nrec; base += kLdsThreads) {
            const uint32_t r = base + lane;
            uint4 rec = make_uint4(0, 0, 0, 0);
            if (r < nrec) rec = table[r];
            const uint32_t dst = rec.z, off = rec.w & 0xffffu, m = rec.w >> 16;
            const uint32_t src = dst - off;
            const uint32_t need = off < m ? off : m;          // distinct source bytes
            bool pending = m > 0u;
            uint32_t spins = 0;
            while (ballot64(pending) != 0ull) {
                bool ready = false;
                if (pending) ready = bits_ready(s_bits, src, src + need);
                if (ready && m <= kShortMax) {
                    if (off >= 8u) {
                        uint32_t k = 0;
                        for (; k + 8u <= m; k += 8u) {
                            uint8_t t[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) t[q] = s_out[src + k + q];
#pragma unroll
                            for (int q = 0; q < 8; q++) s_out[dst + k + q] = t[q];
                        }
                        for (; k < m; k++) s_out[dst + k] = s_out[src + k];
                    } else {
                        for (uint32_t k = 0; k < m; k++) s_out[dst + k] = s_out[src + k];
                    }
                    bits_set(s_bits, dst, dst + m);
                    pending = false;
                }
                uint64_t longm = ballot64(ready && m > kShortMax);
                while (longm) {
                    const uint32_t l = ctz64(longm);
                    longm &= longm - 1u;
                    const uint32_t lm = rdlane(m, l), lo = rdlane(off, l), ld = rdlane(dst, l);
                    const uint32_t ls = ld - lo;
                    // periodic pattern read: every byte comes from the lo bytes before ld (already final)
                    uint32_t rr = lane, step = 64u;
                    if (lo <= 64u) { rr = lane % lo; step = 64u % lo; }
                    for (uint32_t k = lane; k < lm; k += 64u) {
                        s_out[ld + k] = s_out[ls + (lo >= lm ? k : rr)];
                        rr += step;
                        if (rr >= lo) rr -= lo;
                    }
                    wave_bits_set(s_bits, ld, ld + lm);
                    if (lane == l) pending = false;
                }
                if (++spins > kSpinLimit) { *s_fail = 1u; break; }
            }
        }
        __syncthreads();
    }

    // ---- D4: stream the window out (16 B per lane), exact tail ----
    {
        const uint32_t nvec = U >> 4;
        const uint4* src = reinterpret_cast<const uint4*>(s_out);
        for (uint32_t i = tid; i < nvec; i += kLdsThreads) st16u(out + 16u * i, src[i]);
        for (uint32_t i = (nvec << 4) + tid; i < U; i += kLdsThreads) out[i] = s_out[i];
    }
    if (tid == 0 && *s_fail) a.result[c] = CJ_E_CORRUPT;    // cannot happen for a stream the parse kernel accepted
}

void launch_lz4_decode_lds(const BatchArgs& a, const void* sync, const void* meta, hipStream_t s) {
    if (a.n_chunks == 0) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);   // per device; cheap
    hipLaunchKernelGGL(lz4_decode_lds_kernel, dim3(a.n_chunks), dim3(kLdsThreads), kLdsBytes, s, a,
                       (const uint2*)sync, (const ParseMeta*)meta);
}

size_t lz4_lds_scratch_sync_bytes(size_t n_chunks) { return n_chunks * (size_t)kSyncStride * sizeof(uint2); }
size_t lz4_lds_scratch_meta_bytes(size_t n_chunks) { return n_chunks * sizeof(ParseMeta); }

}  // namespace cj
