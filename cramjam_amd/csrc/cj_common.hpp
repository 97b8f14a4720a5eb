// cj_common.hpp — shared device helpers and launch declarations for the gfx950 block-codec kernels.
// Wave = 64 lanes (CDNA4).  Every kernel here maps ONE WAVEFRONT to ONE independent chunk; all
// stream-position state is wave-uniform (SGPR) and the lanes only do the wide byte moves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cramjam_hip_debug.h"      // (the drop-in ABI + the test / benchmark exports)

namespace cj {

struct BatchArgs {
    const uint8_t* in_base;
    const uint64_t* in_off;
    const uint64_t* in_len;
    uint8_t* out_base;
    const uint64_t* out_off;
    const uint64_t* out_cap;
    int64_t* result;
    uint32_t n_chunks;
    uint32_t flags;
    const uint32_t* hist;       // linked LZ4-frame blocks only (frame.hip): bytes of history before out_off[i]; nullptr = none
};

// internal flag bits (never part of the C-ABI; CJ_FLAG_DEBUG_PROFILE 0x1000 is public): 0x2000 = linked-frame parse
// (bit 63 of in_len marks a STORED block; no minimum sequence count for the LDS decoder)
constexpr uint32_t kFlagLinkedFrame = 0x2000u;
constexpr uint32_t kFlagSplitPieces = 0x8000u;     // encoders (large.hip): the chunks are consecutive sub-pieces of 64 KiB pieces, `per` of them per piece
                                                   // (flag bits 16..19 = log2(per): 4 = quarters of 16 KiB, 16 = sub-pieces of 4 KiB); position 0 of a
                                                   // sub-piece's walk = the start of its piece, what lies before it is indexed first (ht_preindex)
constexpr uint32_t kFlagSplitShift = 16;
__host__ __device__ inline uint32_t split_per(uint32_t flags) { return 1u << ((flags >> kFlagSplitShift) & 15u); }
constexpr uint32_t kFlagReportTail = 0x4000u;      // LZ4 encoder (large.hip): result = size | length of the final literal run << 32

constexpr int kWavesPerBlock = 4;
constexpr int kBlockThreads = 64 * kWavesPerBlock;

void launch_lz4_decode(const BatchArgs& a, hipStream_t s);        // one wavefront per chunk
void launch_lz4_decode_lanes(const BatchArgs& a, hipStream_t s);  // one lane per chunk
// parse (lane per chunk) + decode (workgroup per chunk, LDS-resident window); sync/meta are engine scratch
void launch_lz4_parse(const BatchArgs& a, void* sync, void* meta, hipStream_t s);
void launch_lz4_decode_routed(const BatchArgs& a, const void* meta, hipStream_t s);   // wave kernel on chunks the parse kernel routed to it
// variant 2: persistent grid (2 workgroups per CU), record tables in the global scratch `tabs`, chunk indices from *counter
// codec: CJ_CODEC_LZ4_BLOCK or CJ_CODEC_SNAPPY_RAW (only the record expansion D1 differs)
void launch_lz4_decode_lds2(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter, uint32_t grid, hipStream_t s, int codec = 0, uint32_t win = 65536u);
uint32_t lz4_lds2_wgs_per_cu(uint32_t win);
// linked LZ4-frame blocks: one workgroup walks the blocks of a frame in order, previous block kept as a second LDS window
void launch_lz4_decode_lds2_linked(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter,
                                   const void* frames, uint32_t n_frames, uint32_t grid, hipStream_t s);
// parse + decode in ONE kernel: the segmented parse runs inside the workgroup on the staged chunk; meta[c] = kRouteWave for chunks it leaves to the wave kernel
void launch_lz4_decode_fused(const BatchArgs& a, void* meta, void* tabs, uint32_t* counter, uint32_t grid, hipStream_t s, int codec = 0, uint32_t win = 65536u);
size_t lz4_lds2_tab_bytes(uint32_t grid, uint32_t win = 65536u);
// the batch decoders' chunk counters (lz4_decode_lds.hip: a workgroup's first chunk is its own index, the others come from counter blockIdx % kClaimCounters)
#ifndef CJ_CLAIM_COUNTERS
#define CJ_CLAIM_COUNTERS 32u
#endif
constexpr uint32_t kClaimCounters = CJ_CLAIM_COUNTERS, kClaimStride = 64u;      // (stride in 4-byte words: one counter per 256 bytes)
constexpr size_t kClaimBytes = (size_t)kClaimCounters * kClaimStride * 4u;
size_t lz4_lds_scratch_sync_bytes(size_t n_chunks);
size_t lz4_lds_scratch_meta_bytes(size_t n_chunks);
// encoders: one workgroup of two wavefronts per chunk (one wavefront per sub-piece of a split piece, large.hip)
hipError_t launch_lz4_encode(const BatchArgs& a, hipStream_t s);
void launch_snappy_decode(const BatchArgs& a, hipStream_t s);                                   // one wavefront per chunk
void launch_snappy_decode_lanes(const BatchArgs& a, hipStream_t s);                            // one lane per chunk
void launch_snappy_parse(const BatchArgs& a, void* sync, void* meta, hipStream_t s);
// large.hip: one large buffer cut into pieces that are compressed as a batch and joined into one stream
int64_t large_snappy_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
int64_t large_lz4_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, bool prefix);
int64_t large_lz4_frame_blocks(const uint8_t* in, size_t n, uint8_t* out, size_t cap);      // <= large_split_max() bytes
int64_t large_snappy_frame(const uint8_t* in, size_t n, uint8_t* out, size_t cap);           // <= large_split_max() bytes
size_t large_split_max();
int large_decompress_many(::cj_engine* e, int codec, size_t nj, const uint8_t* const* ins, const size_t* lens, const size_t* starts, uint8_t* const* outs, const size_t* caps, int64_t* result);
// the large chunks of a host batch: prologue on the host, then large_decompress_many; result[i] set for every listed chunk
int large_decompress_listed(::cj_engine* e, int codec, uint32_t flags, size_t n_listed, const size_t* idx, const uint8_t* const* in_ptrs, const size_t* in_lens,
                            uint8_t* const* out_ptrs, const size_t* out_caps, int64_t* result);
int64_t large_decompress(int codec, uint32_t flags, const uint8_t* in, size_t n, uint8_t* out, size_t cap);
bool large_few_elements(int codec, uint32_t flags, const uint8_t* in, size_t n, size_t cap);   // a small stream of a handful of long runs: the one-wavefront kernel is quicker
void launch_lz4_decode_lds2_slabs(const BatchArgs& a, const void* sync, const void* meta, void* tabs, uint32_t* counter,
                                  const void* first, uint32_t stream_len, uint32_t* done, void* cross, uint32_t tab_stride,
                                  uint32_t cross_stride, uint32_t grid, hipStream_t s, int codec, bool rel = false);
void launch_snappy_decode_routed(const BatchArgs& a, const void* meta, hipStream_t s);       // wave kernel on chunks the parse kernel routed to it
hipError_t launch_snappy_encode(const BatchArgs& a, hipStream_t s);

#if defined(__HIPCC__)

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
#ifdef CJ_BALLOT_BUILTIN
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#else
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
#endif
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ uint32_t ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }

// compiler-only ordering point between cooperative stores and the loads that read them back.
// Lanes of one wavefront share the CU's vector L1 and issue in order, so no hardware wait is needed
// (LLVM AMDGPU memory model: wavefront-scope fences lower to nothing).
__device__ __forceinline__ void wave_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// unaligned 32-bit global load (gfx950 amdhsa runs in unaligned-access mode)
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// ---- 512-byte register window over the compressed stream -------------------------------------
// Two VGPRs hold 128 consecutive dwords of the (dword-aligned) input; any 4 bytes inside it are
// reachable with v_readlane (≈10 cycles) instead of an LDS/global round trip (≥64/≥500 cycles).
struct InWindow {
    const uint8_t* base;   // dword-aligned start of the input (in - misalignment)
    uint32_t iend;         // end position relative to base
    uint32_t wpos;         // window start (multiple of 4) relative to base
    uint32_t w0, w1;       // VGPR: lane l holds dword at base + wpos + 4*l (w0) / + 256 + 4*l (w1)

    __device__ __forceinline__ uint32_t load_row(uint32_t start) const {
        uint32_t off = start + 4u * lane_id();
        uint32_t v = 0;
        if (off < iend) v = *reinterpret_cast<const uint32_t*>(base + off);
        return v;
    }
    __device__ __forceinline__ void anchor(uint32_t pos) {
        wpos = pos & ~3u;
        w0 = load_row(wpos);
        w1 = load_row(wpos + 256u);
    }
    // make sure [pos, pos+need) with need <= 8 lies inside the window and pos - wpos < 256 when cheap
    __device__ __forceinline__ void ensure(uint32_t pos) {
        uint32_t q = pos - wpos;
        if (q >= 256u) {
            if (q < 504u) {
                w0 = w1;
                wpos += 256u;
                w1 = load_row(wpos + 256u);
            } else {
                anchor(pos);
            }
        }
    }
    // 4 bytes at pos (little endian); requires pos - wpos <= 507
    __device__ __forceinline__ uint32_t fetch32(uint32_t pos) const {
        uint32_t q = pos - wpos;
        uint32_t idx = q >> 2, sh = (q & 3u) * 8u;
        uint32_t a0 = rdlane(w0, idx & 63u), a1 = rdlane(w1, idx & 63u);
        uint32_t b0 = rdlane(w0, (idx + 1u) & 63u), b1 = rdlane(w1, (idx + 1u) & 63u);
        uint32_t lo = idx < 64u ? a0 : a1;
        uint32_t hi = (idx + 1u) < 64u ? b0 : b1;
        uint64_t v = ((uint64_t)hi << 32) | lo;
        return (uint32_t)(v >> sh);
    }
    // safe anywhere: re-anchors when pos is outside the comfortable range
    __device__ __forceinline__ uint32_t fetch32_any(uint32_t pos) {
        if (pos - wpos > 500u) anchor(pos);
        return fetch32(pos);
    }
};

// cooperative forward copy of n bytes, non-overlapping (src, dst global)
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, uint32_t n) {
    const uint32_t lane = lane_id();
    if (n <= 64u) {
        if (lane < n) dst[lane] = src[lane];
        return;
    }
    uint32_t k = 0;
    if (n >= 2048u) {   // long runs (incompressible data): 16 B per lane per step
        for (; k + 1024u <= n; k += 1024u) {
            uint4 v;
            __builtin_memcpy(&v, src + k + 16u * lane, 16);
            __builtin_memcpy(dst + k + 16u * lane, &v, 16);
        }
    }
    for (; k < n; k += 64u) {
        uint32_t j = k + lane;
        if (j < n) dst[j] = src[j];
    }
}

// cooperative LZ77 match copy: dst[j] = dst[j - d] for j in [0, m), d >= 1; the d source bytes
// before dst are already written.  Overlap (d < m) is resolved by reading the periodic pattern.
__device__ __forceinline__ void wave_match_copy(uint8_t* dst, uint32_t d, uint32_t m) {
    const uint32_t lane = lane_id();
    const uint8_t* src = dst - d;
    if (d >= m) {
        if (m <= 64u) {
            if (lane < m) dst[lane] = src[lane];
        } else {
            for (uint32_t k = 0; k < m; k += 64u) {
                uint32_t j = k + lane;
                if (j < m) dst[j] = src[j];
            }
        }
        return;
    }
    uint32_t r = lane, step = 64u;
    if (d <= 64u) {
        r = lane % d;
        step = 64u % d;
    }
    for (uint32_t k = 0; k < m; k += 64u) {
        uint32_t j = k + lane;
        if (j < m) dst[j] = src[r];
        r += step;
        if (r >= d) r -= d;
    }
}

#endif  // __HIPCC__
}  // namespace cj
