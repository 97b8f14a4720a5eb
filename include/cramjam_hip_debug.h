/* cramjam_hip_debug.h — test and benchmark utilities that libcramjam_hip.so exports NEXT TO the drop-in ABI of cramjam_hip.h
 * (cramjam_amd/csrc/bench_util.hip and debug counters of the decoders).  A binding of the reference's call sites needs none of them;
 * tests/test_cabi.py pins the two export lists separately. */
#ifndef CRAMJAM_HIP_DEBUG_H
#define CRAMJAM_HIP_DEBUG_H
#include "cramjam_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* n chunks of S bytes at d_out + i*stride = synth-v1(S, first_index + i, seed), generated on the device */
CJ_API int cj_bench_synth_v1(void* d_out, uint64_t stride, uint64_t S, uint64_t first_index, uint64_t n, uint64_t seed, void* stream);
/* *d_mismatches += chunks i in [0, n) with got[got_off[i] .. +S) != want[(i % n_unique)*want_stride .. +S) */
CJ_API int cj_bench_compare(const void* d_got, const uint64_t* d_got_off, const void* d_want, uint64_t want_stride,
                            uint32_t n_unique, uint64_t S, uint32_t n, void* d_mismatches, void* stream);
/* per-phase cycle counters of the workgroup decoder (CJ_FLAG_DEBUG_PROFILE on a device batch): S0, D1, D2, D3, D4, chunks, 6.. sub-phases.
 * out16 must hold SIXTEEN 64-bit slots (128 bytes; it was eight until round 3) */
CJ_API int cj_debug_lds_phase_cycles(unsigned long long* out16, int reset);
/* chunks of the one-kernel decode path (batches with CJ_FLAG_DEBUG_PROFILE) by how their parse finished: from the listed walks, walked P4, walked P3 + P4 (lds_shared.hpp: fused_parse) */
CJ_API int cj_debug_fused_parse_paths(unsigned long long* out3, int reset);
CJ_API long long cj_debug_forwarded_chunks(int reset);            /* chunks / slabs (of calls with CJ_FLAG_DEBUG_PROFILE) that went through the forwarding phase */
CJ_API unsigned long long cj_debug_linked_lds_frames(void);       /* linked-block LZ4 frames decoded by the two-window decoder */
/* large-stream path with its parse stage's absolute sync points handed back (tests compare them with a serial walk) */
CJ_API int64_t cj_debug_big_parse(int codec, uint32_t flags, const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                  uint32_t* sync_pairs, size_t max_pairs, uint64_t* n_seq);

/* bytes of device scratch the engine holds for CJ_FLAG_BIG_CHUNKS batches (list, record areas, summaries, the slab decoder's tables) */
CJ_API uint64_t cj_debug_big_scratch_bytes(cj_engine* e);

#ifdef __cplusplus
}
#endif
#endif
