// seg_parse.hpp — interface between the SEGMENTED parse kernel (seg_parse.hip) and the workgroup decoder's record mode
// (lz4_decode_lds.hip, kRecs): where a chunk's records live and how the decoder finds record i.
//
// The parse kernel gives every chunk k = 1, 2, 4 or 8 LANES; lane j walks the sequences whose token lies in its SEGMENT of the
// compressed bytes and writes their 8-byte records { lit_src | lit << 16, lit_start | offset << 16 } — the decoder's own
// compact format — into ITS region of the chunk's record area, numbered from 0 and with output positions counted from 0:
// what lies in front of its segment (how many sequences, how many output bytes) is only known when the lanes before it have
// finished.  The kernel's epilogue adds that up (SegMeta), and the decoder translates: record i of the chunk is entry
// i - first[j] of region j, and its output position is the stored one + opb[j] (mod 2^16).
#pragma once
#include "lz4_lane_walk.hpp"

namespace cj {

constexpr uint32_t kSegMax = 8;                                             // lanes (segments) per chunk at most
constexpr uint32_t kRecPitch = kSyncStride * kSyncEvery + kSegMax * 64u;    // record slots per chunk (16 896 x 8 bytes): 16 384 records + a local sentinel per region
struct SegMeta {
    uint32_t first[kSegMax];     // first[j] = index (in the chunk) of the first record of region j; a lane without records has the next lane's value.
                                 // first[0] is 0 by definition and carries the chunk's count of matches with an offset below 4 KiB instead
    uint32_t opb[kSegMax];       // bits 0..19: output position of region j's first record; bits 20..21: the region slot that holds it (the parse
                                 // kernel stores records in groups of four slots whose phase is the wavefront's, not the lane's)
};
__host__ __device__ inline uint32_t seg_region_slots(uint32_t k) { return kRecPitch / k; }

size_t seg_recs_bytes(size_t n_chunks);
size_t seg_meta_bytes(size_t n_chunks);
// klog = log2(lanes per chunk); recs: n_chunks x kRecPitch x 8 bytes; segmeta: n_chunks entries; meta: n_chunks entries (zeroed by the caller)
void launch_seg_parse(const BatchArgs& a, int codec, uint32_t klog, void* recs, void* segmeta, void* meta, hipStream_t s);
// the workgroup decoder on the parse kernel's records (no staging of the compressed chunk, no record expansion)
void launch_lz4_decode_recs(const BatchArgs& a, const void* recs, const void* segmeta, const void* meta, void* tabs, uint32_t* counter,
                            uint32_t klog, uint32_t grid, hipStream_t s, int codec);

}  // namespace cj
