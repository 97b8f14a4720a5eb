// big_chunks.hpp — chunks of 64 KiB .. 256 KiB in a device batch (BASELINE configs[4]): interface between the segmented parse
// kernels for such chunks (big_chunks.hip) and the workgroup decoder's slab mode fed with records (lz4_decode_lds.hip, kRecFeed).
//
// One chunk above 64 KiB is one token stream but several windows of output.  Its SLABS of output (kBigSlabBytes each) are decoded by the slab
// mode of the workgroup decoder (one window in LDS, kBigSlabWgsPerCu workgroups per CU): a match whose source lies before the slab is
// copied from the finished output of the earlier slabs in global memory (L2), and the work items are claimed in SLAB-MAJOR
// order over all listed chunks — when slab s of a chunk is claimed, its slab s - 1 finished thousands of claims ago, so the
// completion flags of the slab mode never make anybody wait.  What the slab decoder needs is the chunk's sequences as RECORDS,
// and which of them touch each slab:
//   * the parse kernel gives the chunk kBigLanes lanes; lane j finds the token chain by a lead-in in front of its segment of the
//     compressed bytes and walks its part of it (the scheme of tools/experiments/seg_parse, where it was measured for 64 KiB
//     chunks: bit-exact, but no faster than one lane per chunk THERE because 100 000 chunks fill the GPU anyway — 8 192 chunks
//     of 256 KiB are 128 wavefronts of a 4x longer chain, and here the lanes are what fills the GPU);
//   * a lane writes 16-byte records { lit_src | mlen_hi << 24, lit, lit_start, offset | mlen_lo << 16 } into its REGION of the
//     chunk's record area, numbered and positioned from its own start; the epilogue sums up what lies in front of each region
//     (BigMeta: first record, first output byte) and finds, per slab boundary, the record that holds that output byte;
//   * the decoder's D1 reads records instead of walking tokens: one thread per record, clipped to the slab.
#pragma once
#include "lz4_lane_walk.hpp"

namespace cj {

constexpr uint32_t kBigLanesLog = 5, kBigLanes = 1u << kBigLanesLog;        // lanes (segments) per chunk: 8 192 chunks x 32 lanes = 4 096 wavefronts, what sixteen per CU hold
constexpr uint32_t kBigOutMax = 4u * 65536u;                                // decoded size of a chunk on this path, at most
constexpr uint32_t kBigInMax = kBigOutMax + kBigOutMax / 128u + 64u;         // compressed size, at most (an incompressible LZ4 block: n + n / 255 + 16)
constexpr uint32_t kBigRegion = 2048u;                                       // record slots per region (a chunk of text has ~1 300 sequences per lane); a lane with more hands the chunk to the wavefront kernel
constexpr uint32_t kBigRecPitch = kBigLanes * kBigRegion;                    // record slots per chunk (16 bytes each: 1 MiB)
// The slabs are 32 KiB since round 6: four workgroups of four wavefronts per CU instead of two of eight — four slabs' dependency chains in
// flight for the same 16 wavefronts (profiles/r06/experiments h01-h04: +32 % on 32 KiB pieces; this mode has no S0 and a D1 of one thread
// per record, so a smaller piece does not pay a whole chunk's fixed latencies as it would on the 64 KiB path).
#ifndef CJ_BIG_SLAB_BYTES
#define CJ_BIG_SLAB_BYTES 32768
#endif
constexpr uint32_t kBigSlabBytes = CJ_BIG_SLAB_BYTES;                       // the slab decoder's window
constexpr uint32_t kBigSlabThreads = kBigSlabBytes >= 65536u ? 512u : 256u; // its workgroup ...
constexpr uint32_t kBigSlabWgsPerCu = kBigSlabBytes >= 65536u ? 2u : 4u;    // ... and how many of them share a CU (LDS: window + bitmap; registers: 16 wavefronts of 128)
constexpr uint32_t kBigSlabs = kBigOutMax / kBigSlabBytes;
constexpr uint32_t kBigSlabRecs = kBigSlabBytes / 4u;                      // records per slab the slab decoder's tables are sized for (+ 64 of slack; an LZ4 sequence with a match covers four bytes): a slab with more keeps its chunk off this path

struct BigMeta {                 // one per listed chunk, written by the parse kernel
    uint32_t chunk;              // index of the chunk in the batch
    uint32_t nseq;               // records in all regions (0: the chunk went to the wavefront kernel)
    uint32_t in_skip;            // bytes in front of the element stream (size prefix / length preamble)
    uint32_t U;                  // decoded size
    uint32_t first[kBigLanes];   // index (in the chunk) of region j's first record; a lane without records has the next lane's value
    uint32_t opb[kBigLanes];     // bits 0..27: output position of region j's first record; bits 28..29: the region slot that holds it
    uint32_t slab_first[kBigSlabs];   // the record that holds (or is the first behind) output byte kBigSlabBytes * s
};

// the engine's scratch for a batch that may hold big chunks (CJ_FLAG_BIG_CHUNKS): a list of at most `cap` chunks gets record areas
size_t big_recs_bytes(size_t cap);
size_t big_meta_bytes(size_t cap);
// list[0] = number of listed chunks (counted on the device), list[4 + i] = chunk index, one slot per chunk of the batch.  The engine
// walks the list in GROUPS of `cap` chunks (as many as it has record areas for): entries [base, base + cap).  The small-chunk
// pipeline has flagged every chunk above 64 KiB kRouteWave in `meta`; a chunk this stage takes gets meta = {0, 0} and its result.
size_t big_walk_scratch_bytes(size_t cap);
void launch_big_list(const BatchArgs& a, int codec, uint32_t* list, hipStream_t s);
void launch_big_parse(const BatchArgs& a, int codec, const uint32_t* list, uint32_t base, uint32_t cap, void* recs, void* bigmeta, void* meta, void* scratch, hipStream_t s);
// the slab work items of the listed chunks in slab-major order (item w = slab w / cap of listed chunk w % cap): descriptor rows
// in_off | in_len | out_off | out_cap | result (8 bytes x items each, in that order from `rows`), their ParseMeta, zeroed flags
constexpr size_t kBigItemRows = 5;
void launch_big_items(const BatchArgs& a, const uint32_t* list, uint32_t base, const void* bigmeta, const void* recs, uint32_t cap, uint64_t* rows, void* item_meta, uint32_t* done, hipStream_t s);
void launch_lz4_decode_big_slabs(const BatchArgs& items, const void* meta, const void* recs, const void* bigmeta, uint32_t cap, void* tabs, uint32_t* counter,
                                 uint32_t* done, void* cross, uint32_t tab_stride, uint32_t cross_stride, uint32_t grid, hipStream_t s, int codec);

}  // namespace cj
