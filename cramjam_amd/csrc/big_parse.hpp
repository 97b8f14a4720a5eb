// big_parse.hpp — argument blocks of the large-stream parse (big_parse.hip) shared with its host side (large.hip).
#pragma once
#include "cj_common.hpp"

namespace cj {

// input bytes per parse piece (one wavefront walks it as 64 sub-segments): small streams take small pieces — one call on a few hundred KB
// is a chain of latency-bound kernels, and a lane's walk is as long as its sub-segment; large ones 64 KiB (fewer serial steps in K2)
#ifndef CJ_BIG_PIECE_TINY
#define CJ_BIG_PIECE_TINY 4096
#endif
constexpr uint32_t kBigPieceTiny = CJ_BIG_PIECE_TINY, kBigPieceSmall = 16384, kBigPieceLarge = 65536;
constexpr uint32_t kBigPieceTinyMax = 1u << 20, kBigPieceSwitch = 4u << 20;       // stream length: below -> tiny, below -> small, else large
inline uint32_t big_piece_for(uint32_t stream_len) { return stream_len < kBigPieceTinyMax ? kBigPieceTiny : stream_len < kBigPieceSwitch ? kBigPieceSmall : kBigPieceLarge; }

struct BigParse {
    const uint8_t* in;       // stream position 0 (device; padded by >= 16 readable bytes)
    uint32_t iend;           // stream length
    uint32_t start;          // position of the first element (Snappy: after the length header)
    uint32_t piece;          // bytes per piece (a multiple of 2048)
    uint32_t np;             // pieces covering [start, iend)
    uint64_t cap;            // LZ4: output capacity; Snappy: the decoded length the header announces
    uint32_t* bits;          // np * piece / 32 words: positions visited by the lanes' own walks
    uint32_t* merge;         // np * 64
    uint32_t* exitp;         // np * 64
    uint32_t* next_tab;      // np * 64: entry of the next piece for a true entry at byte j of this piece
    uint32_t* fe_tab;        // np * 64: end of the entry lane's part of the chain for that entry
    uint2* entry;            // np
    uint32_t* lane_idx;      // np * 64
    uint64_t* lane_op;       // np * 64
    uint64_t* totals;        // np * 2 (count, bytes) -> exclusive prefix after the scan
    uint2* sync;             // absolute (ip, op) of every 8th sequence; iend / 16 + 2 entries cover any stream
    const BigParse* jobs;    // several streams in one run of launches (launch_big_parse_many): the kernels' argument is a header —
    const uint2* piece_map;  // block b works on piece piece_map[b].y of jobs[piece_map[b].x]; nullptr = this struct is the one stream
    uint32_t* status;        // 16 words, zeroed: [0] chain did not end at the input's end, [1] violation, [2,3] sequences,
                             // [4,5] output bytes counted, [6,7] decoded size, [8] last sequence seen
};

struct BigSlabs {
    const uint2* sync;
    uint32_t n_sync;
    uint64_t n_seq, total;   // sequences, decoded bytes
    uint32_t iend;
    uint64_t in_base_off;    // offset of stream position 0 from the batch's in_base
    uint64_t out_base_off;   // offset of the stream's first output byte from the batch's out_base
    uint32_t sync_index_base;// index of sync[0] in the array the decoder is given (several streams share one launch)
    uint32_t n_slabs;
    uint64_t* in_off; uint64_t* in_len; uint64_t* out_off; uint64_t* out_cap; int64_t* result;
    uint2* meta;             // ParseMeta {records, 0}
    uint2* first;            // {first sync index, stream position of that sync point}
    uint32_t* max_rec;       // atomicMax of the slabs' record counts
};

void launch_big_parse(const BigParse& a, int codec, hipStream_t s);
// jobs / piece_map: device arrays (n_jobs structs; n_pieces entries); max_piece = the largest piece size among the jobs
void launch_big_parse_many(const BigParse* jobs, uint32_t n_jobs, const uint2* piece_map, uint32_t n_pieces, uint32_t max_piece, int codec, hipStream_t s);
void launch_big_slabs(const BigSlabs& d, hipStream_t s);
// jobs: device array; slab_job[i] = {job, slab index in that job} for every slab of the chunk list; the jobs' array pointers are
// already offset to their first slab
void launch_big_slabs_many(const BigSlabs* jobs, const uint2* slab_job, uint32_t n_slabs, hipStream_t s);

}  // namespace cj
