/*
 * cramjam_hip.h — the C-ABI of libcramjam_hip.so: an MI355X (gfx950) batched block-codec engine for
 * cramjam's LZ4-block and Snappy-raw hot path.
 *
 * This is the drop-in boundary.  The reference (milesgranger/cramjam) has no plugin registry; its
 * hot path sits behind the crate-call boundary between src/{lz4,snappy}.rs and libcramjam, i.e.
 * plain functions over borrowed byte slices.  Each export below names the reference call site it
 * replaces (paths relative to the reference repository).  A Rust/pyo3 host binds them with
 * `extern "C"` exactly as INTEGRATION.md shows; the Python host in cramjam_amd/ binds the same
 * symbols.  No torch / HIP types appear in any signature: pointers, sizes and ints only.
 *
 * Conventions
 *   - return int64_t >= 0: bytes written / decoded;  < 0: one of CJ_E_* (cj_strerror gives the
 *     message the reference's Rust error would have carried).
 *   - inputs are borrowed for the duration of the call, never retained or freed; outputs are
 *     written into caller memory (the host layer allocates with the bound/len helpers and truncates).
 *   - every export is thread-safe and re-entrant (the reference calls with the GIL released,
 *     src/lz4.rs:84,126,163,205; src/snappy.rs:57,75,97,106).
 *   - all codec arithmetic (and the Snappy framing CRC-32C) runs in HIP kernels on the GPU.  There is NO CPU
 *     fallback: without a usable HIP device every compute entry point returns CJ_E_NO_DEVICE.  The one
 *     piece of checksum arithmetic on the host is the LZ4 frame format's XXH32 (a serial recurrence per frame;
 *     it runs on a thread concurrently with the device batch — see DESIGN.md §5.5).
 */
#ifndef CRAMJAM_HIP_H
#define CRAMJAM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared here are exported */
#define CJ_API __attribute__((visibility("default")))

#define CJ_ABI_VERSION 1

/* ---- error codes ---- */
#define CJ_E_INPUT_TOO_LARGE   (-1)  /* lz4 crate: "Compression input too long." */
#define CJ_E_COMPRESS_FAILED   (-2)  /* lz4 crate: "Compression failed" (output buffer too small) */
#define CJ_E_NO_PREFIX         (-3)  /* lz4 crate: "Source buffer must at least contain size prefix." */
#define CJ_E_NEG_PREFIX        (-4)  /* lz4 crate: "Parsed size prefix in buffer must not be negative." */
#define CJ_E_PREFIX_TOO_BIG    (-5)  /* lz4 crate: "Given size parameter is too big" */
#define CJ_E_OUT_TOO_SMALL     (-6)  /* lz4 crate: "buffer isn't large enough to hold decompressed data" */
#define CJ_E_CORRUPT           (-7)  /* lz4 crate: "Decompression failed. Input invalid or too long?" */
#define CJ_E_SNAPPY_EMPTY      (-8)  /* snap::Error::Empty */
#define CJ_E_SNAPPY_HEADER     (-9)  /* snap::Error::Header */
#define CJ_E_SNAPPY_TOO_BIG    (-10) /* snap::Error::TooBig */
#define CJ_E_SNAPPY_BUF_SMALL  (-11) /* snap::Error::BufferTooSmall */
#define CJ_E_SNAPPY_CORRUPT    (-12) /* snap::Error::{Literal,CopyRead,CopyWrite,Offset,HeaderMismatch} */
#define CJ_E_FRAME_EOF         (-13) /* io::ErrorKind::UnexpectedEof "failed to fill whole buffer" (truncated framed stream) */
#define CJ_E_FRAME_WRITE       (-14) /* io::ErrorKind::WriteZero "failed to write whole buffer" (framed output does not fit) */
#define CJ_E_SNAPPY_STREAM_HEADER (-15) /* snap::Error::{StreamHeader,StreamHeaderMismatch} */
#define CJ_E_SNAPPY_CHUNK_TYPE (-16) /* snap::Error::UnsupportedChunkType */
#define CJ_E_SNAPPY_CHUNK_LEN  (-17) /* snap::Error::UnsupportedChunkLength */
#define CJ_E_SNAPPY_CHECKSUM   (-18) /* snap::Error::Checksum */
#define CJ_E_LZ4F_FRAME_TYPE    (-20) /* LZ4F ERROR_frameType_unknown (bad magic number) */
#define CJ_E_LZ4F_HEADER        (-21) /* LZ4F ERROR_headerVersion_wrong / reservedFlag_set / headerChecksum_invalid */
#define CJ_E_LZ4F_BLOCK_SIZE    (-22) /* LZ4F ERROR_maxBlockSize_invalid */
#define CJ_E_LZ4F_BLOCK_CHECKSUM (-23) /* LZ4F ERROR_blockChecksum_invalid */
#define CJ_E_LZ4F_CONTENT_CHECKSUM (-24) /* LZ4F ERROR_contentChecksum_invalid */
#define CJ_E_LZ4F_CONTENT_SIZE  (-25) /* LZ4F ERROR_frameSize_wrong */
#define CJ_E_LZ4F_INCOMPLETE    (-26) /* lz4 crate Decoder::finish: "Finish runned before read end of compressed stream" */
#define CJ_E_LZ4F_DECOMPRESS    (-27) /* LZ4F ERROR_decompressionFailed (malformed block) */
#define CJ_E_NO_DEVICE         (-100) /* no HIP device / HIP runtime failure (see cj_last_hip_error) */
#define CJ_E_BAD_ARG           (-101)
#define CJ_E_OOM               (-102) /* device or pinned-host allocation failed */

CJ_API const char* cj_strerror(int64_t code);
/* text of the last HIP runtime error seen by the calling thread ("" if none) */
CJ_API const char* cj_last_hip_error(void);
CJ_API int cj_abi_version(void);
/* number of visible HIP devices (0 if none / runtime unusable) */
CJ_API int cj_device_count(void);

/* =====================================================================================
 * Single-buffer entry points — exactly what a pyo3/Rust host would bind in place of the
 * libcramjam calls.  Host pointers.  Run on the default engine of device 0
 * (CJ_DEVICE env var overrides), created lazily.
 * ===================================================================================== */

/* Single-buffer entry points.  A buffer above 64 KiB (input, or announced output) is not run as one serial stream on one
 * wavefront: compress cuts it into 64 KiB pieces, compresses them as a batch and joins them into ONE valid block / raw
 * stream; decompress parses the stream in parallel and decodes it in 64 KiB slabs of output.  Same
 * arguments, results and error codes either way. */

/* src/lz4.rs:228  libcramjam::lz4::block::compress_bound(len, Some(prepend))
 * = LZ4_compressBound(len) (+4 when prepend); 0 when len > 0x7E000000. Pure arithmetic, no device. */
CJ_API size_t cj_lz4_block_compress_bound(size_t len, int prepend);

/* src/lz4.rs:127,206  libcramjam::lz4::block::compress_into(in, out, level, accel, prepend)
 * level/accel: -1 = None.  The reference forwards them but libcramjam always runs the DEFAULT
 * mode, so they do not change the output; accepted and ignored here too.  prepend: -1 = None -> 1. */
CJ_API int64_t cj_lz4_block_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                              int level, int accel, int prepend);

/* src/lz4.rs:88,164,168  libcramjam::lz4::block::decompress_into(in, out, Some(size_prepended))
 * size_prepended=1: u32-LE length prefix expected, decode capacity = that length;
 * size_prepended=0: raw block, decode capacity = cap.  Returns decoded byte count. */
CJ_API int64_t cj_lz4_block_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int size_prepended);

/* src/lz4.rs:90  libcramjam::lz4::block::decompress_vec reads this before allocating:
 * the u32-LE prefix, or CJ_E_NO_PREFIX when n < 4. Pure arithmetic, no device. */
CJ_API int64_t cj_lz4_block_prefixed_len(const uint8_t* in, size_t n);

/* src/snappy.rs:114  snap::raw::max_compress_len(len) = 32 + len + len/6 (0 = too big). No device. */
CJ_API size_t cj_snappy_raw_max_compress_len(size_t len);
/* src/snappy.rs:121  snap::raw::decompress_len(in): varint preamble; 0 for empty input. No device. */
CJ_API int64_t cj_snappy_raw_decompress_len(const uint8_t* in, size_t n);
/* src/snappy.rs:75,97  libcramjam::snappy::raw::compress(in, out); needs cap >= max_compress_len(n) */
CJ_API int64_t cj_snappy_raw_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
/* src/snappy.rs:57,106 libcramjam::snappy::raw::decompress(in, out); needs cap >= decompress_len(in) */
CJ_API int64_t cj_snappy_raw_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);

/* ---- Snappy FRAMING format: a stream of independent <= 64 KiB pieces, each with a masked
 * CRC-32C; pieces are de/compressed and checksummed on the GPU as one batch. ---- */
/* upper bound of cj_snappy_frame_compress's output: 10 + 8 * ceil(n / 65536) + n (0 for n == 0). No device. */
CJ_API size_t cj_snappy_frame_max_compress_len(size_t n);
/* src/snappy.rs:38,82  libcramjam::snappy::compress (snap read::FrameEncoder): stream identifier + one chunk per
 * 65536 input bytes, stored uncompressed when it does not shrink by 1/8; empty input -> empty output. */
CJ_API int64_t cj_snappy_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
/* decoded length of a framed stream from its chunk headers alone (what a caller allocates before
 * cj_snappy_frame_decompress), or the first header-level error. No device. */
CJ_API int64_t cj_snappy_frame_decompress_len(const uint8_t* in, size_t n);
/* src/snappy.rs:24,88  libcramjam::snappy::decompress (snap read::FrameDecoder): errors are reported in stream
 * order like the sequential decoder would (block error, checksum, output full, then header errors).
 * out == NULL: validate only — decode and checksum on the device, return the decoded length or the first error. */
CJ_API int64_t cj_snappy_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);

/* ---- LZ4 FRAME format: blocks de/compressed on the GPU — independent-block and single-block
 * frames as one batch, linked-block frames by a chain kernel; XXH32 frame checksums on a concurrent host thread. ---- */
/* upper bound of cj_lz4_frame_compress's output: 15 + 4 * ceil(n / 65536) + n. No device. */
CJ_API size_t cj_lz4_frame_compress_bound(size_t n);
/* src/lz4.rs:43,56  libcramjam::lz4::compress(input, output, level) (lz4 crate EncoderBuilder -> LZ4F): 64 KiB blocks,
 * content checksum, no content size — like the reference — but INDEPENDENT blocks (the reference links them) and one
 * matcher for every `level` (the reference's default level 4 is LZ4HC): any LZ4F decoder reads the result. */
CJ_API int64_t cj_lz4_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int level);
/* only the block sequence of such a frame (u32 size word + data per 64 KiB of input; no header, EndMark or checksum):
 * what a streaming encoder (reference src/lz4.rs:231-292 `Compressor`) emits per flush.  cap >= n + 4 * ceil(n / 65536). */
CJ_API int64_t cj_lz4_frame_compress_blocks(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
/* upper bound of the decoded size from the headers alone (content size if stored, else blocks x max block size), or the
 * first header-level error. No device. */
CJ_API int64_t cj_lz4_frame_decompress_bound(const uint8_t* in, size_t n);
/* src/lz4.rs:28,63  libcramjam::lz4::decompress (lz4 crate Decoder -> LZ4F_decompress): all block sizes, linked and
 * independent blocks, block / content checksums, content size; stops after the first frame like the crate's Decoder.
 * out == NULL: validate only (block structure and block decode; the content checksum needs the bytes on the host). */
CJ_API int64_t cj_lz4_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);

/* =====================================================================================
 * Batch extension (no reference equivalent: the reference API is one buffer per call; a GPU only
 * pays off on batches of independent chunks).  One engine per GPU; chunks of a batch are
 * independent, so multi-GPU use is host-side round-robin sharding over engines, no collective.
 * ===================================================================================== */
typedef struct cj_engine cj_engine;

typedef enum { CJ_CODEC_LZ4_BLOCK = 0, CJ_CODEC_SNAPPY_RAW = 1 } cj_codec;
typedef enum { CJ_OP_DECOMPRESS = 0, CJ_OP_COMPRESS = 1 } cj_op;

/* flags */
#define CJ_FLAG_LZ4_SIZE_PREFIX 1u   /* lz4: blocks carry / get the u32-LE length prefix (store_size) */
/* decode mapping overrides (tests and comparisons; results are identical): one wavefront per chunk, one lane per chunk, or
 * the workgroup decoder for every chunk it can take.  Default: the workgroup decoder, the wave kernel for what it leaves over */
#define CJ_FLAG_FORCE_WAVE_PER_CHUNK 0x100u
#define CJ_FLAG_FORCE_LANE_PER_CHUNK 0x200u
#define CJ_FLAG_FORCE_LDS_PER_CHUNK  0x400u
/* ... and, for the workgroup decoder, where its parse stage runs: inside the decoder kernel / as the lane-per-chunk kernel in front of
 * it, at any batch size (default: by CJ_FUSED_MAX_CHUNKS below).  Tests exercise both sides of the threshold with them. */
#define CJ_FLAG_FORCE_FUSED_PARSE   0x10u
#define CJ_FLAG_FORCE_PARSE_KERNEL  0x20u
/* decompress, a promise about the batch: every chunk's output capacity (LZ4) / announced length (Snappy) is at most 32 KiB / 16 KiB.
 * The workgroup decoder then runs on windows of that size — four workgroups of four wavefronts / eight of two per CU instead of two of
 * eight: more chunks' dependency chains in flight for the same wavefronts (32 KiB chunks 635 -> 850 GB/s, 16 KiB 430 -> 810; batches of
 * up to CJ_FUSED_MAX_CHUNKS chunks: the one-kernel path on the same windows, 4 096 x 32 / 16 KiB 233 / 153 -> 401 / 309).  A chunk
 * that breaks the promise is still decoded correctly (one wavefront).  cj_batch_host sets them itself. */
#define CJ_FLAG_CHUNKS_LE_32K       0x40u
#define CJ_FLAG_CHUNKS_LE_16K       0x80u
/* decompress: the batch may hold chunks of 64 KiB .. 256 KiB (capacity / announced length) and they matter — the engine lists them on
 * the device, parses them with 32 lanes each into record areas (1 MiB per listed chunk, groups of up to 8 192) and decodes them slab by
 * slab with workgroups.  How many there are is known on the device only: every flagged call copies its count back
 * WITHOUT waiting for it, and a call reserves for the largest of the last eight counts that have arrived; what a batch holds beyond
 * that — and every such chunk without the flag — takes one wavefront: correct, ~2.5x slower in bulk.  Only the first flagged call
 * on an engine waits for the stream once (it reads its own count); like every call, one that has to GROW the engine's scratch waits
 * for the device while it reallocates.  After that cj_batch_device with this flag only enqueues. */
#define CJ_FLAG_BIG_CHUNKS          0x800u
/* debug aid: the workgroup decoder accumulates per-phase cycle counters (read with cj_debug_lds_phase_cycles, cramjam_hip_debug.h; results unchanged) */
#define CJ_FLAG_DEBUG_PROFILE        0x1000u
/* decode batches up to this many chunks run parse + decode as ONE kernel (the segmented parse inside the workgroup decoder:
 * 1 chunk 0.16 ms instead of 0.25, 8 192 chunks 348 instead of 178 GB/s); above, a lane-per-chunk parse kernel in front of the
 * decoder is cheaper per chunk.  The crossover depends on the data — the parse kernel's fixed latency is one lane walking one
 * chunk, i.e. its sequence count: benchmark data (2.7 k sequences per 64 KiB) ~13 000 chunks for LZ4 and ~10 500 for Snappy,
 * the reference's corpus (~8 k) above 24 000 (profiles/r04/experiments, c02); 16 384 loses at most ~11 % on either side */
#define CJ_FUSED_MAX_CHUNKS 16384
/* decode batches larger than this are submitted in slices of this many chunks */
#define CJ_SLICE_CHUNKS_DEFAULT 131072

CJ_API int  cj_engine_create(int device, cj_engine** out);
CJ_API void cj_engine_destroy(cj_engine* e);
CJ_API int  cj_engine_device(const cj_engine* e);

/* Device-resident batch: every pointer is a DEVICE pointer on the engine's GPU.
 * chunk i reads  in_base + in_off[i] .. + in_len[i]   and writes  out_base + out_off[i] .. + out_cap[i];
 * result[i] = bytes produced (>= 0) or CJ_E_* (< 0); one bad chunk never affects another.
 * hip_stream: a hipStream_t (NULL = the engine's own stream).  Asynchronous: returns after enqueue;
 * call cj_engine_sync (or synchronise the stream yourself) before reading results.
 * Input addressing: chunks may start at any byte alignment, but the kernels fetch the stream with aligned vector loads —
 * the 16-byte granules that hold a chunk's first and last byte are read whole (up to 15 bytes before in_base + in_off[i]
 * and up to 15 bytes past its end; never used, never written).  Both granules must therefore lie inside the device
 * allocation: true for every chunk of a buffer that comes from hipMalloc / a caching allocator (allocations start on
 * 256-byte boundaries and are padded to their granule), NOT for a chunk that begins or ends flush with a page the
 * caller carved up itself.  cj_batch_host pads to 16 bytes on its own staging buffers. */
CJ_API int cj_batch_device(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n_chunks,
                    const uint8_t* in_base, const uint64_t* in_off, const uint64_t* in_len,
                    uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                    int64_t* result, void* hip_stream);
CJ_API int cj_engine_sync(cj_engine* e);
/* the same for a caller's stream (a hipStream_t; NULL = the engine's own): what a binding without its own HIP runtime handle needs
 * to wait for a batch it submitted on a stream it was handed (cramjam_amd.batch.*_device(stream=...)) */
CJ_API int cj_stream_sync(cj_engine* e, void* hip_stream);

/* Host batch: host pointers; the engine packs inputs into pinned staging, copies H2D, runs the
 * kernels, copies D2H and scatters — a batch of 128 MiB and more in slices of ~64 MiB, so that those
 * steps overlap (same results as one pass).  Synchronous. result[i] as above. */
CJ_API int cj_batch_host(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n_chunks,
                  const uint8_t* const* in_ptrs, const size_t* in_lens,
                  uint8_t* const* out_ptrs, const size_t* out_caps, int64_t* result);

/* Timing aid for benchmarks: runs the same device batch `reps` times on the engine stream between
 * two hipEvents and returns the mean kernel time per rep in milliseconds (< 0 on error). */
CJ_API double cj_batch_device_timed(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n_chunks,
                             const uint8_t* in_base, const uint64_t* in_off, const uint64_t* in_len,
                             uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                             int64_t* result, int reps);

/* Thin device-memory helpers so C / ctypes callers need no HIP binding of their own. */
CJ_API void* cj_device_alloc(cj_engine* e, size_t bytes);
CJ_API void  cj_device_free(cj_engine* e, void* p);
CJ_API int   cj_memcpy_h2d(cj_engine* e, void* dst_dev, const void* src_host, size_t bytes);
CJ_API int   cj_memcpy_d2h(cj_engine* e, void* dst_host, const void* src_dev, size_t bytes);
CJ_API int   cj_memcpy_d2d(cj_engine* e, void* dst_dev, const void* src_dev, size_t bytes);
CJ_API int   cj_memset_dev(cj_engine* e, void* dst_dev, int value, size_t bytes);

/* The test / benchmark utilities and debug counters the library also exports (cj_bench_*, cj_debug_*) are declared in
 * cramjam_hip_debug.h: they are NOT part of the drop-in ABI. */

#ifdef __cplusplus
}
#endif
#endif
