/*
 * cj_oracle.h — CPU ORACLE for the cramjam LZ4-block / Snappy-raw hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call it.  The product path
 * (cramjam_amd/, libcramjam_hip.so) never includes, links or dlopens anything here.
 *
 * What it restates (plain C, scalar, single thread):
 *   - the wrapper semantics of /root/reference/src/lz4.rs:78-229 and src/snappy.rs:52-122
 *     (prefix handling, output_len meaning, bound/len helpers, error conditions);
 *   - the codec arithmetic those wrappers forward to.  That arithmetic lives in third-party
 *     crates that are NOT under /root/reference (Cargo.lock:407-473,744-747):
 *       libcramjam 0.8.0 -> lz4 1.28.1 -> lz4-sys 1.11.1+lz4-1.10.0 (C liblz4 1.10.0)
 *       libcramjam 0.8.0 -> snap 1.1.1 (Rust)
 *     so the algorithms are restated from the published formats (lz4_Block_format.md,
 *     snappy format_description.txt) and the published encoder algorithms
 *     (LZ4_compress_default greedy hash-chainless matcher; snappy/snap compress_fragment).
 *
 * Parity pinning (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
 *   - the reference's byte-exact known-answer vectors, tests/test_variants.py:329-334;
 *   - the raw blocks inside the reference's third-party-produced fixtures
 *     tests/data/integration/plaintext.txt.{lz4,snappy} (copied as data under tests/golden/);
 *   - golden vectors minted in the build container from the same C code family the reference
 *     executes (system liblz4 1.9.3, libsnappy 1.1.8, pyarrow lz4_raw/snappy): the encoders
 *     here are BIT-IDENTICAL to LZ4_compress_default / snappy::RawCompress on the whole
 *     benchmark corpus, and the decoders agree on every vector incl. malformed inputs.
 *   The reference's own Rust path cannot be built or imported in this environment (no
 *   cargo/rustc, no cramjam wheel), so there is no oracle/_ref.
 */
#ifndef CJ_ORACLE_H
#define CJ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* error codes (shared numbering with include/cramjam_hip.h) */
#define CJO_E_INPUT_TOO_LARGE   (-1)  /* "Compression input too long." */
#define CJO_E_COMPRESS_FAILED   (-2)  /* "Compression failed" (output too small) */
#define CJO_E_NO_PREFIX         (-3)  /* "Source buffer must at least contain size prefix." */
#define CJO_E_NEG_PREFIX        (-4)  /* "Parsed size prefix in buffer must not be negative." */
#define CJO_E_PREFIX_TOO_BIG    (-5)  /* "Given size parameter is too big" */
#define CJO_E_OUT_TOO_SMALL     (-6)  /* "buffer isn't large enough to hold decompressed data" */
#define CJO_E_CORRUPT           (-7)  /* "Decompression failed. Input invalid or too long?" */
#define CJO_E_SNAPPY_EMPTY      (-8)  /* snap Error::Empty */
#define CJO_E_SNAPPY_HEADER     (-9)  /* snap Error::Header */
#define CJO_E_SNAPPY_TOO_BIG    (-10) /* snap Error::TooBig */
#define CJO_E_SNAPPY_BUF_SMALL  (-11) /* snap Error::BufferTooSmall */
#define CJO_E_SNAPPY_CORRUPT    (-12) /* snap Error::{Literal,CopyRead,CopyWrite,Offset,HeaderMismatch} */
#define CJO_E_FRAME_EOF         (-13) /* io::ErrorKind::UnexpectedEof "failed to fill whole buffer" (truncated frame) */
#define CJO_E_FRAME_WRITE       (-14) /* io::ErrorKind::WriteZero "failed to write whole buffer" (output too small) */
#define CJO_E_SNAPPY_STREAM_HEADER (-15) /* snap Error::{StreamHeader,StreamHeaderMismatch} */
#define CJO_E_SNAPPY_CHUNK_TYPE (-16) /* snap Error::UnsupportedChunkType */
#define CJO_E_SNAPPY_CHUNK_LEN  (-17) /* snap Error::UnsupportedChunkLength */
#define CJO_E_SNAPPY_CHECKSUM   (-18) /* snap Error::Checksum */
#define CJO_E_LZ4F_FRAME_TYPE   (-20) /* LZ4F ERROR_frameType_unknown (bad magic) */
#define CJO_E_LZ4F_HEADER       (-21) /* LZ4F ERROR_headerVersion_wrong / reservedFlag_set / headerChecksum_invalid */
#define CJO_E_LZ4F_BLOCK_SIZE   (-22) /* LZ4F ERROR_maxBlockSize_invalid */
#define CJO_E_LZ4F_BLOCK_CHECKSUM (-23) /* LZ4F ERROR_blockChecksum_invalid */
#define CJO_E_LZ4F_CONTENT_CHECKSUM (-24) /* LZ4F ERROR_contentChecksum_invalid */
#define CJO_E_LZ4F_CONTENT_SIZE (-25) /* LZ4F ERROR_frameSize_wrong */
#define CJO_E_LZ4F_INCOMPLETE   (-26) /* lz4 crate: "Finish runned before read end of compressed stream" */
#define CJO_E_LZ4F_DECOMPRESS   (-27) /* LZ4F ERROR_decompressionFailed (malformed block) */

/* ---- LZ4 block: raw codec (liblz4 semantics) ---- */
/* LZ4_compressBound: n + n/255 + 16, 0 if n > 0x7E000000 */
size_t  cjo_lz4_compress_bound_raw(size_t n);
/* LZ4_compress_default(src, dst, n, cap): bytes written, 0 if it does not fit */
int64_t cjo_lz4_compress_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);
/* LZ4_decompress_safe(src, dst, n, cap): decoded bytes, <0 on malformed input */
int64_t cjo_lz4_decompress_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

/* ---- LZ4 block: libcramjam::lz4::block wrapper semantics (reference src/lz4.rs call sites) ---- */
/* src/lz4.rs:228  compress_bound(len, Some(prepend)) */
size_t  cjo_lz4_block_compress_bound(size_t n, int prepend);
/* src/lz4.rs:127,206  compress_into(in,out,level,accel,prepend) — level/accel do not change output */
int64_t cjo_lz4_block_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int prepend);
/* src/lz4.rs:88,164,168  decompress_into(in,out,Some(size_prepended)) */
int64_t cjo_lz4_block_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int size_prepended);

/* ---- Snappy raw (snap 1.1.1 semantics; reference src/snappy.rs call sites) ---- */
/* src/snappy.rs:114  max_compress_len: 32 + n + n/6 (0 if too big) */
size_t  cjo_snappy_max_compress_len(size_t n);
/* src/snappy.rs:121  decompress_len: varint preamble; empty input -> 0 */
int64_t cjo_snappy_decompress_len(const uint8_t* in, size_t n);
/* src/snappy.rs:75,97  raw::compress(in,out) — needs cap >= max_compress_len(n) */
int64_t cjo_snappy_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
/* src/snappy.rs:57,106 raw::decompress(in,out) */
int64_t cjo_snappy_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);

/* ---- LZ4 block with a contiguous history prefix (linked LZ4-frame blocks) ---- */
int64_t cjo_lz4_decompress_with_prefix(const uint8_t* src, size_t n, uint8_t* buf, size_t hist, size_t cap);
int64_t cjo_lz4_compress_with_prefix(const uint8_t* base, size_t hist, size_t len, uint8_t* dst, size_t cap);

/* ---- LZ4 frame format (lz4 crate Encoder/Decoder -> LZ4F_*; reference src/lz4.rs:28-66) ---- */
uint32_t cjo_xxh32(const uint8_t* p, size_t n, uint32_t seed);
size_t  cjo_lz4_frame_compress_bound(size_t n, int bs_code);
/* bs_code 4..7 = 64 KiB..4 MiB blocks; flags: 1 linked blocks, 2 block checksums, 4 content size, 8 NO content checksum */
int64_t cjo_lz4_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int bs_code, int flags);
int64_t cjo_lz4_frame_decompress_bound(const uint8_t* in, size_t n);
int64_t cjo_lz4_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);

/* ---- Snappy framing format (snap 1.1.1 read::FrameEncoder/FrameDecoder; reference src/snappy.rs:24,38,82,88) ---- */
uint32_t cjo_crc32c(const uint8_t* p, size_t n);
uint32_t cjo_crc32c_masked(const uint8_t* p, size_t n);
size_t  cjo_snappy_frame_max_compress_len(size_t n);
int64_t cjo_snappy_frame_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
/* same with a smaller piece size (tests mint ragged foreign streams with it) */
int64_t cjo_snappy_frame_compress_bs(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t block_size);
/* grammar walk only: decoded length, or the first header-level error */
int64_t cjo_snappy_frame_decompress_len(const uint8_t* in, size_t n);
int64_t cjo_snappy_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap);

/* ---- deterministic synthetic chunk generator (SURVEY.md §8d "synth-v1") ---- */
void cjo_synth_v1(uint8_t* dst, size_t chunk_bytes, uint64_t index, uint64_t seed);

/* ---- cpu baseline helpers: run one op over a batch with T pthreads (bench.py cpu_baseline) ---- */
/* op: 0 lz4 decompress (raw, cap = out_stride), 1 lz4 compress raw, 2 snappy decompress, 3 snappy compress.
 * Chunk i: in = in_base + in_off[i], in_len[i]; out = out_base + i*out_stride. res[i] = return value. */
int cjo_batch_run(int op, int threads, size_t n_chunks, const uint8_t* in_base, const uint64_t* in_off,
                  const uint64_t* in_len, uint8_t* out_base, size_t out_stride, int64_t* res);
/* the same, `reps` passes over the batch with ONE pool (threads created once, barrier between passes).
 * op 4 = the host's liblz4 LZ4_decompress_safe (dlopen; res = -1 when the library is absent, see cjo_have_liblz4);
 * op 6 = its LZ4_compress_default, op 7 = the host's libsnappy snappy_compress (the compress legs of bench.py's cpu_baseline). */
int cjo_batch_run_reps(int op, int threads, int reps, size_t n_chunks, const uint8_t* in_base, const uint64_t* in_off,
                       const uint64_t* in_len, uint8_t* out_base, size_t out_stride, int64_t* res);
int cjo_have_liblz4(void);
int cjo_have_libsnappy(void);      /* op 5 = the host's libsnappy snappy_uncompress (dlopen; res = -1 when absent) */

#ifdef __cplusplus
}
#endif
#endif
