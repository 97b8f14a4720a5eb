// frame_kernels.hip — device side of the framed formats (frame.hip): per-piece CRC-32C and the segment mover
// that assembles / unpacks a framed stream in HBM.  One wavefront per piece or segment, 4 per block.
#include "cj_common.hpp"
#include "crc32c_lanes.hpp"

namespace cj {

namespace {
__device__ const Crc32cTables d_crc_tables = make_crc32c_tables();
}

// out[i] = masked CRC-32C of base[off[i] .. off[i] + len[i])      (framing_format.txt §3; len < 2^32)
__global__ __launch_bounds__(kBlockThreads) void crc32c_pieces_kernel(const uint8_t* base, const uint64_t* off,
                                                                      const uint64_t* len, uint32_t* out, uint32_t n) {
    __shared__ uint32_t adv[1024];
    for (uint32_t i = threadIdx.x; i < 1024u; i += kBlockThreads) adv[i] = (&d_crc_tables.adv256[0][0])[i];
    __syncthreads();
    const uint32_t piece = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (piece >= n) return;
    uint32_t c = crc32c_lane(base + off[piece], (uint32_t)len[piece], lane_id(), adv, d_crc_tables.xpow8,
                             [](const uint8_t* p) { return ld32u(p); });
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) c ^= __shfl_xor(c, sh, 64);
    if (lane_id() == 0) out[piece] = crc32c_mask(~c);
}

// segment i: len[i] bytes from the device address src[i] to dst_base + dst_off[i]; when hdr is non-null the low
// hdr_len (<= 8) bytes of hdr[i], little endian, are written just before the segment (Snappy: chunk type, u24 length,
// u32 checksum; LZ4 frame: u32 block size word).
__global__ __launch_bounds__(kBlockThreads) void copy_segments_kernel(const uint64_t* src, uint8_t* dst_base,
                                                                      const uint64_t* dst_off, const uint64_t* len,
                                                                      const uint64_t* hdr, uint32_t hdr_len, uint32_t n) {
    const uint32_t seg = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (seg >= n) return;
    uint8_t* dst = dst_base + dst_off[seg];
    if (hdr != nullptr) {
        const uint64_t h = hdr[seg];
        if (lane_id() < hdr_len) dst[(int)lane_id() - (int)hdr_len] = (uint8_t)(h >> (8u * lane_id()));
    }
    wave_copy(dst, reinterpret_cast<const uint8_t*>(src[seg]), (uint32_t)len[seg]);
}

void launch_crc32c_pieces(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t* out, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(crc32c_pieces_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlockThreads), 0, s, base, off, len, out, n);
}

void launch_copy_segments(const uint64_t* src, uint8_t* dst_base, const uint64_t* dst_off, const uint64_t* len,
                          const uint64_t* hdr, uint32_t hdr_len, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(copy_segments_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlockThreads), 0, s, src, dst_base, dst_off, len, hdr, hdr_len, n);
}

}  // namespace cj
