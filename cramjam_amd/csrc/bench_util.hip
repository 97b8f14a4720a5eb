// bench_util.hip — benchmark/test utilities that live next to the engine but are NOT codec code:
//   * synth-v1 generator on the device, one thread per chunk, so benchmarks can build
//     multi-GB device-resident workloads without host RAM or the CPU oracle;
//   * chunk replicate + compare helpers for device-resident verification.
#include "cj_common.hpp"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_v1_kernel(uint8_t* out, uint64_t stride, uint64_t S, uint64_t first_index, uint64_t n, uint64_t seed) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    uint8_t* dst = out + c * stride;
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + (first_index + c);
    uint64_t pos = 0;
    while (pos < S) {
        uint64_t r = splitmix64(st);
        uint64_t lit = 1 + r % 24;
        uint64_t bits = 0; int have = 0;
        for (uint64_t i = 0; i < lit && pos < S; i++) {
            if (have == 0) { bits = splitmix64(st); have = 10; }
            dst[pos++] = (uint8_t)(0x20 + (bits & 63));
            bits >>= 6; have--;
        }
        if (pos >= 8 && pos < S) {
            uint64_t r2 = splitmix64(st);
            uint64_t r3 = splitmix64(st);
            uint64_t mlen = 4 + r2 % 29;
            uint64_t lim = pos < 65535 ? pos : 65535;
            uint64_t dist = 1 + r3 % lim;
            for (uint64_t i = 0; i < mlen && pos < S; i++, pos++) dst[pos] = dst[pos - dist];
        }
    }
}

// count chunks whose decoded bytes differ from the expected unique chunk (16 B per lane compare)
__global__ void compare_kernel(const uint8_t* got, const uint64_t* got_off, const uint8_t* want, uint64_t want_stride,
                               uint32_t n_unique, uint64_t S, uint32_t n, unsigned long long* mismatches) {
    uint32_t c = blockIdx.x;
    if (c >= n) return;
    const uint8_t* g = got + got_off[c];
    const uint8_t* w = want + (uint64_t)(c % n_unique) * want_stride;
    bool bad = false;
    for (uint64_t i = threadIdx.x; i < S; i += blockDim.x) bad |= g[i] != w[i];
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicAdd(mismatches, 1ull);
}

}  // namespace

extern "C" {

// fills n chunks of S bytes at out + i*stride with synth-v1(S, first_index + i, seed); device pointer
int cj_bench_synth_v1(void* d_out, uint64_t stride, uint64_t S, uint64_t first_index, uint64_t n, uint64_t seed, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(synth_v1_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       (uint8_t*)d_out, stride, S, first_index, n, seed);
    return hipGetLastError() == hipSuccess ? 0 : CJ_E_NO_DEVICE;
}

// *d_mismatches += number of chunks i in [0,n) with got[got_off[i] .. +S) != want[(i % n_unique)*want_stride .. +S)
int cj_bench_compare(const void* d_got, const uint64_t* d_got_off, const void* d_want, uint64_t want_stride,
                     uint32_t n_unique, uint64_t S, uint32_t n, void* d_mismatches, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(compare_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_got, d_got_off,
                       (const uint8_t*)d_want, want_stride, n_unique, S, n, (unsigned long long*)d_mismatches);
    return hipGetLastError() == hipSuccess ? 0 : CJ_E_NO_DEVICE;
}

}  // extern "C"
