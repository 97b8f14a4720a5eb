// probe: LDS *pipe* cycles per wave-instruction with k active lanes, 16 wavefronts of one CU issuing back to back
// (the decoder's resolver is bound by exactly this).  Scattered addresses in a 64 KiB region, aligned vs byte-misaligned (+3;
// the 16-byte operations at +8, the 2-byte store at +2: a 16-byte DS access that is not 8-byte aligned faults).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_throughput_probe.hip -o /tmp/lds_tp && /tmp/lds_tp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPS8(STR) asm volatile(STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n s_waitcnt lgkmcnt(0)" 

template <int OP>
__device__ __forceinline__ void run(uint32_t a, int iters) {
    uint32_t v0 = a; uint64_t v1 = a; 
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v2 = {a, a, a, a};
    for (int i = 0; i < iters; i++) {
        if constexpr (OP == 0) OPS8("ds_read_b32 %0, %1") : "=v"(v0) : "v"(a) : "memory");
        if constexpr (OP == 1) OPS8("ds_read2_b32 %0, %1 offset1:1") : "=v"(v1) : "v"(a) : "memory");
        if constexpr (OP == 2) OPS8("ds_read_b64 %0, %1") : "=v"(v1) : "v"(a) : "memory");
        if constexpr (OP == 3) OPS8("ds_read_b128 %0, %1") : "=v"(v2) : "v"(a) : "memory");
        if constexpr (OP == 4) OPS8("ds_write_b32 %0, %1") :: "v"(a), "v"(v0) : "memory");
        if constexpr (OP == 5) OPS8("ds_write_b64 %0, %1") :: "v"(a), "v"(v1) : "memory");
        if constexpr (OP == 6) OPS8("ds_write_b128 %0, %1") :: "v"(a), "v"(v2) : "memory");
        if constexpr (OP == 7) OPS8("ds_write_b8 %0, %1") :: "v"(a), "v"(v0) : "memory");
        if constexpr (OP == 8) OPS8("ds_or_b32 %0, %1") :: "v"(a), "v"(v0) : "memory");
        if constexpr (OP == 9) OPS8("ds_write_b16 %0, %1") :: "v"(a), "v"(v0) : "memory");
    }
    if (v0 == 0x12345u && (uint32_t)v1 == 7u && v2.x == 9u) asm volatile("s_nop 0");
}

__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 1024) ((uint32_t*)s)[i] = i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t h = (lane * 2654435761u) ^ (wave * 40503u + 977u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const uint32_t base = (uint32_t)(uintptr_t)s + ((h % 4000u) * 16u);          // 16 B aligned, scattered over 64 KiB
    int slot = 0;
    for (int k = 1; k <= 64; k *= 2) {
        for (int mis = 0; mis < 3; mis++) {
            const bool act = (int)lane < k;
            const uint32_t a = base + (mis == 1 ? 3u : mis == 2 ? 4u : 0u);
#define RUN(OP) { const uint32_t aa = (mis && (OP == 3 || OP == 6)) ? base + 8u : (mis && OP == 9) ? base + 2u : (OP == 1 || OP == 8) ? base + (mis == 2 ? 4u : 0u) : a; __syncthreads(); unsigned long long t0 = __builtin_readcyclecounter(); if (act) run<OP>(aa, iters); __syncthreads(); unsigned long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[slot * 16 + OP] = t1 - t0; }
            RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
            slot++;
        }
    }
}

int main() {
    unsigned long long* c; hipMalloc(&c, 32 * 16 * 8); hipMemset(c, 0, 32 * 16 * 8);
    const int iters = 64;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 65536 + 64, 0, c, iters);
    unsigned long long h[32 * 16];
    hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    const char* names[10] = {"rd_b32", "rd2_b32", "rd_b64", "rd_b128", "wr_b32", "wr_b64", "wr_b128", "wr_b8", "or_b32", "wr_b16"};
    printf("LDS pipe cycles per wave-instruction (16 waves x %d x 8 ops each, one CU)\n%-22s", iters, "");
    for (int o = 0; o < 10; o++) printf("%9s", names[o]);
    printf("\n");
    int slot = 0;
    for (int k = 1; k <= 64; k *= 2) for (int mis = 0; mis < 3; mis++, slot++) {
        printf("lanes %2d %-13s", k, mis == 1 ? "misaligned+3" : mis == 2 ? "dword+4" : "aligned");
        for (int o = 0; o < 10; o++) printf("%9.1f", (double)h[slot * 16 + o] / (16.0 * iters * 8));
        printf("\n");
    }
    return 0;
}
