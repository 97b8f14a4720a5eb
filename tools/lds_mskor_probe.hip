// probe for the level-ordered decoder (lz4_decode_lvl.hip):
//  1. ds_mskor_b32 (MEM = (MEM & ~D0) | D1, an atomic byte-masked dword store) — semantics on gfx950, including several lanes of
//     ONE instruction hitting different bytes of the same dword, and lanes of different waves doing so concurrently;
//  2. LDS-pipe cycles per wave-instruction with RANDOM dword addresses (the throughput probe's 16-byte aligned bases use a
//     quarter of the banks): ds_read_b32, ds_read2_b32, ds_read_b64 (8-aligned), ds_write_b32, ds_write_b8, ds_mskor_b32,
//     ds_add_u32, ds_add_rtn_u32 — 8 waves of one workgroup issuing back to back, 64 / 16 / 4 lanes active;
//  3. the price of s_barrier in a 512-thread workgroup (back to back, and behind one dependent LDS round trip per wave).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_mskor_probe.hip -o /tmp/lds_mskor && /tmp/lds_mskor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define OPS8(STR) asm volatile(STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n s_waitcnt lgkmcnt(0)"

template <int OP>
__device__ __forceinline__ void run(uint32_t a, int iters) {
    uint32_t v0 = a; uint64_t v1 = a;
    for (int i = 0; i < iters; i++) {
        if constexpr (OP == 0) OPS8("ds_read_b32 %0, %1") : "=v"(v0) : "v"(a) : "memory");
        if constexpr (OP == 1) OPS8("ds_read2_b32 %0, %1 offset1:1") : "=v"(v1) : "v"(a) : "memory");
        if constexpr (OP == 2) OPS8("ds_read_b64 %0, %1") : "=v"(v1) : "v"(a & ~7u) : "memory");
        if constexpr (OP == 3) OPS8("ds_write_b32 %0, %1") :: "v"(a), "v"(v0) : "memory");
        if constexpr (OP == 4) OPS8("ds_write_b8 %0, %1") :: "v"(a + 1u), "v"(v0) : "memory");
        if constexpr (OP == 5) OPS8("ds_mskor_b32 %0, %1, %2") :: "v"(a), "v"(0xffff00u), "v"(v0 & 0xffff00u) : "memory");
        if constexpr (OP == 6) OPS8("ds_add_u32 %0, %1") :: "v"(a), "v"(1u) : "memory");
        if constexpr (OP == 7) OPS8("ds_add_rtn_u32 %0, %1, %2") : "=v"(v0) : "v"(a), "v"(1u) : "memory");
        if constexpr (OP == 8) OPS8("ds_write_b16 %0, %1") :: "v"(a + 2u), "v"(v0) : "memory");
    }
    if (v0 == 0x12345u && (uint32_t)v1 == 7u) asm volatile("s_nop 0");
}

__global__ __launch_bounds__(512) void probe(unsigned long long* out, uint32_t* sem, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s[];
    uint32_t* w = (uint32_t*)s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // ---- 1. semantics ----
    for (uint32_t i = tid; i < 16384u; i += 512u) w[i] = 0xA0B0C0D0u + i;
    __syncthreads();
    {   // every thread owns byte (tid & 3) of dword tid >> 2: four lanes of one instruction per dword
        const uint32_t a = (uint32_t)(uintptr_t)(w + (tid >> 2)), sh = 8u * (tid & 3u);
        asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(a), "v"(0xffu << sh), "v"((0x40u + (tid & 63u)) << sh) : "memory");
    }
    {   // dwords 1024..: byte k of dword 1024 + lane written by wave k, k + 4 (two waves per byte, the same value) — cross-wave
        const uint32_t a = (uint32_t)(uintptr_t)(w + 1024u + lane), sh = 8u * (wave & 3u);
        asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(a), "v"(0xffu << sh), "v"((0x11u * ((wave & 3u) + 1u)) << sh) : "memory");
    }
    {   // a two-byte mask that leaves the other two bytes alone
        const uint32_t a = (uint32_t)(uintptr_t)(w + 2048u + tid);
        asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(a), "v"(0x00ffff00u), "v"(0x00123400u) : "memory");
    }
    __syncthreads();
    uint32_t bad = 0;
    if (tid < 128u) {
        uint32_t want = 0;
        for (uint32_t k = 0; k < 4u; k++) want |= (0x40u + ((4u * tid + k) & 63u)) << (8u * k);
        bad += w[tid] != want;
    }
    if (tid < 64u) bad += w[1024u + tid] != 0x44332211u;
    bad += w[2048u + tid] != (((0xA0B0C0D0u + 2048u + tid) & 0xff0000ffu) | 0x00123400u);
    if (bad) atomicAdd(sem, bad);
    __syncthreads();
    // ---- 2. pipe cycles, random dword addresses ----
    uint32_t h = (tid * 2654435761u) ^ 0x9e3779b9u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const uint32_t base = (uint32_t)(uintptr_t)s + ((h % 16000u) * 4u);
    int slot = 0;
    for (int k = 64; k >= 4; k /= 4) {
        const bool act = (int)lane < k;
#define RUN(OP) { __syncthreads(); unsigned long long t0 = __builtin_readcyclecounter(); if (act) run<OP>(base, iters); __syncthreads(); unsigned long long t1 = __builtin_readcyclecounter(); if (tid == 0) out[slot * 16 + OP] = t1 - t0; }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
        slot++;
    }
    // ---- 3. barriers ----
    {
        __syncthreads();
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 256; i++) __syncthreads();
        unsigned long long t1 = __builtin_readcyclecounter();
        if (tid == 0) out[15 * 16 + 0] = (t1 - t0) / 256;
        uint32_t a = base, v = 0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 256; i++) {                      // one dependent read -> write -> barrier per round ("a level")
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v + 1u) : "memory");
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
        if (tid == 0) out[15 * 16 + 1] = (t1 - t0) / 256;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 256; i++) {                      // the same without a barrier: one wave's dependent chain
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v + 1u) : "memory");
        }
        t1 = __builtin_readcyclecounter();
        if (tid == 0) out[15 * 16 + 2] = (t1 - t0) / 256;
        // only wave 0 works, the others wait at the barrier
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 256; i++) {
            if (wave == 0) {
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
                asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v + 1u) : "memory");
            }
            __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
        if (tid == 0) out[15 * 16 + 3] = (t1 - t0) / 256;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;             // > 1: several workgroups (two per CU with 512: the decoder's residency)
    unsigned long long* c; uint32_t* sem;
    hipMalloc(&c, 16 * 16 * 8); hipMemset(c, 0, 16 * 16 * 8);
    hipMalloc(&sem, 4); hipMemset(sem, 0, 4);
    const int iters = 64;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 65536 + 64, 0, c, sem, iters);
    unsigned long long h[16 * 16]; uint32_t bad = 0;
    hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(&bad, sem, 4, hipMemcpyDeviceToHost);
    printf("ds_mskor_b32 semantics: %s (%u mismatches)\n", bad ? "WRONG" : "ok", bad);
    const char* names[9] = {"rd_b32", "rd2_b32", "rd_b64", "wr_b32", "wr_b8", "mskor", "add", "add_rtn", "wr_b16"};
    printf("LDS pipe cycles per wave-instruction, random dword addresses (8 waves x %d x 8 ops, %d workgroup(s))\n%-10s", iters, blocks, "");
    for (int o = 0; o < 9; o++) printf("%9s", names[o]);
    printf("\n");
    int slot = 0;
    for (int k = 64; k >= 4; k /= 4, slot++) {
        printf("lanes %2d  ", k);
        for (int o = 0; o < 9; o++) printf("%9.1f", (double)h[slot * 16 + o] / (8.0 * iters * 8));
        printf("\n");
    }
    printf("s_barrier back to back: %llu cycles;  read->write->barrier round: %llu;  read->write chain without barrier: %llu;  wave 0 works, 7 wait: %llu\n",
           h[15 * 16 + 0], h[15 * 16 + 1], h[15 * 16 + 2], h[15 * 16 + 3]);
    return bad ? 1 : 0;
}
