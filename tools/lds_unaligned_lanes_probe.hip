// probe: cost of unaligned LDS ops with only k lanes active
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t lds_rd32(uint32_t addr) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ uint64_t lds_rd64(uint32_t addr) { uint64_t v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_wr64(uint32_t addr, uint64_t v) { asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wr32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wr8(uint32_t addr, uint32_t v) { asm volatile("ds_write_b8 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory"); }
__global__ void probe(unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) uint8_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = 0;
    __syncthreads();
    uint32_t base = (uint32_t)(uintptr_t)s;
    int slot = 0;
    for (int k = 1; k <= 64; k *= 4) {
        for (int sh = 0; sh < 2; sh++) {
            const bool act = (int)threadIdx.x < k;
            uint32_t a = base + threadIdx.x * 72 + sh * 3, acc = 0;
            unsigned long long t0 = __builtin_readcyclecounter();
            if (act) for (int i = 0; i < 256; i++) { uint32_t v = lds_rd32(a + (acc & 8)); acc += v & 8; }
            unsigned long long t1 = __builtin_readcyclecounter();
            if (act) for (int i = 0; i < 256; i++) { uint64_t v = lds_rd64(a + (acc & 8)); acc += (uint32_t)v & 8; }
            unsigned long long t2 = __builtin_readcyclecounter();
            if (act) for (int i = 0; i < 256; i++) { lds_wr64(a + (acc & 8), acc); }
            unsigned long long t3 = __builtin_readcyclecounter();
            if (act) for (int i = 0; i < 256; i++) { lds_wr32(a + (acc & 8), acc); }
            unsigned long long t4 = __builtin_readcyclecounter();
            if (act) for (int i = 0; i < 256; i++) { lds_wr8(a + (acc & 8), acc); }
            unsigned long long t5 = __builtin_readcyclecounter();
            if (threadIdx.x == 0) { cyc[slot * 8 + 0] = t1 - t0; cyc[slot * 8 + 1] = t2 - t1; cyc[slot * 8 + 2] = t3 - t2; cyc[slot * 8 + 3] = t4 - t3; cyc[slot * 8 + 4] = t5 - t4; }
            slot++;
            if (acc == 12345) cyc[63] = acc;
        }
    }
}
int main() {
    unsigned long long* c; hipMalloc(&c, 64 * 8 * 8); hipMemset(c, 0, 64 * 8 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c);
    unsigned long long h[64 * 8];
    hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    int slot = 0;
    for (int k = 1; k <= 64; k *= 4) for (int sh = 0; sh < 2; sh++, slot++)
        printf("lanes %2d %s: per op  rd32 %5.1f  rd64 %5.1f  wr64 %5.1f  wr32 %5.1f  wr8 %5.1f cycles\n", k, sh ? "unaligned(+3)" : "aligned      ",
               h[slot * 8] / 256.0, h[slot * 8 + 1] / 256.0, h[slot * 8 + 2] / 256.0, h[slot * 8 + 3] / 256.0, h[slot * 8 + 4] / 256.0);
    return 0;
}
