// probe: do unaligned ds_read_b32 / ds_write_b32 / ds_read_b64 work on gfx950, and what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t lds_rd32(uint32_t addr) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_wr32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint64_t lds_rd64(uint32_t addr) { uint64_t v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v; }
__global__ void probe(uint32_t* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t base = (uint32_t)(uintptr_t)s;   // LDS address
    for (int sh = 0; sh < 4; sh++) {
        uint32_t a = base + threadIdx.x * 12 + sh;
        out[sh * 64 + threadIdx.x] = lds_rd32(a);
        uint64_t v8 = lds_rd64(a);
        out[256 + sh * 128 + threadIdx.x * 2] = (uint32_t)v8; out[256 + sh * 128 + threadIdx.x * 2 + 1] = (uint32_t)(v8 >> 32);
    }
    __syncthreads();
    // unaligned writes
    lds_wr32(base + 2048 + threadIdx.x * 5 + 1, 0xA0B0C0D0u + threadIdx.x);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[1024 + i] = s[2048 + i];
    // timing: 256 dependent unaligned vs aligned reads
    for (int sh = 0; sh < 2; sh++) {
        uint32_t a = base + threadIdx.x * 8 + sh, acc = 0;
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < 256; i++) { uint32_t v = lds_rd32(a + (acc & 4)); acc += v & 4; }
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[sh] = t1 - t0;
        out[2000 + sh * 64 + threadIdx.x] = acc;
    }
}
int main() {
    uint32_t* d; unsigned long long* c; hipMalloc(&d, 16384); hipMalloc(&c, 64); hipMemset(d, 0, 16384);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c);
    std::vector<uint32_t> h(4096); unsigned long long hc[2];
    if (hipMemcpy(h.data(), d, 16384, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 1; }
    hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    auto byte = [](int i) { return (uint32_t)(uint8_t)(i * 7 + 3); };
    int bad32 = 0, bad64 = 0, badw = 0;
    for (int sh = 0; sh < 4; sh++) for (int t = 0; t < 64; t++) {
        int a = t * 12 + sh; uint32_t e = byte(a) | byte(a + 1) << 8 | byte(a + 2) << 16 | byte(a + 3) << 24;
        if (h[sh * 64 + t] != e) bad32++;
        uint32_t e2 = byte(a + 4) | byte(a + 5) << 8 | byte(a + 6) << 16 | byte(a + 7) << 24;
        if (h[256 + sh * 128 + t * 2] != e || h[256 + sh * 128 + t * 2 + 1] != e2) bad64++;
    }
    std::vector<uint8_t> exp(512); for (int i = 0; i < 512; i++) exp[i] = (uint8_t)byte(2048 + i);
    for (int t = 0; t < 64; t++) { uint32_t v = 0xA0B0C0D0u + t; for (int k = 0; k < 4; k++) exp[t * 5 + 1 + k] = (uint8_t)(v >> (8 * k)); }
    // later lanes overwrite earlier ones where ranges overlap (5-byte stride, 4-byte write: no overlap)
    for (int i = 0; i < 512; i++) if ((uint8_t)h[1024 + i] != exp[i]) badw++;
    printf("unaligned ds_read_b32 mismatches: %d/256, ds_read_b64: %d/256, ds_write_b32 byte mismatches: %d/512\n", bad32, bad64, badw);
    printf("256 dependent ds_read_b32: aligned %llu cycles, unaligned(+1) %llu cycles\n", hc[0], hc[1]);
    return 0;
}
