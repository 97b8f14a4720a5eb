// lz4_decode_lanes.hip — LZ4 *block* decoder for gfx950, ONE LANE per independent chunk.
//
// Same contract and accept/reject rules as lz4_decode.hip (reference call sites
// /root/reference/src/lz4.rs:88,90,164,168 -> LZ4_decompress_safe), different mapping: for large
// batches of small-sequence data the wave-per-chunk kernel is bound by the serial token chain of each
// chunk (one dependent HBM/L2 round trip per copy, ~3.7k cycles per sequence measured).  Here every
// lane walks its own chunk, so a wavefront retires 64 sequences per step and the dependent-latency
// chain is amortised 64x; the price is uncoalesced 16 B accesses (each lane streams its own chunk),
// which is why the launcher only picks this kernel for batches large enough to fill the chip.
// All copies are 16 B per lane "wild" copies while at least 16 B of slack remain in both buffers and
// exact byte copies at the chunk's tail, so no byte past the decoded length is ever written.
#include "cj_common.hpp"

namespace cj {

__device__ __forceinline__ uint4 ld16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16u(uint8_t* p, const uint4& v) { __builtin_memcpy(p, &v, 16); }

// up to 4 bytes at in[ip..], zero-filled past iend
__device__ __forceinline__ uint32_t ld_le_tail(const uint8_t* in, uint32_t ip, uint32_t iend) {
    if (ip + 4u <= iend) return ld32u(in + ip);
    uint32_t v = 0;
    for (uint32_t i = 0; i < 4u && ip + i < iend; i++) v |= (uint32_t)in[ip + i] << (8u * i);
    return v;
}

// dst[0..n) = src[0..n), non-overlapping; room_dst/room_src = bytes that may be touched from dst/src
__device__ __forceinline__ void lane_copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t room_dst, uint32_t room_src) {
    uint32_t k = 0;
    const uint32_t room = room_dst < room_src ? room_dst : room_src;
    for (; k < n && k + 16u <= room; k += 16u) st16u(dst + k, ld16u(src + k));
    for (; k < n; k++) dst[k] = src[k];
}

// 16-byte vector whose byte i is pat[i % d], for 1 <= d < 16 (pat = the d bytes before dst)
__device__ __forceinline__ uint4 splat_pattern(const uint8_t* pat, uint32_t d) {
    uint32_t w[4];
    if (d == 1u) {
        uint32_t b = pat[0] * 0x01010101u;
        w[0] = w[1] = w[2] = w[3] = b;
    } else if (d == 2u) {
        uint32_t h = (uint32_t)pat[0] | ((uint32_t)pat[1] << 8);
        h |= h << 16;
        w[0] = w[1] = w[2] = w[3] = h;
    } else if (d == 4u) {
        uint32_t v = ld32u(pat);
        w[0] = w[1] = w[2] = w[3] = v;
    } else if (d == 8u) {
        w[0] = w[2] = ld32u(pat);
        w[1] = w[3] = ld32u(pat + 4);
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        uint32_t r = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16u; i++) {
            w[i >> 2] |= (uint32_t)pat[r] << (8u * (i & 3u));
            r += 1u;
            if (r == d) r = 0;
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// dst[j] = dst[j - d], j in [0, m); room = bytes that may be touched from dst
__device__ __forceinline__ void lane_match(uint8_t* dst, uint32_t d, uint32_t m, uint32_t room) {
    uint32_t k = 0;
    const uint8_t* src = dst - d;
    if (d >= 16u) {
        for (; k < m && k + 16u <= room; k += 16u) st16u(dst + k, ld16u(src + k));
    } else if (m >= 16u && room >= 32u) {
        const uint4 p = splat_pattern(src, d);
        const uint32_t s = (16u / d) * d;          // advance by whole periods so the phase stays aligned
        for (; k < m && k + 16u <= room; k += s) st16u(dst + k, p);
        if (k > m) k = m;
    }
    for (; k < m; k++) dst[k] = src[k];      // src[k] == dst[k - d]; the pointer form avoids u32 wrap
}

__global__ __launch_bounds__(64) void lz4_decode_lanes_kernel(BatchArgs a) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= a.n_chunks) return;
    const uint8_t* in = a.in_base + a.in_off[c];
    uint64_t n64 = a.in_len[c];
    uint8_t* out = a.out_base + a.out_off[c];
    uint64_t cap64 = a.out_cap[c];
    int64_t status = 0;

    if (a.flags & CJ_FLAG_LZ4_SIZE_PREFIX) {
        if (n64 < 4) status = CJ_E_NO_PREFIX;
        else {
            int32_t size = (int32_t)ld32u(in);
            if (size < 0) status = CJ_E_NEG_PREFIX;
            else if ((uint32_t)size > 0x7E000000u) status = CJ_E_PREFIX_TOO_BIG;
            else if ((uint64_t)size > cap64) status = CJ_E_OUT_TOO_SMALL;
            else { in += 4; n64 -= 4; cap64 = (uint64_t)size; }
        }
    } else {
        int32_t size = (int32_t)(uint32_t)cap64;
        if (cap64 > 0xFFFFFFFFull || size < 0) status = CJ_E_NEG_PREFIX;
        else if ((uint32_t)size > 0x7E000000u) status = CJ_E_PREFIX_TOO_BIG;
    }
    if (status == 0 && n64 > 0x7FFFFFF0ull) status = CJ_E_CORRUPT;
    if (status != 0) { a.result[c] = status; return; }
    const uint32_t cap = (uint32_t)cap64, iend = (uint32_t)n64;
    if (cap == 0) { a.result[c] = (iend == 1 && in[0] == 0) ? 0 : (int64_t)CJ_E_CORRUPT; return; }
    if (iend == 0) { a.result[c] = CJ_E_CORRUPT; return; }

    uint32_t ip = 0, op = 0;
    bool bad = false;
    for (;;) {
        const uint32_t t4 = ld_le_tail(in, ip, iend);
        const uint32_t token = t4 & 0xffu;
        ip += 1;
        uint64_t lit = token >> 4;
        if (lit == 15u) {
            if (ip + 15u >= iend) { bad = true; break; }
            uint32_t b = (t4 >> 8) & 0xffu;
            ip += 1; lit += b;
            if (ip + 15u > iend) { bad = true; break; }
            while (b == 255u) {
                b = in[ip];
                ip += 1; lit += b;
                if (ip + 15u > iend) { bad = true; break; }
            }
            if (bad) break;
        }
        const uint32_t rem_out = cap - op, rem_in = iend - ip;
        if ((uint64_t)rem_out < lit + 12u || (uint64_t)rem_in < lit + 8u) {
            if ((uint64_t)rem_in != lit || (uint64_t)rem_out < lit) { bad = true; break; }
            lane_copy(out + op, in + ip, (uint32_t)lit, (uint32_t)lit, (uint32_t)lit);   // exact: last bytes of both buffers
            op += (uint32_t)lit;
            break;
        }
        lane_copy(out + op, in + ip, (uint32_t)lit, rem_out, rem_in);
        ip += (uint32_t)lit; op += (uint32_t)lit;

        const uint32_t o4 = ld_le_tail(in, ip, iend);       // >= 8 input bytes remain here
        const uint32_t offset = o4 & 0xffffu;
        ip += 2;
        uint64_t mlen = token & 15u;
        if (mlen == 15u) {
            uint32_t b = (o4 >> 16) & 0xffu;
            ip += 1; mlen += b;
            if (ip + 4u > iend) { bad = true; break; }
            while (b == 255u) {
                b = in[ip];
                ip += 1; mlen += b;
                if (ip + 4u > iend) { bad = true; break; }
            }
            if (bad) break;
        }
        mlen += 4u;
        if (offset == 0u || offset > op) { bad = true; break; }
        if ((uint64_t)(cap - op) < mlen + 5u) { bad = true; break; }
        lane_match(out + op, offset, (uint32_t)mlen, cap - op);
        op += (uint32_t)mlen;
    }
    a.result[c] = bad ? (int64_t)CJ_E_CORRUPT : (int64_t)op;
}

void launch_lz4_decode_lanes(const BatchArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return;
    dim3 grid((a.n_chunks + 63u) / 64u), block(64);
    hipLaunchKernelGGL(lz4_decode_lanes_kernel, grid, block, 0, s, a);
}

}  // namespace cj
