// cj_engine.hpp — private host-side state shared by engine.hip (C-ABI, batch submission) and frame.hip
// (framed formats on top of the batch engine).  Not part of the C-ABI.
#pragma once
#include "cj_common.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace cj {

std::string& hip_err_slot();     // thread-local text behind cj_last_hip_error()

inline bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    hip_err_slot() = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return false;
}
#define HIP_TRY(expr, ret) do { if (!cj::hip_ok((expr), #expr)) return (ret); } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 4096;
        if (!hip_ok(hipMalloc(&p, want), "hipMalloc")) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    // for buffers of gigabytes (the record areas of big chunks): no growth slack
    bool reserve_exact(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (!hip_ok(hipMalloc(&p, n), "hipMalloc")) { p = nullptr; return false; }
        cap = n;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinnedBuf {          // page-locked host staging (full PCIe rate, truly asynchronous copies)
    uint8_t* p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 4096;
        if (!hip_ok(hipHostMalloc((void**)&p, want, hipHostMallocDefault), "hipHostMalloc")) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// host-side pack/scatter of many small buffers is memcpy-bound on one core (~20 GB/s); split it over a few threads
template <class F>
void parallel_chunks(size_t n, size_t total_bytes, F&& fn) {
    unsigned t = total_bytes > (32u << 20) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (t <= 1 || n < 2 * t) { fn(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + t - 1) / t;
    for (unsigned k = 0; k < t; k++) {
        const size_t a = k * per, b = std::min(n, a + per);
        if (a >= b) break;
        th.emplace_back([=, &fn] { fn(a, b); });
    }
    for (auto& x : th) x.join();
}

}  // namespace cj

struct cj_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;                 // serialises host-batch staging on this engine
    cj::DevBuf d_in, d_out, d_meta;
    std::mutex scratch_mu;         // LZ4 parse->decode scratch (sync points, per-chunk meta), reused across calls
    cj::DevBuf d_sync, d_pmeta, d_lanelist;   // d_lanelist: word [2] = the workgroup decoder's chunk counter
    hipEvent_t scratch_free = nullptr;    // recorded after the last kernel that reads the scratch
    cj::PinnedBuf h_in, h_out, h_res;   // h_res: the results of a sliced host batch (engine.hip: batch_host_sliced)
    hipStream_t stream_back = nullptr;  // ... its copies back to the host (the engine's stream keeps uploading and decoding the next slice)
    std::vector<hipEvent_t> slice_ev;
    std::vector<uint64_t> h_meta;
    cj::DevBuf d_frame;            // frame.hip: assembled / staged framed stream
    cj::DevBuf d_tab;              // LDS decoder variant 2: per-workgroup record tables
    cj::DevBuf d_biglist, d_bigrecs, d_bigmisc, d_bigslabtab;   // chunks of 64 KiB .. 256 KiB in a device batch (big_chunks.hpp, CJ_FLAG_BIG_CHUNKS): record areas; list + summaries + slab items; the slab decoder's tables
    uint32_t* h_count = nullptr;   // pinned words: the number of big chunks of the last flagged batches, copied back without waiting (engine.hip plan_big)
    hipEvent_t big_ev[8] = {};     // ... one event per slot (the copy has landed)
    uint32_t big_obs[8] = {};      // ... the counts that have
    int big_state[8] = {};         // 0 = empty, 1 = copy in flight, 2 = count known
    int big_next = 0;
    cj::DevBuf d_big, d_bigtab;    // large.hip: parse scratch / record tables of one large stream (under `mu`)
    int n_cu = 0;
};

namespace cj {
cj_engine* default_engine();     // lazily created on device $CJ_DEVICE (default 0); nullptr when no device is usable
// submit one batch on stream s (slices very large decode batches); 0 or CJ_E_*
int launch(cj_engine* e, cj_codec codec, cj_op op, const BatchArgs& a, hipStream_t s);
void fill_args(BatchArgs& a, uint32_t flags, size_t n, const uint8_t* in_base, const uint64_t* in_off,
               const uint64_t* in_len, uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
               int64_t* result);
}  // namespace cj
