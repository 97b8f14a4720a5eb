// engine.hip — the C-ABI of libcramjam_hip.so (include/cramjam_hip.h): per-GPU engine, batch
// submission, host staging and the single-buffer drop-in entry points.  Host-side C++ over the HIP
// runtime; all codec arithmetic is in the four *_decode/_encode.hip kernels.  There is no CPU codec
// in this library: if no device is usable the entry points return CJ_E_NO_DEVICE.
#include "cj_engine.hpp"
#include "big_chunks.hpp"

using cj::hip_ok;
using cj::parallel_chunks;

namespace {
thread_local std::string g_hip_err;
}
std::string& cj::hip_err_slot() { return g_hip_err; }

void cj::fill_args(cj::BatchArgs& a, uint32_t flags, size_t n, const uint8_t* in_base, const uint64_t* in_off,
               const uint64_t* in_len, uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
               int64_t* result) {
    a.in_base = in_base; a.in_off = in_off; a.in_len = in_len;
    a.out_base = out_base; a.out_off = out_off; a.out_cap = out_cap;
    a.result = result; a.n_chunks = (uint32_t)n; a.flags = flags; a.hist = nullptr;
}

namespace {

#ifndef CJ_L2_WGS_PER_CU
#define CJ_L2_WGS_PER_CU 2
#endif
constexpr uint32_t kWgsPerCu = CJ_L2_WGS_PER_CU;       // persistent workgroups of the LDS decoder per CU (tuning variants change it together with CJ_L2_WINDOW)

// scratch of the workgroup decoders (per-chunk verdicts, the chunk counter, record tables), shared by every call on the
// engine: a call waits (on the stream) for the previous user before it overwrites them
int lds_scratch(cj_engine* e, const cj::BatchArgs& a, hipStream_t s, bool with_sync) {
    const size_t list_bytes = 256 + cj::kClaimBytes + (size_t)a.n_chunks * 8;
    const size_t sync_bytes = with_sync ? cj::lz4_lds_scratch_sync_bytes(a.n_chunks) : 0;
    const bool grow = sync_bytes > e->d_sync.cap || cj::lz4_lds_scratch_meta_bytes(a.n_chunks) > e->d_pmeta.cap || list_bytes > e->d_lanelist.cap;
    if (grow && e->scratch_free) HIP_TRY(hipEventSynchronize(e->scratch_free), CJ_E_NO_DEVICE);
    if (!e->d_sync.reserve(sync_bytes) || !e->d_pmeta.reserve(cj::lz4_lds_scratch_meta_bytes(a.n_chunks)) || !e->d_lanelist.reserve(list_bytes)) return CJ_E_OOM;
    if (!e->scratch_free) HIP_TRY(hipEventCreateWithFlags(&e->scratch_free, hipEventDisableTiming), CJ_E_NO_DEVICE);
    else HIP_TRY(hipStreamWaitEvent(s, e->scratch_free, 0), CJ_E_NO_DEVICE);   // previous user of the scratch
    HIP_TRY(hipMemsetAsync(e->d_lanelist.p, 0, 256 + cj::kClaimBytes, s), CJ_E_NO_DEVICE);        // words [64 ..): the decoders' chunk counters
    if (e->n_cu == 0) HIP_TRY(hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, e->device), CJ_E_NO_DEVICE);
    if (!e->d_tab.reserve(cj::lz4_lds2_tab_bytes(kWgsPerCu * (uint32_t)e->n_cu))) return CJ_E_OOM;
    return 0;
}

// LZ4 block / Snappy raw decode of a batch of independent chunks:
//   up to CJ_FUSED_MAX_CHUNKS chunks   parse + decode in ONE kernel (two persistent workgroups per CU: segmented parse on the staged
//                      chunk, records, literals, matches through the LDS window)
//   larger batches     lane-per-chunk parse kernel (validation + sync points in memory) -> the same workgroup decoder: with every
//                      CU's LDS pipe saturated by the resolver the in-kernel parse's dependent LDS reads cost more than a second
//                      pass over the input (100 k chunks: 18 ms fused, 10.6 ms with the parse kernel)
//   then               the wavefront-per-chunk kernel on what the parse left over (errors, chunks above 64 KiB, few long runs)
//   flags              CJ_FLAG_FORCE_WAVE_PER_CHUNK / _LANE_PER_CHUNK: one mapping for every chunk (tests, comparisons)
constexpr size_t kBigCap = 8192;
constexpr int kBigObs = 8;                // counts of big chunks the engine remembers (cj_engine::big_obs)

// CJ_FLAG_BIG_CHUNKS: which chunks lie in (64 KiB, 256 KiB] is known on the device only (big_list_kernel), but the record areas
// (1 MiB per listed chunk), the number of groups and the slab decoder's tables are decided here.  The engine sizes them from what it
// has SEEN: every flagged call copies its list's count to a pinned slot without waiting for it, and the next calls plan for the largest of
// the last kBigObs counts that have arrived.  What a batch holds beyond the plan stays with the wavefront kernel (slower, never wrong)
// and raises the plan of the calls behind it.  Only an engine that has not seen any count yet reads its first one back (the call
// waits for the stream once, like every call that grows the engine's scratch); after that a flagged call only enqueues.
int plan_big(cj_engine* e, cj_codec codec, const cj::BatchArgs& a, hipStream_t s, uint32_t** list_out, uint32_t* n_plan) {
    *n_plan = 0; *list_out = nullptr;
    const size_t big_list_bytes = ((4 + (size_t)a.n_chunks) * 4 + 255) & ~(size_t)255;
    if (!e->d_biglist.reserve(big_list_bytes)) { (void)hipGetLastError(); return 0; }      // no room for the list: the chunks stay with the wavefront kernel
    uint32_t* big_list = (uint32_t*)e->d_biglist.p;
    if (!e->h_count) {
        HIP_TRY(hipHostMalloc((void**)&e->h_count, 64, hipHostMallocDefault), CJ_E_OOM);
        for (int i = 0; i < kBigObs; i++) HIP_TRY(hipEventCreateWithFlags(&e->big_ev[i], hipEventDisableTiming), CJ_E_NO_DEVICE);
    }
    cj::launch_big_list(a, codec, big_list, s);
    // counts that have arrived since the last call
    bool any = false;
    uint32_t plan = 0;
    for (int i = 0; i < kBigObs; i++) {
        if (e->big_state[i] == 1 && hipEventQuery(e->big_ev[i]) == hipSuccess) { e->big_obs[i] = e->h_count[i]; e->big_state[i] = 2; }
        if (e->big_state[i] == 2) { any = true; plan = std::max(plan, e->big_obs[i]); }
    }
    (void)hipGetLastError();                                  // (hipErrorNotReady from the queries)
    const int slot = e->big_next;
    e->big_next = (slot + 1) % kBigObs;
    if (e->big_state[slot] == 1) HIP_TRY(hipEventSynchronize(e->big_ev[slot]), CJ_E_NO_DEVICE);   // (eight flagged calls in flight: the oldest copy must have landed before its slot is reused)
    HIP_TRY(hipMemcpyAsync(e->h_count + slot, big_list, 4, hipMemcpyDeviceToHost, s), CJ_E_NO_DEVICE);
    HIP_TRY(hipEventRecord(e->big_ev[slot], s), CJ_E_NO_DEVICE);
    e->big_state[slot] = 1;
    if (!any) {                                                // an engine that has seen nothing yet: this batch's own count
        HIP_TRY(hipStreamSynchronize(s), CJ_E_NO_DEVICE);
        e->big_obs[slot] = e->h_count[slot]; e->big_state[slot] = 2;
        plan = e->big_obs[slot];
    }
    *n_plan = (uint32_t)std::min<size_t>(plan, a.n_chunks);
    *list_out = big_list;
    return 0;
}          // big chunks (CJ_FLAG_BIG_CHUNKS) decoded per group: each holds a record area of 1 MiB while its group is in flight

int launch_decode(cj_engine* e, cj_codec codec, const cj::BatchArgs& a_in, hipStream_t s) {
    const bool lz4 = codec == CJ_CODEC_LZ4_BLOCK;
    cj::BatchArgs a = a_in;                                   // the batch as the kernels see it (the window promise only where it applies)
    const uint32_t small = a.flags & (CJ_FLAG_CHUNKS_LE_32K | CJ_FLAG_CHUNKS_LE_16K);
    a.flags &= ~(CJ_FLAG_CHUNKS_LE_32K | CJ_FLAG_CHUNKS_LE_16K);
    int mode = 2;
    if (a.flags & CJ_FLAG_FORCE_WAVE_PER_CHUNK) mode = 0;
    if (a.flags & CJ_FLAG_FORCE_LANE_PER_CHUNK) mode = 1;
    if (a.flags & CJ_FLAG_FORCE_LDS_PER_CHUNK) mode = 2;
    if (mode == 0) { if (lz4) cj::launch_lz4_decode(a, s); else cj::launch_snappy_decode(a, s); return 0; }
    if (mode == 1) { if (lz4) cj::launch_lz4_decode_lanes(a, s); else cj::launch_snappy_decode_lanes(a, s); return 0; }
    // the workgroup decoder (lz4_decode_lds.hip): its parse stage inside the decoder kernel for small batches, as a kernel of its own
    // in front of it for large ones (CJ_FLAG_FORCE_FUSED_PARSE / _PARSE_KERNEL: one of them at any batch size — tests, comparisons)
    bool fused = a.n_chunks <= (uint32_t)CJ_FUSED_MAX_CHUNKS;
    // (a batch of small chunks — CJ_FLAG_CHUNKS_LE_32K / _16K — runs either pipeline on windows of that size: four / eight workgroups per CU.  The
    //  one-kernel path and the parse kernel cross at ~16 k chunks for 64, 32, 16 and 8 KiB chunks alike: profiles/r06/experiments f04 / f05)
    if (a.flags & CJ_FLAG_FORCE_FUSED_PARSE) fused = true;
    if (a.flags & CJ_FLAG_FORCE_PARSE_KERNEL) fused = false;
    a.flags |= small;                                         // (round 6, f05: the one-kernel path runs on the batch's window too)
    const uint32_t win = cj::lds_window(a.flags);
    std::lock_guard<std::mutex> lock(e->scratch_mu);
    // (first: it makes `s` wait for the previous user of the engine's shared scratch — the big-chunk list below is part of it)
    const int rc = lds_scratch(e, a, s, !fused);
    if (rc != 0) return rc;
    uint32_t n_big = 0;
    uint32_t* big_list = nullptr;
    if (a.flags & CJ_FLAG_BIG_CHUNKS) {
        const int brc = plan_big(e, codec, a, s, &big_list, &n_big);
        if (brc != 0) return brc;
    }
    uint32_t* lists = (uint32_t*)e->d_lanelist.p;
    const uint32_t grid = (win >= 65536u ? kWgsPerCu : cj::lz4_lds2_wgs_per_cu(win)) * (uint32_t)e->n_cu;          // persistent workgroups: two per CU on 64 KiB windows
    if (fused) {
        cj::launch_lz4_decode_fused(a, e->d_pmeta.p, e->d_tab.p, lists + 64, grid, s, codec, win);
    } else {
        HIP_TRY(hipMemsetAsync(e->d_pmeta.p, 0, cj::lz4_lds_scratch_meta_bytes(a.n_chunks), s), CJ_E_NO_DEVICE);   // no chunk is pre-routed
        // validate, size, count sequences, sync points, route
        if (lz4) cj::launch_lz4_parse(a, e->d_sync.p, e->d_pmeta.p, s);
        else cj::launch_snappy_parse(a, e->d_sync.p, e->d_pmeta.p, s);
        cj::launch_lz4_decode_lds2(a, e->d_sync.p, e->d_pmeta.p, e->d_tab.p, lists + 64, grid, s, codec, win);
    }
    if (n_big != 0u) {
        // chunks of 64 KiB .. 256 KiB (flagged kRouteWave above): listed, parsed by 32 lanes each into records, decoded slab by slab
        // with two workgroups per CU (big_chunks.hpp) — in GROUPS of `cap` listed chunks, as many as there are record areas for
        // (1 MiB each); what that stage does not take stays flagged for the wavefront kernel
        const uint32_t cap = (uint32_t)std::min<size_t>(n_big, kBigCap);
        const uint32_t items = cj::kBigSlabs * cap;
        // (per workgroup: a slab's records + at most one extra record each — the literals of a match that is cut at the slab's start — and at most one
        //  cross copy each; no forwarding in this mode: nothing stages the input)
        const uint32_t tab_stride = 2u * (cj::kBigSlabRecs + 64u), cross_stride = cj::kBigSlabRecs + 64u;
        // d_bigmisc: BigMeta x cap | item rows (5 x 8 bytes x items) | item meta | done flags | counter | walk scratch
        const size_t o_meta = 0, o_rows = (o_meta + cj::big_meta_bytes(cap) + 255) & ~(size_t)255,
                     o_imeta = o_rows + cj::kBigItemRows * 8 * (size_t)items, o_done = o_imeta + 8 * (size_t)items, o_ctr = o_done + 4 * (size_t)items,
                     o_walk = o_ctr + 256, total = o_walk + cj::big_walk_scratch_bytes(cap);
        // no room for the record areas: the chunks stay with the wavefront kernel (slower, never wrong)
        const uint32_t sgrid = std::min(cj::kBigSlabWgsPerCu * (uint32_t)e->n_cu, items);          // the slab decoder's persistent workgroups (their tables are 0.9 MiB each)
        const bool room = e->d_bigrecs.reserve_exact(cj::big_recs_bytes(cap)) && e->d_bigmisc.reserve(total)
            && e->d_bigslabtab.reserve((size_t)sgrid * tab_stride * 16 + (size_t)sgrid * cross_stride * 16 + (size_t)sgrid * (tab_stride + 512u) * 4);
        if (!room) (void)hipGetLastError();
        uint8_t* m = (uint8_t*)e->d_bigmisc.p;
        for (uint32_t base = 0; room && base < n_big; base += cap) {
            HIP_TRY(hipMemsetAsync(m + o_ctr, 0, 256, s), CJ_E_NO_DEVICE);
            cj::launch_big_parse(a, codec, big_list, base, cap, e->d_bigrecs.p, m + o_meta, e->d_pmeta.p, m + o_walk, s);
            uint64_t* rows = (uint64_t*)(m + o_rows);
            cj::launch_big_items(a, big_list, base, m + o_meta, e->d_bigrecs.p, cap, rows, m + o_imeta, (uint32_t*)(m + o_done), s);
            cj::BatchArgs it = a;
            it.in_off = rows; it.in_len = rows + items; it.out_off = rows + 2 * (size_t)items; it.out_cap = rows + 3 * (size_t)items;
            it.result = (int64_t*)(rows + 4 * (size_t)items); it.n_chunks = items; it.flags = a.flags & CJ_FLAG_DEBUG_PROFILE;
            uint8_t* t = (uint8_t*)e->d_bigslabtab.p;
            cj::launch_lz4_decode_big_slabs(it, m + o_imeta, e->d_bigrecs.p, m + o_meta, cap, t, (uint32_t*)(m + o_ctr), (uint32_t*)(m + o_done),
                                            t + (size_t)sgrid * tab_stride * 16, tab_stride, cross_stride, sgrid, s, codec);
        }
    }
    if (lz4) cj::launch_lz4_decode_routed(a, e->d_pmeta.p, s);                // few long runs / oversize chunks / errors
    else cj::launch_snappy_decode_routed(a, e->d_pmeta.p, s);
    HIP_TRY(hipEventRecord(e->scratch_free, s), CJ_E_NO_DEVICE);
    return 0;
}

// LZ4 block / Snappy raw encode of a batch of independent chunks: one workgroup of two wavefronts per chunk (lz4_encode.hip)
int launch_encode(cj_engine*, cj_codec codec, const cj::BatchArgs& a, hipStream_t s) {
    HIP_TRY(codec == CJ_CODEC_LZ4_BLOCK ? cj::launch_lz4_encode(a, s) : cj::launch_snappy_encode(a, s), CJ_E_NO_DEVICE);
    return 0;
}

int launch_slice(cj_engine* e, cj_codec codec, cj_op op, const cj::BatchArgs& a, hipStream_t s) {
    if (codec != CJ_CODEC_LZ4_BLOCK && codec != CJ_CODEC_SNAPPY_RAW) return CJ_E_BAD_ARG;
    const int rc = op == CJ_OP_DECOMPRESS ? launch_decode(e, codec, a, s) : launch_encode(e, codec, a, s);
    if (rc != 0) return rc;
    HIP_TRY(hipGetLastError(), CJ_E_NO_DEVICE);
    return 0;
}

}  // namespace

// Very large batches are submitted in slices: the lane-per-chunk kernels lose efficiency when several hundred
// thousand chunks are in flight at once (their random match reads thrash L2 and the write amplification grows:
// 1 M chunks in one go ran at 299 GB/s vs 430 GB/s at 100 k), and the parse/LDS scratch stays bounded.

int cj::launch(cj_engine* e, cj_codec codec, cj_op op, const cj::BatchArgs& a, hipStream_t s) {
#ifdef CJ_DEBUG_KNOBS
    static const size_t kSliceChunks = [] {                   // (tuning builds only: the shipped library reads no environment)
        const char* v = std::getenv("CJ_SLICE_CHUNKS");
        size_t x = v ? (size_t)std::strtoull(v, nullptr, 10) : (size_t)CJ_SLICE_CHUNKS_DEFAULT;
        return x < 8192 ? (size_t)8192 : x;
    }();
#else
    constexpr size_t kSliceChunks = CJ_SLICE_CHUNKS_DEFAULT;
#endif
    if (op != CJ_OP_DECOMPRESS || a.n_chunks <= kSliceChunks) return launch_slice(e, codec, op, a, s);
    for (size_t start = 0; start < a.n_chunks; start += kSliceChunks) {
        cj::BatchArgs b = a;
        b.in_off += start; b.in_len += start; b.out_off += start; b.out_cap += start; b.result += start;
        b.n_chunks = (uint32_t)std::min(kSliceChunks, (size_t)a.n_chunks - start);
        int rc = launch_slice(e, codec, op, b, s);
        if (rc != 0) return rc;
    }
    return 0;
}

namespace {
std::once_flag g_default_once;
cj_engine* g_default = nullptr;
int g_default_rc = CJ_E_NO_DEVICE;
}  // namespace

cj_engine* cj::default_engine() {
    std::call_once(g_default_once, [] {
        int dev = 0;                                          // (the single-buffer entry points: device 0 of what HIP_VISIBLE_DEVICES shows)
#ifdef CJ_DEBUG_KNOBS
        if (const char* s = std::getenv("CJ_DEVICE")) dev = std::atoi(s);
#endif
        g_default_rc = cj_engine_create(dev, &g_default);
    });
    return g_default_rc == 0 ? g_default : nullptr;
}

namespace {

using cj::default_engine;
using cj::fill_args;
using cj::launch;

// single buffers above this size take large.hip's piece-parallel path (compress)
constexpr size_t kLargeMin = 65536;
// compress: already above two 4 KiB sub-pieces (sixteen wavefronts on a 64 KiB buffer instead of one: 1.7 -> 0.3 ms)
constexpr size_t kLargeMinCompress = 8192;

int64_t single(cj_codec codec, cj_op op, uint32_t flags, const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    cj_engine* e = default_engine();
    if (!e) return CJ_E_NO_DEVICE;
    int64_t res = CJ_E_NO_DEVICE;
    const uint8_t* ins[1] = { in };
    uint8_t* outs[1] = { out };
    int rc = cj_batch_host(e, codec, op, flags, 1, ins, &n, outs, &cap, &res);
    return rc != 0 ? (int64_t)rc : res;
}

// A LARGE host batch, in slices: while slice k is uploaded and decoded on the engine's stream and slice k - 1 travels back
// on a second one, the host packs slice k + 1 into the pinned staging and scatters the slices that have arrived — the one-shot
// path below does these five things one after the other (pack, H2D, kernels, D2H, scatter: 18.8 GB/s of output for 16 384 x 64 KiB).
// m = the engine's meta rows (in_off | in_len | out_off | out_cap | result), already laid out by the caller; e->mu is held.
int batch_host_sliced(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n, const uint8_t* const* in_ptrs, const size_t* in_lens,
                      uint8_t* const* out_ptrs, const size_t* out_caps, int64_t* result, uint64_t in_total, uint64_t out_total) {
    const std::vector<uint64_t>& m = e->h_meta;
    if (!e->h_in.reserve(in_total) || !e->h_out.reserve(out_total) || !e->h_res.reserve(n * 8)) return CJ_E_OOM;
    if (!e->stream_back) HIP_TRY(hipStreamCreateWithFlags(&e->stream_back, hipStreamNonBlocking), CJ_E_NO_DEVICE);
    size_t K = (size_t)((in_total + out_total) >> 26);               // ~64 MiB of traffic per slice
    K = K < 2 ? 2 : (K > 16 ? 16 : K);
    while (e->slice_ev.size() < 2 * K) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming), CJ_E_NO_DEVICE);
        e->slice_ev.push_back(ev);
    }
    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint8_t* d_out = (uint8_t*)e->d_out.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    int64_t* h_res = (int64_t*)e->h_res.p;
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), 4 * n * 8, hipMemcpyHostToDevice, e->stream), CJ_E_NO_DEVICE);
    std::vector<size_t> c(K + 1, n);                                 // slice k = chunks [c[k], c[k + 1]): equal shares of the output space (compress: of the bounds)
    c[0] = 0;
    for (size_t k = 1; k < K; k++) {
        const uint64_t want = out_total / K * k;
        size_t lo = c[k - 1], hi = n;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (m[2 * n + mid] < want) lo = mid + 1; else hi = mid; }
        c[k] = lo;
    }
    const auto in_at = [&](size_t i) { return i < n ? m[i] : in_total; };
    const auto out_at = [&](size_t i) { return i < n ? m[2 * n + i] : out_total; };
    const auto scatter = [&](size_t k) {
        const size_t a0 = c[k], b0 = c[k + 1];
        parallel_chunks(b0 - a0, out_at(b0) - out_at(a0), [&](size_t a, size_t b) {
            for (size_t i = a0 + a; i < a0 + b; i++) {
                result[i] = h_res[i];
                if (result[i] <= 0) continue;
                if ((uint64_t)result[i] > out_caps[i]) { result[i] = CJ_E_COMPRESS_FAILED; continue; }
                std::memcpy(out_ptrs[i], e->h_out.p + m[2 * n + i], (size_t)result[i]);
            }
        });
    };
    const auto bail = [&](int rc) { (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->stream_back); return rc; };
    size_t scattered = 0;
    for (size_t k = 0; k < K; k++) {
        const size_t a0 = c[k], b0 = c[k + 1];
        if (b0 > a0) {
            parallel_chunks(b0 - a0, in_at(b0) - in_at(a0), [&](size_t a, size_t b) {
                for (size_t i = a0 + a; i < a0 + b; i++)
                    if (in_lens[i]) std::memcpy(e->h_in.p + m[i], in_ptrs[i], in_lens[i]);
            });
            if (in_at(b0) > in_at(a0) && !hip_ok(hipMemcpyAsync(d_in + in_at(a0), e->h_in.p + in_at(a0), in_at(b0) - in_at(a0), hipMemcpyHostToDevice, e->stream), "hipMemcpyAsync")) return bail(CJ_E_NO_DEVICE);
            cj::BatchArgs a;
            fill_args(a, flags, b0 - a0, d_in, d_meta + a0, d_meta + n + a0, d_out, d_meta + 2 * n + a0, d_meta + 3 * n + a0, (int64_t*)(d_meta + 4 * n + a0));
            const int rc = cj::launch(e, codec, op, a, e->stream);
            if (rc != 0) return bail(rc);
        }
        if (!hip_ok(hipEventRecord(e->slice_ev[2 * k], e->stream), "hipEventRecord") || !hip_ok(hipStreamWaitEvent(e->stream_back, e->slice_ev[2 * k], 0), "hipStreamWaitEvent")) return bail(CJ_E_NO_DEVICE);
        if (b0 > a0) {
            if (!hip_ok(hipMemcpyAsync(h_res + a0, d_meta + 4 * n + a0, (b0 - a0) * 8, hipMemcpyDeviceToHost, e->stream_back), "hipMemcpyAsync")) return bail(CJ_E_NO_DEVICE);
            if (out_at(b0) > out_at(a0) && !hip_ok(hipMemcpyAsync(e->h_out.p + out_at(a0), d_out + out_at(a0), out_at(b0) - out_at(a0), hipMemcpyDeviceToHost, e->stream_back), "hipMemcpyAsync")) return bail(CJ_E_NO_DEVICE);
        }
        if (!hip_ok(hipEventRecord(e->slice_ev[2 * k + 1], e->stream_back), "hipEventRecord")) return bail(CJ_E_NO_DEVICE);
        while (scattered < k && hipEventQuery(e->slice_ev[2 * scattered + 1]) == hipSuccess) scatter(scattered++);
    }
    for (; scattered < K; scattered++) {
        if (!hip_ok(hipEventSynchronize(e->slice_ev[2 * scattered + 1]), "hipEventSynchronize")) return bail(CJ_E_NO_DEVICE);
        scatter(scattered);
    }
    HIP_TRY(hipStreamSynchronize(e->stream), CJ_E_NO_DEVICE);
    return 0;
}

}  // namespace

extern "C" {

const char* cj_strerror(int64_t code) {
    switch (code) {
    case CJ_E_INPUT_TOO_LARGE: return "Compression input too long.";
    case CJ_E_COMPRESS_FAILED: return "Compression failed";
    case CJ_E_NO_PREFIX: return "Source buffer must at least contain size prefix.";
    case CJ_E_NEG_PREFIX: return "Parsed size prefix in buffer must not be negative.";
    case CJ_E_PREFIX_TOO_BIG: return "Given size parameter is too big";
    case CJ_E_OUT_TOO_SMALL: return "buffer isn't large enough to hold decompressed data";
    case CJ_E_CORRUPT: return "Decompression failed. Input invalid or too long?";
    case CJ_E_SNAPPY_EMPTY: return "snappy: corrupt input (empty)";
    case CJ_E_SNAPPY_HEADER: return "snappy: corrupt input (invalid header)";
    case CJ_E_SNAPPY_TOO_BIG: return "snappy: input buffer (size) is bigger than the maximum allowed";
    case CJ_E_SNAPPY_BUF_SMALL: return "snappy: output buffer is too small";
    case CJ_E_SNAPPY_CORRUPT: return "snappy: corrupt input";
    case CJ_E_FRAME_EOF: return "failed to fill whole buffer";
    case CJ_E_FRAME_WRITE: return "failed to write whole buffer";
    case CJ_E_SNAPPY_STREAM_HEADER: return "snappy: corrupt input (expected stream header but got unexpected chunk type byte)";
    case CJ_E_SNAPPY_CHUNK_TYPE: return "snappy: corrupt input (unsupported chunk type)";
    case CJ_E_SNAPPY_CHUNK_LEN: return "snappy: corrupt input (unsupported chunk length)";
    case CJ_E_SNAPPY_CHECKSUM: return "snappy: corrupt input (bad checksum)";
    case CJ_E_LZ4F_FRAME_TYPE: return "LZ4 error: ERROR_frameType_unknown";
    case CJ_E_LZ4F_HEADER: return "LZ4 error: ERROR_headerChecksum_invalid";
    case CJ_E_LZ4F_BLOCK_SIZE: return "LZ4 error: ERROR_maxBlockSize_invalid";
    case CJ_E_LZ4F_BLOCK_CHECKSUM: return "LZ4 error: ERROR_blockChecksum_invalid";
    case CJ_E_LZ4F_CONTENT_CHECKSUM: return "LZ4 error: ERROR_contentChecksum_invalid";
    case CJ_E_LZ4F_CONTENT_SIZE: return "LZ4 error: ERROR_frameSize_wrong";
    case CJ_E_LZ4F_INCOMPLETE: return "Finish runned before read end of compressed stream";
    case CJ_E_LZ4F_DECOMPRESS: return "LZ4 error: ERROR_decompressionFailed";
    case CJ_E_NO_DEVICE: return "cramjam_hip: no usable HIP device (no CPU fallback exists)";
    case CJ_E_BAD_ARG: return "cramjam_hip: bad argument";
    case CJ_E_OOM: return "cramjam_hip: out of memory";
    default: return code >= 0 ? "ok" : "cramjam_hip: unknown error";
    }
}

const char* cj_last_hip_error(void) { return g_hip_err.c_str(); }
int cj_abi_version(void) { return CJ_ABI_VERSION; }

int cj_device_count(void) {
    int n = 0;
    if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount")) return 0;
    return n;
}

size_t cj_lz4_block_compress_bound(size_t len, int prepend) {
    size_t b = len > 0x7E000000u ? 0 : len + len / 255 + 16;
    return prepend ? b + 4 : b;
}

int64_t cj_lz4_block_prefixed_len(const uint8_t* in, size_t n) {
    if (n < 4) return CJ_E_NO_PREFIX;
    return (int64_t)((uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24));
}

size_t cj_snappy_raw_max_compress_len(size_t len) {
    if ((uint64_t)len > 0xFFFFFFFFull) return 0;
    uint64_t m = 32 + (uint64_t)len + (uint64_t)len / 6;
    return m > 0xFFFFFFFFull ? 0 : (size_t)m;
}

int64_t cj_snappy_raw_decompress_len(const uint8_t* in, size_t n) {
    if (n == 0) return 0;
    uint64_t v = 0;
    unsigned shift = 0;
    for (size_t i = 0; i < n && i < 10; i++) {
        uint8_t b = in[i];
        if (b < 0x80) {
            if (i == 9 && b > 1) return CJ_E_SNAPPY_HEADER;
            v |= (uint64_t)b << shift;
            return v > 0xFFFFFFFFull ? (int64_t)CJ_E_SNAPPY_TOO_BIG : (int64_t)v;
        }
        v |= (uint64_t)(b & 0x7f) << shift;
        shift += 7;
    }
    return CJ_E_SNAPPY_HEADER;
}

int cj_engine_create(int device, cj_engine** out) {
    if (!out) return CJ_E_BAD_ARG;
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n), CJ_E_NO_DEVICE);
    if (device < 0 || device >= n) { g_hip_err = "device index out of range"; return CJ_E_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device), CJ_E_NO_DEVICE);
    cj_engine* e = new cj_engine();
    e->device = device;
    if (!hip_ok(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate")) { delete e; return CJ_E_NO_DEVICE; }
    *out = e;
    return 0;
}

void cj_engine_destroy(cj_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    e->d_in.release(); e->d_out.release(); e->d_meta.release(); e->d_sync.release(); e->d_bigrecs.release(); e->d_bigmisc.release(); e->d_biglist.release(); if (e->h_count) { (void)hipHostFree(e->h_count); for (auto& ev : e->big_ev) if (ev) (void)hipEventDestroy(ev); } e->d_bigslabtab.release(); e->d_pmeta.release(); e->d_lanelist.release(); e->d_frame.release(); e->d_tab.release(); e->d_big.release(); e->d_bigtab.release();
    e->h_in.release(); e->h_out.release(); e->h_res.release();
    for (hipEvent_t ev : e->slice_ev) (void)hipEventDestroy(ev);
    if (e->stream_back) (void)hipStreamDestroy(e->stream_back);
    if (e->scratch_free) (void)hipEventDestroy(e->scratch_free);
   
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int cj_engine_device(const cj_engine* e) { return e ? e->device : -1; }

// the flag bits a C-ABI caller may set; everything else (piece splitting, tail reports, linked-frame parse: cj_common.hpp) belongs
// to large.hip / frame.hip, which call cj::launch directly — a stray bit would make a kernel read descriptors that are not there
static constexpr uint32_t kPublicFlags = CJ_FLAG_LZ4_SIZE_PREFIX | CJ_FLAG_FORCE_FUSED_PARSE | CJ_FLAG_FORCE_PARSE_KERNEL | CJ_FLAG_CHUNKS_LE_32K | CJ_FLAG_CHUNKS_LE_16K | CJ_FLAG_FORCE_WAVE_PER_CHUNK | CJ_FLAG_FORCE_LANE_PER_CHUNK | CJ_FLAG_FORCE_LDS_PER_CHUNK | CJ_FLAG_BIG_CHUNKS
                                         | CJ_FLAG_DEBUG_PROFILE;

int cj_batch_device(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n_chunks,
                    const uint8_t* in_base, const uint64_t* in_off, const uint64_t* in_len,
                    uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                    int64_t* result, void* hip_stream) {
    if (!e || n_chunks > 0xFFFFFFF0ull || (flags & ~kPublicFlags)) return CJ_E_BAD_ARG;
    if (n_chunks == 0) return 0;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    cj::BatchArgs a;
    fill_args(a, flags, n_chunks, in_base, in_off, in_len, out_base, out_off, out_cap, result);
    return launch(e, codec, op, a, hip_stream ? (hipStream_t)hip_stream : e->stream);
}

int cj_engine_sync(cj_engine* e) {
    if (!e) return CJ_E_BAD_ARG;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(e->stream), CJ_E_NO_DEVICE);
    return 0;
}

int cj_stream_sync(cj_engine* e, void* hip_stream) {
    if (!e) return CJ_E_BAD_ARG;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : e->stream), CJ_E_NO_DEVICE);
    return 0;
}

double cj_batch_device_timed(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n_chunks,
                             const uint8_t* in_base, const uint64_t* in_off, const uint64_t* in_len,
                             uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                             int64_t* result, int reps) {
    if (!e || reps < 1 || n_chunks == 0 || n_chunks > 0xFFFFFFF0ull || (flags & ~kPublicFlags)) return -1.0;
    HIP_TRY(hipSetDevice(e->device), -1.0);
    cj::BatchArgs a;
    fill_args(a, flags, n_chunks, in_base, in_off, in_len, out_base, out_off, out_cap, result);
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0), -1.0);
    HIP_TRY(hipEventCreate(&t1), -1.0);
    HIP_TRY(hipEventRecord(t0, e->stream), -1.0);
    for (int r = 0; r < reps; r++)
        if (launch(e, codec, op, a, e->stream) != 0) return -1.0;
    HIP_TRY(hipEventRecord(t1, e->stream), -1.0);
    HIP_TRY(hipEventSynchronize(t1), -1.0);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t0, t1), -1.0);
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    return (double)ms / reps;
}

int cj_batch_host(cj_engine* e, cj_codec codec, cj_op op, uint32_t flags, size_t n,
                  const uint8_t* const* in_ptrs, const size_t* in_lens,
                  uint8_t* const* out_ptrs, const size_t* out_caps, int64_t* result) {
    if (!e || (n && (!in_ptrs || !in_lens || !out_ptrs || !out_caps || !result)) || (flags & ~kPublicFlags)) return CJ_E_BAD_ARG;
    if (n == 0) return 0;
    if (n > 0xFFFFFFF0ull) return CJ_E_BAD_ARG;
    // decompress: chunks above 64 KiB (by input or by capacity) would each be one serial stream on one wavefront; up to
    // 256 MiB of them take the large-stream path together (large.hip) and the rest of the batch follows as usual
    if (op == CJ_OP_DECOMPRESS && n > 1 && !(flags & (CJ_FLAG_FORCE_WAVE_PER_CHUNK | CJ_FLAG_FORCE_LANE_PER_CHUNK | CJ_FLAG_FORCE_LDS_PER_CHUNK))) {
        std::vector<size_t> big, rest;
        size_t big_bytes = 0;
        for (size_t i = 0; i < n; i++) {
            const bool is_big = in_ptrs[i] && out_ptrs[i] && (in_lens[i] > kLargeMin || (codec == CJ_CODEC_SNAPPY_RAW
                                    ? cj_snappy_raw_decompress_len(in_ptrs[i], in_lens[i]) > (int64_t)kLargeMin
                                    : (out_caps[i] > kLargeMin && in_lens[i] > 4096)));
            if (is_big) { big.push_back(i); big_bytes += in_lens[i]; } else rest.push_back(i);
        }
        if (!big.empty() && big.size() <= 512 && big_bytes <= (256u << 20)) {      // (per-chunk host work: not for thousands of chunks)
            const int rc = cj::large_decompress_listed(e, codec, flags, big.size(), big.data(), in_ptrs, in_lens, out_ptrs, out_caps, result);
            if (rc != 0) return rc;
            if (rest.empty()) return 0;
            std::vector<const uint8_t*> ip(rest.size()); std::vector<size_t> il(rest.size()), oc(rest.size()); std::vector<uint8_t*> opp(rest.size());
            std::vector<int64_t> rr(rest.size());
            for (size_t k = 0; k < rest.size(); k++) { ip[k] = in_ptrs[rest[k]]; il[k] = in_lens[rest[k]]; opp[k] = out_ptrs[rest[k]]; oc[k] = out_caps[rest[k]]; }
            const int rc2 = cj_batch_host(e, codec, op, flags, rest.size(), ip.data(), il.data(), opp.data(), oc.data(), rr.data());
            if (rc2 != 0) return rc2;
            for (size_t k = 0; k < rest.size(); k++) result[rest[k]] = rr[k];
            return 0;
        }
        // many big chunks: the batch's own big-chunk path takes those of up to 256 KiB (big_chunks.hpp) — asked for only if there is one
        bool any_mid = false;
        for (size_t i : big) {
            const int64_t u = codec == CJ_CODEC_SNAPPY_RAW ? cj_snappy_raw_decompress_len(in_ptrs[i], in_lens[i]) : (int64_t)out_caps[i];
            if (u > (int64_t)kLargeMin && u <= (int64_t)(4u * 65536u)) { any_mid = true; break; }
        }
        if (any_mid) flags |= CJ_FLAG_BIG_CHUNKS;
    }
    if (op == CJ_OP_DECOMPRESS && n > 0) {                   // the capacities are on the host here: a batch of small chunks gets windows of its size
        size_t cap_max = 0;
        for (size_t i = 0; i < n; i++) cap_max = std::max(cap_max, out_caps[i]);
        flags &= ~(CJ_FLAG_CHUNKS_LE_32K | CJ_FLAG_CHUNKS_LE_16K);
        if (cap_max <= 16384) flags |= CJ_FLAG_CHUNKS_LE_16K; else if (cap_max <= 32768) flags |= CJ_FLAG_CHUNKS_LE_32K;
    }
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);

    // meta layout (u64 each, n entries per row): in_off | in_len | out_off | out_cap | result
    std::vector<uint64_t>& m = e->h_meta;
    m.assign(5 * n, 0);
    uint64_t in_total = 0, out_total = 0;
    for (size_t i = 0; i < n; i++) {
        m[i] = in_total;
        m[n + i] = in_lens[i];
        in_total += (in_lens[i] + 15u) & ~(uint64_t)15u;
        // LZ4 compress: the kernel wants a full LZ4_compressBound of room; give it that on the device and
        // apply the caller's capacity when copying back (fits -> ok, else "Compression failed").
        uint64_t dcap = out_caps[i];
        if (codec == CJ_CODEC_LZ4_BLOCK && op == CJ_OP_COMPRESS) {
            uint64_t b = cj_lz4_block_compress_bound(in_lens[i], (flags & CJ_FLAG_LZ4_SIZE_PREFIX) ? 1 : 0);
            if (b > dcap) dcap = b;
        }
        m[2 * n + i] = out_total;
        m[3 * n + i] = dcap;
        out_total += (dcap + 15u) & ~(uint64_t)15u;
    }
    if (!e->d_in.reserve(in_total + 16) || !e->d_out.reserve(out_total + 16) || !e->d_meta.reserve(5 * n * 8)) return CJ_E_OOM;

    // (a large batch: sliced, so that packing, the two directions of the link, the kernels and the scattering overlap)
    if (n >= 512 && in_total + out_total >= (128ull << 20))
        return batch_host_sliced(e, codec, op, flags, n, in_ptrs, in_lens, out_ptrs, out_caps, result, in_total, out_total);

    uint8_t* d_in = (uint8_t*)e->d_in.p;
    uint8_t* d_out = (uint8_t*)e->d_out.p;
    uint64_t* d_meta = (uint64_t*)e->d_meta.p;
    if (n == 1) {
        if (in_lens[0]) HIP_TRY(hipMemcpyAsync(d_in, in_ptrs[0], in_lens[0], hipMemcpyHostToDevice, e->stream), CJ_E_NO_DEVICE);
    } else {
        if (!e->h_in.reserve(in_total)) return CJ_E_OOM;
        parallel_chunks(n, in_total, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++)
                if (in_lens[i]) std::memcpy(e->h_in.p + m[i], in_ptrs[i], in_lens[i]);
        });
        if (in_total) HIP_TRY(hipMemcpyAsync(d_in, e->h_in.p, in_total, hipMemcpyHostToDevice, e->stream), CJ_E_NO_DEVICE);
    }
    HIP_TRY(hipMemcpyAsync(d_meta, m.data(), 4 * n * 8, hipMemcpyHostToDevice, e->stream), CJ_E_NO_DEVICE);

    cj::BatchArgs a;
    fill_args(a, flags, n, d_in, d_meta, d_meta + n, d_out, d_meta + 2 * n, d_meta + 3 * n, (int64_t*)(d_meta + 4 * n));
    int rc = launch(e, codec, op, a, e->stream);
    if (rc != 0) return rc;
    HIP_TRY(hipMemcpyAsync(result, d_meta + 4 * n, n * 8, hipMemcpyDeviceToHost, e->stream), CJ_E_NO_DEVICE);
    HIP_TRY(hipStreamSynchronize(e->stream), CJ_E_NO_DEVICE);

    if (n == 1) {
        if (result[0] > 0) {
            if ((uint64_t)result[0] > out_caps[0]) result[0] = CJ_E_COMPRESS_FAILED;
            else HIP_TRY(hipMemcpy(out_ptrs[0], d_out, (size_t)result[0], hipMemcpyDeviceToHost), CJ_E_NO_DEVICE);
        }
        return 0;
    }
    // copy back only the span that was produced
    uint64_t span = 0;
    for (size_t i = 0; i < n; i++)
        if (result[i] > 0 && m[2 * n + i] + (uint64_t)result[i] > span) span = m[2 * n + i] + (uint64_t)result[i];
    if (!e->h_out.reserve(span)) return CJ_E_OOM;
    if (span) HIP_TRY(hipMemcpy(e->h_out.p, d_out, span, hipMemcpyDeviceToHost), CJ_E_NO_DEVICE);
    parallel_chunks(n, span, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            if (result[i] <= 0) continue;
            if ((uint64_t)result[i] > out_caps[i]) { result[i] = CJ_E_COMPRESS_FAILED; continue; }
            std::memcpy(out_ptrs[i], e->h_out.p + m[2 * n + i], (size_t)result[i]);
        }
    });
    return 0;
}

int64_t cj_lz4_block_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int level, int accel, int prepend) {
    (void)level; (void)accel;   // libcramjam always runs LZ4's DEFAULT mode (see header)
    const bool pre = prepend != 0;   // -1 (None) and 1 -> prefix
    if (n > 0x7FFFFFFFull || cj_lz4_block_compress_bound(n, 0) == 0) return CJ_E_INPUT_TOO_LARGE;
    if (pre && cap < 4) return CJ_E_COMPRESS_FAILED;
    if (n > kLargeMinCompress && in && out) return cj::large_lz4_compress(in, n, out, cap, pre);      // pieces compressed as a batch, stitched into one block
    return single(CJ_CODEC_LZ4_BLOCK, CJ_OP_COMPRESS, pre ? CJ_FLAG_LZ4_SIZE_PREFIX : 0u, in, n, out, cap);
}

int64_t cj_lz4_block_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int size_prepended) {
    // one large stream (by its input, or by the output size it announces): parsed and decoded slab-parallel (large.hip)
    uint64_t announced = cap;
    if (size_prepended && in && n >= 4) {
        uint32_t u; std::memcpy(&u, in, 4);
        if ((int32_t)u >= 0 && u <= cap) {
            announced = u;
            // an LZ4 length byte stands for at most 255 bytes: a block cannot decode to more than 255 n + 64 — a prefix that
            // announces more belongs to a stream whose decode fails; answer without staging gigabytes for it
            if (announced > 255ull * (uint64_t)n + 64ull) return CJ_E_CORRUPT;
        }
    }
    if ((n > kLargeMin || announced > kLargeMin) && in && out && !cj::large_few_elements(CJ_CODEC_LZ4_BLOCK, size_prepended ? CJ_FLAG_LZ4_SIZE_PREFIX : 0u, in, n, cap))
        return cj::large_decompress(CJ_CODEC_LZ4_BLOCK, size_prepended ? CJ_FLAG_LZ4_SIZE_PREFIX : 0u, in, n, out, cap);
    return single(CJ_CODEC_LZ4_BLOCK, CJ_OP_DECOMPRESS, size_prepended ? CJ_FLAG_LZ4_SIZE_PREFIX : 0u, in, n, out, cap);
}

int64_t cj_snappy_raw_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if (n > kLargeMinCompress && in && out) return cj::large_snappy_compress(in, n, out, cap);
    return single(CJ_CODEC_SNAPPY_RAW, CJ_OP_COMPRESS, 0u, in, n, out, cap);
}

int64_t cj_snappy_raw_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    // a 3-byte copy element stands for at most 64 bytes: n bytes cannot decode to more than 22 n + 64 — a preamble that
    // announces more (and fits the caller's buffer) belongs to a stream whose decode fails
    if (in && n > 0) {
        const int64_t dn = cj_snappy_raw_decompress_len(in, n);
        if (dn > 0 && (uint64_t)dn <= cap && (uint64_t)dn > 22ull * (uint64_t)n + 64ull) return CJ_E_SNAPPY_CORRUPT;
    }
    if (in && out && (n > kLargeMin || (n > 0 && cj_snappy_raw_decompress_len(in, n) > (int64_t)kLargeMin)) && !cj::large_few_elements(CJ_CODEC_SNAPPY_RAW, 0u, in, n, cap))
        return cj::large_decompress(CJ_CODEC_SNAPPY_RAW, 0u, in, n, out, cap);
    return single(CJ_CODEC_SNAPPY_RAW, CJ_OP_DECOMPRESS, 0u, in, n, out, cap);
}

void* cj_device_alloc(cj_engine* e, size_t bytes) {
    if (!e) return nullptr;
    if (!hip_ok(hipSetDevice(e->device), "hipSetDevice")) return nullptr;
    void* p = nullptr;
    if (!hip_ok(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc")) return nullptr;
    return p;
}

void cj_device_free(cj_engine* e, void* p) {
    if (!e || !p) return;
    (void)hipSetDevice(e->device);
    (void)hipFree(p);
}

int cj_memcpy_h2d(cj_engine* e, void* d, const void* s, size_t n) {
    if (!e) return CJ_E_BAD_ARG;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    if (n) HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice), CJ_E_NO_DEVICE);
    return 0;
}

int cj_memcpy_d2h(cj_engine* e, void* d, const void* s, size_t n) {
    if (!e) return CJ_E_BAD_ARG;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    if (n) HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost), CJ_E_NO_DEVICE);
    return 0;
}

int cj_memcpy_d2d(cj_engine* e, void* d, const void* s, size_t n) {
    if (!e) return CJ_E_BAD_ARG;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    if (n) HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice), CJ_E_NO_DEVICE);
    return 0;
}

int cj_memset_dev(cj_engine* e, void* d, int v, size_t n) {
    if (!e) return CJ_E_BAD_ARG;
    HIP_TRY(hipSetDevice(e->device), CJ_E_NO_DEVICE);
    if (n) HIP_TRY(hipMemset(d, v, n), CJ_E_NO_DEVICE);
    return 0;
}

}  // extern "C"

extern "C" uint64_t cj_debug_big_scratch_bytes(cj_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lock(e->scratch_mu);
    return (uint64_t)e->d_biglist.cap + e->d_bigrecs.cap + e->d_bigmisc.cap + e->d_bigslabtab.cap;
}
