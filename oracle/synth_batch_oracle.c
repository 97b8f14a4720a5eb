/*
 * synth_batch_oracle.c — CPU ORACLE support (test infrastructure only; see cj_oracle.h).
 *  - cjo_synth_v1: the deterministic synthetic chunk generator of SURVEY.md §8(d) ("synth-v1").
 *    The product has its own device-side copy of this generator only inside bench.py's data
 *    preparation (host side, via this file) — it is never part of the codec path.
 *  - cjo_batch_run: pthread pool that runs one oracle codec over a batch of chunks; this is the
 *    `cpu_baseline` leg of bench.py (mirrors what a cramjam user can do today: the reference
 *    releases the GIL around each call, /root/reference/src/lz4.rs:84,126, src/snappy.rs:57,75,
 *    so a thread pool over chunks is the reference's best multi-core configuration).
 */
#define _GNU_SOURCE
#include "cj_oracle.h"
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void cjo_synth_v1(uint8_t* dst, size_t S, uint64_t index, uint64_t seed) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + index;
    size_t pos = 0;
    while (pos < S) {
        uint64_t r = splitmix64(&st);
        size_t lit = 1 + (size_t)(r % 24);
        uint64_t bits = 0; int have = 0;
        for (size_t i = 0; i < lit && pos < S; i++) {
            if (have == 0) { bits = splitmix64(&st); have = 10; }
            dst[pos++] = (uint8_t)(0x20 + (bits & 63));
            bits >>= 6; have--;
        }
        if (pos >= 8 && pos < S) {
            uint64_t r2 = splitmix64(&st);
            uint64_t r3 = splitmix64(&st);
            size_t mlen = 4 + (size_t)(r2 % 29);
            size_t lim = pos < 65535 ? pos : 65535;
            size_t dist = 1 + (size_t)(r3 % lim);
            for (size_t i = 0; i < mlen && pos < S; i++, pos++) dst[pos] = dst[pos - dist];
        }
    }
}

/* liblz4's own decoder, when the host has the library (same C code lineage the reference executes through lz4-sys):
 * resolved once with dlopen, never linked.  op 4 below; -1 for every chunk when it is absent. */
typedef int (*lz4_safe_fn)(const char*, char*, int, int);
static lz4_safe_fn g_lz4_safe;
static int g_lz4_tried;
static lz4_safe_fn liblz4_decoder(void) {
    if (!g_lz4_tried) {
        static const char* names[] = { "liblz4.so.1", "/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1", "liblz4.so" };
        for (unsigned k = 0; k < sizeof names / sizeof names[0] && !g_lz4_safe; k++) {
            void* h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
            if (h) g_lz4_safe = (lz4_safe_fn)dlsym(h, "LZ4_decompress_safe");
        }
        g_lz4_tried = 1;
    }
    return g_lz4_safe;
}
int cjo_have_liblz4(void) { return liblz4_decoder() != 0; }
/* LZ4_compress_default of the same library (op 6 below): the C code the reference's compress_block executes through lz4-sys */
typedef int (*lz4_comp_fn)(const char*, char*, int, int);
static lz4_comp_fn g_lz4_comp;
static lz4_comp_fn liblz4_encoder(void) {
    if (!g_lz4_comp && liblz4_decoder()) {
        static const char* names[] = { "liblz4.so.1", "/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1", "liblz4.so" };
        for (unsigned k = 0; k < sizeof names / sizeof names[0] && !g_lz4_comp; k++) {
            void* h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
            if (h) g_lz4_comp = (lz4_comp_fn)dlsym(h, "LZ4_compress_default");
        }
    }
    return g_lz4_comp;
}

/* libsnappy's decoder through its C API (snappy-c.h: snappy_uncompress), when the host has the library: the C++ code the
 * reference's `snap` crate is a port of.  op 5 below; -1 for every chunk when it is absent. */
typedef int (*snappy_unc_fn)(const char*, size_t, char*, size_t*);
static snappy_unc_fn g_sn_unc;
static int g_sn_tried;
static snappy_unc_fn libsnappy_decoder(void) {
    if (!g_sn_tried) {
        static const char* names[] = { "libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1", "/usr/lib/x86_64-linux-gnu/libsnappy.so.1", "libsnappy.so" };
        for (unsigned k = 0; k < sizeof names / sizeof names[0] && !g_sn_unc; k++) {
            void* h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
            if (h) g_sn_unc = (snappy_unc_fn)dlsym(h, "snappy_uncompress");
        }
        g_sn_tried = 1;
    }
    return g_sn_unc;
}
int cjo_have_libsnappy(void) { return libsnappy_decoder() != 0; }
/* snappy_compress of the same library (op 7 below) */
typedef int (*snappy_comp_fn)(const char*, size_t, char*, size_t*);
static snappy_comp_fn g_sn_comp;
static snappy_comp_fn libsnappy_encoder(void) {
    if (!g_sn_comp && libsnappy_decoder()) {
        static const char* names[] = { "libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1", "/usr/lib/x86_64-linux-gnu/libsnappy.so.1", "libsnappy.so" };
        for (unsigned k = 0; k < sizeof names / sizeof names[0] && !g_sn_comp; k++) {
            void* h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
            if (h) g_sn_comp = (snappy_comp_fn)dlsym(h, "snappy_compress");
        }
    }
    return g_sn_comp;
}

/* A pool that lives for the whole call: `reps` passes over the batch, the threads are created ONCE and meet at a barrier
 * between passes (round 1 created and joined 255 threads per 33 ms pass and reported a tenth of what the cores can do). */
typedef struct {
    int op, reps; size_t n; const uint8_t* in_base; const uint64_t* in_off; const uint64_t* in_len;
    uint8_t* out_base; size_t out_stride; int64_t* res; atomic_size_t* next; pthread_barrier_t* bar; int use_bar;
    atomic_int go;          /* 0: the pool is still being started (workers wait), 1: run */
} job_t;

static void* worker(void* p) {
    job_t* j = (job_t*)p;
    while (atomic_load_explicit(&j->go, memory_order_acquire) == 0) sched_yield();      /* the barrier is sized once every thread exists */
    lz4_safe_fn lz4 = j->op == 4 ? liblz4_decoder() : 0;
    snappy_unc_fn snu = j->op == 5 ? libsnappy_decoder() : 0;
    lz4_comp_fn lzc = j->op == 6 ? liblz4_encoder() : 0;
    snappy_comp_fn snc = j->op == 7 ? libsnappy_encoder() : 0;
    for (int r = 0; r < j->reps; r++) {
        for (;;) {
            size_t i = atomic_fetch_add(&j->next[r], 8);
            if (i >= j->n) break;
            size_t e = i + 8 < j->n ? i + 8 : j->n;
            for (; i < e; i++) {
                const uint8_t* in = j->in_base + j->in_off[i];
                uint8_t* out = j->out_base + i * j->out_stride;
                size_t n = (size_t)j->in_len[i];
                switch (j->op) {
                case 0: j->res[i] = cjo_lz4_decompress_raw(in, n, out, j->out_stride); break;
                case 1: j->res[i] = cjo_lz4_compress_raw(in, n, out, j->out_stride); break;
                case 2: j->res[i] = cjo_snappy_decompress(in, n, out, j->out_stride); break;
                case 3: j->res[i] = cjo_snappy_compress(in, n, out, j->out_stride); break;
                case 5: { size_t on = j->out_stride; j->res[i] = snu && snu((const char*)in, n, (char*)out, &on) == 0 ? (int64_t)on : -1; } break;
                case 6: j->res[i] = lzc ? lzc((const char*)in, (char*)out, (int)n, (int)j->out_stride) : -1; break;
                case 7: { size_t on = j->out_stride; j->res[i] = snc && snc((const char*)in, n, (char*)out, &on) == 0 ? (int64_t)on : -1; } break;
                default: j->res[i] = lz4 ? lz4((const char*)in, (char*)out, (int)n, (int)j->out_stride) : -1; break;
                }
            }
        }
        if (j->use_bar) pthread_barrier_wait(j->bar);      /* pass r is complete before anyone overwrites its outputs */
    }
    return 0;
}

int cjo_batch_run_reps(int op, int threads, int reps, size_t n_chunks, const uint8_t* in_base, const uint64_t* in_off,
                       const uint64_t* in_len, uint8_t* out_base, size_t out_stride, int64_t* res) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    if (reps < 1) reps = 1;
    atomic_size_t* next = (atomic_size_t*)calloc((size_t)reps, sizeof(atomic_size_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    if (!next || !th) { free(next); free(th); return -1; }
    pthread_barrier_t bar;
    job_t j = { op, reps, n_chunks, in_base, in_off, in_len, out_base, out_stride, res, next, &bar, 0, 0 };
    int started = 1;
    for (int t = 1; t < threads; t++) {
        if (pthread_create(&th[t], 0, worker, &j) != 0) break;
        started++;
    }
    /* the workers wait for `go`: the barrier is sized for the threads that actually exist (a pool that could only be started in
     * part still finishes — with fewer threads — and the call reports -2), and nothing of the job changes after a worker has read it */
    pthread_barrier_init(&bar, 0, (unsigned)started);
    j.use_bar = started > 1;
    atomic_store_explicit(&j.go, 1, memory_order_release);
    worker(&j);
    for (int t = 1; t < started; t++) pthread_join(th[t], 0);
    pthread_barrier_destroy(&bar);
    free(next); free(th);
    return started == threads ? 0 : -2;
}

int cjo_batch_run(int op, int threads, size_t n_chunks, const uint8_t* in_base, const uint64_t* in_off,
                  const uint64_t* in_len, uint8_t* out_base, size_t out_stride, int64_t* res) {
    return cjo_batch_run_reps(op, threads, 1, n_chunks, in_base, in_off, in_len, out_base, out_stride, res);
}
