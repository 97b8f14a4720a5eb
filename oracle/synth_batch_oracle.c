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
#include "cj_oracle.h"
#include <pthread.h>
#include <stdatomic.h>
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

typedef struct {
    int op; size_t n; const uint8_t* in_base; const uint64_t* in_off; const uint64_t* in_len;
    uint8_t* out_base; size_t out_stride; int64_t* res; atomic_size_t next;
} job_t;

static void* worker(void* p) {
    job_t* j = (job_t*)p;
    for (;;) {
        size_t i = atomic_fetch_add(&j->next, 16);
        if (i >= j->n) break;
        size_t e = i + 16 < j->n ? i + 16 : j->n;
        for (; i < e; i++) {
            const uint8_t* in = j->in_base + j->in_off[i];
            uint8_t* out = j->out_base + i * j->out_stride;
            size_t n = (size_t)j->in_len[i];
            switch (j->op) {
            case 0: j->res[i] = cjo_lz4_decompress_raw(in, n, out, j->out_stride); break;
            case 1: j->res[i] = cjo_lz4_compress_raw(in, n, out, j->out_stride); break;
            case 2: j->res[i] = cjo_snappy_decompress(in, n, out, j->out_stride); break;
            default: j->res[i] = cjo_snappy_compress(in, n, out, j->out_stride); break;
            }
        }
    }
    return 0;
}

int cjo_batch_run(int op, int threads, size_t n_chunks, const uint8_t* in_base, const uint64_t* in_off,
                  const uint64_t* in_len, uint8_t* out_base, size_t out_stride, int64_t* res) {
    job_t j = { op, n_chunks, in_base, in_off, in_len, out_base, out_stride, res, 0 };
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    for (int t = 1; t < threads; t++) pthread_create(&th[t], 0, worker, &j);
    worker(&j);
    for (int t = 1; t < threads; t++) pthread_join(th[t], 0);
    return 0;
}
