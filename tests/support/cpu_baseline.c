/*
 * tests/support/cpu_baseline.c -- TEST / BENCH INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg).
 *
 * The reference's CPU path for one entry, driven from C the way crates/zarc drives it:
 *
 *   pack    one ZSTD_CCtx per Encoder (crates/zarc/src/encode.rs:60-62), sticky parameters checksumFlag = 1 and
 *           compressionLevel (crates/zarc-cli/src/pack.rs:227-232); per entry blake3::hash (encode/content_frame.rs:26), then
 *           ZSTD_CCtx_reset(session_only) + ZSTD_compress2 into a buffer of len + max(1024, len / 10)
 *           (content_frame.rs:37-41, lowlevel_frames.rs:21-31)
 *   unpack  a fresh ZSTD_DCtx per frame (decode/zstd_iterator.rs:28-29), ZSTD_decompressStream fed input slabs of
 *           ZSTD_DStreamInSize() = 131 075 bytes and output steps of ZSTD_DStreamOutSize() = 131 072 bytes
 *           (zstd_iterator.rs:88-153), every chunk into an incremental BLAKE3 (decode/frame_iterator.rs:94-103), digest compared
 *
 * libzstd is the reference's own codec (zstd-sys 2.0.9+zstd.1.5.5, Cargo.lock:2480-2481) at whatever pin the box has: it is
 * dlopen'ed by path and its version reported.  BLAKE3 is the oracle's portable port (the blake3 crate's AVX-512 code is faster).
 * The reference is single-threaded; `threads` > 1 gives every thread its own context and every T-th entry: the embarrassingly
 * parallel bound of the same work on the host's cores.  Entries are the benchmark corpus (zarc_amd/csrc/corpus.h), generated
 * before the clock starts.
 */
#define _GNU_SOURCE
#include "../../oracle/oracle.h"
#include "../../zarc_amd/csrc/corpus.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { void *dst; size_t size; size_t pos; } zbuf_out;
typedef struct { const void *src; size_t size; size_t pos; } zbuf_in;

typedef struct {
    void *(*createCCtx)(void);
    size_t (*freeCCtx)(void *);
    size_t (*setParameter)(void *, int, int);
    size_t (*reset)(void *, int);
    size_t (*compress2)(void *, void *, size_t, const void *, size_t);
    void *(*createDCtx)(void);
    size_t (*freeDCtx)(void *);
    size_t (*decompressStream)(void *, zbuf_out *, zbuf_in *);
    size_t (*inSize)(void);
    size_t (*outSize)(void);
    unsigned (*isError)(size_t);
    const char *(*versionString)(void);
} zapi;

typedef struct {
    const zapi *z;
    int level, tid, threads, fail;
    size_t n, entry_bytes;
    uint8_t **raw;      /* n entries */
    uint8_t **frame;    /* n frames (allocated by the pack pass)  */
    size_t *frame_len;
    uint8_t (*digest)[32];
    int phase;          /* 0 pack, 1 unpack */
    double hash_s;      /* seconds this thread spent inside BLAKE3 (the oracle's portable port: the blake3 crate's SIMD code is several times
                           faster, so the codec's share is reported beside the total) */
} job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg)
{
    job *j = (job *)arg;
    const zapi *z = j->z;
    size_t i;
    if (j->phase == 0) {
        void *c = z->createCCtx();                                  /* encode.rs:60-62 */
        if (!c) { j->fail = 1; return NULL; }
        z->setParameter(c, 201 /* checksumFlag */, 1);              /* pack.rs:227 */
        z->setParameter(c, 100 /* compressionLevel */, j->level);   /* pack.rs:229-232 */
        for (i = (size_t)j->tid; i < j->n; i += (size_t)j->threads) {
            const size_t len = j->entry_bytes, cap = len + (len / 10 > 1024 ? len / 10 : 1024); /* lowlevel_frames.rs:21 */
            size_t r;
            { const double h0 = now_s(); oracle_blake3(j->raw[i], len, j->digest[i]); j->hash_s += now_s() - h0; } /* content_frame.rs:26 */
            z->reset(c, 1 /* ZSTD_reset_session_only */);           /* content_frame.rs:37-39 */
            /* (the frame buffers are allocated and touched before the clock starts, cpu_baseline_run: a malloc per frame inside the timed loop
             * is a run of page faults under one lock when hundreds of threads do it at once) */
            r = z->compress2(c, j->frame[i], cap, j->raw[i], len);  /* lowlevel_frames.rs:29-31 */
            if (z->isError(r)) { j->fail = 1; break; }
            j->frame_len[i] = r;
        }
        z->freeCCtx(c);
    } else {
        const size_t in_step = z->inSize(), out_step = z->outSize();
        uint8_t *out = (uint8_t *)malloc(out_step);
        for (i = (size_t)j->tid; i < j->n; i += (size_t)j->threads) {
            void *d = z->createDCtx();                              /* zstd_iterator.rs:28-29: one per frame */
            oracle_blake3_hasher h;
            uint8_t dig[32];
            size_t fed = 0, produced = 0, hint = 1;
            if (!d) { j->fail = 1; break; }
            oracle_blake3_init(&h);
            while (hint != 0 && fed < j->frame_len[i]) {            /* zstd_iterator.rs:88-153 */
                zbuf_in in;
                in.src = j->frame[i] + fed;
                in.size = j->frame_len[i] - fed < in_step ? j->frame_len[i] - fed : in_step;
                in.pos = 0;
                while (in.pos < in.size || hint != 0) {
                    zbuf_out ob;
                    ob.dst = out; ob.size = out_step; ob.pos = 0;
                    hint = z->decompressStream(d, &ob, &in);
                    if (z->isError(hint)) { j->fail = 1; hint = 0; break; }
                    { const double h0 = now_s(); oracle_blake3_update(&h, out, ob.pos); j->hash_s += now_s() - h0; } /* frame_iterator.rs:99 */
                    produced += ob.pos;
                    if (hint == 0 || (ob.pos < ob.size && in.pos == in.size)) break;
                }
                fed += in.pos;
            }
            oracle_blake3_finalize(&h, dig);
            if (produced != j->entry_bytes || memcmp(dig, j->digest[i], 32) != 0) j->fail = 1;  /* frame_iterator.rs:75-88 */
            z->freeDCtx(d);
        }
        free(out);
    }
    return NULL;
}

typedef struct { job *j; int tid, threads, kind, fail; uint64_t first_index; } prep;
static void *prepare(void *arg)
{
    prep *p = (prep *)arg;
    job *j = p->j;
    size_t i;
    const size_t cap = j->entry_bytes + (j->entry_bytes / 10 > 1024 ? j->entry_bytes / 10 : 1024);
    for (i = (size_t)p->tid; i < j->n; i += (size_t)p->threads) {
        j->raw[i] = (uint8_t *)malloc(j->entry_bytes + 16);
        j->frame[i] = (uint8_t *)malloc(cap);
        if (!j->raw[i] || !j->frame[i]) { p->fail = 1; break; }
        zarc_corpus_entry(j->raw[i], j->entry_bytes, p->first_index + i, p->kind);
        memset(j->frame[i], 0, cap);
    }
    return NULL;
}

static int run_phase(job *proto, int threads, int phase, double *seconds, double *hash_seconds)
{
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    job *jobs = (job *)calloc((size_t)threads, sizeof *jobs);
    int t, fail = 0;
    const double t0 = now_s();
    for (t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].tid = t; jobs[t].threads = threads; jobs[t].phase = phase; jobs[t].fail = 0; jobs[t].hash_s = 0;
        if (pthread_create(&th[t], NULL, worker, &jobs[t]) != 0) { jobs[t].fail = 1; th[t] = 0; }
    }
    *hash_seconds = 0;
    for (t = 0; t < threads; t++) { if (th[t]) pthread_join(th[t], NULL); fail |= jobs[t].fail; if (jobs[t].hash_s > *hash_seconds) *hash_seconds = jobs[t].hash_s; }
    *seconds = now_s() - t0;
    free(th); free(jobs);
    return fail;
}

/* Returns 0 on success.  info receives "libzstd <version>; <cpu model>; <online cores> cores". */
int cpu_baseline_run(const char *libzstd_path, int level, int threads, size_t n, size_t entry_bytes, uint64_t first_index, int kind,
                     double *pack_seconds, double *unpack_seconds, uint64_t *compressed_bytes, char *info, size_t info_cap,
                     double *pack_hash_seconds, double *unpack_hash_seconds /* the slowest thread's time inside BLAKE3, part of the totals */)
{
    zapi z;
    job j;
    size_t i;
    int rc = 0;
    void *lib = dlopen(libzstd_path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return -1;
#define SYM(field, name) do { *(void **)(&z.field) = dlsym(lib, name); if (!z.field) { dlclose(lib); return -2; } } while (0)
    SYM(createCCtx, "ZSTD_createCCtx"); SYM(freeCCtx, "ZSTD_freeCCtx"); SYM(setParameter, "ZSTD_CCtx_setParameter");
    SYM(reset, "ZSTD_CCtx_reset"); SYM(compress2, "ZSTD_compress2"); SYM(createDCtx, "ZSTD_createDCtx"); SYM(freeDCtx, "ZSTD_freeDCtx");
    SYM(decompressStream, "ZSTD_decompressStream"); SYM(inSize, "ZSTD_DStreamInSize"); SYM(outSize, "ZSTD_DStreamOutSize");
    SYM(isError, "ZSTD_isError"); SYM(versionString, "ZSTD_versionString");
#undef SYM
    memset(&j, 0, sizeof j);
    j.z = &z; j.level = level; j.n = n; j.entry_bytes = entry_bytes;
    j.raw = (uint8_t **)calloc(n, sizeof *j.raw);
    j.frame = (uint8_t **)calloc(n, sizeof *j.frame);
    j.frame_len = (size_t *)calloc(n, sizeof *j.frame_len);
    j.digest = (uint8_t(*)[32])calloc(n, 32);
    if (threads < 1) threads = 1;
    { /* inputs and frame buffers (lowlevel_frames.rs:21 sizes them) are made and touched before the clock starts, on all threads */
        pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
        prep *pj = (prep *)calloc((size_t)threads, sizeof *pj);
        int t;
        for (t = 0; t < threads; t++) {
            pj[t].j = &j; pj[t].tid = t; pj[t].threads = threads; pj[t].first_index = first_index; pj[t].kind = kind; pj[t].fail = 0;
            if (pthread_create(&th[t], NULL, prepare, &pj[t]) != 0) { th[t] = 0; prepare(&pj[t]); }
        }
        for (t = 0; t < threads; t++) { if (th[t]) pthread_join(th[t], NULL); rc |= pj[t].fail; }
        free(th); free(pj);
    }
    *pack_hash_seconds = *unpack_hash_seconds = 0;
    if (!rc) rc = run_phase(&j, threads, 0, pack_seconds, pack_hash_seconds);
    if (!rc) rc = run_phase(&j, threads, 1, unpack_seconds, unpack_hash_seconds);
    *compressed_bytes = 0;
    for (i = 0; i < n; i++) *compressed_bytes += j.frame_len[i];
    if (info && info_cap) {
        char model[160] = "unknown cpu";
        FILE *f = fopen("/proc/cpuinfo", "r");
        if (f) {
            char line[256];
            while (fgets(line, sizeof line, f))
                if (strncmp(line, "model name", 10) == 0) {
                    char *c = strchr(line, ':');
                    if (c) { size_t l; c += 2; l = strlen(c); if (l && c[l - 1] == '\n') c[l - 1] = 0; snprintf(model, sizeof model, "%s", c); }
                    break;
                }
            fclose(f);
        }
        snprintf(info, info_cap, "libzstd %s (in %zu / out %zu byte steps); %s", z.versionString(), z.inSize(), z.outSize(), model);
    }
    for (i = 0; i < n; i++) { free(j.raw[i]); free(j.frame[i]); }
    free(j.raw); free(j.frame); free(j.frame_len); free(j.digest);
    dlclose(lib);
    return rc ? -3 : 0;
}
