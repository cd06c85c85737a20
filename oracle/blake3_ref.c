/*
 * oracle/blake3_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Portable scalar restatement of BLAKE3 (default hash mode, 32-byte digest) following the public
 * BLAKE3 specification (section 2: compression function, chunk chaining, binary tree, ROOT flag).
 * It stands in for the `blake3` crate 1.5.0 that the reference calls at
 *   crates/zarc/src/encode/content_frame.rs:26   (blake3::hash)
 *   crates/zarc/src/integrity.rs:107-117         (verify_data)
 *   crates/zarc/src/decode/frame_iterator.rs:54,77,99 (Hasher::new/finalize/update)
 * Pinned by the published known-answer vectors in tests/golden/blake3_kat.json.
 */
#include "oracle.h"
#include <string.h>

enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                               0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t MSG_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

static uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void g(uint32_t *v, int a, int b, int c, int d, uint32_t x, uint32_t y)
{
    v[a] = v[a] + v[b] + x;
    v[d] = rotr32(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];
    v[b] = rotr32(v[b] ^ v[c], 12);
    v[a] = v[a] + v[b] + y;
    v[d] = rotr32(v[d] ^ v[a], 8);
    v[c] = v[c] + v[d];
    v[b] = rotr32(v[b] ^ v[c], 7);
}

/* out[0..7] = new chaining value (first 8 words of the 16-word output) */
static void compress(const uint32_t cv[8], const uint32_t block_words[16], uint64_t counter,
                     uint32_t block_len, uint32_t flags, uint32_t out[8])
{
    uint32_t v[16], m[16], t[16];
    int r, i;
    for (i = 0; i < 8; i++) v[i] = cv[i];
    v[8] = IV[0]; v[9] = IV[1]; v[10] = IV[2]; v[11] = IV[3];
    v[12] = (uint32_t)counter;
    v[13] = (uint32_t)(counter >> 32);
    v[14] = block_len;
    v[15] = flags;
    memcpy(m, block_words, sizeof m);
    for (r = 0; r < 7; r++) {
        g(v, 0, 4, 8, 12, m[0], m[1]);
        g(v, 1, 5, 9, 13, m[2], m[3]);
        g(v, 2, 6, 10, 14, m[4], m[5]);
        g(v, 3, 7, 11, 15, m[6], m[7]);
        g(v, 0, 5, 10, 15, m[8], m[9]);
        g(v, 1, 6, 11, 12, m[10], m[11]);
        g(v, 2, 7, 8, 13, m[12], m[13]);
        g(v, 3, 4, 9, 14, m[14], m[15]);
        if (r != 6) {
            for (i = 0; i < 16; i++) t[i] = m[MSG_PERM[i]];
            memcpy(m, t, sizeof m);
        }
    }
    for (i = 0; i < 8; i++) out[i] = v[i] ^ v[i + 8];
}

static void load_block(const uint8_t *p, int len, uint32_t w[16])
{
    uint8_t buf[64];
    int i;
    memset(buf, 0, 64);
    memcpy(buf, p, (size_t)len);
    for (i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
               ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}

void oracle_blake3_init(oracle_blake3_hasher *h)
{
    memset(h, 0, sizeof *h);
    memcpy(h->chunk_cv, IV, sizeof IV);
}

static void parent_cv(const uint32_t l[8], const uint32_t r[8], uint32_t flags, uint32_t out[8])
{
    uint32_t w[16];
    memcpy(w, l, 32);
    memcpy(w + 8, r, 32);
    compress(IV, w, 0, 64, PARENT | flags, out);
}

/* push a finished chunk CV; merge completed subtrees (one merge per trailing zero bit of the
 * number of chunks finished so far) */
static void push_chunk_cv(oracle_blake3_hasher *h, const uint32_t cv[8], uint64_t total_chunks)
{
    uint32_t cur[8];
    memcpy(cur, cv, 32);
    while ((total_chunks & 1) == 0) {
        uint32_t merged[8];
        h->cv_stack_len--;
        parent_cv(&h->cv_stack[h->cv_stack_len * 8], cur, 0, merged);
        memcpy(cur, merged, 32);
        total_chunks >>= 1;
    }
    memcpy(&h->cv_stack[h->cv_stack_len * 8], cur, 32);
    h->cv_stack_len++;
}

void oracle_blake3_update(oracle_blake3_hasher *h, const void *data, size_t len)
{
    const uint8_t *p = (const uint8_t *)data;
    while (len > 0) {
        /* the buffered block is only compressed once we know more input follows it */
        if (h->block_len == 64) {
            uint32_t w[16], out[8];
            if (h->blocks_done == 15) {
                /* last block of a full chunk, and more data follows: finish the chunk */
                load_block(h->block, 64, w);
                compress(h->chunk_cv, w, h->chunk_counter, 64, CHUNK_END, out);
                push_chunk_cv(h, out, h->chunk_counter + 1);
                h->chunk_counter++;
                memcpy(h->chunk_cv, IV, sizeof IV);
                h->blocks_done = 0;
            } else {
                load_block(h->block, 64, w);
                compress(h->chunk_cv, w, h->chunk_counter, 64,
                         h->blocks_done == 0 ? CHUNK_START : 0, out);
                memcpy(h->chunk_cv, out, 32);
                h->blocks_done++;
            }
            h->block_len = 0;
        }
        {
            size_t take = 64 - (size_t)h->block_len;
            if (take > len) take = len;
            memcpy(h->block + h->block_len, p, take);
            h->block_len += (int)take;
            p += take;
            len -= take;
        }
    }
}

void oracle_blake3_finalize(const oracle_blake3_hasher *h, uint8_t out[32])
{
    uint32_t w[16], cur[8];
    uint32_t flags = CHUNK_END | (h->blocks_done == 0 ? CHUNK_START : 0);
    int i, sp = h->cv_stack_len;
    load_block(h->block, h->block_len, w);
    if (sp == 0) {
        compress(h->chunk_cv, w, h->chunk_counter, (uint32_t)h->block_len, flags | ROOT, cur);
    } else {
        compress(h->chunk_cv, w, h->chunk_counter, (uint32_t)h->block_len, flags, cur);
        while (sp > 0) {
            uint32_t merged[8];
            sp--;
            parent_cv(&h->cv_stack[sp * 8], cur, sp == 0 ? ROOT : 0, merged);
            memcpy(cur, merged, 32);
        }
    }
    for (i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)cur[i];
        out[4 * i + 1] = (uint8_t)(cur[i] >> 8);
        out[4 * i + 2] = (uint8_t)(cur[i] >> 16);
        out[4 * i + 3] = (uint8_t)(cur[i] >> 24);
    }
}

/* ---- eight chunks side by side (GCC vector extensions; an AVX2 clone is picked at run time where the CPU has it) ----
 * Only used by the one-shot entry point, for whole 1 KiB chunks that are followed by more input: same chaining values as the
 * scalar path, pushed onto the same stack.  It exists so that the CPU baseline of bench.py is not dominated by a scalar digest
 * (the blake3 crate the reference calls is SIMD code); the known-answer tests cover it through oracle_blake3(). */
typedef uint32_t v8u __attribute__((vector_size(32)));

#define ROTR8(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
#define G8(a, b, c, d, x, y)                                     \
    do {                                                         \
        v[a] = v[a] + v[b] + (x); v[d] = ROTR8(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];       v[b] = ROTR8(v[b] ^ v[c], 12); \
        v[a] = v[a] + v[b] + (y); v[d] = ROTR8(v[d] ^ v[a], 8);  \
        v[c] = v[c] + v[d];       v[b] = ROTR8(v[b] ^ v[c], 7);  \
    } while (0)

__attribute__((target_clones("avx2", "default")))
void oracle_blake3_chunks8(const uint8_t *p, uint64_t counter0, uint32_t out[8][8])
{
    uint8_t sched[7][16];
    v8u cv[8];
    int r, i, j, blk;
    for (i = 0; i < 16; i++) sched[0][i] = (uint8_t)i;
    for (r = 1; r < 7; r++) for (i = 0; i < 16; i++) sched[r][i] = sched[r - 1][MSG_PERM[i]];
    for (i = 0; i < 8; i++) for (j = 0; j < 8; j++) cv[i][j] = IV[i];
    for (blk = 0; blk < 16; blk++) {
        v8u m[16], v[16];
        const uint32_t flags = (blk == 0 ? CHUNK_START : 0) | (blk == 15 ? CHUNK_END : 0);
        for (i = 0; i < 16; i++)
            for (j = 0; j < 8; j++) {
                const uint8_t *q = p + (size_t)j * 1024 + (size_t)blk * 64 + 4 * (size_t)i;
                m[i][j] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            }
        for (i = 0; i < 8; i++) v[i] = cv[i];
        for (j = 0; j < 8; j++) {
            v[8][j] = IV[0]; v[9][j] = IV[1]; v[10][j] = IV[2]; v[11][j] = IV[3];
            v[12][j] = (uint32_t)(counter0 + (uint64_t)j);
            v[13][j] = (uint32_t)((counter0 + (uint64_t)j) >> 32);
            v[14][j] = 64;
            v[15][j] = flags;
        }
        for (r = 0; r < 7; r++) {
            const uint8_t *sc = sched[r];
            G8(0, 4, 8, 12, m[sc[0]], m[sc[1]]);
            G8(1, 5, 9, 13, m[sc[2]], m[sc[3]]);
            G8(2, 6, 10, 14, m[sc[4]], m[sc[5]]);
            G8(3, 7, 11, 15, m[sc[6]], m[sc[7]]);
            G8(0, 5, 10, 15, m[sc[8]], m[sc[9]]);
            G8(1, 6, 11, 12, m[sc[10]], m[sc[11]]);
            G8(2, 7, 8, 13, m[sc[12]], m[sc[13]]);
            G8(3, 4, 9, 14, m[sc[14]], m[sc[15]]);
        }
        for (i = 0; i < 8; i++) cv[i] = v[i] ^ v[i + 8];
    }
    for (j = 0; j < 8; j++) for (i = 0; i < 8; i++) out[j][i] = cv[i][j];
}

void oracle_blake3(const void *data, size_t len, uint8_t out[32])
{
    oracle_blake3_hasher h;
    const uint8_t *p = (const uint8_t *)data;
    oracle_blake3_init(&h);
    while (len > 8 * 1024) { /* eight whole chunks with input left behind them: none of them is the last chunk */
        uint32_t cvs[8][8];
        int j;
        oracle_blake3_chunks8(p, h.chunk_counter, cvs);
        for (j = 0; j < 8; j++) { push_chunk_cv(&h, cvs[j], h.chunk_counter + 1); h.chunk_counter++; }
        p += 8 * 1024;
        len -= 8 * 1024;
    }
    oracle_blake3_update(&h, p, len);
    oracle_blake3_finalize(&h, out);
}
