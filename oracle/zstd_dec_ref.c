/*
 * oracle/zstd_dec_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * A complete Zstandard frame decoder restated from RFC 8878 / zstd_compression_format.md.
 * It stands in for libzstd 1.5.5's ZSTD_decompressStream, which the reference drives at
 *   crates/zarc/src/decode/zstd_iterator.rs:88-153 (decompress_step)
 * and defines what "valid Zstandard" means for frames produced by the engine's encoder
 * (crates/zarc/src/encode/lowlevel_frames.rs:19-39 is the call site being replaced).
 * Wire layouts follow crates/ozarc/src/framing.rs:106-405.
 * Pinned by decoding frames produced by real libzstd builds (tests/golden/zstd_frames).
 */
#include "oracle.h"
/* optional sequence trace (tools/seqdiff.py: which sequence two encoders first disagree on) */
void (*oracle_zstd_trace)(void *ctx, uint64_t pos, uint32_t ll, uint32_t ml, uint32_t offset) = 0;
void *oracle_zstd_trace_ctx = 0;
#include <stdlib.h>
#include <string.h>

#define BLOCK_MAX (128 * 1024)
#define MAX_SEQ_TABLE 512

typedef struct { uint8_t sym, nbits; uint16_t base; } fse_cell;
typedef struct { fse_cell cell[MAX_SEQ_TABLE]; int al; int valid; } fse_table;
typedef struct { uint8_t sym[1 << 11], nbits[1 << 11]; int max_bits; int valid; } huf_table;

/* ---------------------------------------------------------------- forward bit reader -------- */
typedef struct { const uint8_t *p; size_t len; size_t bitpos; int overrun; } fbits;
static uint32_t fb_peek(fbits *b, int n)
{
    uint32_t v = 0;
    int i;
    for (i = 0; i < n; i++) {
        size_t bp = b->bitpos + (size_t)i;
        size_t byte = bp >> 3;
        uint32_t bit = byte < b->len ? (uint32_t)(b->p[byte] >> (bp & 7)) & 1u : 0u;
        v |= bit << i;
    }
    return v;
}
static void fb_skip(fbits *b, int n)
{
    b->bitpos += (size_t)n;
    if (b->bitpos > b->len * 8) b->overrun = 1;
}

/* ---------------------------------------------------------------- backward bit reader ------- */
typedef struct { const uint8_t *p; int64_t bits; } bbits; /* bits = unread bits below cursor */
static int bb_init(bbits *b, const uint8_t *p, size_t len)
{
    uint8_t last;
    int hi;
    if (len == 0) return -1;
    last = p[len - 1];
    if (last == 0) return -1;
    hi = 7;
    while (!((last >> hi) & 1)) hi--;
    b->p = p;
    b->bits = (int64_t)(len - 1) * 8 + hi;
    return 0;
}
/* read n (<=32) bits; bits below position 0 read as zero and drive b->bits negative */
static uint32_t bb_read(bbits *b, int n)
{
    uint64_t v = 0;
    int i;
    int64_t start = b->bits - n;
    for (i = 0; i < n; i++) {
        int64_t bp = start + i;
        if (bp >= 0) v |= (uint64_t)((b->p[bp >> 3] >> (bp & 7)) & 1u) << i;
    }
    b->bits = start;
    return (uint32_t)v;
}

/* ---------------------------------------------------------------- FSE ------------------------ */
static int highbit(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

static int fse_build(fse_table *t, const int16_t *norm, int nsym, int al)
{
    int size = 1 << al, high = size - 1, s, i, pos = 0;
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    uint16_t next[256];
    for (s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            t->cell[high--].sym = (uint8_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint16_t)norm[s];
        }
    }
    for (s = 0; s < nsym; s++) {
        if (norm[s] <= 0) continue;
        for (i = 0; i < norm[s]; i++) {
            t->cell[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return -1;
    for (i = 0; i < size; i++) {
        uint16_t x = next[t->cell[i].sym]++;
        int nb = al - highbit(x);
        t->cell[i].nbits = (uint8_t)nb;
        t->cell[i].base = (uint16_t)(((uint32_t)x << nb) - (uint32_t)size);
    }
    t->al = al;
    t->valid = 1;
    return 0;
}

/* parse an FSE table description; returns bytes consumed or <0 */
static int fse_read_desc(const uint8_t *src, size_t len, int max_al, int max_sym, int16_t *norm,
                         int *nsym_out, int *al_out)
{
    fbits b = {src, len, 0, 0};
    int al, remaining, threshold, nb, sym = 0;
    if (len == 0) return -1;
    al = (int)fb_peek(&b, 4) + 5;
    fb_skip(&b, 4);
    if (al > max_al) return -1;
    remaining = (1 << al) + 1;
    threshold = 1 << al;
    nb = al + 1;
    memset(norm, 0, sizeof(int16_t) * 256);
    while (remaining > 1 && sym <= max_sym) {
        int max = 2 * threshold - 1 - remaining;
        int low = (int)fb_peek(&b, nb - 1), val, count;
        if (low < max) {
            val = low;
            fb_skip(&b, nb - 1);
        } else {
            val = (int)fb_peek(&b, nb);
            if (val >= threshold) val -= max;
            fb_skip(&b, nb);
        }
        count = val - 1;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        if (count == 0) {
            for (;;) {
                int rep = (int)fb_peek(&b, 2);
                fb_skip(&b, 2);
                sym += rep;
                if (rep != 3) break;
            }
            if (sym > max_sym + 1) return -1;
        }
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (b.overrun) return -1;
    }
    if (remaining != 1 || b.overrun || sym > max_sym + 1) return -1;
    *nsym_out = sym;
    *al_out = al;
    return (int)((b.bitpos + 7) >> 3);
}

/* ---------------------------------------------------------------- Huffman -------------------- */
static int huf_build(huf_table *h, const uint8_t *weights, int nweights /*explicit*/)
{
    uint32_t sum = 0, left;
    int i, max_bits, last_w, nsym = nweights + 1, w;
    uint8_t wt[256];
    uint32_t rank_start[16], rank_count[16];
    if (nweights < 1 || nweights > 255) return -1;
    for (i = 0; i < nweights; i++) {
        if (weights[i] > 11) return -1;
        wt[i] = weights[i];
        if (weights[i]) sum += 1u << (weights[i] - 1);
    }
    if (sum == 0) return -1;
    max_bits = highbit(sum) + 1;
    if (max_bits > 11) return -1;
    left = (1u << max_bits) - sum;
    if (left & (left - 1)) return -1; /* must be a power of two */
    last_w = highbit(left) + 1;
    wt[nweights] = (uint8_t)last_w;
    memset(rank_count, 0, sizeof rank_count);
    for (i = 0; i < nsym; i++) rank_count[wt[i]]++;
    {
        uint32_t pos = 0;
        for (w = 1; w <= max_bits; w++) {
            rank_start[w] = pos;
            pos += rank_count[w] << (w - 1);
        }
        if (pos != (1u << max_bits)) return -1;
    }
    for (i = 0; i < nsym; i++) {
        uint32_t len, start, k;
        w = wt[i];
        if (!w) continue;
        len = 1u << (w - 1);
        start = rank_start[w];
        rank_start[w] += len;
        for (k = 0; k < len; k++) {
            h->sym[start + k] = (uint8_t)i;
            h->nbits[start + k] = (uint8_t)(max_bits + 1 - w);
        }
    }
    h->max_bits = max_bits;
    h->valid = 1;
    return 0;
}

/* read Huffman tree description, returns bytes consumed or <0 */
static int huf_read_desc(huf_table *h, const uint8_t *src, size_t len)
{
    uint8_t weights[256];
    int hb, n, i;
    if (len < 1) return -1;
    hb = src[0];
    if (hb >= 128) {
        n = hb - 127;
        if ((size_t)(1 + (n + 1) / 2) > len) return -1;
        for (i = 0; i < n; i++) {
            uint8_t byte = src[1 + i / 2];
            weights[i] = (i & 1) ? (byte & 15) : (byte >> 4);
        }
        if (huf_build(h, weights, n) < 0) return -1;
        return 1 + (n + 1) / 2;
    } else {
        int16_t norm[256];
        int nsym, al, used;
        fse_table t;
        bbits b;
        uint32_t s1, s2;
        if ((size_t)(1 + hb) > len || hb == 0) return -1;
        used = fse_read_desc(src + 1, (size_t)hb, 6, 255, norm, &nsym, &al);
        if (used < 0 || used >= hb) return -1;
        if (fse_build(&t, norm, nsym, al) < 0) return -1;
        if (bb_init(&b, src + 1 + used, (size_t)(hb - used)) < 0) return -1;
        s1 = bb_read(&b, al);
        s2 = bb_read(&b, al);
        if (b.bits < 0) return -1;
        n = 0;
        for (;;) {
            if (n > 253) return -1;
            weights[n++] = t.cell[s1].sym;
            s1 = t.cell[s1].base + bb_read(&b, t.cell[s1].nbits);
            if (b.bits < 0) { weights[n++] = t.cell[s2].sym; break; }
            if (n > 253) return -1;
            weights[n++] = t.cell[s2].sym;
            s2 = t.cell[s2].base + bb_read(&b, t.cell[s2].nbits);
            if (b.bits < 0) { weights[n++] = t.cell[s1].sym; break; }
        }
        if (huf_build(h, weights, n) < 0) return -1;
        return 1 + hb;
    }
}

static int huf_decode_stream(const huf_table *h, const uint8_t *src, size_t len, uint8_t *out,
                             size_t nout)
{
    bbits b;
    size_t i;
    if (bb_init(&b, src, len) < 0) return -1;
    for (i = 0; i < nout; i++) {
        /* peek max_bits (zero-filled below the start), consume the code length */
        int64_t save = b.bits;
        uint32_t idx = bb_read(&b, h->max_bits);
        b.bits = save - h->nbits[idx];
        if (b.bits < 0) return -1;
        out[i] = h->sym[idx];
    }
    return b.bits == 0 ? 0 : -1;
}

/* ---------------------------------------------------------------- sequences tables ----------- */
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                     20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                     4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                    1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
                                     20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                                     35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515,
                                     1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                    2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

typedef struct {
    huf_table huf;
    fse_table ll, of, ml;
    uint32_t rep[3];
    uint8_t *lit; /* BLOCK_MAX + 32 */
} dctx;

/* mode: 0 predefined, 1 rle, 2 fse, 3 repeat; returns bytes consumed or <0 */
static int seq_table(fse_table *t, int mode, const uint8_t *src, size_t len, const int16_t *def,
                     int def_n, int def_al, int max_al, int max_sym)
{
    if (mode == 0) {
        return fse_build(t, def, def_n, def_al) < 0 ? -1 : 0;
    } else if (mode == 1) {
        if (len < 1 || src[0] > max_sym) return -1;
        t->cell[0].sym = src[0];
        t->cell[0].nbits = 0;
        t->cell[0].base = 0;
        t->al = 0;
        t->valid = 1;
        return 1;
    } else if (mode == 2) {
        int16_t norm[256];
        int nsym, al, used = fse_read_desc(src, len, max_al, max_sym, norm, &nsym, &al);
        if (used < 0) return -1;
        if (fse_build(t, norm, nsym, al) < 0) return -1;
        return used;
    }
    return t->valid ? 0 : -1;
}

static int decode_literals(dctx *d, const uint8_t *src, size_t len, size_t *lit_len,
                           size_t *consumed)
{
    int type, sf;
    size_t regen, comp, hdr;
    if (len < 1) return ORACLE_ZSTD_E_CORRUPT;
    type = src[0] & 3;
    sf = (src[0] >> 2) & 3;
    if (type < 2) {
        if (sf == 0 || sf == 2) { regen = src[0] >> 3; hdr = 1; }
        else if (sf == 1) {
            if (len < 2) return ORACLE_ZSTD_E_CORRUPT;
            regen = (src[0] >> 4) | ((size_t)src[1] << 4); hdr = 2;
        } else {
            if (len < 3) return ORACLE_ZSTD_E_CORRUPT;
            regen = (src[0] >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); hdr = 3;
        }
        if (regen > BLOCK_MAX) return ORACLE_ZSTD_E_CORRUPT;
        if (type == 0) {
            if (hdr + regen > len) return ORACLE_ZSTD_E_CORRUPT;
            memcpy(d->lit, src + hdr, regen);
            *consumed = hdr + regen;
        } else {
            if (hdr + 1 > len) return ORACLE_ZSTD_E_CORRUPT;
            memset(d->lit, src[hdr], regen);
            *consumed = hdr + 1;
        }
        *lit_len = regen;
        return 0;
    } else {
        int streams;
        const uint8_t *p;
        size_t rem;
        uint64_t v;
        if (sf <= 1) {
            if (len < 3) return ORACLE_ZSTD_E_CORRUPT;
            v = src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16);
            regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; hdr = 3;
            streams = sf == 0 ? 1 : 4;
        } else if (sf == 2) {
            if (len < 4) return ORACLE_ZSTD_E_CORRUPT;
            v = src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24);
            regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; hdr = 4; streams = 4;
        } else {
            if (len < 5) return ORACLE_ZSTD_E_CORRUPT;
            v = src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) |
                ((uint64_t)src[3] << 24) | ((uint64_t)src[4] << 32);
            regen = (v >> 4) & 0x3FFFF; comp = (v >> 22) & 0x3FFFF; hdr = 5; streams = 4;
        }
        if (regen > BLOCK_MAX || hdr + comp > len) return ORACLE_ZSTD_E_CORRUPT;
        p = src + hdr;
        rem = comp;
        if (type == 2) {
            int used = huf_read_desc(&d->huf, p, rem);
            if (used < 0) return ORACLE_ZSTD_E_CORRUPT;
            p += used;
            rem -= (size_t)used;
        } else if (!d->huf.valid) {
            return ORACLE_ZSTD_E_CORRUPT;
        }
        if (streams == 1) {
            if (huf_decode_stream(&d->huf, p, rem, d->lit, regen) < 0) return ORACLE_ZSTD_E_CORRUPT;
        } else {
            size_t s1, s2, s3, s4, per = (regen + 3) / 4;
            if (rem < 6) return ORACLE_ZSTD_E_CORRUPT;
            s1 = p[0] | ((size_t)p[1] << 8);
            s2 = p[2] | ((size_t)p[3] << 8);
            s3 = p[4] | ((size_t)p[5] << 8);
            if (6 + s1 + s2 + s3 > rem) return ORACLE_ZSTD_E_CORRUPT;
            s4 = rem - 6 - s1 - s2 - s3;
            if (per * 3 > regen) return ORACLE_ZSTD_E_CORRUPT;
            p += 6;
            if (huf_decode_stream(&d->huf, p, s1, d->lit, per) < 0 ||
                huf_decode_stream(&d->huf, p + s1, s2, d->lit + per, per) < 0 ||
                huf_decode_stream(&d->huf, p + s1 + s2, s3, d->lit + 2 * per, per) < 0 ||
                huf_decode_stream(&d->huf, p + s1 + s2 + s3, s4, d->lit + 3 * per, regen - 3 * per) < 0)
                return ORACLE_ZSTD_E_CORRUPT;
        }
        *lit_len = regen;
        *consumed = hdr + comp;
        return 0;
    }
}

static int decode_block(dctx *d, const uint8_t *src, size_t len, uint8_t *dst_base, size_t dst_cap,
                        size_t *dst_pos)
{
    size_t lit_len = 0, used = 0, lit_pos = 0, pos = *dst_pos;
    int rc = decode_literals(d, src, len, &lit_len, &used), nseq, modes, r;
    const uint8_t *p;
    size_t rem;
    bbits b;
    uint32_t sl, so, sm;
    int i;
    if (rc) return rc;
    p = src + used;
    rem = len - used;
    if (rem < 1) return ORACLE_ZSTD_E_CORRUPT;
    if (p[0] < 128) { nseq = p[0]; p += 1; rem -= 1; }
    else if (p[0] < 255) {
        if (rem < 2) return ORACLE_ZSTD_E_CORRUPT;
        nseq = ((p[0] - 128) << 8) + p[1]; p += 2; rem -= 2;
    } else {
        if (rem < 3) return ORACLE_ZSTD_E_CORRUPT;
        nseq = p[1] + (p[2] << 8) + 0x7F00; p += 3; rem -= 3;
    }
    if (nseq == 0) {
        if (rem != 0) return ORACLE_ZSTD_E_CORRUPT;
        if (pos + lit_len > dst_cap) return ORACLE_ZSTD_E_DSTSIZE;
        memcpy(dst_base + pos, d->lit, lit_len);
        *dst_pos = pos + lit_len;
        return 0;
    }
    if (rem < 1) return ORACLE_ZSTD_E_CORRUPT;
    modes = p[0]; p++; rem--;
    if (modes & 3) return ORACLE_ZSTD_E_CORRUPT;
    r = seq_table(&d->ll, modes >> 6, p, rem, LL_DEFAULT, 36, 6, 9, 35);
    if (r < 0) return ORACLE_ZSTD_E_CORRUPT;
    p += r; rem -= (size_t)r;
    r = seq_table(&d->of, (modes >> 4) & 3, p, rem, OF_DEFAULT, 29, 5, 8, 31);
    if (r < 0) return ORACLE_ZSTD_E_CORRUPT;
    p += r; rem -= (size_t)r;
    r = seq_table(&d->ml, (modes >> 2) & 3, p, rem, ML_DEFAULT, 53, 6, 9, 52);
    if (r < 0) return ORACLE_ZSTD_E_CORRUPT;
    p += r; rem -= (size_t)r;
    if (bb_init(&b, p, rem) < 0) return ORACLE_ZSTD_E_CORRUPT;
    sl = bb_read(&b, d->ll.al);
    so = bb_read(&b, d->of.al);
    sm = bb_read(&b, d->ml.al);
    if (b.bits < 0) return ORACLE_ZSTD_E_CORRUPT;
    for (i = 0; i < nseq; i++) {
        int ofc = d->of.cell[so].sym, mlc = d->ml.cell[sm].sym, llc = d->ll.cell[sl].sym;
        uint32_t ofv, ml, ll, offset;
        size_t k;
        if (ofc > 31 || mlc > 52 || llc > 35) return ORACLE_ZSTD_E_CORRUPT;
        ofv = (1u << ofc) + bb_read(&b, ofc);
        ml = ML_BASE[mlc] + bb_read(&b, ML_BITS[mlc]);
        ll = LL_BASE[llc] + bb_read(&b, LL_BITS[llc]);
        if (b.bits < 0) return ORACLE_ZSTD_E_CORRUPT;
        if (ofv > 3) {
            offset = ofv - 3;
            d->rep[2] = d->rep[1]; d->rep[1] = d->rep[0]; d->rep[0] = offset;
        } else {
            uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u);
            if (idx == 0) {
                offset = d->rep[0];
            } else {
                offset = idx == 3 ? d->rep[0] - 1 : d->rep[idx];
                if (offset == 0) return ORACLE_ZSTD_E_CORRUPT;
                if (idx > 1) d->rep[2] = d->rep[1];
                d->rep[1] = d->rep[0];
                d->rep[0] = offset;
            }
        }
        if (oracle_zstd_trace) oracle_zstd_trace(oracle_zstd_trace_ctx, (uint64_t)pos + ll, ll, ml, offset); /* test tooling: where a match starts, its lengths and offset */
        if (lit_pos + ll > lit_len) return ORACLE_ZSTD_E_CORRUPT;
        if (pos + ll + ml > dst_cap) return ORACLE_ZSTD_E_DSTSIZE;
        memcpy(dst_base + pos, d->lit + lit_pos, ll);
        pos += ll;
        lit_pos += ll;
        if (offset > pos) return ORACLE_ZSTD_E_CORRUPT;
        for (k = 0; k < ml; k++) dst_base[pos + k] = dst_base[pos + k - offset];
        pos += ml;
        if (i + 1 < nseq) {
            sl = d->ll.cell[sl].base + bb_read(&b, d->ll.cell[sl].nbits);
            sm = d->ml.cell[sm].base + bb_read(&b, d->ml.cell[sm].nbits);
            so = d->of.cell[so].base + bb_read(&b, d->of.cell[so].nbits);
            if (b.bits < 0) return ORACLE_ZSTD_E_CORRUPT;
        }
    }
    if (b.bits != 0) return ORACLE_ZSTD_E_CORRUPT;
    if (pos + (lit_len - lit_pos) > dst_cap) return ORACLE_ZSTD_E_DSTSIZE;
    memcpy(dst_base + pos, d->lit + lit_pos, lit_len - lit_pos);
    pos += lit_len - lit_pos;
    if (pos - *dst_pos > BLOCK_MAX) return ORACLE_ZSTD_E_CORRUPT;
    *dst_pos = pos;
    return 0;
}

int oracle_zstd_frame_header(const void *src_, size_t len, int64_t *fcs, uint64_t *window,
                             int *has_checksum, int *single_segment)
{
    const uint8_t *src = (const uint8_t *)src_;
    int desc, fcs_flag, ss, did_flag, pos = 5, fcs_bytes, did_bytes;
    uint64_t win = 0, content = 0;
    if (len < 5) return ORACLE_ZSTD_E_TRUNCATED;
    if (!(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD))
        return ORACLE_ZSTD_E_MAGIC;
    desc = src[4];
    fcs_flag = desc >> 6;
    ss = (desc >> 5) & 1;
    did_flag = desc & 3;
    if (desc & 0x08) return ORACLE_ZSTD_E_CORRUPT; /* reserved bit */
    if (!ss) {
        int wd;
        if ((size_t)pos >= len) return ORACLE_ZSTD_E_TRUNCATED;
        wd = src[pos++];
        win = (1ULL << (10 + (wd >> 3)));
        win += (win >> 3) * (uint64_t)(wd & 7);
    }
    did_bytes = did_flag == 3 ? 4 : did_flag;
    if ((size_t)(pos + did_bytes) > len) return ORACLE_ZSTD_E_TRUNCATED;
    {
        uint32_t did = 0;
        int i;
        for (i = 0; i < did_bytes; i++) did |= (uint32_t)src[pos + i] << (8 * i);
        if (did != 0) return ORACLE_ZSTD_E_UNSUPPORTED;
    }
    pos += did_bytes;
    fcs_bytes = fcs_flag == 0 ? (ss ? 1 : 0) : (1 << fcs_flag);
    if ((size_t)(pos + fcs_bytes) > len) return ORACLE_ZSTD_E_TRUNCATED;
    {
        int i;
        for (i = 0; i < fcs_bytes; i++) content |= (uint64_t)src[pos + i] << (8 * i);
        if (fcs_bytes == 2) content += 256;
    }
    pos += fcs_bytes;
    if (ss) win = content;
    if (fcs) *fcs = fcs_bytes ? (int64_t)content : -1;
    if (window) *window = win;
    if (has_checksum) *has_checksum = (desc >> 2) & 1;
    if (single_segment) *single_segment = ss;
    return pos;
}

int oracle_zstd_decode_frame(const void *src_, size_t src_len, void *dst_, size_t dst_cap,
                             size_t *dst_len, size_t *consumed)
{
    const uint8_t *src = (const uint8_t *)src_;
    uint8_t *dst = (uint8_t *)dst_;
    int64_t fcs;
    uint64_t window;
    int has_ck, ss, hl = oracle_zstd_frame_header(src, src_len, &fcs, &window, &has_ck, &ss);
    size_t pos, out = 0;
    dctx *d;
    int rc = 0, last = 0;
    if (hl < 0) return hl;
    d = (dctx *)calloc(1, sizeof *d);
    if (!d) return ORACLE_ZSTD_E_DSTSIZE;
    d->lit = (uint8_t *)malloc(BLOCK_MAX + 64);
    d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;
    pos = (size_t)hl;
    while (!last) {
        uint32_t bh, type, size;
        if (pos + 3 > src_len) { rc = ORACLE_ZSTD_E_TRUNCATED; break; }
        bh = src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        pos += 3;
        last = bh & 1;
        type = (bh >> 1) & 3;
        size = bh >> 3;
        if (type == 0) {
            if (pos + size > src_len) { rc = ORACLE_ZSTD_E_TRUNCATED; break; }
            if (out + size > dst_cap) { rc = ORACLE_ZSTD_E_DSTSIZE; break; }
            if (size > BLOCK_MAX) { rc = ORACLE_ZSTD_E_CORRUPT; break; }
            memcpy(dst + out, src + pos, size);
            out += size;
            pos += size;
        } else if (type == 1) {
            if (pos + 1 > src_len) { rc = ORACLE_ZSTD_E_TRUNCATED; break; }
            if (out + size > dst_cap) { rc = ORACLE_ZSTD_E_DSTSIZE; break; }
            if (size > BLOCK_MAX) { rc = ORACLE_ZSTD_E_CORRUPT; break; }
            memset(dst + out, src[pos], size);
            out += size;
            pos += 1;
        } else if (type == 2) {
            if (pos + size > src_len) { rc = ORACLE_ZSTD_E_TRUNCATED; break; }
            if (size > BLOCK_MAX) { rc = ORACLE_ZSTD_E_CORRUPT; break; }
            rc = decode_block(d, src + pos, size, dst, dst_cap, &out);
            if (rc) break;
            pos += size;
        } else {
            rc = ORACLE_ZSTD_E_CORRUPT;
            break;
        }
    }
    if (!rc && fcs >= 0 && (uint64_t)fcs != out) rc = ORACLE_ZSTD_E_CORRUPT;
    if (!rc && has_ck) {
        if (pos + 4 > src_len) rc = ORACLE_ZSTD_E_TRUNCATED;
        else {
            uint32_t want = src[pos] | ((uint32_t)src[pos + 1] << 8) |
                            ((uint32_t)src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
            if ((uint32_t)oracle_xxh64(dst, out, 0) != want) rc = ORACLE_ZSTD_E_CHECKSUM;
            pos += 4;
        }
    }
    free(d->lit);
    free(d);
    if (!rc) {
        if (dst_len) *dst_len = out;
        if (consumed) *consumed = pos;
    }
    return rc;
}

int oracle_zstd_frame_stats(const void *src_, size_t src_len, int counts[3])
{
    const uint8_t *src = (const uint8_t *)src_;
    int hl = oracle_zstd_frame_header(src, src_len, 0, 0, 0, 0), last = 0;
    size_t pos;
    if (hl < 0) return hl;
    counts[0] = counts[1] = counts[2] = 0;
    pos = (size_t)hl;
    while (!last) {
        uint32_t bh, type, size;
        if (pos + 3 > src_len) return ORACLE_ZSTD_E_TRUNCATED;
        bh = src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        pos += 3;
        last = bh & 1;
        type = (bh >> 1) & 3;
        size = bh >> 3;
        if (type == 3) return ORACLE_ZSTD_E_CORRUPT;
        counts[type]++;
        pos += type == 1 ? 1 : size;
    }
    return 0;
}
