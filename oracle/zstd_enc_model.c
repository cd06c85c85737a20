/*
 * oracle/zstd_enc_model.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar CPU model of the engine's Zstandard *encoder* ("ZGE").  libzstd's exact parse is not part of
 * the contract at the reference's compress call site (crates/zarc/src/encode/lowlevel_frames.rs:29-31:
 * any valid frame that round-trips is acceptable, and the pinned libzstd 1.5.5 is not available here),
 * so the encoder is the engine's own design: a tile-parallel double-hash match finder plus Huffman /
 * FSE entropy stages laid out for wave64 execution.  This file states that design
 * as plain sequential C with every tie-break made explicit, so that
 *   (1) its output is checked for *validity* by oracle_zstd_decode_frame and by real libzstd builds,
 *   (2) the HIP kernels are checked *bit-exactly* against it on the same inputs.
 * The frame/ block / section layouts follow RFC 8878 (and crates/ozarc/src/framing.rs:106-405).
 */
#include "oracle.h"
#include "zge_model.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers ------------------ */
static int hb32(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; } /* floor(log2 v), v>0 */
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

typedef struct { uint8_t *p; size_t pos, cap; uint64_t acc; int nb; int overflow; } bitw;
static void bw_init(bitw *b, uint8_t *p, size_t cap) { b->p = p; b->pos = 0; b->cap = cap; b->acc = 0; b->nb = 0; b->overflow = 0; }
static void bw_add(bitw *b, uint32_t v, int n)
{
    if (n == 0) return;
    b->acc |= (uint64_t)(v & (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1))) << b->nb;
    b->nb += n;
    while (b->nb >= 8) {
        if (b->pos < b->cap) b->p[b->pos] = (uint8_t)b->acc; else b->overflow = 1;
        b->pos++;
        b->acc >>= 8;
        b->nb -= 8;
    }
}
/* end mark: a single 1 bit then zero padding to a byte */
static size_t bw_close(bitw *b)
{
    bw_add(b, 1, 1);
    if (b->nb > 0) {
        if (b->pos < b->cap) b->p[b->pos] = (uint8_t)b->acc; else b->overflow = 1;
        b->pos++;
        b->nb = 0;
        b->acc = 0;
    }
    return b->pos;
}

/* ------------------------------------------------------------------ code tables -------------- */
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
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

uint32_t zge_ll_code(uint32_t ll)
{
    if (ll < 16) return ll;
    if (ll < 64) { /* codes 16..24 */
        static const uint8_t t[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 16, 16, 17, 17, 18, 18,
                                      19, 19, 20, 20, 20, 20, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 22, 22,
                                      23, 23, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24,
                                      24, 24, 24, 24, 24, 24};
        return t[ll];
    }
    return (uint32_t)hb32(ll) + 19;
}
uint32_t zge_ml_code(uint32_t ml) /* ml >= 3 */
{
    uint32_t b = ml - 3;
    if (b < 32) return b;
    if (b < 128) {
        /* base-3 values 32..127 -> codes 32..42 */
        static const uint8_t lim[11] = {34, 36, 38, 40, 44, 48, 56, 64, 80, 96, 128};
        uint32_t c = 0;
        while (b >= lim[c]) c++;
        return 32 + c;
    }
    return (uint32_t)hb32(b) + 36;
}

/* ------------------------------------------------------------------ FSE (encoder side) ------- */
typedef struct {
    uint16_t state_tab[512];     /* cumul-ordered cell -> state value (T + cell position)        */
    int32_t  delta_nb[64];       /* (maxBitsOut<<16) - minStatePlus                                */
    int32_t  delta_find[64];
    int al;
} fse_ctab;

/* Build the encoding table from normalized counts (norm: -1 = "less than one"). */
static void fse_build_ctab(fse_ctab *t, const int16_t *norm, int nsym, int al)
{
    int T = 1 << al, high = T - 1, s, i, pos = 0, step = (T >> 1) + (T >> 3) + 3, mask = T - 1;
    uint8_t cellsym[512];
    int cumul[65], total;
    for (s = 0; s < nsym; s++)
        if (norm[s] == -1) cellsym[high--] = (uint8_t)s;
    for (s = 0; s < nsym; s++) {
        if (norm[s] <= 0) continue;
        for (i = 0; i < norm[s]; i++) {
            cellsym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    cumul[0] = 0;
    for (s = 0; s < nsym; s++) cumul[s + 1] = cumul[s] + (norm[s] == -1 ? 1 : norm[s]);
    {
        int next[64];
        for (s = 0; s < nsym; s++) next[s] = cumul[s];
        for (i = 0; i < T; i++) t->state_tab[next[cellsym[i]]++] = (uint16_t)(T + i);
    }
    total = 0;
    for (s = 0; s < nsym; s++) {
        int n = norm[s];
        if (n == 0) { t->delta_nb[s] = ((al + 1) << 16) - T; t->delta_find[s] = 0; continue; }
        if (n == -1 || n == 1) {
            t->delta_nb[s] = (al << 16) - T;
            t->delta_find[s] = total - 1;
            total += 1;
        } else {
            int max_bits_out = al - hb32((uint32_t)(n - 1));
            int min_state_plus = n << max_bits_out;
            t->delta_nb[s] = (max_bits_out << 16) - min_state_plus;
            t->delta_find[s] = total - n;
            total += n;
        }
    }
    t->al = al;
}
static uint32_t fse_init_state(const fse_ctab *t, int s)
{
    /* the symbol's lowest state: guarantees the largest bit count on its first transition */
    int nb = (t->delta_nb[s] + (1 << 15)) >> 16;
    int value = (nb << 16) - t->delta_nb[s];
    return t->state_tab[(value >> nb) + t->delta_find[s]];
}
static uint32_t fse_encode(const fse_ctab *t, bitw *b, uint32_t state, int s)
{
    int nb = (int)((state + (uint32_t)t->delta_nb[s]) >> 16);
    bw_add(b, state, nb);
    return t->state_tab[(int)(state >> nb) + t->delta_find[s]];
}
static void fse_flush(const fse_ctab *t, bitw *b, uint32_t state) { bw_add(b, state, t->al); }

/* fixed-point log2: returns 256*log2(x) for x>=1 (linear interpolation of the mantissa) */
static uint32_t log2_fp8(uint32_t x)
{
    int h = hb32(x);
    uint32_t frac = h >= 8 ? (x >> (h - 8)) - 256 : (x << (8 - h)) - 256; /* 0..255 */
    return (uint32_t)h * 256 + frac;
}

/* Normalize counts to sum 2^al.  Every present symbol gets >= 1.  Deterministic:
 * round-to-nearest, then the surplus/deficit is applied to the largest entries. */
static void fse_normalize(const uint32_t *count, int nsym, uint32_t total, int al, int16_t *norm)
{
    int T = 1 << al, sum = 0, s;
    for (s = 0; s < nsym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        {
            uint64_t v = ((uint64_t)count[s] * (uint64_t)T + total / 2) / total;
            if (v < 1) v = 1;
            norm[s] = (int16_t)v;
            sum += (int)v;
        }
    }
    while (sum != T) {
        /* pick the entry with the largest norm (lowest symbol wins ties) */
        int best = -1;
        for (s = 0; s < nsym; s++)
            if (norm[s] > 0 && (best < 0 || norm[s] > norm[best])) best = s;
        if (sum > T) {
            int take = sum - T, room = norm[best] - 1;
            if (take > room) take = room;
            if (take <= 0) break; /* cannot happen while T >= number of present symbols */
            norm[best] = (int16_t)(norm[best] - take);
            sum -= take;
        } else {
            norm[best] = (int16_t)(norm[best] + (T - sum));
            sum = T;
        }
    }
}

/* Write an FSE table description (RFC 8878 4.1.1); returns bytes written. */
static size_t fse_write_desc(uint8_t *dst, size_t cap, const int16_t *norm, int nsym, int al)
{
    bitw b;
    int remaining = (1 << al) + 1, threshold = 1 << al, nb = al + 1, s = 0;
    bw_init(&b, dst, cap);
    bw_add(&b, (uint32_t)(al - 5), 4);
    while (remaining > 1 && s < nsym) {
        int count = norm[s++], max = 2 * threshold - 1 - remaining, val = count + 1;
        remaining -= count < 0 ? -count : count;
        if (val >= threshold) val += max;          /* large values are shifted up by max   */
        if (val < max) bw_add(&b, (uint32_t)val, nb - 1);   /* small values use nb-1 bits  */
        else bw_add(&b, (uint32_t)val, nb);
        if (count == 0) {
            /* run of further zero-probability symbols, 2-bit repeat fields */
            int run = 0;
            while (s + run < nsym && norm[s + run] == 0) run++;
            s += run;
            while (run >= 3) { bw_add(&b, 3, 2); run -= 3; }
            bw_add(&b, (uint32_t)run, 2);
        }
        while (remaining < threshold) { nb--; threshold >>= 1; }
    }
    if (b.nb > 0) { /* flush partial byte, no end mark in forward streams */
        if (b.pos < b.cap) b.p[b.pos] = (uint8_t)b.acc; else b.overflow = 1;
        b.pos++;
    }
    return b.overflow ? 0 : b.pos;
}

/* ------------------------------------------------------------------ Huffman ------------------- */
#define HUF_MAXBITS 11
/* Code lengths for `count[256]`, limited to HUF_MAXBITS.  Returns number of present symbols. */
static int huf_build_lengths(const uint32_t *count, uint8_t *len)
{
    int order[256], n = 0, i, j;
    uint32_t w[512];
    int parent[512], depth[512];
    memset(len, 0, 256);
    for (i = 0; i < 256; i++) if (count[i]) order[n++] = i;
    if (n < 2) { if (n == 1) len[order[0]] = 1; return n; }
    /* sort ascending by (count, symbol): rank-by-counting in the kernel, insertion sort here */
    for (i = 1; i < n; i++) {
        int s = order[i];
        for (j = i; j > 0 && (count[order[j - 1]] > count[s]); j--) order[j] = order[j - 1];
        order[j] = s;
    }
    /* two-queue Huffman: leaves 0..n-1 (sorted), internal nodes n..2n-2 created in order */
    for (i = 0; i < n; i++) w[i] = count[order[i]];
    {
        int leaf = 0, inode = n, next = n;
        while (next < 2 * n - 1) {
            int a, b;
            /* ties: prefer the leaf (keeps the tree shallow) */
            if (leaf < n && (inode >= next || w[leaf] <= w[inode])) a = leaf++; else a = inode++;
            if (leaf < n && (inode >= next || w[leaf] <= w[inode])) b = leaf++; else b = inode++;
            w[next] = w[a] + w[b];
            parent[a] = next;
            parent[b] = next;
            next++;
        }
        depth[2 * n - 2] = 0;
        for (i = 2 * n - 3; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
    }
    {
        /* histogram of leaf depths, clamp to HUF_MAXBITS and repair the Kraft sum */
        int num[64], k;
        uint32_t total = 0;
        memset(num, 0, sizeof num);
        for (i = 0; i < n; i++) num[depth[i] > 63 ? 63 : depth[i]]++;
        for (k = HUF_MAXBITS + 1; k < 64; k++) { num[HUF_MAXBITS] += num[k]; num[k] = 0; }
        for (k = 1; k <= HUF_MAXBITS; k++) total += (uint32_t)num[k] << (HUF_MAXBITS - k);
        while (total != (1u << HUF_MAXBITS)) {
            num[HUF_MAXBITS]--;
            for (k = HUF_MAXBITS - 1; k > 0; k--)
                if (num[k]) { num[k]--; num[k + 1] += 2; break; }
            total--;
        }
        /* most frequent symbols (end of `order`) get the shortest codes */
        i = n - 1;
        for (k = 1; k <= HUF_MAXBITS; k++) {
            int c;
            for (c = 0; c < num[k]; c++) len[order[i--]] = (uint8_t)k;
        }
    }
    return n;
}

typedef struct { uint16_t code[256]; uint8_t len[256]; int max_bits; int nsym_last; } huf_ctab;

/* Canonical code assignment in zstd's weight order (RFC 8878 4.2.1). */
static void huf_assign_codes(huf_ctab *h)
{
    int maxlen = 0, i, w;
    uint32_t rank_start[16], rank_count[16];
    h->nsym_last = 0;
    for (i = 0; i < 256; i++) if (h->len[i]) { if (h->len[i] > maxlen) maxlen = h->len[i]; h->nsym_last = i; }
    h->max_bits = maxlen;
    memset(rank_count, 0, sizeof rank_count);
    for (i = 0; i < 256; i++) if (h->len[i]) rank_count[maxlen + 1 - h->len[i]]++;
    {
        uint32_t pos = 0;
        for (w = 1; w <= maxlen; w++) { rank_start[w] = pos; pos += rank_count[w] << (w - 1); }
    }
    for (i = 0; i < 256; i++) {
        if (!h->len[i]) { h->code[i] = 0; continue; }
        w = maxlen + 1 - h->len[i];
        h->code[i] = (uint16_t)(rank_start[w] >> (w - 1));
        rank_start[w] += 1u << (w - 1);
    }
}

/* Huffman tree description.  Returns bytes written (0 = cannot represent). */
static size_t huf_write_desc(const huf_ctab *h, uint8_t *dst, size_t cap)
{
    uint8_t wt[256];
    int n = h->nsym_last, i; /* explicit weights for symbols 0..n-1; symbol n is implied */
    for (i = 0; i < n; i++) wt[i] = h->len[i] ? (uint8_t)(h->max_bits + 1 - h->len[i]) : 0;
    /* FSE-compressed weights */
    if (n > 1) {
        uint32_t count[16];
        int16_t norm[16];
        int nsym = 0, distinct = 0, al = 6, s;
        uint32_t maxc = 0;
        uint8_t tmp[160];
        size_t hdr;
        memset(count, 0, sizeof count);
        for (i = 0; i < n; i++) count[wt[i]]++;
        for (s = 0; s < 13; s++) if (count[s]) { nsym = s + 1; distinct++; if (count[s] > maxc) maxc = count[s]; }
        if (distinct > 1 && maxc > 1) {
            fse_ctab ct;
            bitw b;
            uint32_t s1, s2;
            int ip = n;
            size_t body;
            /* accuracy: at most 6, lower for few weights */
            { int lim = hb32((uint32_t)(n - 1)) - 2; if (lim < al) al = lim; if (al < 5) al = 5; }
            while ((1 << al) < distinct) al++;
            fse_normalize(count, nsym, (uint32_t)n, al, norm);
            hdr = fse_write_desc(tmp, sizeof tmp, norm, nsym, al);
            if (hdr) {
                fse_build_ctab(&ct, norm, nsym, al);
                bw_init(&b, tmp + hdr, sizeof tmp - hdr);
                if (n & 1) {
                    s1 = fse_init_state(&ct, wt[--ip]);
                    s2 = fse_init_state(&ct, wt[--ip]);
                    s1 = fse_encode(&ct, &b, s1, wt[--ip]);
                } else {
                    s2 = fse_init_state(&ct, wt[--ip]);
                    s1 = fse_init_state(&ct, wt[--ip]);
                }
                while (ip > 0) {
                    s2 = fse_encode(&ct, &b, s2, wt[--ip]);
                    s1 = fse_encode(&ct, &b, s1, wt[--ip]);
                }
                fse_flush(&ct, &b, s2);
                fse_flush(&ct, &b, s1);
                body = bw_close(&b);
                if (!b.overflow && hdr + body < 128 && (n > 128 || hdr + body < (size_t)(n + 1) / 2) && 1 + hdr + body <= cap) {
                    dst[0] = (uint8_t)(hdr + body);
                    memcpy(dst + 1, tmp, hdr + body);
                    return 1 + hdr + body;
                }
            }
        }
    }
    if (n > 128) return 0;
    if ((size_t)(1 + (n + 1) / 2) > cap) return 0;
    dst[0] = (uint8_t)(127 + n);
    for (i = 0; i < n; i += 2) dst[1 + i / 2] = (uint8_t)((wt[i] << 4) | (i + 1 < n ? wt[i + 1] : 0));
    return (size_t)(1 + (n + 1) / 2);
}

/* one backward Huffman stream over lit[0..n): symbols are written last-to-first */
static size_t huf_encode_stream(const huf_ctab *h, const uint8_t *lit, size_t n, uint8_t *dst, size_t cap)
{
    bitw b;
    size_t i;
    bw_init(&b, dst, cap);
    for (i = n; i > 0; i--) bw_add(&b, h->code[lit[i - 1]], h->len[lit[i - 1]]);
    bw_close(&b);
    return b.overflow ? 0 : b.pos;
}

/* Literals section.  Returns bytes written. */
static size_t encode_literals(const uint8_t *lit, size_t n, uint8_t *dst, size_t cap, zge_stats *st)
{
    uint32_t count[256];
    size_t i, raw_hdr = n < 32 ? 1 : (n < 4096 ? 2 : 3);
    int distinct = 0;
    memset(count, 0, sizeof count);
    for (i = 0; i < n; i++) count[lit[i]]++;
    for (i = 0; i < 256; i++) if (count[i]) distinct++;
    if (n > 0 && distinct == 1 && n >= 2) {            /* RLE literals */
        if (raw_hdr + 1 > cap) return 0;
        if (raw_hdr == 1) dst[0] = (uint8_t)(1 | (n << 3));
        else if (raw_hdr == 2) { dst[0] = (uint8_t)(1 | (1 << 2) | ((n & 15) << 4)); dst[1] = (uint8_t)(n >> 4); }
        else { dst[0] = (uint8_t)(1 | (3 << 2) | ((n & 15) << 4)); dst[1] = (uint8_t)(n >> 4); dst[2] = (uint8_t)(n >> 12); }
        dst[raw_hdr] = lit[0];
        if (st) st->lit_rle++;
        return raw_hdr + 1;
    }
    {   /* a flat histogram is not worth a Huffman attempt (the rule libzstd's huf_compress uses: "probably not compressible") */
        uint32_t largest = 0;
        for (i = 0; i < 256; i++) if (count[i] > largest) largest = count[i];
        if ((size_t)largest <= (n >> 7) + 4) distinct = 0;
    }
    if (n >= ZGE_MIN_HUF_LITERALS && distinct >= 2) {
        huf_ctab h;
        uint8_t desc[160];
        size_t dlen, est_bits = 0, est;
        huf_build_lengths(count, h.len);
        huf_assign_codes(&h);
        dlen = huf_write_desc(&h, desc, sizeof desc);
        for (i = 0; i < 256; i++) est_bits += (size_t)count[i] * h.len[i];
        est = dlen + (est_bits + 7) / 8 + (n >= 256 ? 6 + 4 : 1);
        if (dlen && est + 3 < n) {
            int single = n < 256;
            size_t hdr = single ? 3 : (n < 1024 ? 3 : (n < 16384 ? 4 : 5));
            uint8_t *body = dst + hdr;
            size_t bcap = cap > hdr ? cap - hdr : 0, pos, comp;
            if (bcap < dlen + 6) return 0;
            memcpy(body, desc, dlen);
            pos = dlen;
            if (single) {
                size_t s = huf_encode_stream(&h, lit, n, body + pos, bcap - pos);
                if (!s) return 0;
                pos += s;
            } else {
                size_t per = (n + 3) / 4, s[4], jt = pos, k;
                pos += 6;
                for (k = 0; k < 4; k++) {
                    size_t beg = k * per, cnt = k < 3 ? per : n - 3 * per;
                    s[k] = huf_encode_stream(&h, lit + beg, cnt, body + pos, bcap - pos);
                    if (!s[k]) return 0;
                    pos += s[k];
                }
                for (k = 0; k < 3; k++) { body[jt + 2 * k] = (uint8_t)s[k]; body[jt + 2 * k + 1] = (uint8_t)(s[k] >> 8); }
                if (s[0] > 65535 || s[1] > 65535 || s[2] > 65535) return 0;
            }
            comp = pos;
            if (comp + hdr < n + raw_hdr) {
                /* sizes must fit the chosen header; hdr was chosen from n, comp < n here */
                uint64_t v;
                if (hdr == 3) { v = 2u | ((single ? 0u : 1u) << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << 14); }
                else if (hdr == 4) { v = 2u | (2u << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << 18); }
                else { v = 2u | (3u << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << 22); }
                for (i = 0; i < hdr; i++) dst[i] = (uint8_t)(v >> (8 * i));
                if (st) st->lit_huf++;
                return hdr + comp;
            }
        }
    }
    /* raw literals */
    if (raw_hdr + n > cap) return 0;
    if (raw_hdr == 1) dst[0] = (uint8_t)(0 | (n << 3));
    else if (raw_hdr == 2) { dst[0] = (uint8_t)(0 | (1 << 2) | ((n & 15) << 4)); dst[1] = (uint8_t)(n >> 4); }
    else { dst[0] = (uint8_t)(0 | (3 << 2) | ((n & 15) << 4)); dst[1] = (uint8_t)(n >> 4); dst[2] = (uint8_t)(n >> 12); }
    memcpy(dst + raw_hdr, lit, n);
    if (st) st->lit_raw++;
    return raw_hdr + n;
}

/* ------------------------------------------------------------------ sequences section -------- */
/* cost in 1/256 bit of coding `count` with the distribution `norm` of accuracy `al` */
static uint64_t dist_cost(const uint32_t *count, const int16_t *norm, int nsym, int al)
{
    uint64_t c = 0;
    int s;
    for (s = 0; s < nsym; s++) {
        uint32_t n;
        if (!count[s]) continue;
        if (norm[s] == 0) return (uint64_t)-1;     /* cannot be coded */
        n = norm[s] < 0 ? 1u : (uint32_t)norm[s];
        c += (uint64_t)count[s] * ((uint32_t)al * 256 - log2_fp8(n));
    }
    return c;
}

typedef struct { int mode; int16_t norm[64]; int nsym; int al; uint8_t rle_sym; uint8_t desc[80]; size_t desc_len; uint64_t cost; } seq_table_choice;

static void choose_table(seq_table_choice *c, const uint32_t *count, int maxsym, uint32_t nseq,
                         const int16_t *def, int def_n, int def_al, int max_al)
{
    int s, distinct = 0, last = 0, al;
    uint64_t cost_def, cost_dyn;
    for (s = 0; s <= maxsym; s++) if (count[s]) { distinct++; last = s; }
    c->cost = 0;
    if (distinct == 1) { c->mode = 1; c->rle_sym = (uint8_t)last; c->desc_len = 0; return; }
    cost_def = last < def_n ? dist_cost(count, def, def_n, def_al) : (uint64_t)-1;
    /* dynamic table */
    al = hb32(nseq > 1 ? nseq - 1 : 1) - 2;
    if (al > max_al) al = max_al;
    if (al < 5) al = 5;
    while ((1 << al) < distinct) al++;
    c->nsym = last + 1;
    c->al = al;
    fse_normalize(count, c->nsym, nseq, al, c->norm);
    c->desc_len = fse_write_desc(c->desc, sizeof c->desc, c->norm, c->nsym, al);
    cost_dyn = c->desc_len ? dist_cost(count, c->norm, c->nsym, al) + (uint64_t)c->desc_len * 8 * 256 : (uint64_t)-1;
    if (cost_def <= cost_dyn) {
        c->mode = 0;
        c->nsym = def_n;
        c->al = def_al;
        memcpy(c->norm, def, sizeof(int16_t) * (size_t)def_n);
        c->desc_len = 0;
        c->cost = cost_def;
    } else {
        c->mode = 2;
        c->cost = cost_dyn;
    }
}

/* Shared sequence tables (Repeat_Mode, RFC 8878 3.1.1.3.2.1: "the table used in the previous Compressed_Block with
 * Number_of_Sequences > 0 will be used again").  The blocks of a frame are taken in GROUPS of ZGE_TABLE_GROUP (block index / 16: the sixteen
 * 64 KiB blocks of a 1 MiB entry are one group).  Per table type (LL, OF, ML) a group may code all its blocks with ONE table, normalised from
 * the SUM of the blocks' code histograms: the first block describes it, the others say Repeat.  That saves fifteen descriptions, and --
 * the point for the engine's decoder -- the blocks of an entry share one table set, which zstd_decode.hip keeps in LDS for the whole
 * wave instead of one 2.5 KiB table set per block in HBM.  Taken when it costs at most 1/64 more table-coded bits than the blocks' own
 * best choices (the sum's cost is linear in the counts: one dist_cost call).
 * The engine decides this per group in a small kernel between two passes of the entropy stage (zge_entropy.hip: zarc_zge_plan),
 * before any sequence is coded, so what the decoder "has" must be PROVEN from the histograms: a block hands the group's table on only
 * if an upper bound of its coded size is below its raw size (it cannot end up a raw block); the block behind any other block
 * describes the table again.  A block whose codes of one type are all equal keeps RLE mode for that type and breaks that type's chain
 * the same way.  Blocks without sequences, RLE blocks and blocks whose literals failed to code take no part: they leave the decoder's
 * tables alone. */
#define ZGE_TABLE_GROUP 16
typedef struct {
    int active;                /* the block has sequences to code (and its literals section exists) */
    uint32_t nseq;
    uint32_t count[3][64];     /* code histograms: LL, OF, ML */
    seq_table_choice ch[3];    /* own best choice (choose_table), then the final one */
    uint64_t extra_bits;       /* bits of the sequences' extra-bits fields */
    size_t lsz, blen;          /* literals section in front, raw size of the block (for the size bound) */
    int guaranteed;
} seq_plan;

static uint32_t fse_max_bits(int n, int al) { return (n == -1 || n == 1) ? (uint32_t)al : (uint32_t)(al - hb32((uint32_t)(n - 1))); }
static const int SEQ_MAXSYM[3] = {35, 31, 52}, SEQ_MAX_AL[3] = {9, 8, 9};

static void seq_plan_block(seq_plan *pl, const zge_seq *seq, uint32_t nseq, size_t lsz, size_t blen)
{
    uint32_t i;
    int s;
    memset(pl, 0, sizeof *pl);
    pl->nseq = nseq; pl->lsz = lsz; pl->blen = blen;
    pl->active = nseq != 0 && lsz != 0;
    if (!pl->active) return;
    for (i = 0; i < nseq; i++) {
        pl->count[0][zge_ll_code(seq[i].ll)]++;
        pl->count[1][hb32(seq[i].ofv)]++;
        pl->count[2][zge_ml_code(seq[i].ml)]++;
    }
    choose_table(&pl->ch[0], pl->count[0], 35, nseq, LL_DEFAULT, 36, 6, 9);
    choose_table(&pl->ch[1], pl->count[1], 31, nseq, OF_DEFAULT, 29, 5, 8);
    choose_table(&pl->ch[2], pl->count[2], 52, nseq, ML_DEFAULT, 53, 6, 9);
    for (s = 0; s < 36; s++) pl->extra_bits += (uint64_t)pl->count[0][s] * LL_BITS[s];
    for (s = 0; s < 32; s++) pl->extra_bits += (uint64_t)pl->count[1][s] * (uint32_t)s;
    for (s = 0; s < 53; s++) pl->extra_bits += (uint64_t)pl->count[2][s] * ML_BITS[s];
}

/* the blocks of one group, in order */
static void seq_plan_group(seq_plan *pl, int nb)
{
    int use_group[3] = {0, 0, 0}, have[3] = {0, 0, 0}, t, b, s;
    seq_table_choice g[3];
    for (t = 0; t < 3; t++) {
        uint32_t sum[64], total = 0;
        uint64_t cost_own = 0, cost_group;
        int np = 0, distinct = 0, last = 0, al;
        memset(sum, 0, sizeof sum);
        memset(&g[t], 0, sizeof g[t]);
        for (b = 0; b < nb; b++) {
            if (!pl[b].active || pl[b].ch[t].mode == 1) continue;
            for (s = 0; s <= SEQ_MAXSYM[t]; s++) sum[s] += pl[b].count[t][s];
            total += pl[b].nseq; cost_own += pl[b].ch[t].cost; np++;
        }
        if (np < 2) continue;
        for (s = 0; s <= SEQ_MAXSYM[t]; s++) if (sum[s]) { distinct++; last = s; }
        al = hb32(total > 1 ? total - 1 : 1) - 2;
        if (al > SEQ_MAX_AL[t]) al = SEQ_MAX_AL[t];
        if (al < 5) al = 5;
        while ((1 << al) < distinct) al++;
        g[t].mode = 2; g[t].nsym = last + 1; g[t].al = al;
        fse_normalize(sum, g[t].nsym, total, al, g[t].norm);
        g[t].desc_len = fse_write_desc(g[t].desc, sizeof g[t].desc, g[t].norm, g[t].nsym, al);
        if (!g[t].desc_len) continue;
        cost_group = dist_cost(sum, g[t].norm, g[t].nsym, al) + (uint64_t)g[t].desc_len * 8 * 256;
        use_group[t] = cost_group <= cost_own + (cost_own >> 6);
    }
    for (b = 0; b < nb; b++) {
        uint64_t ub_bits = 1; /* the end mark */
        size_t ub;
        seq_plan *p = &pl[b];
        if (!p->active) continue;
        for (t = 0; t < 3; t++) {
            seq_table_choice *c = &p->ch[t];
            if (c->mode == 1 || !use_group[t]) continue;
            { const uint8_t rle = c->rle_sym; *c = g[t]; c->rle_sym = rle; }
            if (have[t]) { c->mode = 3; c->desc_len = 0; }
        }
        /* upper bound of the block's coded size: every state transition at its symbol's larger bit count */
        for (t = 0; t < 3; t++) {
            const seq_table_choice *c = &p->ch[t];
            if (c->mode == 1) continue;
            for (s = 0; s < c->nsym && s <= SEQ_MAXSYM[t]; s++) if (p->count[t][s]) ub_bits += (uint64_t)p->count[t][s] * fse_max_bits(c->norm[s], c->al);
            ub_bits += (uint64_t)c->al;
        }
        ub_bits += p->extra_bits;
        ub = p->lsz + (p->nseq < 128 ? 1 : (p->nseq < 0x7F00 ? 2 : 3)) + 1 + (size_t)((ub_bits + 7) / 8);
        for (t = 0; t < 3; t++) ub += p->ch[t].mode == 1 ? 1 : p->ch[t].desc_len;
        p->guaranteed = ub < p->blen;
        for (t = 0; t < 3; t++) have[t] = (p->ch[t].mode != 1 && use_group[t]) ? p->guaranteed : 0;
    }
}

static size_t encode_sequences(const zge_seq *seq, uint32_t nseq, uint8_t *dst, size_t cap, zge_stats *st, const seq_plan *pl)
{
    size_t pos = 0;
    seq_table_choice tl, to, tm;
    fse_ctab ctl, cto, ctm;
    bitw b;
    if (cap < 4) return 0;
    if (nseq < 128) dst[pos++] = (uint8_t)nseq;
    else if (nseq < 0x7F00) { dst[pos++] = (uint8_t)((nseq >> 8) + 128); dst[pos++] = (uint8_t)nseq; }
    else { dst[pos++] = 255; dst[pos++] = (uint8_t)(nseq - 0x7F00); dst[pos++] = (uint8_t)((nseq - 0x7F00) >> 8); }
    if (nseq == 0) return pos;
    tl = pl->ch[0]; to = pl->ch[1]; tm = pl->ch[2];
    if (st) { st->seq_mode[tl.mode]++; st->seq_mode[to.mode]++; st->seq_mode[tm.mode]++; }
    if (pos + 1 + 3 + tl.desc_len + to.desc_len + tm.desc_len > cap) return 0;
    dst[pos++] = (uint8_t)((tl.mode << 6) | (to.mode << 4) | (tm.mode << 2));
    if (tl.mode == 1) dst[pos++] = tl.rle_sym; else { memcpy(dst + pos, tl.desc, tl.desc_len); pos += tl.desc_len; }
    if (to.mode == 1) dst[pos++] = to.rle_sym; else { memcpy(dst + pos, to.desc, to.desc_len); pos += to.desc_len; }
    if (tm.mode == 1) dst[pos++] = tm.rle_sym; else { memcpy(dst + pos, tm.desc, tm.desc_len); pos += tm.desc_len; }
    if (tl.mode != 1) fse_build_ctab(&ctl, tl.norm, tl.nsym, tl.al);
    if (to.mode != 1) fse_build_ctab(&cto, to.norm, to.nsym, to.al);
    if (tm.mode != 1) fse_build_ctab(&ctm, tm.norm, tm.nsym, tm.al);
    bw_init(&b, dst + pos, cap - pos);
    {
        uint32_t sl = 0, so = 0, sm = 0;
        uint32_t n = nseq - 1;
        uint32_t llc = zge_ll_code(seq[n].ll), mlc = zge_ml_code(seq[n].ml), ofc = (uint32_t)hb32(seq[n].ofv);
        if (tm.mode != 1) sm = fse_init_state(&ctm, (int)mlc);
        if (to.mode != 1) so = fse_init_state(&cto, (int)ofc);
        if (tl.mode != 1) sl = fse_init_state(&ctl, (int)llc);
        bw_add(&b, seq[n].ll - LL_BASE[llc], LL_BITS[llc]);
        bw_add(&b, seq[n].ml - ML_BASE[mlc], ML_BITS[mlc]);
        bw_add(&b, seq[n].ofv - (1u << ofc), (int)ofc);
        while (n > 0) {
            n--;
            llc = zge_ll_code(seq[n].ll); mlc = zge_ml_code(seq[n].ml); ofc = (uint32_t)hb32(seq[n].ofv);
            if (to.mode != 1) so = fse_encode(&cto, &b, so, (int)ofc);
            if (tm.mode != 1) sm = fse_encode(&ctm, &b, sm, (int)mlc);
            if (tl.mode != 1) sl = fse_encode(&ctl, &b, sl, (int)llc);
            bw_add(&b, seq[n].ll - LL_BASE[llc], LL_BITS[llc]);
            bw_add(&b, seq[n].ml - ML_BASE[mlc], ML_BITS[mlc]);
            bw_add(&b, seq[n].ofv - (1u << ofc), (int)ofc);
        }
        if (tm.mode != 1) fse_flush(&ctm, &b, sm);
        if (to.mode != 1) fse_flush(&cto, &b, so);
        if (tl.mode != 1) fse_flush(&ctl, &b, sl);
    }
    bw_close(&b);
    if (b.overflow) return 0;
    return pos + b.pos;
}

/* ------------------------------------------------------------------ match finder ------------- */
/* Hashes built from 32-bit multiplies only (a 64-bit multiply costs four quarter-rate VALU ops on gfx950):
 * two odd multipliers, wrapping sum, top `bits` bits. */
static uint32_t hash_long(uint64_t v, int bits)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return (lo * 0x9E3779B1u + hi * 0x85EBCA77u) >> (32 - bits);
}
/* far long-hash table: 12 bytes (the 8 at the position and the next 4).  Far offsets are expensive to code, so only repeats of
 * some length are worth finding there, and a table keyed by 12 bytes is not crowded by the short repeats of text-like data. */
static uint32_t hash_far(uint64_t v, uint32_t w, int bits)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return (lo * 0x9E3779B1u + hi * 0x85EBCA77u + w * 0xC2B2AE3Du) >> (32 - bits);
}
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t hash_short(uint64_t v, int bits, int nbytes)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (nbytes < 8) hi &= (nbytes > 4) ? ((1u << (8 * (nbytes - 4))) - 1) : 0u;
    if (nbytes < 4) lo &= (1u << (8 * nbytes)) - 1;
    return (lo * 0xC2B2AE3Du + hi * 0x27D4EB2Fu) >> (32 - bits);
}

/* common prefix length of src[p..] and src[q..], q < p, at most `limit` bytes */
static uint32_t match_len(const uint8_t *src, size_t p, size_t q, uint32_t limit)
{
    uint32_t n = 0;
    while (n + 8 <= limit) {
        uint64_t x = rd64(src + p + n) ^ rd64(src + q + n);
        if (x) return n + (uint32_t)(__builtin_ctzll(x) >> 3);
        n += 8;
    }
    while (n < limit && src[p + n] == src[q + n]) n++;
    return n;
}

typedef struct { uint32_t len, off; uint8_t back, is_rep; } cand;

static int32_t score_of(const zge_params *P, uint32_t len, uint32_t off, int is_rep)
{
    if (is_rep) return (int32_t)(P->lit_cost * len) - P->rep_cost;
    return (int32_t)(P->lit_cost * len) - P->match_cost - hb32(off);
}

typedef struct {
    const zge_params *P;
    const uint8_t *src;
    size_t n;
    uint32_t *tl, *ts;       /* long / short tables: value = position+1, 0 = empty */
    uint16_t *t16;           /* near16: the one near table, value = low 16 bits of the position */
    size_t window;
    cand *M, *M2;            /* per-tile candidates: own best, then after backward propagation */
    uint32_t *next;          /* per-tile successor of each position on the parse path */
    uint8_t *take, *mark;
    zge_stats *st;
    /* Cold stretches (incompressible data): `cold` counts the searched tiles in a row in which no position found a match;
     * from the second one on, the next 1, 3, then 7 tiles are not searched at all (all literals, nothing inserted), the way
     * libzstd's search step grows while it finds nothing.  Any match in a searched tile ends the stretch.  The state lives
     * for the whole frame. */
    uint32_t cold, skip_left;
    uint32_t erep0, erep1;   /* recent-offset guesses: offsets of the last two matches selected so far in the frame (0 = none) */
    uint32_t *fl, *fs;       /* far tables (long / short hash), far_ways entries per bucket */
    uint32_t *farc;          /* per-tile far candidates: ZGE_FAR_MAX per position, value = position+1 */
    size_t far_pending;      /* start of the most recent searched tile whose far inserts are still to be made (ZGE_NO_TILE = none) */
} mf_ctx;
#define ZGE_FAR_MAX 8 /* far_ways * (1 + far_short) <= 8 */
#define ZGE_NO_TILE ((size_t)-1)

/* Resolve explicit offsets against the repcode history (RFC 8878 3.1.1.5).  The history starts UNKNOWN
 * (0 never equals a real offset) in every block: whether the previous block ends up raw/RLE -- which leaves the
 * decoder's history untouched -- is only known after entropy coding, and blocks are coded independently.
 * The cost is at most 3 repcodes per block.  In the engine this pass runs at the start of the entropy stage. */
/* A sequence without literals that continues the previous match at the same offset is the tail of ONE longer match: the finder
 * caps a match at `cap` bytes per position (its compare loops are per-thread work in the kernel) and never looks past a tile's
 * overrun window, so long matches arrive in pieces.  Joined here (the engine does it at the start of the entropy stage), match
 * lengths reach the format's 131 074. */
static uint32_t merge_sequences(zge_seq *seq, uint32_t nseq)
{
    uint32_t i, o = 0;
    for (i = 0; i < nseq; i++) {
        if (o > 0 && seq[i].ll == 0 && seq[i].off == seq[o - 1].off) seq[o - 1].ml += seq[i].ml;
        else seq[o++] = seq[i];
    }
    return o;
}

static void resolve_repcodes(zge_seq *seq, uint32_t nseq, zge_stats *st)
{
    uint32_t r[3] = {0, 0, 0}, i;
    for (i = 0; i < nseq; i++) {
        uint32_t off = seq[i].off, ofv;
        if (seq[i].ll > 0) {
            if (off == r[0]) ofv = 1;
            else if (off == r[1]) { ofv = 2; r[1] = r[0]; r[0] = off; }
            else if (off == r[2]) { ofv = 3; r[2] = r[1]; r[1] = r[0]; r[0] = off; }
            else { ofv = off + 3; r[2] = r[1]; r[1] = r[0]; r[0] = off; }
        } else {
            if (off == r[1]) { ofv = 1; r[1] = r[0]; r[0] = off; }
            else if (off == r[2]) { ofv = 2; r[2] = r[1]; r[1] = r[0]; r[0] = off; }
            else if (r[0] > 1 && off == r[0] - 1) { ofv = 3; r[2] = r[1]; r[1] = r[0]; r[0] = off; }
            else { ofv = off + 3; r[2] = r[1]; r[1] = r[0]; r[0] = off; }
        }
        seq[i].ofv = ofv;
        if (st) { if (ofv <= 3) st->rep_seqs++; st->seqs++; st->match_bytes += seq[i].ml; }
    }
}

/* Is position p (12 readable bytes) a far-table position?  Positional scheme: p mod 2^far_step_log < 2^far_res_log is inserted,
 * p mod 2^far_res_log == 0 is looked up.  Content-defined scheme (far_cdc_log > 0): inserted and looked up iff the far_cdc_log bits
 * of the hash below the bucket and check bits are zero. */
static int far_sampled(const zge_params *P, const uint8_t *src, size_t p, int insert)
{
    if (P->far_cdc_log > 0) {
        /* windows of period 1, 2 or 4 (three equal 4-byte words: runs, zero padding) are no far positions: every position of such a
         * stretch has the same hash -- hundreds of same-bucket inserts per tile -- and the near table and the recent offsets find them */
        const uint64_t v = rd64(src + p);
        if ((uint32_t)v == (uint32_t)(v >> 32) && (uint32_t)v == rd32(src + p + 8)) return 0;
        return (hash_far(v, rd32(src + p + 8), P->far_log + P->tag_bits + P->far_cdc_log) & ((1u << P->far_cdc_log) - 1)) == 0;
    }
    if (insert) return ((uint32_t)p & ((1u << P->far_step_log) - 1)) < (1u << P->far_res_log);
    return ((uint32_t)p & ((1u << P->far_res_log) - 1)) == 0;
}

/* Far inserts of the searched tile starting at `tile`: every 2^far_step_log-th position, into the way of this tile; the highest
 * position wins (atomic max in the kernel). */
static void far_insert_tile(mf_ctx *c, size_t tile)
{
    const zge_params *P = c->P;
    const uint8_t *src = c->src;
    const uint32_t tmask = (1u << P->tag_bits) - 1;
    const size_t segbase = tile & ~(((size_t)1 << P->seg_log) - 1);
    const size_t way = (tile / (size_t)P->tile) % (size_t)P->far_ways;
    const size_t far_end = c->n >= 12 ? c->n - 11 : 0;
    /* a tile never straddles a block: its end is the block's (or the frame's) */
    size_t tend = tile + (size_t)P->tile, bend = (tile / ZGE_BLOCK + 1) * (size_t)ZGE_BLOCK, p;
    if (tend > bend) tend = bend;
    if (tend > c->n) tend = c->n;
    for (p = tile; p < tend && p < far_end; p++) {
        uint64_t v;
        uint32_t hf, code = (uint32_t)(p - segbase) + 1, *e;
        if (!far_sampled(P, src, p, 1)) continue;
        v = rd64(src + p);
        hf = hash_far(v, rd32(src + p + 8), P->far_log + P->tag_bits);
        e = &c->fl[(size_t)(hf >> P->tag_bits) * (size_t)P->far_ways + way];
        if (((code << P->tag_bits) | (hf & tmask)) > *e) *e = (code << P->tag_bits) | (hf & tmask);
        if (P->far_short) {
            uint32_t hg = hash_short(v, P->far_log + P->tag_bits, P->short_bytes);
            e = &c->fs[(size_t)(hg >> P->tag_bits) * (size_t)P->far_ways + way];
            if (((code << P->tag_bits) | (hg & tmask)) > *e) *e = (code << P->tag_bits) | (hg & tmask);
        }
    }
}

/* Process one block [bs, be): fills seq[] (ll, ml, off) and lit[]; returns nseq, *nlit.
 * Every step below is a data-parallel operation over the positions of a 1024-position tile, except the
 * ordered table update (64 positions at a time) and the tile-to-tile carry of the parse cursor. */
static uint32_t matchfind_block(mf_ctx *c, size_t bs, size_t be, zge_seq *seq, uint8_t *lit, size_t *nlit)
{
    const zge_params *P = c->P;
    const uint8_t *src = c->src;
    size_t pos = bs, tile, lp = 0, prev_lp = 0;
    uint32_t nseq = 0, erep0 = c->erep0, erep1 = c->erep1, i;
    size_t cont_pos = 0, cont_end = 0; uint32_t cont_off = 0;
    /* positions with fewer than 8 readable bytes are never hashed */
    size_t hash_end = c->n >= 8 ? c->n - 7 : 0; /* p < hash_end is hashable */
    size_t far_end = c->n >= 12 ? c->n - 11 : 0; /* the far tables' long hash reads 12 bytes */
    for (tile = bs; tile < be; tile += (size_t)P->tile) {
        size_t tend = tile + (size_t)P->tile < be ? tile + (size_t)P->tile : be, p, sub;
        uint32_t tcount = (uint32_t)(tend - tile), t;
        int tile_any = 0;
        /* Table entries hold (position inside the current 2^seg_log segment + 1) << tag_bits | tag; the frame loop
         * clears the tables at every segment boundary. */
        if (pos >= tend) continue; /* whole tile already covered by a match: skip it (nothing is inserted) */
        if (c->skip_left) { /* cold stretch: this tile is not searched */
            c->skip_left--;
            for (t = (uint32_t)((pos > tile ? pos : tile) - tile); t < tcount; t++) lit[lp++] = src[tile + t];
            pos = tend;
            continue;
        }
        /* S2: ordered lookup + insert, 64 positions at a time (lookups of a group see inserts of earlier groups).
         * The extra tag bits of the hash reject most false candidates without touching memory. */
        for (sub = tile; sub < tend; sub += (size_t)P->sub) {
            size_t send = sub + (size_t)P->sub < tend ? sub + (size_t)P->sub : tend;
            const uint32_t tmask = (1u << P->tag_bits) - 1;
            const size_t segbase = tile & ~(((size_t)1 << P->seg_log) - 1);
            if (P->near16) {
                /* one table of 16-bit entries: the candidate lies d = 1 .. 65536 bytes back, d = (p - entry) mod 2^16 (0 -> 65536);
                 * whatever the entry holds (a stale position, the zeros of a fresh table) is a candidate like any other: S3 compares */
                for (p = sub; p < send; p++) {
                    cand *m = &c->M[p - tile];
                    m->len = 0; m->off = 0; m->back = 0; m->is_rep = 0;
                    if (p < hash_end) {
                        uint32_t hs = hash_short(rd64(src + p), P->short_log, P->short_bytes);
                        uint32_t d = (((uint32_t)p - c->t16[hs] - 1u) & 0xFFFFu) + 1u;
                        m->len = d <= p ? (uint32_t)p - d + 1 : 0; /* candidate position + 1 */
                    }
                }
                for (p = sub; p < send && p < hash_end; p++)
                    c->t16[hash_short(rd64(src + p), P->short_log, P->short_bytes)] = (uint16_t)p; /* ascending: the highest position stays */
                continue;
            }
            for (p = sub; p < send; p++) {
                cand *m = &c->M[p - tile];
                m->len = 0; m->off = 0; m->back = 0; m->is_rep = 0;
                if (p < hash_end) {
                    uint64_t v = rd64(src + p);
                    uint32_t hl = hash_long(v, P->long_log + P->tag_bits), hs = hash_short(v, P->short_log + P->tag_bits, P->short_bytes);
                    uint32_t el = c->tl[hl >> P->tag_bits], es = c->ts[hs >> P->tag_bits];
                    /* stash candidate positions (+1), 0 = none */
                    m->off = (el && (el & tmask) == (hl & tmask)) ? (uint32_t)segbase + (el >> P->tag_bits) : 0;
                    m->len = (es && (es & tmask) == (hs & tmask)) ? (uint32_t)segbase + (es >> P->tag_bits) : 0;
                }
            }
            for (p = sub; p < send && p < hash_end; p++) {
                uint64_t v = rd64(src + p);
                uint32_t hl = hash_long(v, P->long_log + P->tag_bits), hs = hash_short(v, P->short_log + P->tag_bits, P->short_bytes);
                uint32_t code = (uint32_t)(p - segbase) + 1;
                c->tl[hl >> P->tag_bits] = (code << P->tag_bits) | (hl & tmask);   /* ascending: max wins */
                c->ts[hs >> P->tag_bits] = (code << P->tag_bits) | (hs & tmask);
            }
        }
        /* Far tables: every position of the tile is looked up.  The kernel requests a tile's entries while it works on the tile before
         * (the round trip to HBM is off the critical path that way), ahead of that tile's own inserts: a lookup sees the inserts of
         * every searched tile except the one directly in front of it. */
        if (P->far_log) {
            const uint32_t tmask = (1u << P->tag_bits) - 1;
            const size_t segbase = tile & ~(((size_t)1 << P->seg_log) - 1);
            int w;
            if (c->far_pending != ZGE_NO_TILE && c->far_pending + (size_t)P->tile != tile) { far_insert_tile(c, c->far_pending); c->far_pending = ZGE_NO_TILE; }
            memset(c->farc, 0, sizeof(uint32_t) * ZGE_FAR_MAX * (size_t)P->tile);
            for (p = tile; p < tend && p < far_end; p++) {
                uint64_t v = rd64(src + p);
                uint32_t hf = hash_far(v, rd32(src + p + 8), P->far_log + P->tag_bits), *fc = c->farc + (p - tile) * ZGE_FAR_MAX;
                if (!far_sampled(P, src, p, 0)) continue;
                for (w = 0; w < P->far_ways; w++) {
                    uint32_t e = c->fl[(size_t)(hf >> P->tag_bits) * (size_t)P->far_ways + (size_t)w];
                    if (e && (e & tmask) == (hf & tmask)) fc[w] = (uint32_t)segbase + (e >> P->tag_bits);
                }
                if (P->far_short) {
                    uint32_t hg = hash_short(v, P->far_log + P->tag_bits, P->short_bytes);
                    for (w = 0; w < P->far_ways; w++) {
                        uint32_t e = c->fs[(size_t)(hg >> P->tag_bits) * (size_t)P->far_ways + (size_t)w];
                        if (e && (e & tmask) == (hg & tmask)) fc[P->far_ways + w] = (uint32_t)segbase + (e >> P->tag_bits);
                    }
                }
            }
        }
        /* S3: every position scores its own candidates {near (long, else short), far ..., guess0, guess1}; ties keep the earlier candidate.
         * near16, runs: inside a repeat every position's hash leads to the same distance; a position whose validated near distance
         * equals its left neighbour's (same 64-position chunk) is a FOLLOWER: its near length is its neighbour's minus one, nothing is
         * compared (the kernel requests no source for it), and a follower's near match gets no backward extension. */
        uint32_t prev_on = 0, prev_near = 0;
        for (p = tile; p < tend; p++) {
            cand *m = &c->M[p - tile];
            uint32_t limit = (uint32_t)(be - p), cap = limit < (uint32_t)P->cap ? limit : (uint32_t)P->cap;
            uint32_t best_len = 0, best_off = 0; int best_rep = 0, best_far = 0, best_fol = 0, fol = 0; int32_t best_score = -1000000;
            uint32_t k, offs[4 + ZGE_FAR_MAX];
            const uint32_t nfar = P->far_log ? (uint32_t)(P->far_ways * (1 + (P->far_short ? 1 : 0))) : 0, ntab = 1 + nfar;
            /* one near candidate: the long table's when it has a hit, the short table's only otherwise (measured: with a long-hash hit
             * at hand the short-hash candidate changes 0.003 % of the output, and it costs a source fetch per position) */
            offs[0] = m->off ? (uint32_t)p - (m->off - 1) : (m->len ? (uint32_t)p - (m->len - 1) : 0);
            if (!m->off && offs[0] > ((uint32_t)1 << P->short_window_log)) offs[0] = 0;
            for (k = 0; k < nfar; k++) { uint32_t fc = c->farc[(p - tile) * ZGE_FAR_MAX + k]; offs[1 + k] = fc ? (uint32_t)p - (fc - 1) : 0; }
            offs[ntab] = P->rep_search > 0 ? erep0 : 0;
            offs[ntab + 1] = (P->rep_search > 1 && erep1 != erep0) ? erep1 : 0;
            /* recent-offset guesses are only tried when their source lies in the tile window the kernel keeps in LDS
             * (rep_back bytes before the tile): far guesses almost never match and would cost an HBM access each */
            if (offs[ntab] > (uint32_t)(p - tile) + (uint32_t)P->rep_back) offs[ntab] = 0;
            if (offs[ntab + 1] > (uint32_t)(p - tile) + (uint32_t)P->rep_back) offs[ntab + 1] = 0;
            /* table candidates need 8 bytes in front of their source: the kernel fetches source[-8 .. 8) in one load (the
             * first half feeds the backward extension); sources in the first 8 bytes of a frame are skipped */
            for (k = 0; k < ntab; k++) if (offs[k] + 8 > p) offs[k] = 0;
            if (offs[0] > p || offs[0] > c->window) offs[0] = 0; /* the validated near distance */
            if (P->near16) {
                fol = ((p - tile) & 63) != 0 && offs[0] != 0 && offs[0] == prev_on;
                prev_near = fol ? (prev_near > 0 ? prev_near - 1 : 0) : (offs[0] ? match_len(src, p, p - offs[0], cap) : 0);
                prev_on = offs[0];
            }
            m->len = 0; m->off = 0;
            for (k = 0; k < ntab + 2; k++) {
                uint32_t off = offs[k], len; int is_rep; int32_t sc;
                if (off == 0 || off > p || off > c->window) continue;
                is_rep = off == erep0 || off == erep1;
                /* with 64 bytes in hand already, a far candidate is not looked at: comparing it would be the longest compare of the tile */
                if (k >= 1 && k < ntab && P->far_skip && best_len >= (uint32_t)P->far_skip) continue;
                len = (P->near16 && k == 0) ? prev_near : match_len(src, p, p - off, cap);
                if (len < (uint32_t)(is_rep ? P->min_rep : P->min_match)) continue;
                sc = score_of(P, len, off, is_rep);
                if (sc > best_score) { best_score = sc; best_len = len; best_off = off; best_rep = is_rep; best_far = k >= 1 && k < ntab; best_fol = k == 0 && fol; }
            }
            /* A far candidate that WON with all `cap` bytes equal is compared on, up to far_cap bytes (the kernel: the whole wave, one
             * trip): a long repeat MiB back is found at one sampled position, and every piece boundary would have to find it again. */
            if (P->far_cap > P->cap && best_far && best_len == (uint32_t)P->cap && limit > (uint32_t)P->cap) {
                best_len = match_len(src, p, p - best_off, limit < (uint32_t)P->far_cap ? limit : (uint32_t)P->far_cap);
                best_score = score_of(P, best_len, best_off, best_rep);
            }
            /* Continuation guess (cont_cap > 0): when the last selected match of the previous searched tile ends exactly at the parse
             * cursor, the position at the cursor tries that match's offset once more, over cont_cap bytes -- a match cut at `cap` (or
             * far_cap) goes on at the same offset however far back its source lies (the recent-offset guesses above only look inside the
             * staged window); the entropy stage joins the pieces into one sequence.  Ranks last, no backward extension (the bytes in front
             * of it are the previous piece). */
            if (P->cont_cap > 0 && cont_off && cont_pos == p && cont_off <= p && cont_off <= c->window) {
                uint32_t ccap = limit < (uint32_t)P->cont_cap ? limit : (uint32_t)P->cont_cap, len = match_len(src, p, p - cont_off, ccap);
                if (len >= (uint32_t)P->min_rep) {
                    int32_t sc = score_of(P, len, cont_off, 1);
                    if (sc > best_score) { best_score = sc; best_len = len; best_off = cont_off; best_rep = 1; best_far = 0; best_fol = 1; }
                }
            }
            if (best_len && best_score > 0) {
                uint32_t back = 0, back_cap = best_far ? (uint32_t)P->far_back : (uint32_t)P->back_cap;
                m->len = best_len; m->off = best_off; m->is_rep = (uint8_t)best_rep;
                /* backward extension potential: equal bytes just before the match and its source */
                if (p - best_off >= back_cap && !best_fol) /* same rule for recent-offset guesses: no extension next to the frame start */
                    while (back < back_cap && p - back > bs && p - back > best_off &&
                           src[p - back - 1] == src[p - back - 1 - best_off]) back++;
                m->back = (uint8_t)back;
            }
        }
        /* far inserts: the tile in front of this one (if it was searched) now, this tile's after the next tile's lookups */
        if (P->far_log) {
            if (c->far_pending != ZGE_NO_TILE) far_insert_tile(c, c->far_pending);
            c->far_pending = tile;
        }
        {
            int any = 0;
            for (t = 0; t < tcount; t++) any |= c->M[t].len != 0;
            tile_any = any;
            if (any) c->cold = 0;
            else {
                c->cold++;
                if (c->cold >= 2) c->skip_left = c->cold >= 4 ? 7u : (1u << (c->cold - 1)) - 1;
            }
        }
        {
        uint32_t it;
        for (it = 0; ; it++) {
        /* S4: backward propagation -- position t may start the match of t+k, k bytes earlier */
        for (t = 0; t < tcount; t++) {
            cand best = c->M[t];
            int32_t best_score = best.len ? score_of(P, best.len, best.off, best.is_rep) : 0;
            uint32_t k;
            for (k = 1; k <= (uint32_t)(P->far_back > P->back_cap ? P->far_back : P->back_cap) && t + k < tcount; k++) {
                const cand *nb = &c->M[t + k];
                int32_t sc;
                if (!nb->len || nb->back < k) continue;
                sc = score_of(P, nb->len + k, nb->off, nb->is_rep);
                if (sc > best_score) { best_score = sc; best.len = nb->len + k; best.off = nb->off; best.is_rep = nb->is_rep; }
            }
            best.back = 0;
            c->M2[t] = best;
        }
        /* S5: take flags (one-byte lazy lookahead inside the tile) and successors */
        for (t = 0; t < tcount; t++) {
            const cand *m = &c->M2[t];
            int tk = m->len != 0;
            if (tk && P->lazy && t + 1 < tcount && c->M2[t + 1].len) {
                const cand *m2 = &c->M2[t + 1];
                if (score_of(P, m2->len, m2->off, m2->is_rep) > score_of(P, m->len, m->off, m->is_rep) + P->lazy_delta) tk = 0;
            }
            if (tk && P->lazy2_delta && t + 2 < tcount && c->M2[t + 2].len) { /* two bytes ahead (libzstd's lazy2) */
                const cand *m3 = &c->M2[t + 2];
                if (score_of(P, m3->len, m3->off, m3->is_rep) > score_of(P, m->len, m->off, m->is_rep) + P->lazy2_delta) tk = 0;
            }
            c->take[t] = (uint8_t)tk;
            c->next[t] = tk ? t + m->len : t + 1;
            c->mark[t] = 0;
        }
        /* S6: the parse path from the entry cursor (pointer doubling in the kernel) */
        t = (uint32_t)((pos > tile ? pos : tile) - tile);
        while (t < tcount) { c->mark[t] = 1; t = c->next[t]; }
        if (it >= (uint32_t)P->rep_pass || !tile_any) break;
        /* Live recent offsets (rep_pass > 0, the level >= 9 finder): libzstd's lazy parsers try the offsets of the matches they took
         * last at every position; the guesses above are the offsets the PREVIOUS tile ended with.  With a parse of the tile in hand,
         * every position tries the last two different offsets of the selected matches on the path in front of it (before the tile's
         * first ones: the entry guesses) as two more candidates at recent-offset cost, and the tile is propagated and parsed again;
         * rep_pass rounds.  Same rules as a table candidate: 8 bytes in front of the source, at most back_cap bytes of backward
         * extension; of two candidates with the same score the more recent offset stays.  Only positions ON the path (literals and
         * selected matches) and the position after one (same 64-position chunk) are tried: the positions inside a selected match
         * would compare the rest of that very match at its own offset -- most of the work for none of the gain (model: elf slices
         * +0.1 %). */
        {
            uint32_t live = erep0, live1 = erep1, lv;
            for (t = 0; t < tcount; t++) {
                size_t p = tile + t;
                for (lv = 0; lv < 2; lv++) {
                uint32_t lo = lv ? live1 : live;
                if (lv && live1 == live) continue;
                if (P->live_reps && lo && lo + 8 <= p && lo <= c->window && (c->mark[t] || ((t & 63) != 0 && c->mark[t - 1]))) {
                    cand *m = &c->M[t];
                    uint32_t limit = (uint32_t)(be - p), cap = limit < (uint32_t)P->cap ? limit : (uint32_t)P->cap;
                    uint32_t len = match_len(src, p, p - lo, cap);
                    if (len >= (uint32_t)P->min_rep) {
                        int32_t sc = score_of(P, len, lo, 1), cur = m->len ? score_of(P, m->len, m->off, m->is_rep) : 0;
                        if (sc > cur) {
                            uint32_t back = 0;
                            m->len = len; m->off = lo; m->is_rep = 1;
                            while (back < (uint32_t)P->back_cap && p - back > bs && p - back > lo && src[p - back - 1] == src[p - back - 1 - lo]) back++;
                            m->back = (uint8_t)back;
                        }
                    }
                }
                }
                /* a SELECTED match that was cut at the cap goes on at its offset (ext_cap bytes at most): the pieces of a long repeat stay
                 * one match even where the next piece's position would not have found the offset again (the kernel: one trip of the
                 * whole wave per such match) */
                if (P->ext_cap > 0 && c->mark[t] && c->take[t] && c->M2[t].len >= (uint32_t)P->cap) {
                    uint32_t limit = (uint32_t)(be - p), xcap = limit < (uint32_t)P->ext_cap ? limit : (uint32_t)P->ext_cap;
                    uint32_t xl = xcap > c->M2[t].len ? match_len(src, p, p - c->M2[t].off, xcap) : 0;
                    if (xl > c->M2[t].len) { cand *m = &c->M[t]; m->len = xl; m->off = c->M2[t].off; m->is_rep = c->M2[t].is_rep; m->back = 0; }
                }
                if (c->mark[t] && c->take[t] && c->M2[t].off != live) { live1 = live; live = c->M2[t].off; }
            }
        }
        }
        }
        pos = tile + t;
        /* S7: emission in position order (prefix sums in the kernel) */
        {
            uint32_t last0 = 0, last1 = 0, nsel = 0;
            for (t = 0; t < tcount; t++) {
                if (!c->mark[t]) continue;
                if (c->take[t]) {
                    seq[nseq].ll = (uint32_t)(lp - prev_lp); seq[nseq].ml = c->M2[t].len; seq[nseq].off = c->M2[t].off; seq[nseq].ofv = 0;
                    prev_lp = lp;
                    nseq++;
                    last1 = last0; last0 = c->M2[t].off; nsel++; cont_end = tile + t + c->M2[t].len;
                } else {
                    lit[lp++] = src[tile + t];
                }
            }
            /* offset guesses for the next tile: offsets of the last two matches selected so far */
            cont_off = (nsel && cont_end == pos) ? last0 : 0; cont_pos = pos;
            if (nsel >= 2) { erep0 = last0; erep1 = last1; }
            else if (nsel == 1) { erep1 = erep0; erep0 = last0; }
        }
    }
    (void)i;
    c->erep0 = erep0; c->erep1 = erep1;
    *nlit = lp;
    nseq = merge_sequences(seq, nseq);
    resolve_repcodes(seq, nseq, c->st);
    return nseq;
}

/* ------------------------------------------------------------------ frame --------------------- */
size_t zge_bound(size_t n)
{
    size_t blocks = (n + ZGE_BLOCK - 1) / ZGE_BLOCK;
    if (blocks == 0) blocks = 1;
    return n + 3 * blocks + 18;
}

void zge_default_params(zge_params *P, int level)
{
    memset(P, 0, sizeof *P);
    P->level = level;
    P->checksum = 1;
    P->long_log = 13; P->short_log = 13; P->short_bytes = 5; P->tag_bits = 10; P->seg_log = 21; P->rep_back = 256;
    P->tile = 1024; P->sub = 64; P->cap = 256;
    P->min_match = 5; P->min_rep = 3; P->rep_search = 2;
    P->back_cap = 8; P->lazy = level >= 2 || level == 0 ? 1 : 0; P->lazy_delta = 5; /* engine.hip: derive_params */
    P->lit_cost = 5; P->match_cost = 12; P->rep_cost = 9;
    P->window_log = 21; P->short_window_log = 30;
    P->far_log = 16; P->far_ways = 1; P->far_step_log = 5; P->far_res_log = 2; P->far_short = 0; P->far_skip = 64; P->far_back = 48;
    P->far_min_frame = 65536;
    /* long matches: one more parse round per tile in which selected matches that were cut at `cap` go on at their offset (no live
     * recent offsets below level 9); the kernel runs the round only in tiles that have such a match */
    P->rep_pass = 1; P->live_reps = 0; P->ext_cap = 960;
    P->seq_repeat = 1; /* round 4: the blocks of a group share their sequence tables (seq_plan_group) */
    P->near16 = 1; P->short_log = 15; P->far_cdc_log = 4; /* round 3: one 16-bit near table of 2^15 entries, content-defined far sampling */
    if (level >= 9) { /* the engine's deep finder (engine.hip: derive_params) */
        P->near16 = 0; P->far_cdc_log = 0; P->far_min_frame = 0;
        P->long_log = 14; P->short_log = 14; P->short_bytes = 4; P->min_match = 4; P->match_cost = 10; P->window_log = 22;
        P->far_log = 16; P->far_ways = 2; P->far_step_log = 1; P->far_res_log = 0; P->far_short = 1; P->far_skip = 0; P->far_back = 48;
        /* round 3: the parse -- live recent offsets (two rounds), a second lazy step, literals priced at 6; near tables of 2^13 entries
         * (the far tables hold what they forget: no ratio lost on any item of tests/support/realdata.py, and two workgroups fit a CU) */
        P->rep_pass = 2; P->live_reps = 1; P->lazy2_delta = 5; P->lit_cost = 6; P->long_log = 13; P->short_log = 13;
        P->cont_cap = 960; /* a match cut at `cap` goes on at the tile's cursor (the level-3 kernel has no register to spare for it: DESIGN.md 4.1) */
        if (level >= 15) P->rep_pass = 4; /* round 4: levels 15 .. 22 run two more rounds of the live recent-offset pass */
    }
    if (level != 0 && level <= 1) { /* round 4: level 1 and the negative levels -- the near table only: no far table, no lazy step, no extension round */
        P->far_log = 0; P->rep_pass = 0; P->lazy = 0;
    }
}

int zge_encode_frame(const zge_params *P_in, const void *src_, size_t n, void *dst_, size_t cap,
                     size_t *out_len, zge_stats *st)
{
    zge_params Pn = *P_in;
    const zge_params *P = &Pn;
    const uint8_t *src = (const uint8_t *)src_;
    uint8_t *dst = (uint8_t *)dst_;
    size_t pos = 0, bs, gs;
    mf_ctx c;
    zge_seq *gseq[ZGE_TABLE_GROUP];
    uint8_t *lit, *gblk[ZGE_TABLE_GROUP];
    seq_plan *plans;
    int wlog, single, bad_bound = 0, g;
    if (cap < zge_bound(n)) return -1;
    if (st) memset(st, 0, sizeof *st);
    if (n <= (size_t)Pn.far_min_frame) Pn.far_log = 0; /* small frames: no far table */
    /* frame header */
    dst[pos++] = 0x28; dst[pos++] = 0xB5; dst[pos++] = 0x2F; dst[pos++] = 0xFD;
    wlog = P->window_log;
    single = n <= ((size_t)1 << wlog);
    {
        int fcs_flag = n < 256 ? 0 : (n < 65536 + 256 ? 1 : (n <= 0xFFFFFFFFu ? 2 : 3));
        int i, fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
        uint64_t v = fcs_flag == 1 ? n - 256 : n;
        if (!single && fcs_flag == 0) { fcs_flag = 1; fcs_bytes = 2; v = n - 256; } /* unreachable: !single => n > 2^wlog */
        dst[pos++] = (uint8_t)((fcs_flag << 6) | (single << 5) | ((P->checksum ? 1 : 0) << 2));
        if (!single) dst[pos++] = (uint8_t)((wlog - 10) << 3);
        for (i = 0; i < fcs_bytes; i++) dst[pos++] = (uint8_t)(v >> (8 * i));
    }
    c.P = P; c.src = src; c.n = n; c.st = st; c.cold = 0; c.skip_left = 0; c.erep0 = 0; c.erep1 = 0; c.far_pending = ZGE_NO_TILE;
    c.window = single ? (n ? n : 1) : ((size_t)1 << wlog);
    {
        const size_t fw = P->far_log ? ((size_t)1 << P->far_log) * (size_t)P->far_ways : 1;
        if (P->far_log && (P->far_ways < 1 || P->far_ways * (1 + (P->far_short ? 1 : 0)) > ZGE_FAR_MAX)) return -2;
        c.fl = (uint32_t *)calloc(fw, 4);
        c.fs = (uint32_t *)calloc(fw, 4);
        c.farc = (uint32_t *)calloc((size_t)P->tile * ZGE_FAR_MAX, 4);
    }
    c.t16 = (uint16_t *)calloc((size_t)1 << P->short_log, 2);
    c.tl = (uint32_t *)calloc((size_t)1 << P->long_log, 4);
    c.ts = (uint32_t *)calloc((size_t)1 << P->short_log, 4);
    c.M = (cand *)calloc((size_t)P->tile, sizeof(cand));
    c.M2 = (cand *)calloc((size_t)P->tile, sizeof(cand));
    c.next = (uint32_t *)calloc((size_t)P->tile, 4);
    c.take = (uint8_t *)calloc((size_t)P->tile, 1);
    c.mark = (uint8_t *)calloc((size_t)P->tile, 1);
    for (g = 0; g < ZGE_TABLE_GROUP; g++) {
        gseq[g] = (zge_seq *)malloc(sizeof(zge_seq) * (ZGE_BLOCK / 3 + 8));
        gblk[g] = (uint8_t *)malloc(ZGE_BLOCK + 1024);
    }
    plans = (seq_plan *)calloc(ZGE_TABLE_GROUP, sizeof *plans);
    lit = (uint8_t *)malloc(ZGE_BLOCK + 64);
    if (n == 0) { dst[pos++] = 1; dst[pos++] = 0; dst[pos++] = 0; }
    for (gs = 0; gs < n; gs += (size_t)ZGE_TABLE_GROUP * ZGE_BLOCK) {
      /* first pass over the group's blocks: match finding, literals section, code histograms and each block's own table choices */
      int nb = 0, rle[ZGE_TABLE_GROUP];
      size_t nlits[ZGE_TABLE_GROUP];
      for (bs = gs; bs < n && nb < ZGE_TABLE_GROUP; bs += ZGE_BLOCK, nb++) {
        size_t be = bs + ZGE_BLOCK < n ? bs + ZGE_BLOCK : n, blen = be - bs, nlit = 0, i;
        int all_same = 1;
        /* table positions are relative to 2^seg_log segments (multiples of the block size): at a boundary the
         * tables are cleared, so candidates never cross it (only frames larger than a segment notice) */
        if (bs > 0 && (bs & (((size_t)1 << P->seg_log) - 1)) == 0) {
            if (P->far_log) {
                c.far_pending = ZGE_NO_TILE; /* inserts of the old segment's last tile die with its tables */
                memset(c.fl, 0, (sizeof(uint32_t) << P->far_log) * (size_t)P->far_ways);
                memset(c.fs, 0, (sizeof(uint32_t) << P->far_log) * (size_t)P->far_ways);
            }
            memset(c.tl, 0, sizeof(uint32_t) << P->long_log);
            memset(c.ts, 0, sizeof(uint32_t) << P->short_log);
            memset(c.t16, 0, sizeof(uint16_t) << P->short_log);
            /* in a frame larger than ZGE_SPLIT_MIN a segment is a unit of work of its own in the engine (zge_match.hip: one workgroup
             * per segment, so that the segments of a large frame are searched side by side): what the finder carries from tile to
             * tile starts afresh as well */
            if (n > ZGE_SPLIT_MIN) { c.erep0 = c.erep1 = 0; c.cold = 0; c.skip_left = 0; }
        }
        for (i = 1; i < blen; i++) if (src[bs + i] != src[bs]) { all_same = 0; break; }
        rle[nb] = all_same && blen >= 2;
        memset(&plans[nb], 0, sizeof plans[nb]);
        nlits[nb] = 0;
        if (rle[nb]) continue; /* RLE block: the engine inserts nothing for it (cheap and deterministic) */
        {
            const uint32_t nseq = matchfind_block(&c, bs, be, gseq[nb], lit, &nlit);
            const size_t lsz = encode_literals(lit, nlit, gblk[nb], ZGE_BLOCK + 1024, st);
            seq_plan_block(&plans[nb], gseq[nb], nseq, lsz, blen);
            nlits[nb] = nlit;
        }
      }
      if (P->seq_repeat) seq_plan_group(plans, nb);
      /* second pass: the sequences sections, and the blocks in order */
      for (g = 0, bs = gs; g < nb; g++, bs += ZGE_BLOCK) {
        const size_t be = bs + ZGE_BLOCK < n ? bs + ZGE_BLOCK : n, blen = be - bs;
        const int last = be == n;
        uint32_t hdr;
        if (rle[g]) {
            hdr = (uint32_t)last | (1u << 1) | ((uint32_t)blen << 3);
            dst[pos++] = (uint8_t)hdr; dst[pos++] = (uint8_t)(hdr >> 8); dst[pos++] = (uint8_t)(hdr >> 16);
            dst[pos++] = src[bs];
            if (st) st->blk_rle++;
            continue;
        }
        {
            const size_t lsz = plans[g].lsz;
            size_t ssz = 0, csz;
            uint8_t *blk = gblk[g];
            if (lsz) ssz = encode_sequences(gseq[g], plans[g].nseq, blk + lsz, ZGE_BLOCK + 1024 - lsz, st, &plans[g]);
            csz = lsz && ssz ? lsz + ssz : 0;
            if (plans[g].guaranteed && !(csz && csz < blen)) bad_bound = 1; /* the size bound that lets a successor repeat this block's tables was no bound */
            if (csz && csz < blen) {
                hdr = (uint32_t)last | (2u << 1) | ((uint32_t)csz << 3);
                dst[pos++] = (uint8_t)hdr; dst[pos++] = (uint8_t)(hdr >> 8); dst[pos++] = (uint8_t)(hdr >> 16);
                memcpy(dst + pos, blk, csz);
                pos += csz;
                if (st) { st->blk_comp++; st->lit_bytes += nlits[g]; st->lit_section += lsz; st->seq_section += ssz; }
            } else {
                /* raw block: the decoder's repcode history is not advanced by it (see matchfind_block) */
                hdr = (uint32_t)last | (0u << 1) | ((uint32_t)blen << 3);
                dst[pos++] = (uint8_t)hdr; dst[pos++] = (uint8_t)(hdr >> 8); dst[pos++] = (uint8_t)(hdr >> 16);
                memcpy(dst + pos, src + bs, blen);
                pos += blen;
                if (st) st->blk_raw++;
            }
        }
      }
    }
    if (P->checksum) {
        uint32_t x = (uint32_t)oracle_xxh64(src, n, 0);
        dst[pos++] = (uint8_t)x; dst[pos++] = (uint8_t)(x >> 8); dst[pos++] = (uint8_t)(x >> 16); dst[pos++] = (uint8_t)(x >> 24);
    }
    free(c.t16); free(c.fl); free(c.fs); free(c.farc); free(c.tl); free(c.ts); free(c.M); free(c.M2); free(c.next); free(c.take); free(c.mark); free(lit); free(plans);
    for (g = 0; g < ZGE_TABLE_GROUP; g++) { free(gseq[g]); free(gblk[g]); }
    *out_len = pos;
    return bad_bound ? -3 : 0;
}
