// zarc_amd/csrc/corpus.h -- deterministic synthetic corpus.
//
// One generator, integer arithmetic only, compiled for the host (tests, CPU baseline) and for the
// device (bench.py fills HBM directly, so no host staging of 10 GiB).  Entry i of a batch uses
//   seed = 0x5A41524300000000 ^ i        kind = i mod 4
//   K0 text    : Zipf-like words from a 4096-word lowercase vocabulary, joined by spaces
//   K1 records : 64-byte records {u32 counter, 8 B of 2-bit noise, 4 B slow field, 16 B zero,
//                8 B random, 24 B zero}
//   K2 lzgen   : literal runs over a skewed 64-symbol alphabet alternating with copies from the
//                last 32 KiB (geometric lengths, mean about 12)
//   K3 random  : raw splitmix64 output (incompressible -> raw blocks)
// Each entry is generated serially from its own counter-based stream, so any entry can be produced
// independently on any device.
#ifndef ZARC_CORPUS_H
#define ZARC_CORPUS_H
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ZARC_HD __host__ __device__ inline
#else
#define ZARC_HD static inline
#endif

#define ZARC_CORPUS_SEED_BASE 0x5A41524300000000ULL

ZARC_HD uint64_t zarc_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef struct { uint64_t s; } zarc_rng;
ZARC_HD uint64_t zarc_next(zarc_rng *r)
{
    r->s += 0x9E3779B97F4A7C15ULL;
    return zarc_mix64(r->s);
}

// piecewise-uniform Zipf(~1.3)-like rank in [0,4096): bucket b covers ranks [2^b - 1, 2^(b+1) - 1)
ZARC_HD uint32_t zarc_zipf_rank(zarc_rng *r)
{
    // cumulative bucket thresholds out of 65536 for buckets 0..11 (s ~ 1.1-1.3 shaped)
    const uint16_t cdf[12] = {9200, 18000, 26000, 33200, 39700, 45500, 50600, 55000,
                              58700, 61700, 64000, 65535};
    uint64_t x = zarc_next(r);
    uint32_t u = (uint32_t)(x & 0xFFFF), b = 0, lo, span;
    while (b < 11 && u >= cdf[b]) b++;
    lo = (1u << b) - 1;
    span = 1u << b;
    return (lo + (uint32_t)((x >> 16) % span)) & 4095u;
}

ZARC_HD size_t zarc_gen_text(uint8_t *dst, size_t n, zarc_rng *r)
{
    size_t pos = 0;
    while (pos < n) {
        uint32_t w = zarc_zipf_rank(r);
        uint64_t h = zarc_mix64(0xC0FFEE00ULL + w);
        uint32_t len = 2 + (uint32_t)(h & 7), j;
        h >>= 3;
        for (j = 0; j < len && pos < n; j++) {
            // skewed letter choice: 5 bits -> 26 letters with common ones doubled
            const char *alpha = "etaoinshrdlucmfwypvbgkqjxzetaoin";
            dst[pos++] = (uint8_t)alpha[h & 31];
            h >>= 5;
        }
        if (pos < n) {
            uint64_t p = zarc_next(r) & 63;
            dst[pos++] = p == 0 ? (uint8_t)'\n' : (p < 4 ? (uint8_t)',' : (uint8_t)' ');
        }
    }
    return pos;
}

ZARC_HD size_t zarc_gen_records(uint8_t *dst, size_t n, zarc_rng *r)
{
    size_t pos = 0;
    uint32_t counter = (uint32_t)zarc_next(r), slow = (uint32_t)zarc_next(r);
    while (pos < n) {
        uint8_t rec[64];
        uint64_t noise = zarc_next(r), rnd = zarc_next(r);
        int j;
        for (j = 0; j < 64; j++) rec[j] = 0;
        rec[0] = (uint8_t)counter; rec[1] = (uint8_t)(counter >> 8);
        rec[2] = (uint8_t)(counter >> 16); rec[3] = (uint8_t)(counter >> 24);
        for (j = 0; j < 8; j++) rec[4 + j] = (uint8_t)((noise >> (2 * j)) & 3);
        rec[12] = (uint8_t)slow; rec[13] = (uint8_t)(slow >> 8);
        rec[14] = (uint8_t)(slow >> 16); rec[15] = (uint8_t)(slow >> 24);
        for (j = 0; j < 8; j++) rec[32 + j] = (uint8_t)(rnd >> (8 * j));
        for (j = 0; j < 64 && pos < n; j++) dst[pos++] = rec[j];
        counter++;
        if ((noise >> 60) == 0) slow += 1 + (uint32_t)((noise >> 40) & 0xFF);
    }
    return pos;
}

ZARC_HD size_t zarc_gen_lz(uint8_t *dst, size_t n, zarc_rng *r)
{
    size_t pos = 0;
    while (pos < n) {
        uint64_t x = zarc_next(r);
        // literal run: length 1..16 (skewed short), symbols from a skewed 64-symbol alphabet
        uint32_t ll = 1 + (uint32_t)(x & 7) + (((x >> 3) & 3) == 0 ? (uint32_t)((x >> 5) & 7) : 0);
        uint32_t j;
        for (j = 0; j < ll && pos < n; j++) {
            uint64_t y = zarc_next(r);
            uint32_t a = (uint32_t)(y & 63), b = (uint32_t)((y >> 6) & 63);
            dst[pos++] = (uint8_t)(0x20 + (a < b ? a : b)); // min of two uniforms: skewed
        }
        if (pos >= 64 && pos < n) {
            uint64_t y = zarc_next(r);
            // copy: length 4 + geometric-ish (mean about 12), offset log-uniform up to 32 KiB
            uint32_t ml = 4, obits, off;
            uint64_t g = y >> 20;
            while ((g & 7) != 0 && ml < 120) { ml += 1 + (uint32_t)((g >> 3) & 1); g >>= 4; if (!g) break; }
            obits = 6 + (uint32_t)((y & 0xFF) % 10);           // 6..15
            off = 1 + (uint32_t)(((y >> 8) & 0xFFF) % ((1u << obits) - 1));
            if (off > pos) off = (uint32_t)pos;
            for (j = 0; j < ml && pos < n; j++, pos++) dst[pos] = dst[pos - off];
        }
    }
    return pos;
}

ZARC_HD size_t zarc_gen_random(uint8_t *dst, size_t n, zarc_rng *r)
{
    size_t pos = 0;
    while (pos < n) {
        uint64_t x = zarc_next(r);
        int j;
        for (j = 0; j < 8 && pos < n; j++) dst[pos++] = (uint8_t)(x >> (8 * j));
    }
    return pos;
}

// Generate entry `index` of the corpus: n bytes at dst.  kind < 0 -> index mod 4.
ZARC_HD void zarc_corpus_entry(uint8_t *dst, size_t n, uint64_t index, int kind)
{
    zarc_rng r;
    r.s = zarc_mix64(ZARC_CORPUS_SEED_BASE ^ index);
    if (kind < 0) kind = (int)(index & 3);
    switch (kind) {
    case 0: zarc_gen_text(dst, n, &r); break;
    case 1: zarc_gen_records(dst, n, &r); break;
    case 2: zarc_gen_lz(dst, n, &r); break;
    default: zarc_gen_random(dst, n, &r); break;
    }
}

#endif
