// zarc_amd/csrc/zstd_decode.hip -- batched Zstandard frame decoder for gfx950 (RFC 8878).
//
// Replaces libzstd's ZSTD_decompressStream as driven by the reference at
//   crates/zarc/src/decode/zstd_iterator.rs:88-153 (decompress_step) / :24-36 (one fresh DCtx per frame).
// Frames are independent (no dictionary, no cross-frame window).  Inside a frame the format is a chain of serial
// dependencies (backward bitstreams, FSE states, repeat offsets, LZ window), and a wave that decodes a bitstream on one
// lane spends 64 lanes' worth of issue slots on it.  So decoding is split by the kind of parallelism each part has:
//   zarc_zdec_scan      one LANE per frame: frame header, block headers, where each block's sections are, how many
//                       sequences and Huffman literal bytes it holds (the host sizes the scratch from these counts)
//   zarc_zdec_seqs      one LANE per block: FSE tables (built in HBM scratch) and the sequence bitstream -> 8-byte
//                       (literal length, match length, offset) records; repeat offsets resolved symbolically
//   zarc_zdec_literals  one wave per 16 blocks: canonical Huffman decoders in LDS, one LANE per stream -> literal bytes
//   zarc_zstd_decode    one persistent WAVE per frame (frames come from a queue): prefix sums give every sequence of a
//                       batch of 64 its literal and output position, every byte move is wave-cooperative or one
//                       16-byte head/tail pair per lane
// Anything unusual in a frame clears fast[frame]; zarc_zstd_decode then decodes that frame INLINE (tables in LDS, about
// 7.5 KiB per wave; bitstreams on lane 0 / lanes 0..3), which is also the only place where error statuses are decided.
// Algorithmic traffic per frame: C bytes read + N bytes written (+ the sequence / literal scratch round trip).
#include "zarc_device.h"
#include "zarc_kernels.h"

// timing-only ablation switches exist in the diagnostic build alone (make DIAG=1); in the product library they fold to zero
#ifdef ZARC_GPU_DIAG
#define ZDEC_DBG(x) (x)
#else
#define ZDEC_DBG(x) 0
#endif

namespace {

constexpr int BLOCK_MAX = 128 * 1024;
 // zarc_zdec_seqs_shared: tables per type a wave keeps in LDS
#ifdef ZARC_HIPEMU
typedef const uint16_t *ZDEC_LDS_TAB;
typedef const uint32_t *ZDEC_LDS_INFO;
#else
typedef const uint16_t __attribute__((address_space(3))) *ZDEC_LDS_TAB; // a pointer the compiler knows to be LDS: ds_read_u16, not flat loads
typedef const uint32_t __attribute__((address_space(3))) *ZDEC_LDS_INFO;
#endif
constexpr int SEQ_BATCH = 64;

// (literal-length / match-length baselines and extra bits are worked out arithmetically: ll_code_info / ml_code_info below)
__constant__ const int8_t D_LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                              2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ const int8_t D_ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                              1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                              1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ const int8_t D_OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                              1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

// ---- backward bit reader (one lane) -----------------------------------------------------------
struct BackBits {
    const uint8_t *p;
    int32_t bitpos; // unread bits below the cursor; negative = read past the beginning
    uint64_t c;     // container: the next k bits are (c << used) >> (64 - k)
    int32_t used;
    __device__ __forceinline__ void refill()
    {
        int32_t top_byte = (bitpos + 7) >> 3;
        if (top_byte >= 8) c = zd::load_u64(p + top_byte - 8);
        else if (top_byte > 0) c = zd::load_u64(p) << (8 * (8 - top_byte));
        else c = 0;
        used = 8 * top_byte - bitpos;
    }
    // returns false if the stream is malformed (empty or last byte zero)
    __device__ __forceinline__ bool init(const uint8_t *ptr, uint32_t len)
    {
        p = ptr;
        if (len == 0) return false;
        uint32_t last = ptr[len - 1];
        if (last == 0) return false;
        bitpos = (int32_t)(len - 1) * 8 + zd::hb32(last);
        refill();
        return true;
    }
    __device__ __forceinline__ uint32_t peek(int k) // 1 <= k <= 32
    {
        if (used + k > 64) refill();
        return (uint32_t)((c << used) >> (64 - k));
    }
    __device__ __forceinline__ void skip(int k) { used += k; bitpos -= k; }
    __device__ __forceinline__ uint32_t read(int k) // 0 <= k <= 32
    {
        if (k == 0) return 0;
        uint32_t v = peek(k);
        skip(k);
        return v;
    }
};

// ---- backward bit reader for the sequence stream (lane 0): same interface as BackBits, but refills come out of a
// three-word register window [B-16, B+8) that is reloaded one word ahead, so the L1/L2 latency of the bitstream is
// off the per-sequence dependency chain ----
struct SeqBits {
    const uint8_t *p;
    int32_t bitpos, used, B;
    uint64_t c, w_hi, w_mid, w_lo; // bytes [B,B+8), [B-8,B), [B-16,B-8) of the stream (zero below its start)
    __device__ __forceinline__ uint64_t load8z(int32_t idx) const
    {
        if (idx >= 0) return zd::load_u64(p + idx);
        if (idx > -8) return zd::load_u64(p) << (8 * (-idx));
        return 0;
    }
    __device__ __forceinline__ void refill()
    {
        const int32_t T = (bitpos + 7) >> 3; // the next bits end at byte T (exclusive)
        if (T < B) { w_hi = w_mid; w_mid = w_lo; B -= 8; w_lo = load8z(B - 16); }
        const int32_t s = B + 8 - T;         // 0..8 bytes of w_hi are already consumed
        c = s <= 0 ? w_hi : (s >= 8 ? w_mid : ((w_hi << (8 * s)) | (w_mid >> (64 - 8 * s))));
        used = 8 * T - bitpos;
    }
    __device__ __forceinline__ bool init(const uint8_t *ptr, uint32_t len)
    {
        p = ptr;
        if (len == 0) return false;
        const uint32_t last = ptr[len - 1];
        if (last == 0) return false;
        bitpos = (int32_t)(len - 1) * 8 + zd::hb32(last);
        const int32_t T = (bitpos + 7) >> 3;
        B = T - 8;
        w_hi = load8z(B); w_mid = load8z(B - 8); w_lo = load8z(B - 16);
        refill();
        return true;
    }
    __device__ __forceinline__ uint32_t read(int k) // 0 <= k <= 32
    {
        if (k == 0) return 0;
        if (used + k > 64) refill();
        const uint32_t v = (uint32_t)((c << used) >> (64 - k));
        used += k;
        bitpos -= k;
        return v;
    }
};

// ---- backward bit reader of zarc_zdec_seqs_lds: one 16-byte window per sequence, requested at a fixed point of the loop
// and consumed in two phases of at most 57 bits each; nothing in it depends on what other lanes have consumed ----
struct Bits128 {
    const uint8_t *p;
    int32_t bitpos; // unread bits below the cursor; negative = read past the beginning
    uint64_t hi, lo;
    int32_t u;      // bits of `hi` already consumed
    int32_t top;    // byte (exclusive) where the window ends
    __device__ __forceinline__ bool init(const uint8_t *ptr, uint32_t len)
    {
        p = ptr; hi = lo = 0; u = 0; top = 0;
        if (len == 0) return false;
        const uint32_t last = ptr[len - 1];
        if (last == 0) return false;
        bitpos = (int32_t)(len - 1) * 8 + zd::hb32(last);
        return true;
    }
    __device__ __forceinline__ void request() // the 128 bits that end at the cursor's byte; no branch, no wait
    {
        top = (bitpos + 7) >> 3;
        const int32_t base = top >= 16 ? top - 16 : 0;
        lo = zd::load_u64(p + base);
        hi = zd::load_u64(p + base + 8);
        u = 8 * top - bitpos;
    }
    // The window of a LATER cursor position, asked for now (ahead) and taken over when the cursor is there (adopt): the step of a sequence
    // knows how many bits it will consume as soon as its five table lookups are back, long before it has consumed them.
    __device__ __forceinline__ void ahead(int32_t at_bitpos, uint64_t &a_lo, uint64_t &a_hi, int32_t &a_top) const
    {
        a_top = (at_bitpos + 7) >> 3;
        const int32_t base = a_top >= 16 ? a_top - 16 : 0;
        a_lo = zd::load_u64(p + base);
        a_hi = zd::load_u64(p + base + 8);
    }
    __device__ __forceinline__ void adopt(uint64_t a_lo, uint64_t a_hi, int32_t a_top)
    {
        lo = a_lo; hi = a_hi; top = a_top;
        u = 8 * top - bitpos;
    }
    __device__ __forceinline__ void settle() // within 16 bytes of the stream's start: move the bytes up, zeros come in below
    {
        if (top < 16) {
            const int32_t sb = 16 - top; // bytes to shift by, >= 1
            if (sb >= 16) { hi = 0; lo = 0; }
            else if (sb >= 8) { hi = sb == 8 ? lo : lo << (8 * (sb - 8)); lo = 0; }
            else { hi = (hi << (8 * sb)) | (lo >> (64 - 8 * sb)); lo <<= 8 * sb; }
        }
    }
    __device__ __forceinline__ void second_phase() // bring the next 64 unread bits to the top of `hi`
    {
        if (u) hi = (hi << u) | (lo >> (64 - u));
        u = 0;
    }
    __device__ __forceinline__ uint32_t take(uint32_t k) // 0 <= k <= 32, within the phase's budget
    {
        const uint32_t t = (uint32_t)((hi << u) >> 32);
        const uint32_t v = (t >> 1) >> (31u - k); // (k = 0 -> 0 without a select; k <= 31 everywhere: table logs, offset codes <= 27, length codes <= 16 bits)
        u += (int32_t)k;
        bitpos -= (int32_t)k;
        return v;
    }
};

// literal-length / match-length code -> baseline and number of extra bits, in registers (RFC 8878 tables 15 and 16)
__device__ __forceinline__ void ll_code_info(uint32_t code, uint32_t &base, uint32_t &bits)
{
    if (code < 16) { base = code; bits = 0; }
    else if (code < 25) {
        const uint32_t i = code - 16;
        bits = (uint32_t)(0x433221111ull >> (4 * i)) & 15u;
        base = i == 8 ? 48u : (uint32_t)(0x28201C1816141210ull >> (8 * i)) & 0xFFu;
    } else { bits = code - 19; base = 1u << bits; }
}
__device__ __forceinline__ void ml_code_info(uint32_t code, uint32_t &base, uint32_t &bits)
{
    if (code < 32) { base = code + 3; bits = 0; }
    else if (code < 43) {
        const uint32_t i = code - 32;
        bits = (uint32_t)(0x54433221111ull >> (4 * i)) & 15u;
        base = i >= 8 ? 67u + 16u * (i - 8) : (uint32_t)(0x3B332F2B29272523ull >> (8 * i)) & 0xFFu;
    } else { bits = code - 36; base = (1u << bits) + 3; }
}

// ---- forward bit reader (one lane), for FSE table descriptions -----------------------------------
struct FwdBits {
    const uint8_t *p;
    uint32_t bitpos;
    __device__ __forceinline__ uint32_t peek(int k) { return (zd::load_u32(p + (bitpos >> 3)) >> (bitpos & 7)) & ((1u << k) - 1); }
};

// FSE decode cell, 16 bits: sym (6 bits) | x << 6, where x is the cell's state counter in [count, 2*count).  The
// number of bits to read and the new-state baseline follow from x and the table accuracy (nbits = al - floor(log2 x),
// base = (x << nbits) - 2^al), three ALU ops instead of two more bytes per cell: the three sequence tables take
// 2.5 KiB of LDS instead of 5, which buys decoder occupancy (the kernel is latency-bound: time ~ 1 / waves in flight).
__device__ __forceinline__ uint32_t cell_sym(uint32_t c) { return c & 63u; }
__device__ __forceinline__ uint32_t cell_nbits(uint32_t c, int al) { return (uint32_t)(al - zd::hb32(c >> 6)); }
__device__ __forceinline__ uint32_t cell_base(uint32_t c, int al) { return ((c >> 6) << cell_nbits(c, al)) - (1u << al); }

// Build an FSE decode table from normalized counts (lane-serial).  Returns false on bad distributions.
__device__ bool fse_build_dtable(uint16_t *tab, const int16_t *norm, int nsym, int al, uint16_t *next /*>=nsym*/)
{
    const int T = 1 << al;
    int high = T - 1;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { tab[high--] = (uint16_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    const int step = (T >> 1) + (T >> 3) + 3, mask = T - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tab[pos] = (uint16_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (int i = 0; i < T; i++) {
        const uint32_t s = tab[i];
        const uint32_t x = next[s]++;
        tab[i] = (uint16_t)(s | (x << 6));
    }
    return true;
}

// The same table with ONE scratch array: nn[] holds the normalised counts on entry and the symbols' state counters afterwards.  (For
// the lanes that build tables side by side with the scratch in LDS: two 64-entry arrays of a lane's own end up as registers selected by
// compare chains, ~130 instructions per access.)
// 64 scratch entries of one lane among its workgroup's: entry i of every lane side by side (a row per lane would put the same entry of
// all lanes on one LDS bank)
template <int STRIDE> struct Strided16 {
    int16_t *p;
    __device__ __forceinline__ int16_t &operator[](int i) const { return p[i * STRIDE]; }
};
template <class NN>
__device__ __forceinline__ bool fse_build_dtable_inplace(uint16_t *tab, NN nn, int nsym, int al)
{
    const int T = 1 << al;
    int high = T - 1;
    for (int s = 0; s < nsym; s++) if (nn[s] == -1) tab[high--] = (uint16_t)s;
    const int step = (T >> 1) + (T >> 3) + 3, mask = T - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        const int c = nn[s];
        for (int i = 0; i < c; i++) {
            tab[pos] = (uint16_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
        if (c == -1) nn[s] = 1;
    }
    if (pos != 0) return false;
    for (int i = 0; i < T; i++) {
        const uint32_t s = tab[i];
        const uint32_t x = (uint16_t)nn[s];
        nn[s] = (int16_t)(x + 1);
        tab[i] = (uint16_t)(s | (x << 6));
    }
    return true;
}

// Parse an FSE table description (lane-serial).  Returns bytes consumed, or -1.
// (NN: int16_t * or a Strided16 view of LDS scratch)
template <class NN>
__device__ int fse_read_desc(const uint8_t *src, uint32_t len, int max_al, int max_sym, NN norm, int *nsym_out, int *al_out)
{
    if (len == 0) return -1;
    FwdBits b = {src, 0};
    int al = (int)b.peek(4) + 5;
    b.bitpos += 4;
    if (al > max_al) return -1;
    int remaining = (1 << al) + 1, threshold = 1 << al, nb = al + 1, sym = 0;
    for (int i = 0; i <= max_sym; i++) norm[i] = 0;
    const uint32_t limit_bits = len * 8;
    while (remaining > 1 && sym <= max_sym) {
        if (b.bitpos + (uint32_t)nb > limit_bits + 7) return -1; // a valid description never needs this
        int max = 2 * threshold - 1 - remaining;
        int low = (int)b.peek(nb - 1), val;
        if (low < max) { val = low; b.bitpos += (uint32_t)(nb - 1); }
        else {
            val = (int)b.peek(nb);
            if (val >= threshold) val -= max;
            b.bitpos += (uint32_t)nb;
        }
        int count = val - 1;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        if (count == 0) {
            for (;;) {
                if (b.bitpos + 2 > limit_bits) return -1;
                int rep = (int)b.peek(2);
                b.bitpos += 2;
                sym += rep;
                if (rep != 3) break;
            }
            if (sym > max_sym + 1) return -1;
        }
        while (remaining < threshold) { nb--; threshold >>= 1; }
    }
    if (remaining != 1 || sym > max_sym + 1 || b.bitpos > limit_bits) return -1;
    *nsym_out = sym;
    *al_out = al;
    return (int)((b.bitpos + 7) >> 3);
}

// About 7.5 KiB per wave.  The table-construction scratch (weights, normalised counts)
// and the per-batch sequence buffer are never live at the same time and share storage.
constexpr uint32_t OBUF = 6144; // bytes of batch output staged in LDS on the fast path
struct Lds {
    union {
        struct {
            uint16_t huf[2048];   // sym | nbits << 8
            uint16_t ll[512], ml[512], of[256];
        };
        // Frames whose sequences and literals were decoded ahead need none of the tables: the same storage stages the output
        // of a batch of sequences, so near matches resolve at LDS speed and HBM sees whole lines
        alignas(16) uint8_t obuf[OBUF + 80];
    };
    union {
        struct {
            uint8_t weights[256];
            int16_t norm[64];
            uint16_t next[64];
            uint16_t wtab[64]; // FSE table for Huffman weights (accuracy <= 6)
        } b;
        uint32_t seq[SEQ_BATCH * 3];
    };
    int32_t ctrl[16];
};
enum { C_ERR = 0, C_HUF_BITS = 1, C_HUF_VALID = 2, C_LL_AL = 3, C_OF_AL = 4, C_ML_AL = 5, C_LL_OK = 6, C_OF_OK = 7, C_ML_OK = 8,
       C_TMP0 = 9, C_TMP1 = 10, C_TMP2 = 11 };

// wave-cooperative byte copy; src and dst never overlap forward within one call
__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, int lane)
{
    struct B16 { uint64_t a, b; };
    const uint32_t n16 = n / 16; // 16 bytes per lane and step (no alignment needed), then the tail
    for (uint32_t i = (uint32_t)lane; i < n16; i += 64) { B16 v; __builtin_memcpy(&v, src + 16 * i, 16); __builtin_memcpy(dst + 16 * i, &v, 16); }
    for (uint32_t i = n16 * 16 + (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
}

// One lane copies n <= 32 bytes (source and destination do not overlap): head and tail pieces of the largest power of two
// that fits, both read before either is written -- at most two loads and two stores per lane instead of one per byte.
__device__ __forceinline__ void lane_copy32(uint8_t *d, const uint8_t *s, uint32_t n)
{
    struct B16 { uint64_t a, b; };
    if (n >= 16) {
        B16 x, y;
        __builtin_memcpy(&x, s, 16); __builtin_memcpy(&y, s + n - 16, 16);
        __builtin_memcpy(d, &x, 16); __builtin_memcpy(d + n - 16, &y, 16);
    } else if (n >= 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, s, 8); __builtin_memcpy(&y, s + n - 8, 8);
        __builtin_memcpy(d, &x, 8); __builtin_memcpy(d + n - 8, &y, 8);
    } else if (n >= 4) {
        uint32_t x, y;
        __builtin_memcpy(&x, s, 4); __builtin_memcpy(&y, s + n - 4, 4);
        __builtin_memcpy(d, &x, 4); __builtin_memcpy(d + n - 4, &y, 4);
    } else if (n >= 2) {
        uint16_t x, y;
        __builtin_memcpy(&x, s, 2); __builtin_memcpy(&y, s + n - 2, 2);
        __builtin_memcpy(d, &x, 2); __builtin_memcpy(d + n - 2, &y, 2);
    } else if (n == 1) d[0] = s[0];
}

// Same shape for global -> LDS (8-byte pieces: unaligned ds_write_b64 is native, b128 is not)
__device__ __forceinline__ void lane_stage32(uint8_t *d, const uint8_t *s, uint32_t n)
{
    if (n >= 16) {
        uint64_t a, b, c, e;
        __builtin_memcpy(&a, s, 8); __builtin_memcpy(&b, s + 8, 8); __builtin_memcpy(&c, s + n - 16, 8); __builtin_memcpy(&e, s + n - 8, 8);
        __builtin_memcpy(d, &a, 8); __builtin_memcpy(d + 8, &b, 8); __builtin_memcpy(d + n - 16, &c, 8); __builtin_memcpy(d + n - 8, &e, 8);
    } else if (n >= 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, s, 8); __builtin_memcpy(&y, s + n - 8, 8);
        __builtin_memcpy(d, &x, 8); __builtin_memcpy(d + n - 8, &y, 8);
    } else if (n >= 4) {
        uint32_t x, y;
        __builtin_memcpy(&x, s, 4); __builtin_memcpy(&y, s + n - 4, 4);
        __builtin_memcpy(d, &x, 4); __builtin_memcpy(d + n - 4, &y, 4);
    } else if (n >= 2) {
        uint16_t x, y;
        __builtin_memcpy(&x, s, 2); __builtin_memcpy(&y, s + n - 2, 2);
        __builtin_memcpy(d, &x, 2); __builtin_memcpy(d + n - 2, &y, 2);
    } else if (n == 1) d[0] = s[0];
}

// The same copy in two halves, so that the loads of several copies are in flight together and the wave waits ONCE: lane_stage32
// keeps its loads inside the branch of its length class, and a wave with mixed lengths pays one HBM round trip per class.  Here the
// loads do not depend on the class -- 16 bytes at the start (up to 15 of them beyond the run: the caller checks that they exist) and
// the 16 bytes that end the run when it is longer than that -- and the classes only differ in the LDS stores.
struct Win32 { uint64_t a0, a1, b0, b1; };
__device__ __forceinline__ void lane_load32(Win32 &w, const uint8_t *s, uint32_t n)
{
    __builtin_memcpy(&w.a0, s, 8); __builtin_memcpy(&w.a1, s + 8, 8);
    const uint8_t *t = s + (n > 16 ? n - 16 : 0u);
    __builtin_memcpy(&w.b0, t, 8); __builtin_memcpy(&w.b1, t + 8, 8);
}
__device__ __forceinline__ void lane_store32(uint8_t *d, const Win32 &w, uint32_t n)
{
    if (n >= 16) {
        __builtin_memcpy(d, &w.a0, 8); __builtin_memcpy(d + 8, &w.a1, 8); __builtin_memcpy(d + n - 16, &w.b0, 8); __builtin_memcpy(d + n - 8, &w.b1, 8);
    } else if (n >= 8) {
        const uint32_t sh = 8 * (n - 8); // bytes [n-8, n) of the 16 loaded
        const uint64_t y = sh ? (w.a0 >> sh) | (w.a1 << (64 - sh)) : w.a0;
        __builtin_memcpy(d, &w.a0, 8); __builtin_memcpy(d + n - 8, &y, 8);
    } else if (n >= 4) {
        const uint32_t x = (uint32_t)w.a0, y = (uint32_t)(w.a0 >> (8 * (n - 4)));
        __builtin_memcpy(d, &x, 4); __builtin_memcpy(d + n - 4, &y, 4);
    } else if (n >= 2) {
        const uint16_t x = (uint16_t)w.a0, y = (uint16_t)(w.a0 >> (8 * (n - 2)));
        __builtin_memcpy(d, &x, 2); __builtin_memcpy(d + n - 2, &y, 2);
    } else if (n == 1) d[0] = (uint8_t)w.a0;
}

// LDS -> LDS, n <= 32, ranges do not overlap: same head/tail scheme
__device__ __forceinline__ void lane_move32(uint8_t *d, const uint8_t *s, uint32_t n)
{
    Win32 w; // one LDS round trip whatever the mix of lengths in the wave (the staging buffer has spare bytes behind it for the over-read)
    lane_load32(w, s, n);
    lane_store32(d, w, n);
}

// Huffman tree description -> weights in LDS, returns bytes consumed (or -1); lane-serial part on lane 0.
__device__ int huf_read_weights(Lds &L, const uint8_t *src, uint32_t len, int lane, int *nweights)
{
    if (len < 1) return -1;
    const uint32_t hb = src[0];
    int consumed, n;
    if (hb >= 128) {
        n = (int)hb - 127;
        consumed = 1 + (n + 1) / 2;
        if ((uint32_t)consumed > len) return -1;
        for (int i = lane; i < n; i += 64) {
            uint8_t byte = src[1 + i / 2];
            L.b.weights[i] = (i & 1) ? (byte & 15) : (byte >> 4);
        }
        zd::wave_sync();
    } else {
        if (hb == 0 || 1 + hb > len) return -1;
        if (lane == 0) {
            int nsym = 0, al = 0, cnt = -1;
            int used = fse_read_desc(src + 1, hb, 6, 63, L.b.norm, &nsym, &al); // weights are <= 11; 63 bounds L.b.norm
            BackBits b;
            if (used > 0 && (uint32_t)used < hb && fse_build_dtable(L.b.wtab, L.b.norm, nsym, al, L.b.next) &&
                b.init(src + 1 + used, hb - (uint32_t)used)) {
                uint32_t s1 = b.read(al), s2 = b.read(al);
                if (b.bitpos >= 0) {
                    cnt = 0;
                    for (;;) {
                        if (cnt > 253) { cnt = -1; break; }
                        L.b.weights[cnt++] = (uint8_t)cell_sym(L.b.wtab[s1]);
                        s1 = cell_base(L.b.wtab[s1], al) + b.read((int)cell_nbits(L.b.wtab[s1], al));
                        if (b.bitpos < 0) { L.b.weights[cnt++] = (uint8_t)cell_sym(L.b.wtab[s2]); break; }
                        if (cnt > 253) { cnt = -1; break; }
                        L.b.weights[cnt++] = (uint8_t)cell_sym(L.b.wtab[s2]);
                        s2 = cell_base(L.b.wtab[s2], al) + b.read((int)cell_nbits(L.b.wtab[s2], al));
                        if (b.bitpos < 0) { L.b.weights[cnt++] = (uint8_t)cell_sym(L.b.wtab[s1]); break; }
                    }
                }
            }
            L.ctrl[C_TMP0] = cnt;
        }
        zd::wave_sync();
        n = L.ctrl[C_TMP0];
        zd::wave_sync();
        if (n < 1) return -1;
        consumed = 1 + (int)hb;
    }
    *nweights = n;
    return consumed;
}

// weights[0..n) in LDS -> decode table.  Uniform; returns false on an invalid tree.
__device__ bool huf_build_table(Lds &L, int n, int lane)
{
    // sum of 2^(w-1)
    uint32_t part = 0;
    bool bad = false;
    for (int i = lane; i < n; i += 64) {
        uint32_t w = L.b.weights[i];
        if (w > 11) bad = true;
        else if (w) part += 1u << (w - 1);
    }
    const uint32_t sum = zd::wave_sum(part);
    if (zd::ballot(bad) != 0 || sum == 0) return false;
    const int max_bits = zd::hb32(sum) + 1;
    if (max_bits > 11) return false;
    const uint32_t left = (1u << max_bits) - sum;
    if (left & (left - 1)) return false;
    const uint32_t last_w = (uint32_t)zd::hb32(left) + 1;
    if (lane == 0) L.b.weights[n] = (uint8_t)last_w;
    zd::wave_sync();
    const int nsym = n + 1;
    // rank_start[w] for w = 1..max_bits, computed redundantly by every lane from ballots
    uint32_t run[12];
#pragma unroll
    for (int w = 0; w < 12; w++) run[w] = 0;
    uint32_t my_start[4] = {0, 0, 0, 0}, my_w[4] = {0, 0, 0, 0};
    // pass 1: counts per weight
    uint32_t cnt[12];
#pragma unroll
    for (int w = 0; w < 12; w++) cnt[w] = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int s = r * 64 + lane;
        const uint32_t w = s < nsym ? L.b.weights[s] : 0u;
        my_w[r] = w;
#pragma unroll
        for (int ww = 1; ww < 12; ww++) cnt[ww] += (uint32_t)__popcll(zd::ballot(w == (uint32_t)ww));
    }
    uint32_t start[12], pos = 0;
#pragma unroll
    for (int ww = 1; ww < 12; ww++) { start[ww] = pos; pos += cnt[ww] << (ww - 1); }
    if (pos != (1u << max_bits)) return false;
    // pass 2: each symbol's slot = start[w] + (symbols of the same weight before it) << (w-1)
    const uint64_t lt = (1ull << lane) - 1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t w = my_w[r];
#pragma unroll
        for (int ww = 1; ww < 12; ww++) {
            const uint64_t m = zd::ballot(w == (uint32_t)ww);
            if (w == (uint32_t)ww) my_start[r] = start[ww] + ((run[ww] + (uint32_t)__popcll(m & lt)) << (ww - 1));
            run[ww] += (uint32_t)__popcll(m);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t w = my_w[r];
        if (w) {
            const uint32_t lenr = 1u << (w - 1);
            const uint16_t e = (uint16_t)((uint32_t)(r * 64 + lane) | ((uint32_t)(max_bits + 1 - (int)w) << 8));
            for (uint32_t k = 0; k < lenr; k++) L.huf[my_start[r] + k] = e;
        }
    }
    if (lane == 0) { L.ctrl[C_HUF_BITS] = max_bits; L.ctrl[C_HUF_VALID] = 1; }
    zd::wave_sync();
    return true;
}

// One Huffman stream on the calling lane.  Returns false on corruption.
__device__ bool huf_decode_stream(const uint16_t *tab, int max_bits, const uint8_t *src, uint32_t len, uint8_t *out, uint32_t nout)
{
    BackBits b;
    if (!b.init(src, len)) return false;
    for (uint32_t i = 0; i < nout; i++) {
        const uint32_t e = tab[b.peek(max_bits)];
        b.skip((int)(e >> 8));
        out[i] = (uint8_t)e;
    }
    return b.bitpos == 0;
}

// (Re)build one sequence table according to its mode.  Uniform entry; the work runs on lane 0.
// Returns bytes consumed from src, or -1.
__device__ int seq_table(Lds &L, uint16_t *tab, int ctrl_al, int ctrl_ok, int mode, const uint8_t *src, uint32_t len,
                         const int8_t *def, int def_n, int def_al, int max_al, int max_sym, int lane)
{
    if (lane == 0) {
        int res = -1;
        if (mode == 0) {
            for (int i = 0; i < def_n; i++) L.b.norm[i] = def[i];
            if (fse_build_dtable(tab, L.b.norm, def_n, def_al, L.b.next)) { L.ctrl[ctrl_al] = def_al; L.ctrl[ctrl_ok] = 1; res = 0; }
        } else if (mode == 1) {
            if (len >= 1 && src[0] <= max_sym) { tab[0] = (uint16_t)(src[0] | (1u << 6)); L.ctrl[ctrl_al] = 0; L.ctrl[ctrl_ok] = 1; res = 1; }
        } else if (mode == 2) {
            int nsym = 0, al = 0;
            int used = fse_read_desc(src, len, max_al, max_sym, L.b.norm, &nsym, &al);
            if (used > 0 && fse_build_dtable(tab, L.b.norm, nsym, al, L.b.next)) { L.ctrl[ctrl_al] = al; L.ctrl[ctrl_ok] = 1; res = used; }
        } else {
            res = L.ctrl[ctrl_ok] ? 0 : -1;
        }
        L.ctrl[C_TMP0] = res;
    }
    zd::wave_sync();
    const int r = L.ctrl[C_TMP0];
    zd::wave_sync();
    return r;
}

// ---- decoder fast path: sequences entropy-decoded ahead of the frame pass ---------------------------------------
// The frame pass (one wave per frame) spends most of its time in the lane-0 FSE loop, where a whole wave issues
// instructions for one lane.  Sequence decoding of a block needs nothing from other blocks except (for Repeat mode)
// the table description of an earlier block, so it runs here with one LANE per block: 64 serial decoders per wave,
// each written as plain serial code, tables in HBM scratch (L2-resident).  Anything unusual (odd block structure,
// invalid descriptions, offsets beyond the format's window) clears fast[frame]: the frame pass then decodes that
// frame inline exactly as before, so error statuses are decided in one place.

// literals section header of a compressed block (same checks as the frame pass)
struct LitInfo { uint32_t ltype, lit_len, hdr, comp, streams, lused; };
__device__ bool parse_lit_section(const uint8_t *bp, uint32_t blen, LitInfo *o)
{
    if (blen < 2) return false;
    const uint32_t b0 = bp[0], ltype = b0 & 3, sf = (b0 >> 2) & 3;
    uint32_t lit_len, hdr, comp = 0, streams = 0, lused;
    if (ltype < 2) {
        if (sf == 0 || sf == 2) { lit_len = b0 >> 3; hdr = 1; }
        else if (sf == 1) { lit_len = (b0 >> 4) | ((uint32_t)bp[1] << 4); hdr = 2; }
        else { if (blen < 3) return false; lit_len = (b0 >> 4) | ((uint32_t)bp[1] << 4) | ((uint32_t)bp[2] << 12); hdr = 3; }
        if (lit_len > BLOCK_MAX) return false;
        if (ltype == 0) { if (hdr + lit_len > blen) return false; lused = hdr + lit_len; }
        else { if (hdr + 1 > blen) return false; lused = hdr + 1; }
    } else {
        if (blen < 5 && !(sf <= 1 && blen >= 3)) return false;
        if (sf <= 1) {
            const uint32_t v = b0 | ((uint32_t)bp[1] << 8) | ((uint32_t)bp[2] << 16);
            lit_len = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; hdr = 3; streams = sf == 0 ? 1 : 4;
        } else if (sf == 2) {
            const uint32_t v = b0 | ((uint32_t)bp[1] << 8) | ((uint32_t)bp[2] << 16) | ((uint32_t)bp[3] << 24);
            lit_len = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; hdr = 4; streams = 4;
        } else {
            const uint64_t v = (uint64_t)b0 | ((uint64_t)bp[1] << 8) | ((uint64_t)bp[2] << 16) | ((uint64_t)bp[3] << 24) | ((uint64_t)bp[4] << 32);
            lit_len = (uint32_t)(v >> 4) & 0x3FFFF; comp = (uint32_t)(v >> 22) & 0x3FFFF; hdr = 5; streams = 4;
        }
        if (lit_len > BLOCK_MAX || hdr + comp > blen) return false;
        lused = hdr + comp;
    }
    o->ltype = ltype; o->lit_len = lit_len; o->hdr = hdr; o->comp = comp; o->streams = streams; o->lused = lused;
    return true;
}

// where the three table descriptions of a block's sequences section sit (types 0 LL, 1 OF, 2 ML, in stream order)
struct SeqHeader { uint32_t mode[3], off[3], len[3], bits_off; };
__device__ bool scan_seq_header(const uint8_t *src, uint32_t hdr_off, uint32_t end, SeqHeader *h)
{
    if (hdr_off + 1 > end) return false;
    const uint32_t modes = src[hdr_off];
    if (modes & 3) return false;
    uint32_t q = hdr_off + 1;
    int16_t norm[64];
    for (int t = 0; t < 3; t++) {
        const uint32_t m = (modes >> (6 - 2 * t)) & 3;
        uint32_t len = 0;
        if (m == 1) len = 1;
        else if (m == 2) {
            int nsym = 0, al = 0;
            const int used = fse_read_desc(src + q, end - q, t == 1 ? 8 : 9, t == 0 ? 35 : (t == 1 ? 31 : 52), norm, &nsym, &al);
            if (used <= 0) return false;
            len = (uint32_t)used;
        }
        if (q + len > end) return false;
        h->mode[t] = m; h->off[t] = q; h->len[t] = len;
        q += len;
    }
    h->bits_off = q;
    return true;
}

// build the decode table of type t from a description found by scan_seq_header (mode 0, 1 or 2); returns the accuracy or -1
__device__ int build_seq_table(uint16_t *tab, int t, uint32_t mode, const uint8_t *desc, uint32_t len)
{
    int16_t norm[64];
    uint16_t next[64];
    const int max_sym = t == 0 ? 35 : (t == 1 ? 31 : 52);
    if (mode == 0) {
        const int8_t *def = t == 0 ? D_LL_DEFAULT : (t == 1 ? D_OF_DEFAULT : D_ML_DEFAULT);
        const int def_n = t == 0 ? 36 : (t == 1 ? 29 : 53), def_al = t == 1 ? 5 : 6;
        for (int i = 0; i < def_n; i++) norm[i] = def[i];
        return fse_build_dtable(tab, norm, def_n, def_al, next) ? def_al : -1;
    }
    if (mode == 1) {
        if (len < 1 || desc[0] > max_sym) return -1;
        tab[0] = (uint16_t)(desc[0] | (1u << 6));
        return 0;
    }
    int nsym = 0, al = 0;
    const int used = fse_read_desc(desc, len, t == 1 ? 8 : 9, max_sym, norm, &nsym, &al);
    if (used <= 0 || !fse_build_dtable(tab, norm, nsym, al, next)) return -1;
    return al;
}
// (scratch: 64 entries the caller provides, see fse_build_dtable_inplace)
template <class NN>
__device__ __forceinline__ int build_seq_table(uint16_t *tab, int t, uint32_t mode, const uint8_t *desc, uint32_t len, NN scratch)
{
    const int max_sym = t == 0 ? 35 : (t == 1 ? 31 : 52);
    if (mode == 0) {
        const int8_t *def = t == 0 ? D_LL_DEFAULT : (t == 1 ? D_OF_DEFAULT : D_ML_DEFAULT);
        const int def_n = t == 0 ? 36 : (t == 1 ? 29 : 53), def_al = t == 1 ? 5 : 6;
        for (int i = 0; i < def_n; i++) scratch[i] = def[i];
        return fse_build_dtable_inplace(tab, scratch, def_n, def_al) ? def_al : -1;
    }
    if (mode == 1) {
        if (len < 1 || desc[0] > max_sym) return -1;
        tab[0] = (uint16_t)(desc[0] | (1u << 6));
        return 0;
    }
    int nsym = 0, al = 0;
    const int used = fse_read_desc(desc, len, t == 1 ? 8 : 9, max_sym, scratch, &nsym, &al);
    if (used <= 0 || !fse_build_dtable_inplace(tab, scratch, nsym, al)) return -1;
    return al;
}

} // namespace

// One wave (64-thread workgroup) per frame.  order[] lists frame indices, largest first.
// one frame, one wave.  PRE: the frame's sequences and literals were decoded ahead (fast path); the two instantiations live in
// two kernels so that the fast one does not carry the registers of the inline decoder.
template <bool PRE>
__device__ __forceinline__ void decode_frame(Lds &L, const int lane, const uint32_t f, uint8_t *__restrict__ lit_buf,
                                             const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off,
                                             const uint64_t *__restrict__ frame_len, uint8_t *__restrict__ dst_base,
                                             const uint64_t *__restrict__ dst_off, const uint64_t *__restrict__ raw_len,
                                             int32_t *__restrict__ status, uint32_t *__restrict__ stored_checksum, const int dbg,
                                             const ZdecBlock *__restrict__ fblocks, const uint64_t *__restrict__ fseq_index,
                                             const uint64_t *__restrict__ seqs, const uint64_t *__restrict__ flit_index, const uint8_t *__restrict__ lits,
                                             const ZdecPiece piece /* PRE: the blocks of the frame this call decodes (the whole frame: first 0, count ~0) */)
{
    constexpr bool use_pre = PRE;
    const uint8_t *src = frames_base + frame_off[f];
    const uint32_t slen = (uint32_t)frame_len[f];
    uint8_t *out = dst_base + dst_off[f];
    const uint64_t cap = raw_len[f];
    int err = ZARC_FRAME_OK;

    if (lane == 0) {
        for (int i = 0; i < 16; i++) L.ctrl[i] = 0;
    }
    zd::wave_sync();

    // ---- frame header (every lane computes the same values) ----
    uint32_t pos = 0;
    uint32_t has_ck = 0;
    uint64_t window = 0, fcs = 0; // window is parsed for completeness; offsets are only checked against the output position
    bool have_fcs = false;
    if (slen < 6) err = ZARC_FRAME_SRCSIZE;
    else if (!(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD)) err = ZARC_FRAME_BAD_MAGIC;
    else {
        const uint32_t desc = src[4];
        const uint32_t fcs_flag = desc >> 6, ss = (desc >> 5) & 1, did_flag = desc & 3;
        has_ck = (desc >> 2) & 1;
        pos = 5;
        if (desc & 8) err = ZARC_FRAME_CORRUPT;
        const uint32_t did_bytes = did_flag == 3 ? 4u : did_flag;
        const uint32_t fcs_bytes = fcs_flag == 0 ? ss : (1u << fcs_flag);
        if (!err && pos + (ss ? 0u : 1u) + did_bytes + fcs_bytes > slen) err = ZARC_FRAME_SRCSIZE;
        if (!err) {
            if (!ss) {
                const uint32_t wd = src[pos++];
                window = 1ull << (10 + (wd >> 3));
                window += (window >> 3) * (wd & 7);
            }
            uint32_t did = 0;
            for (uint32_t i = 0; i < did_bytes; i++) did |= (uint32_t)src[pos + i] << (8 * i);
            pos += did_bytes;
            if (did != 0) err = ZARC_FRAME_UNSUPPORTED;
            for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
            pos += fcs_bytes;
            have_fcs = fcs_bytes != 0;
            if (ss) window = fcs;
            if (have_fcs && fcs != cap) err = ZARC_FRAME_SRCSIZE;
        }
    }
    err = (int)zd::uniform((uint32_t)err);

    // A piece starts at block piece.first with the output position and the repeat-offset history the host worked out from stage 2's
    // block summaries (engine.hip: a piece's blocks read nothing in front of it, so the pieces of a frame decode side by side).
    uint64_t opos = PRE ? piece.out_start : 0; // bytes produced
    uint32_t rep0 = PRE ? piece.rep[0] : 1, rep1 = PRE ? piece.rep[1] : 4, rep2 = PRE ? piece.rep[2] : 8;
    bool last = false;
    uint32_t bidx = PRE ? piece.first : 0; // block ordinal inside the frame (slot index of the fast path)
    uint32_t blocks_left = PRE ? piece.count : 0xFFFFFFFFu;
    if (PRE && piece.first) pos = fblocks[piece.first].payload - 3; // the piece's first block header
    while (!err && !last && blocks_left) {
        blocks_left--;
        const uint32_t my_b = bidx++;
        if (pos + 3 > slen) { err = ZARC_FRAME_SRCSIZE; break; }
        const uint32_t bh = zd::uniform((uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16));
        pos += 3;
        last = bh & 1;
        const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3 || bsize > BLOCK_MAX) { err = ZARC_FRAME_CORRUPT; break; }
        if (btype == 0) { // raw
            if (pos + bsize > slen) { err = ZARC_FRAME_SRCSIZE; break; }
            if (opos + bsize > cap) { err = ZARC_FRAME_DSTSIZE; break; }
            wave_copy(out + opos, src + pos, bsize, lane);
            opos += bsize;
            pos += bsize;
            zd::wave_sync_global(); // later blocks may copy from these bytes
            continue;
        }
        if (btype == 1) { // RLE
            if (pos + 1 > slen) { err = ZARC_FRAME_SRCSIZE; break; }
            if (opos + bsize > cap) { err = ZARC_FRAME_DSTSIZE; break; }
            const uint8_t v = src[pos];
            for (uint32_t i = (uint32_t)lane; i < bsize; i += 64) out[opos + i] = v;
            opos += bsize;
            pos += 1;
            zd::wave_sync_global();
            continue;
        }
        // ---- compressed block ----
        if (pos + bsize > slen) { err = ZARC_FRAME_SRCSIZE; break; }
        const uint8_t *bp = src + pos;
        const uint32_t blen = bsize;
        pos += bsize;
        if (blen < 2) { err = ZARC_FRAME_CORRUPT; break; }
        // literals section
        const uint32_t b0 = bp[0];
        const uint32_t ltype = b0 & 3, sf = (b0 >> 2) & 3;
        const uint8_t *lit = lit_buf; // where the literal bytes live
        uint32_t lit_len = 0, lused = 0;
        uint32_t lit_room = 0; // bytes that may be READ from lit (the fast path over-reads short runs)
        bool lit_rle = false;
        uint8_t lit_rle_byte = 0;
        if (ltype < 2) {
            uint32_t hdr;
            if (sf == 0 || sf == 2) { lit_len = b0 >> 3; hdr = 1; }
            else if (sf == 1) { lit_len = (b0 >> 4) | ((uint32_t)bp[1] << 4); hdr = 2; }
            else { if (blen < 3) { err = ZARC_FRAME_CORRUPT; break; } lit_len = (b0 >> 4) | ((uint32_t)bp[1] << 4) | ((uint32_t)bp[2] << 12); hdr = 3; }
            if (lit_len > BLOCK_MAX) { err = ZARC_FRAME_CORRUPT; break; }
            if (ltype == 0) {
                if (hdr + lit_len > blen) { err = ZARC_FRAME_CORRUPT; break; }
                lit = bp + hdr; // used in place
                lit_room = slen - (uint32_t)(lit - src);
                lused = hdr + lit_len;
            } else {
                if (hdr + 1 > blen) { err = ZARC_FRAME_CORRUPT; break; }
                lit_rle = true;
                lit_rle_byte = bp[hdr];
                lused = hdr + 1;
            }
        } else {
            uint32_t hdr, comp, streams;
            if (blen < 5 && !(sf <= 1 && blen >= 3)) { err = ZARC_FRAME_CORRUPT; break; }
            if (sf <= 1) {
                const uint32_t v = b0 | ((uint32_t)bp[1] << 8) | ((uint32_t)bp[2] << 16);
                lit_len = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; hdr = 3; streams = sf == 0 ? 1 : 4;
            } else if (sf == 2) {
                const uint32_t v = b0 | ((uint32_t)bp[1] << 8) | ((uint32_t)bp[2] << 16) | ((uint32_t)bp[3] << 24);
                lit_len = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; hdr = 4; streams = 4;
            } else {
                const uint64_t v = (uint64_t)b0 | ((uint64_t)bp[1] << 8) | ((uint64_t)bp[2] << 16) | ((uint64_t)bp[3] << 24) | ((uint64_t)bp[4] << 32);
                lit_len = (uint32_t)(v >> 4) & 0x3FFFF; comp = (uint32_t)(v >> 22) & 0x3FFFF; hdr = 5; streams = 4;
            }
            if (lit_len > BLOCK_MAX || hdr + comp > blen) { err = ZARC_FRAME_CORRUPT; break; }
            if (use_pre) { // fast path: stage 2 has regenerated these literals
                if (fblocks[my_b].lit_len != lit_len) { err = ZARC_FRAME_CORRUPT; break; } // not reachable
                lit = lits + flit_index[my_b];
                lit_room = lit_len + 64; // the literal scratch ends with 64 spare bytes
                lused = hdr + comp;
            } else {
            const uint8_t *hp = bp + hdr;
            uint32_t rem = comp;
            if (ltype == 2) {
                int nw = 0;
                const int used = huf_read_weights(L, hp, rem, lane, &nw);
                if (used < 0 || !huf_build_table(L, nw, lane)) { err = ZARC_FRAME_CORRUPT; break; }
                hp += used;
                rem -= (uint32_t)used;
            } else if (!L.ctrl[C_HUF_VALID]) { err = ZARC_FRAME_CORRUPT; break; }
            const int max_bits = L.ctrl[C_HUF_BITS];
            bool ok = true;
            if (streams == 1) {
                if (lane == 0 && !(dbg & 4)) ok = huf_decode_stream(L.huf, max_bits, hp, rem, lit_buf, lit_len);
            } else {
                if (rem < 6) { err = ZARC_FRAME_CORRUPT; break; }
                const uint32_t s1 = hp[0] | ((uint32_t)hp[1] << 8), s2 = hp[2] | ((uint32_t)hp[3] << 8), s3 = hp[4] | ((uint32_t)hp[5] << 8);
                const uint32_t per = (lit_len + 3) / 4;
                if (6 + s1 + s2 + s3 > rem || per * 3 > lit_len) { err = ZARC_FRAME_CORRUPT; break; }
                const uint32_t s4 = rem - 6 - s1 - s2 - s3;
                if (lane < 4 && !(dbg & 4)) {
                    const uint32_t so = lane == 0 ? 0 : (lane == 1 ? s1 : (lane == 2 ? s1 + s2 : s1 + s2 + s3));
                    const uint32_t sl = lane == 0 ? s1 : (lane == 1 ? s2 : (lane == 2 ? s3 : s4));
                    const uint32_t cntl = lane < 3 ? per : lit_len - 3 * per;
                    ok = huf_decode_stream(L.huf, max_bits, hp + 6 + so, sl, lit_buf + (uint32_t)lane * per, cntl);
                }
            }
            if (zd::ballot(!ok) != 0) { err = ZARC_FRAME_CORRUPT; break; }
            zd::wave_sync_global(); // literal bytes written by lanes 0..3 are read by every lane below
            lused = hdr + comp;
            }
        }
        // sequences section
        const uint8_t *sp = bp + lused;
        uint32_t srem = blen - lused;
        if (srem < 1) { err = ZARC_FRAME_CORRUPT; break; }
        uint32_t nseq;
        if (sp[0] < 128) { nseq = sp[0]; sp += 1; srem -= 1; }
        else if (sp[0] < 255) { if (srem < 2) { err = ZARC_FRAME_CORRUPT; break; } nseq = ((uint32_t)(sp[0] - 128) << 8) + sp[1]; sp += 2; srem -= 2; }
        else { if (srem < 3) { err = ZARC_FRAME_CORRUPT; break; } nseq = (uint32_t)sp[1] + ((uint32_t)sp[2] << 8) + 0x7F00; sp += 3; srem -= 3; }
        nseq = zd::uniform(nseq);
        const uint64_t block_start = opos;
        uint32_t lp = 0; // literals consumed
        if (nseq > 0) {
            int al_l = 0, al_o = 0, al_m = 0;
            SeqBits b;
            uint32_t sl = 0, so = 0, sm = 0;
            bool okb = true;
            const uint64_t *pre = nullptr; // this block's sequences (literal length, match length, offset value), already decoded
            if (use_pre) {
                const ZdecBlock zb = fblocks[my_b];
                if (zb.state != 1 || zb.nseq != nseq) { err = ZARC_FRAME_CORRUPT; break; } // not reachable: stage 2 covers every block of a fast frame
                pre = seqs + fseq_index[my_b];
            } else {
                if (srem < 1) { err = ZARC_FRAME_CORRUPT; break; }
                const uint32_t modes = sp[0];
                sp++; srem--;
                if (modes & 3) { err = ZARC_FRAME_CORRUPT; break; }
                int r = seq_table(L, L.ll, C_LL_AL, C_LL_OK, (int)(modes >> 6), sp, srem, D_LL_DEFAULT, 36, 6, 9, 35, lane);
                if (r < 0) { err = ZARC_FRAME_CORRUPT; break; }
                sp += r; srem -= (uint32_t)r;
                r = seq_table(L, L.of, C_OF_AL, C_OF_OK, (int)((modes >> 4) & 3), sp, srem, D_OF_DEFAULT, 29, 5, 8, 31, lane);
                if (r < 0) { err = ZARC_FRAME_CORRUPT; break; }
                sp += r; srem -= (uint32_t)r;
                r = seq_table(L, L.ml, C_ML_AL, C_ML_OK, (int)((modes >> 2) & 3), sp, srem, D_ML_DEFAULT, 53, 6, 9, 52, lane);
                if (r < 0) { err = ZARC_FRAME_CORRUPT; break; }
                sp += r; srem -= (uint32_t)r;
                al_l = L.ctrl[C_LL_AL]; al_o = L.ctrl[C_OF_AL]; al_m = L.ctrl[C_ML_AL];
                // lane 0 owns the bitstream and the FSE states
                if (lane == 0) {
                    okb = b.init(sp, srem);
                    if (okb) { sl = b.read(al_l); so = b.read(al_o); sm = b.read(al_m); okb = b.bitpos >= 0; }
                }
                if (zd::ballot(!okb) != 0) { err = ZARC_FRAME_CORRUPT; break; }
            }
            uint64_t q_next = 0; // fast path: the next batch's sequences are requested one batch ahead
            if (PRE && (uint32_t)lane < nseq) q_next = pre[lane];
            for (uint32_t base = 0; base < nseq && !err; base += SEQ_BATCH) {
                const uint32_t cnt = nseq - base < SEQ_BATCH ? nseq - base : SEQ_BATCH;
                if (dbg & 2) continue;
                uint32_t p_ll = 0, p_ml = 0, p_off = 1;
                if (PRE) {
                    // every lane fetches its sequence; stage 2 already resolved the offset as far as the block alone allows, what is
                    // left refers to the history at the start of the block (rep0..2 stay fixed during a pre-decoded block)
                    bool pbad = false;
                    const uint64_t q = q_next;
                    if (base + SEQ_BATCH + (uint32_t)lane < nseq) q_next = pre[base + SEQ_BATCH + (uint32_t)lane];
                    if ((uint32_t)lane < cnt) {
                        const uint32_t llr = zge_seq_ll(q), v = zge_seq_ofv(q);
                        p_ll = llr & (ZDEC_LL_REF - 1);
                        p_ml = zge_seq_ml(q);
                        if (llr & ZDEC_LL_REF) {
                            const uint32_t slot = v & 3, delta = v >> 2, hist = slot == 0 ? rep0 : (slot == 1 ? rep1 : rep2);
                            if (hist <= delta) pbad = true; else p_off = hist - delta;
                        } else p_off = v;
                    }
                    if (zd::ballot(pbad) != 0) okb = false;
                } else if (lane == 0) {
                    for (uint32_t i = 0; i < cnt; i++) {
                        const uint32_t cl = L.ll[sl], co = L.of[so], cm = L.ml[sm];
                        const uint32_t ofc = cell_sym(co), mlc = cell_sym(cm), llc = cell_sym(cl);
                        if (ofc > 31 || mlc > 52 || llc > 35) { okb = false; break; }
                        const uint32_t ofv = (1u << ofc) + b.read((int)ofc);
                        uint32_t mbase, mbits, lbase, lbits;
                        ml_code_info(mlc, mbase, mbits);
                        ll_code_info(llc, lbase, lbits);
                        const uint32_t ml = mbase + b.read((int)mbits);
                        const uint32_t ll = lbase + b.read((int)lbits);
                        uint32_t offset;
                        if (ofv > 3) { offset = ofv - 3; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                        else {
                            const uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u);
                            if (idx == 0) offset = rep0;
                            else {
                                offset = idx == 3 ? rep0 - 1 : (idx == 1 ? rep1 : rep2);
                                if (offset == 0) { okb = false; break; }
                                if (idx > 1) rep2 = rep1;
                                rep1 = rep0;
                                rep0 = offset;
                            }
                        }
                        L.seq[i * 3] = ll; L.seq[i * 3 + 1] = ml; L.seq[i * 3 + 2] = offset;
                        if (base + i + 1 < nseq) {
                            sl = cell_base(cl, al_l) + b.read((int)cell_nbits(cl, al_l));
                            sm = cell_base(cm, al_m) + b.read((int)cell_nbits(cm, al_m));
                            so = cell_base(co, al_o) + b.read((int)cell_nbits(co, al_o));
                        }
                        if (b.bitpos < 0) { okb = false; break; }
                    }
                }
                zd::wave_sync();
                if (zd::ballot(!okb) != 0) { err = ZARC_FRAME_CORRUPT; break; }
                // ---- execute the batch: lane i does the bookkeeping of sequence i; copies of different sequences
                // are independent unless a match source reaches into this batch's own output ("near" matches) ----
                {
                    const bool have = (uint32_t)lane < cnt;
                    const uint32_t ll = !have ? 0u : (PRE ? p_ll : L.seq[lane * 3]), ml = !have ? 0u : (PRE ? p_ml : L.seq[lane * 3 + 1]);
                    const uint32_t offset = !have ? 1u : (PRE ? p_off : L.seq[lane * 3 + 2]);
                    const uint32_t incl_ll = zd::wave_scan_incl(ll), incl_all = zd::wave_scan_incl(ll + ml);
                    const uint32_t tot_ll = zd::readlane(incl_ll, 63), tot_all = zd::readlane(incl_all, 63);
                    if (lp + tot_ll > lit_len) { err = ZARC_FRAME_CORRUPT; break; }
                    if (opos + tot_all > cap) { err = ZARC_FRAME_DSTSIZE; break; }
                    const uint32_t bpos = (uint32_t)opos;                      // output position where this batch starts
                    const uint32_t dlit = bpos + (incl_all - ll - ml);         // this sequence's literals go here
                    const uint32_t dmat = dlit + ll;                           // and its match here
                    const uint32_t slit = lp + (incl_ll - ll);                 // its first literal
                    if (zd::ballot(have && offset > dmat) != 0) { err = ZARC_FRAME_CORRUPT; break; }
                    const uint32_t msrc = dmat - offset;
                    const bool far = have && msrc + ml <= bpos;                // source entirely below this batch's output
                    if (!(dbg & 1) && PRE && tot_all <= OBUF) {
                        // ---- fast path: the batch's output is put together in LDS, then written out as whole lines ----
                        uint8_t *const ob = L.obuf;
                        const uint32_t o_lit = dlit - bpos, o_mat = dmat - bpos;
                        // (1) literal runs and (2) far matches (sources below this batch's output): independent of each other
                        // short runs: all their loads are issued before any is used, one round trip for the batch (the runs that would
                        // read past the end of their buffer take lane_stage32)
                        {
                            const bool lit_short = !lit_rle && ll > 0 && ll <= 32 && !(dbg & 32), far_short = far && ml > 0 && ml <= 32 && !(dbg & 64);
                            const bool lit_w = lit_short && slit + 16 <= lit_room, far_w = far_short && (uint64_t)msrc + 16 <= cap;
                            Win32 wl, wf;
                            if (lit_w) lane_load32(wl, lit + slit, ll);
                            if (far_w) lane_load32(wf, out + msrc, ml);
                            if (lit_rle) { for (uint32_t r = 0; r < ll; r++) ob[o_lit + r] = lit_rle_byte; }
                            if (lit_w) lane_store32(ob + o_lit, wl, ll);
                            else if (lit_short) lane_stage32(ob + o_lit, lit + slit, ll);
                            if (far_w) lane_store32(ob + o_mat, wf, ml);
                            else if (far_short) lane_stage32(ob + o_mat, out + msrc, ml);
                        }
                        uint64_t longs = lit_rle ? 0ull : zd::ballot(ll > 32);
                        while (longs) {
                            const int i = zd::ctz64(longs);
                            longs &= longs - 1;
                            const uint32_t n_ = zd::readlane(ll, (uint32_t)i), d_ = zd::readlane(o_lit, (uint32_t)i), s_ = zd::readlane(slit, (uint32_t)i);
                            for (uint32_t k = (uint32_t)lane; k < n_; k += 64) ob[d_ + k] = lit[s_ + k];
                        }
                        longs = zd::ballot(far && ml > 32);
                        while (longs) {
                            const int i = zd::ctz64(longs);
                            longs &= longs - 1;
                            const uint32_t n_ = zd::readlane(ml, (uint32_t)i), d_ = zd::readlane(o_mat, (uint32_t)i), s_ = zd::readlane(msrc, (uint32_t)i);
                            for (uint32_t k = (uint32_t)lane; k < n_; k += 64) ob[d_ + k] = out[s_ + k];
                        }
                        zd::wave_sync();
                        // (3) near matches.  Those whose source touches no UNRESOLVED near match's output run together, one lane each, round
                        // after round (records: every field copies the previous record's, so a round retires one record's matches and
                        // a batch takes as many rounds as it holds records, not one step per match).  Outputs are ordered like the lanes,
                        // so "the near matches whose output overlaps my source" is a lane range (two binary searches over the match positions,
                        // once per batch) and a round only tests it against the unresolved set.  A round that finds
                        // nothing (long, overlapping or straddling matches) runs the lowest unresolved match wave-wide: all below it is final.
                        uint64_t near = (dbg & 8) ? 0ull : zd::ballot(have && !far);
                        if (near) {
                            const int32_t s_rel = (int32_t)o_mat - (int32_t)offset;   // source start relative to the batch
                            const uint32_t s_end = (uint32_t)s_rel + ml;              // meaningful when s_rel >= 0
                            const bool cand = have && !far && s_rel >= 0 && offset >= ml && ml <= 32 && !(dbg & 128);
                            const uint32_t o_end = o_mat + ml;
                            uint32_t jb = 0;                                          // lanes whose match starts below s_end
#pragma unroll
                            for (uint32_t step = 32; step; step >>= 1) {
                                const uint32_t probe = zd::shfl(o_mat, (int)((jb + step - 1) & 63));
                                if (probe < s_end) jb += step;
                            }
                            if (zd::readlane(o_mat, 63) < s_end) jb = 64;
                            uint32_t jl = 0;                                          // lanes whose match ends at or below my source's start
#pragma unroll
                            for (uint32_t step = 32; step; step >>= 1) {
                                const uint32_t probe = zd::shfl(o_end, (int)((jl + step - 1) & 63));
                                if (probe <= (uint32_t)s_rel) jl += step;
                            }
                            if (zd::readlane(o_end, 63) <= (uint32_t)s_rel) jl = 64;
                            // the lanes whose output overlaps my source: [jl, jb) -- fixed for the batch, so a round is a mask test
                            const uint64_t deps = (jb >= 64 ? ~0ull : ((1ull << jb) - 1)) & ~(jl >= 64 ? ~0ull : ((1ull << jl) - 1));
                            while (near) {
                                const bool indep = cand && ((near >> lane) & 1) && (near & deps) == 0;
                                const uint64_t im = zd::ballot(indep);
                                if (im) {
                                    if (indep) lane_move32(ob + o_mat, ob + s_rel, ml);
                                    near &= ~im;
                                } else {
                                    const int i = zd::ctz64(near);
                                    near &= near - 1;
                                    const uint32_t n_ = zd::readlane(ml, (uint32_t)i), d_ = zd::readlane(o_mat, (uint32_t)i), o_ = zd::readlane(offset, (uint32_t)i);
                                    // all reads come from below the match start: byte k <- source[k mod offset]; the source may begin under the batch
                                    for (uint32_t k = (uint32_t)lane; k < n_; k += 64) {
                                        uint32_t j = k;
                                        if (j >= o_) j = j % o_;
                                        const int32_t sp = (int32_t)d_ - (int32_t)o_ + (int32_t)j; // relative to the batch start
                                        ob[d_ + k] = sp >= 0 ? ob[sp] : out[bpos + sp];
                                    }
                                }
                                zd::wave_sync();
                            }
                        }
                        // (4) the finished bytes leave LDS 16 per lane
                        for (uint32_t k = (uint32_t)lane * 16; k < tot_all && !(dbg & 16); k += 64 * 16) {
                            if (k + 16 <= tot_all) { struct { uint64_t a, b; } v; __builtin_memcpy(&v, ob + k, 16); __builtin_memcpy(out + bpos + k, &v, 16); }
                            else for (uint32_t r = k; r < tot_all; r++) out[bpos + r] = ob[r];
                        }
                        zd::wave_sync_global(); // later batches may copy from anything written here; obuf is reused
                    } else if (!(dbg & 1)) {
                    // (1) literal runs: short ones one lane per sequence, long ones wave-wide
                    {
                        const uint32_t ns = ll <= 32 ? ll : 0u;
                        if (lit_rle) { for (uint32_t r = 0; r < ns; r++) out[dlit + r] = lit_rle_byte; }
                        else lane_copy32(out + dlit, lit + slit, ns);
                        uint64_t longs = zd::ballot(ll > 32);
                        while (longs) {
                            const int i = zd::ctz64(longs);
                            longs &= longs - 1;
                            const uint32_t n_ = zd::readlane(ll, (uint32_t)i), d_ = zd::readlane(dlit, (uint32_t)i), s_ = zd::readlane(slit, (uint32_t)i);
                            if (lit_rle) { for (uint32_t k = (uint32_t)lane; k < n_; k += 64) out[d_ + k] = lit_rle_byte; }
                            else wave_copy(out + d_, lit + s_, n_, lane);
                        }
                    }
                    // (2) far matches: no dependence on this batch (and no self-overlap: offset >= length)
                    {
                        const uint32_t ns = (far && ml <= 32) ? ml : 0u;
                        lane_copy32(out + dmat, out + msrc, ns);
                        uint64_t longs = zd::ballot(far && ml > 32);
                        while (longs) {
                            const int i = zd::ctz64(longs);
                            longs &= longs - 1;
                            const uint32_t n_ = zd::readlane(ml, (uint32_t)i), d_ = zd::readlane(dmat, (uint32_t)i), s_ = zd::readlane(msrc, (uint32_t)i);
                            wave_copy(out + d_, out + s_, n_, lane);
                        }
                    }
                    // (3) near matches, in order; their sources may be bytes written above or by earlier near matches
                    uint64_t near = zd::ballot(have && !far);
                    if (near) {
                        zd::wave_sync_global();
                        uint32_t unfenced_lo = 0xFFFFFFFFu; // lowest output byte written by a near match since the last fence
                        while (near) {
                            const int i = zd::ctz64(near);
                            near &= near - 1;
                            const uint32_t n_ = zd::readlane(ml, (uint32_t)i), d_ = zd::readlane(dmat, (uint32_t)i), o_ = zd::readlane(offset, (uint32_t)i);
                            const uint32_t s_ = d_ - o_;
                            const uint32_t send = s_ + n_ < d_ ? s_ + n_ : d_;
                            if (send > unfenced_lo) { zd::wave_sync_global(); unfenced_lo = 0xFFFFFFFFu; }
                            // all reads come from below the match start: byte k <- source[k mod offset]
                            for (uint32_t k = (uint32_t)lane; k < n_; k += 64) {
                                uint32_t j = k;
                                if (j >= o_) j = j % o_;
                                out[d_ + k] = out[s_ + j];
                            }
                            if (d_ < unfenced_lo) unfenced_lo = d_;
                        }
                    }
                    zd::wave_sync_global(); // later batches may copy from anything written here
                    }
                    lp += tot_ll;
                    opos += tot_all;
                }
                zd::wave_sync(); // L.seq is rewritten by lane 0 in the next batch
            }
            if (err) break;
            if (PRE) { // history after the block, from stage 2's symbolic summary
                const ZdecBlock zb = fblocks[my_b];
                uint32_t nr[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t e = zb.rep[k];
                    if (e & ZDEC_REP_REF) { const uint32_t slot = e & 3, delta = (e & ~ZDEC_REP_REF) >> 2; nr[k] = (slot == 0 ? rep0 : (slot == 1 ? rep1 : rep2)) - delta; }
                    else nr[k] = e;
                }
                rep0 = nr[0]; rep1 = nr[1]; rep2 = nr[2];
            }
            bool endok = true;
            if (lane == 0 && !PRE) endok = b.bitpos == 0;
            if (zd::ballot(!endok) != 0) { err = ZARC_FRAME_CORRUPT; break; }
        } else if (srem != 0) { err = ZARC_FRAME_CORRUPT; break; }
        // trailing literals
        {
            const uint32_t tail = lit_len - lp;
            if (opos + tail > cap) { err = ZARC_FRAME_DSTSIZE; break; }
            if (lit_rle) { for (uint32_t k = (uint32_t)lane; k < tail; k += 64) out[opos + k] = lit_rle_byte; }
            else wave_copy(out + opos, lit + lp, tail, lane);
            opos += tail;
        }
        if (opos - block_start > BLOCK_MAX) { err = ZARC_FRAME_CORRUPT; break; }
        zd::wave_sync_global(); // the next block may reference anything written so far; lit_buf is reused
    }
    if (PRE && !last) { // a piece in front of the frame's last one: it must have produced exactly what the host expected of its blocks
        if (!err && (blocks_left != 0 || opos != piece.out_start + piece.out_len)) err = ZARC_FRAME_CORRUPT;
        if (lane == 0 && err) atomicMax(&status[f], err); // (statuses start at 0 = ok; the highest code of a frame's pieces wins, whatever their order)
        return;
    }
    if (!err && opos != cap) err = ZARC_FRAME_SRCSIZE;
    uint32_t ck = 0;
    if (!err && has_ck) {
        if (pos + 4 > slen) err = ZARC_FRAME_SRCSIZE;
        else { ck = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24); pos += 4; }
    }
    if (!err && pos != slen) err = ZARC_FRAME_SRCSIZE;
    if (lane == 0) {
        if (PRE) { if (err) atomicMax(&status[f], err); }
        else status[f] = err;
        stored_checksum[2 * f] = has_ck;
        stored_checksum[2 * f + 1] = ck;
    }
}

// Persistent waves: the grid is sized to what the chip holds at once and every wave takes the next frame from a
// queue (frames are ordered largest first), so slow and fast frames balance across XCDs whatever their order.
// Every wave leaves as soon as the queue is empty.  zarc_zstd_decode is the general kernel (everything decoded inline);
// when the fast path is on it only takes the frames stage 1/2 turned down, and zarc_zstd_frames takes the others.
__global__ void __launch_bounds__(64, 4) zarc_zstd_decode(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off,
                                                       const uint64_t *__restrict__ frame_len, uint8_t *__restrict__ dst_base,
                                                       const uint64_t *__restrict__ dst_off, const uint64_t *__restrict__ raw_len,
                                                       const uint32_t *__restrict__ order, uint32_t n_frames,
                                                       uint8_t *__restrict__ lit_scratch /* one block per resident wave */, int32_t *__restrict__ status,
                                                       uint32_t *__restrict__ stored_checksum /* 2 words/frame: has, value */,
                                                       int dbg /* timing-only ablations: 1 no copies, 2 no sequence decode, 4 no Huffman decode */,
                                                       uint32_t *__restrict__ queue, const uint32_t *__restrict__ fast /* null: take every frame */)
{
    __shared__ Lds L;
    const int lane = zd::lane_id();
    uint8_t *lit_buf = lit_scratch + (uint64_t)blockIdx.x * (BLOCK_MAX + 64);
    // Beside the fast path this kernel mostly looks at frames that are not its own: in a batch of many small frames a wave takes 64 queue
    // slots at a time and every lane looks at one flag (a million frames one by one: a million dependent atomics, 16 ms on the critical
    // path of `--config small` for nothing to decode); few, large frames stay one per trip so that the largest do not share a wave.
    const uint32_t take = (fast != nullptr && n_frames >= 16u * gridDim.x) ? 64u : 1u; // uniform over the grid
    for (;;) {
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(queue, take);
        slot = zd::uniform(slot); // lane 0 is the first active lane
        if (slot >= n_frames) break;
        uint32_t f_l = 0;
        bool mine = false;
        if ((uint32_t)lane < take && slot + (uint32_t)lane < n_frames) {
            f_l = order[slot + (uint32_t)lane];
            mine = fast == nullptr || fast[f_l] == 0; // the others: zarc_zstd_frames has them
        }
        uint64_t todo = zd::ballot(mine);
        while (todo) { // uniform
            const uint32_t k = (uint32_t)zd::ctz64(todo);
            todo &= todo - 1;
            const uint32_t f = zd::readlane(f_l, k);
            decode_frame<false>(L, lane, f, lit_buf, frames_base, frame_off, frame_len, dst_base, dst_off, raw_len, status, stored_checksum, ZDEC_DBG(dbg), nullptr, nullptr,
                                nullptr, nullptr, nullptr, ZdecPiece{});
            zd::wave_sync_global(); // LDS tables and the literal buffer are reused by the next frame
        }
    }
}

// The frame pass of the fast path: same queue discipline over PIECES of frames whose sequences and literals were decoded ahead (a frame
// that cannot be cut is one piece; pieces are listed longest first).
__global__ void __launch_bounds__(64, 4) zarc_zstd_frames(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off,
                                                       const uint64_t *__restrict__ frame_len, uint8_t *__restrict__ dst_base,
                                                       const uint64_t *__restrict__ dst_off, const uint64_t *__restrict__ raw_len,
                                                       const ZdecPiece *__restrict__ pieces, uint32_t n_listed, uint32_t first_unlisted, uint32_t n_pieces,
                                                       int32_t *__restrict__ status,
                                                       uint32_t *__restrict__ stored_checksum,
                                                       int dbg /* timing-only ablations: 1 no copies; 8 no near matches, 16 no flush, 32 no literal staging, 64 no far-match staging, 128 all near matches in order */,
                                                       uint32_t *__restrict__ queue, const uint32_t *__restrict__ fast,
                                                       const uint64_t *__restrict__ slot_prefix, const ZdecBlock *__restrict__ zblocks,
                                                       const uint64_t *__restrict__ seq_index, const uint64_t *__restrict__ seqs,
                                                       const uint64_t *__restrict__ lit_index, const uint8_t *__restrict__ lits)
{
    __shared__ Lds L;
    const int lane = zd::lane_id();
    for (;;) {
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(queue, 1u);
        slot = zd::uniform(slot);
        if (slot >= n_pieces) break;
        ZdecPiece piece;
        if (slot < n_listed) piece = pieces[slot];
        else { // behind the list: frames first_unlisted, first_unlisted + 1, ... are one piece each (engine.hip: only large frames are cut and listed)
            piece.frame = first_unlisted + (slot - n_listed); piece.first = 0; piece.count = 0xFFFFFFFFu; piece.rep[0] = 1; piece.rep[1] = 4; piece.rep[2] = 8;
            piece.out_start = 0; piece.out_len = raw_len[piece.frame];
        }
        const uint32_t f = piece.frame;
        if (zd::uniform(fast[f]) == 0) continue; // zarc_zstd_decode has it
        const uint64_t first = slot_prefix[f];
        decode_frame<true>(L, lane, f, nullptr, frames_base, frame_off, frame_len, dst_base, dst_off, raw_len, status, stored_checksum, ZDEC_DBG(dbg),
                           zblocks + first, seq_index + first, seqs, lit_index + first, lits, piece);
        zd::wave_sync_global(); // the LDS staging buffer is reused by the next frame
    }
}

// Exclusive prefix sum of n 32-bit values (in[i * stride], at least `floor` each) into out[0 .. n] (64-bit; out[n] = the total, also at
// *total).  ONE workgroup of 1024 threads, four values per thread and trip: the decoder's slot / sequence / literal offsets of a batch
// are sized on the device, so that the host reads back three totals instead of per-slot arrays (a million small frames: 12 MB of
// counts down, 24 MB of offsets up and five host loops over them, a third of the call).
__global__ void __launch_bounds__(1024) zarc_scan_u32(const uint32_t *__restrict__ in, uint32_t stride, uint32_t floor_, uint64_t n,
                                                      uint64_t *__restrict__ out, uint64_t *__restrict__ total)
{
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint64_t carry = 0;
    for (uint64_t base = 0; base < n; base += 4096) {
        const uint64_t i0 = base + 4ull * tid;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint64_t i = i0 + (uint64_t)k; v[k] = i < n ? in[i * stride] : 0u; if (i < n && v[k] < floor_) v[k] = floor_; }
        const uint32_t mine = v[0] + v[1] + v[2] + v[3]; // (a trip's sum stays far below 2^32: 4096 values of at most 2^17)
        const uint32_t incl = zd::wave_scan_incl(mine);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) { const uint32_t x = wsum[w]; all += x; if (w < wave) before += x; }
        uint64_t at = carry + before + (incl - mine);
#pragma unroll
        for (int k = 0; k < 4; k++) { if (i0 + (uint64_t)k < n) out[i0 + k] = at; at += v[k]; }
        carry += all;
        __syncthreads(); // wsum is rewritten by the next trip
    }
    if (tid == 0) { out[n] = carry; *total = carry; }
}

// Stage 0 of the fast path: how many blocks does every frame hold?  (One lane per frame walks the block headers only.)  The slots of
// the fast path are then sized exactly: libzstd 1.5.7 splits blocks below 128 KiB at most levels, so "content / 128 KiB" undercounts,
// and generous guesses would leave the lane-per-slot kernels with mostly empty waves.  0 = not a frame the fast path takes.
__global__ void __launch_bounds__(64) zarc_zdec_count(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off,
                                                      const uint64_t *__restrict__ frame_len, const uint64_t *__restrict__ raw_len, uint32_t n_frames,
                                                      uint32_t *__restrict__ nblocks)
{
    const uint32_t f = blockIdx.x * 64u + threadIdx.x;
    if (f >= n_frames) return;
    nblocks[f] = 0;
    const uint8_t *src = frames_base + frame_off[f];
    const uint32_t slen = (uint32_t)frame_len[f];
    if (slen < 6 || !(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD)) return;
    const uint32_t desc = src[4];
    const uint32_t fcs_flag = desc >> 6, ss = (desc >> 5) & 1, did_flag = desc & 3;
    if (desc & 8) return;
    const uint32_t did_bytes = did_flag == 3 ? 4u : did_flag, fcs_bytes = fcs_flag == 0 ? ss : (1u << fcs_flag);
    uint32_t pos = 5 + (ss ? 0u : 1u) + did_bytes + fcs_bytes;
    const uint64_t cap = raw_len[f] / 512 + 64; // a frame cut into blocks of under 512 bytes on average is left to the general decoder
    uint32_t nb = 0;
    for (bool last = false; !last;) {
        if (pos + 3 > slen || nb >= cap) return;
        const uint32_t bh = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        last = bh & 1;
        const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3 || bsize > BLOCK_MAX) return;
        pos += 3 + (btype == 1 ? 1u : bsize);
        if (pos > slen) return;
        nb++;
    }
    nblocks[f] = nb;
}

// Stage 1 of the fast path: one lane per frame reads the frame header and walks the block headers.
__global__ void __launch_bounds__(64) zarc_zdec_scan(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off,
                                                     const uint64_t *__restrict__ frame_len, const uint64_t *__restrict__ raw_len, uint32_t n_frames,
                                                     const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                     uint32_t *__restrict__ counts, uint32_t *__restrict__ fast)
{
    const uint32_t f = blockIdx.x * 64u + threadIdx.x;
    if (f >= n_frames) return;
    fast[f] = 0;
    const uint8_t *src = frames_base + frame_off[f];
    const uint32_t slen = (uint32_t)frame_len[f];
    const uint64_t first = slot_prefix[f], maxb = slot_prefix[f + 1] - first;
    if (slen < 6 || !(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD)) return;
    const uint32_t desc = src[4];
    const uint32_t fcs_flag = desc >> 6, ss = (desc >> 5) & 1, did_flag = desc & 3;
    if (desc & 8) return;
    const uint32_t did_bytes = did_flag == 3 ? 4u : did_flag, fcs_bytes = fcs_flag == 0 ? ss : (1u << fcs_flag);
    uint32_t pos = 5 + (ss ? 0u : 1u) + did_bytes + fcs_bytes;
    if (pos > slen) return;
    (void)raw_len; // content-size and checksum checks stay with the frame pass
    bool last = false;
    uint64_t bi = 0;
    while (!last) {
        if (pos + 3 > slen || bi >= maxb) return;
        const uint32_t bh = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        pos += 3;
        last = bh & 1;
        const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3 || bsize > BLOCK_MAX) return;
        ZdecBlock zb;
        zb.frame = f; zb.type = btype; zb.payload = pos; zb.size = bsize; zb.nseq = 0; zb.seq_hdr = 0; zb.state = 0;
        for (int k = 0; k < 3; k++) zb.rep[k] = ZDEC_REP_REF | (uint32_t)k;
        zb.lit_type = 0; zb.lit_len = 0; zb.lit_off = 0; zb.lit_comp = 0; zb.lit_streams = 0; zb.pad[0] = zb.pad[1] = zb.pad[2] = 0;
        if (btype == 0) { if (pos + bsize > slen) return; pos += bsize; }
        else if (btype == 1) { if (pos + 1 > slen) return; pos += 1; }
        else {
            if (pos + bsize > slen) return;
            const uint8_t *bp = src + pos;
            LitInfo li;
            if (!parse_lit_section(bp, bsize, &li)) return;
            const uint32_t lused = li.lused;
            zb.lit_type = li.ltype; zb.lit_len = li.lit_len; zb.lit_off = pos + li.hdr; zb.lit_comp = li.comp; zb.lit_streams = li.streams;
            const uint8_t *sp = bp + lused;
            const uint32_t srem = bsize - lused;
            if (srem < 1) return;
            uint32_t nseq, adv;
            if (sp[0] < 128) { nseq = sp[0]; adv = 1; }
            else if (sp[0] < 255) { if (srem < 2) return; nseq = ((uint32_t)(sp[0] - 128) << 8) + sp[1]; adv = 2; }
            else { if (srem < 3) return; nseq = (uint32_t)sp[1] + ((uint32_t)sp[2] << 8) + 0x7F00; adv = 3; }
            if (nseq == 0 && srem != adv) return;
            if (nseq > BLOCK_MAX / 3 + 1) return; // more sequences than a block can hold output for
            zb.nseq = nseq;
            zb.seq_hdr = pos + lused + adv;
            pos += bsize;
        }
        // pad[0] = how far in front of the block its matches reach (ZDEC_REACH_UNKNOWN until stage 2 has seen its sequences),
        // pad[1] = the bytes the block regenerates: what the host needs to cut a frame into pieces that decode side by side
        zb.pad[0] = (btype == 2 && zb.nseq) ? ZDEC_REACH_UNKNOWN : 0u;
        zb.pad[1] = btype == 2 ? zb.lit_len : bsize;
        zblocks[first + bi] = zb;
        counts[2 * (first + bi)] = zb.nseq;
        counts[2 * (first + bi) + 1] = (btype == 2 && zb.lit_type >= 2) ? zb.lit_len : 0u;
        bi++;
    }
    fast[f] = 1;
}

// The sequence chain of one block on one lane (stage 2): FSE states through the three decode tables `tl` / `to` / `tm` (HBM scratch or
// LDS: the pointer's address space is the caller's), repeat offsets kept symbolic, 8 bytes per sequence streamed to outp.  Leaves the
// block's summary in zblocks[s] or, when anything is off, clears fast[f] (the frame pass then decodes the frame inline).
template <typename TAB>
__device__ __forceinline__ void seq_chain(bool ok, const uint8_t *__restrict__ src, const ZdecBlock &zb, const SeqHeader &own, const uint32_t end,
                                          TAB tl, TAB to, TAB tm, const int al_l, const int al_o, const int al_m, uint64_t *__restrict__ outp,
                                          ZdecBlock *__restrict__ zslot, uint32_t *__restrict__ fast_f)
{
    // repeat-offset history, symbolic: hv = offset (hr = 0) or hv = slot | delta << 2 of the history at block start (hr = 1)
    uint32_t hv0 = 0, hv1 = 1, hv2 = 2, hr0 = 1, hr1 = 1, hr2 = 1;
    uint32_t bpos_ = 0, reach_ = 0, msum_ = 0; // output position inside the block, farthest reach in front of it, sum of the match lengths
    if (ok) {
        SeqBits b;
        ok = b.init(src + own.bits_off, end - own.bits_off);
        uint32_t sl = 0, so = 0, sm = 0;
        if (ok) { sl = b.read(al_l); so = b.read(al_o); sm = b.read(al_m); ok = b.bitpos >= 0; }
        for (uint32_t i = 0; i < zb.nseq && ok; i++) {
            const uint32_t cl = tl[sl], co = to[so], cm = tm[sm];
            const uint32_t ofc = cell_sym(co), mlc = cell_sym(cm), llc = cell_sym(cl);
            if (ofc > 27 || mlc > 52 || llc > 35) { ok = false; break; } // offsets past the format's largest window: left to the frame pass
            const uint32_t ofv = (1u << ofc) + b.read((int)ofc);
            uint32_t mbase, mbits, lbase, lbits;
            ml_code_info(mlc, mbase, mbits);
            ll_code_info(llc, lbase, lbits);
            const uint32_t ml = mbase + b.read((int)mbits);
            const uint32_t ll = lbase + b.read((int)lbits);
            uint32_t ov, orf; // this sequence's offset, same symbolic form
            if (ofv > 3) { ov = ofv - 3; orf = 0; hv2 = hv1; hr2 = hr1; hv1 = hv0; hr1 = hr0; hv0 = ov; hr0 = orf; }
            else {
                const uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u);
                if (idx == 0) { ov = hv0; orf = hr0; }
                else {
                    if (idx == 1) { ov = hv1; orf = hr1; }
                    else if (idx == 2) { ov = hv2; orf = hr2; }
                    else { // first history entry minus one
                        ov = hv0; orf = hr0;
                        if (orf) { if ((ov >> 2) >= ZDEC_MAX_DELTA) { ok = false; break; } ov += 4; } // delta + 1 (only absurd chains are left to the frame pass)
                        else { if (ov <= 1) { ok = false; break; } ov -= 1; }
                    }
                    if (idx > 1) { hv2 = hv1; hr2 = hr1; }
                    hv1 = hv0; hr1 = hr0;
                    hv0 = ov; hr0 = orf;
                }
            }
            // where this match's source starts, relative to the block: in front of it by `reach` bytes at most (an offset that still
            // refers to the history at the block's start is not known here)
            bpos_ += ll;
            if (orf) reach_ = ZDEC_REACH_UNKNOWN; else if (ov > bpos_ && ov - bpos_ > reach_) reach_ = ov - bpos_;
            bpos_ += ml; msum_ += ml;
            zd::store_streaming(outp + i, zge_pack_seq(ll | (orf ? ZDEC_LL_REF : 0u), ml, ov)); // written once, read by the frame pass: keep it out of the way of the tables
            if (i + 1 < zb.nseq) {
                sl = cell_base(cl, al_l) + b.read((int)cell_nbits(cl, al_l));
                sm = cell_base(cm, al_m) + b.read((int)cell_nbits(cm, al_m));
                so = cell_base(co, al_o) + b.read((int)cell_nbits(co, al_o));
            }
            if (b.bitpos < 0) ok = false;
        }
        if (ok && b.bitpos != 0) ok = false;
    }
    if (ok) {
        zslot->rep[0] = hr0 ? (ZDEC_REP_REF | hv0) : hv0;
        zslot->rep[1] = hr1 ? (ZDEC_REP_REF | hv1) : hv1;
        zslot->rep[2] = hr2 ? (ZDEC_REP_REF | hv2) : hv2;
        zslot->pad[0] = reach_;
        zslot->pad[1] = zb.lit_len + msum_;
        zslot->state = 1;
    }
    else *fast_f = 0; // the frame pass decodes this frame inline and reports whatever is wrong with it
}

// The same chain with ONE bitstream request per sequence (Bits128: the 16 bytes that end at the cursor, asked for at the top of the step
// and in flight during the table lookups) instead of a refill -- a dependent load and a wait -- behind any of the nine bit fields: with
// 64 blocks in lockstep every refill site is taken by some lane in every step, and a step then is six to nine memory round trips.  For
// tables that cost no memory request (LDS).
template <class TAB, class INFO>
__device__ __forceinline__ void seq_chain128(bool ok, const uint8_t *__restrict__ src, const ZdecBlock &zb, const SeqHeader &own, const uint32_t end,
                                          TAB tl, TAB to, TAB tm, const int al_l, const int al_o, const int al_m, INFO info /* seq_code_info(): [0,64) literal lengths, [64,128) match lengths */,
                                          uint64_t *__restrict__ outp, ZdecBlock *__restrict__ zslot, uint32_t *__restrict__ fast_f)
{
    // The step of a sequence is straight-line code: every lane of the wave walks its own block, so a branch some lane takes is a
    // branch all of them pay for (round 4: 17 branches, 147 + 103 instructions per step -> selects; the code -> baseline / extra-bits
    // rules come from a 128-entry table in LDS).  Whatever goes wrong sets `bad`, which ends the lane's loop at its next test.
    // repeat-offset history, symbolic: an offset, or ZDEC_REP_REF | slot | delta << 2 of the history at block start
    uint32_t h0 = ZDEC_REP_REF | 0u, h1 = ZDEC_REP_REF | 1u, h2 = ZDEC_REP_REF | 2u;
    uint32_t bpos_ = 0, reach_ = 0, msum_ = 0; // output position inside the block, farthest reach in front of it, sum of the match lengths
    if (ok) {
        Bits128 b;
        ok = b.init(src + own.bits_off, end - own.bits_off);
        uint32_t sl = 0, so = 0, sm = 0;
        if (ok) { b.request(); b.settle(); sl = b.take((uint32_t)al_l); so = b.take((uint32_t)al_o); sm = b.take((uint32_t)al_m); ok = b.bitpos >= 0; } // <= 7 + 26 bits
        bool bad = !ok;
        const uint32_t nseq = zb.nseq;
        uint64_t held = 0; // the sequence of an even step, until the odd one's store takes it along
        uint64_t a_lo = 0, a_hi = 0;
        int32_t a_top = 0;
        if (ok) b.ahead(b.bitpos, a_lo, a_hi, a_top);
        for (uint32_t i = 0; i < nseq && !bad; i++) {
            // One bitstream window per sequence, asked for a step ahead: what this step will consume is known once its lookups are back
            // (the codes' extra bits, the three states' bits), ~100 instructions before the next step needs its window -- in a kernel
            // that runs one or two waves per SIMD that is the L2 round trip no other wave would cover.
            b.adopt(a_lo, a_hi, a_top);
            const uint32_t cl = tl[sl], co = to[so], cm = tm[sm];
            const uint32_t ofc = cell_sym(co), mlc = cell_sym(cm), llc = cell_sym(cl);
            const uint32_t mi = info[64 + mlc], li = info[llc];
            const bool more = i + 1 < nseq; // the last sequence reads no state bits
            const uint32_t nb_l = more ? cell_nbits(cl, al_l) : 0u, nb_m = more ? cell_nbits(cm, al_m) : 0u, nb_o = more ? cell_nbits(co, al_o) : 0u;
            b.ahead(b.bitpos - (int32_t)((ofc & 31u) + (mi >> 20) + (li >> 20) + nb_l + nb_m + nb_o), a_lo, a_hi, a_top);
            b.settle();
            bad = ofc > 27 || mlc > 52 || llc > 35;                        // offsets past the format's largest window: left to the frame pass
            const uint32_t ofv = (1u << (ofc & 31u)) + b.take(ofc & 31u);  // phase 1: <= 7 + 27 + 16 bits
            const uint32_t ml = (mi & 0xFFFFFu) + b.take(mi >> 20);
            b.second_phase();                                              // phase 2: <= 16 + 9 + 9 + 8 bits
            const uint32_t ll = (li & 0xFFFFFu) + b.take(li >> 20);
            // this sequence's offset, same symbolic form
            const bool is_new = ofv > 3;
            const uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u); // repeat codes: 0..3
            uint32_t c = idx == 1 ? h1 : h0;
            c = idx == 2 ? h2 : c;
            const bool cref = (c & ZDEC_REP_REF) != 0, is3 = idx == 3 && !is_new; // first history entry minus one: delta + 1, or offset - 1
            const bool bad3 = cref ? ((c & ~ZDEC_REP_REF) >> 2) >= ZDEC_MAX_DELTA : c <= 1u; // (only absurd chains are left to the frame pass)
            c = is3 ? (cref ? c + 4u : c - 1u) : c;
            bad = bad || (is3 && bad3);
            const uint32_t o = is_new ? ofv - 3u : c;
            h2 = (is_new || idx >= 2) ? h1 : h2;
            h1 = (is_new || idx >= 1) ? h0 : h1;
            h0 = o;
            const bool orf = (o & ZDEC_REP_REF) != 0;
            const uint32_t ov = o & ~ZDEC_REP_REF;
            // where this match's source starts, relative to the block: in front of it by `reach` bytes at most (an offset that still
            // refers to the history at the block's start is not known here)
            bpos_ += ll;
            const uint32_t before = orf ? ZDEC_REACH_UNKNOWN : (ov > bpos_ ? ov - bpos_ : 0u);
            reach_ = before > reach_ ? before : reach_;
            bpos_ += ml; msum_ += ml;
            // a plain store: eight steps fill a 64-byte line in L2 before it leaves for HBM.  (The non-temporal form of the HBM-table
            // kernel -- which keeps its tables in L2 -- sends every 8-byte store of every lane to memory on its own, and the step's wait
            // for the bitstream also waits for that store: 40.9 -> 26.6 ms at BASELINE configs[1].)
            // Two sequences per store: every lane writes its own block's sequences, and 16-byte pieces fill a line in half as many
            // requests as 8-byte ones (the literal kernel gained 25 % from the same change: fewer half-written lines leave L2).
            const uint64_t packed = zge_pack_seq(ll | (orf ? ZDEC_LL_REF : 0u), ml, ov);
            if (i & 1u) { // uniform: the lanes of a wave are at the same sequence number
                struct { uint64_t a, b; } two = {held, packed};
                __builtin_memcpy(outp + (i - 1), &two, 16);
            } else held = packed;
            sl = cell_base(cl, al_l) + b.take(nb_l);
            sm = cell_base(cm, al_m) + b.take(nb_m);
            so = cell_base(co, al_o) + b.take(nb_o);
            bad = bad || b.bitpos < 0;
        }
        ok = !bad && b.bitpos == 0;
        if (ok && (nseq & 1u)) outp[nseq - 1] = held;
    }
    if (ok) {
        zslot->rep[0] = h0;
        zslot->rep[1] = h1;
        zslot->rep[2] = h2;
        zslot->pad[0] = reach_;
        zslot->pad[1] = zb.lit_len + msum_;
        zslot->state = 1;
    }
    else *fast_f = 0; // the frame pass decodes this frame inline and reports whatever is wrong with it
}

// Where the table of type t of block slot s is described: the block's own header or, in Repeat mode, the nearest earlier block of the
// frame that has sequences and sets this table.  *owner = the slot that describes it.
__device__ bool seq_table_source(int t, const SeqHeader &own, const uint8_t *src, uint64_t s, uint32_t f, const uint64_t *__restrict__ slot_prefix,
                                 const ZdecBlock *__restrict__ zblocks, uint32_t *mode, uint32_t *off, uint32_t *len, uint64_t *owner)
{
    *mode = own.mode[t]; *off = own.off[t]; *len = own.len[t]; *owner = s;
    if (*mode != 3) return true;
    const uint64_t first = slot_prefix[f];
    for (uint64_t j = s; j > first;) {
        j--;
        const ZdecBlock pb = zblocks[j];
        if (pb.type != 2 || pb.nseq == 0) continue;
        SeqHeader ph;
        if (!scan_seq_header(src, pb.seq_hdr, pb.payload + pb.size, &ph)) return false;
        if (ph.mode[t] == 3) continue;
        *mode = ph.mode[t]; *off = ph.off[t]; *len = ph.len[t]; *owner = j;
        return true;
    }
    return false;
}
__device__ int make_seq_table(uint16_t *tab, int t, const SeqHeader &own, const uint8_t *src, uint64_t s, uint32_t f,
                              const uint64_t *__restrict__ slot_prefix, const ZdecBlock *__restrict__ zblocks)
{
    uint32_t mode, off, len;
    uint64_t owner;
    if (!seq_table_source(t, own, src, s, f, slot_prefix, zblocks, &mode, &off, &len, &owner)) return -1;
    return build_seq_table(tab, t, mode, src + off, mode == 2 ? len : (mode == 1 ? 1u : 0u));
}

// (see zarc_zdec_seqs_lds) a block with a long chain: not this kernel's when the long-block kernel runs beside it
__device__ __forceinline__ bool zdec_long_block(const ZdecBlock &zb) { return zb.type == 2 && zb.nseq >= ZDEC_LONG_NSEQ; }

// Stage 2: one lane per block slot, every lane with its own table set in HBM scratch.  With `wave_flag` (the launch behind
// zarc_zdec_seqs_shared) only the waves that kernel turned down do anything.
__global__ void __launch_bounds__(64) zarc_zdec_seqs(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off, uint64_t n_slots,
                                                     const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                     const uint64_t *__restrict__ seq_index, uint64_t *__restrict__ seqs, uint16_t *__restrict__ tables,
                                                     uint32_t *__restrict__ fast, uint64_t slot_base, const uint32_t *__restrict__ wave_flag,
                                                     const uint16_t *__restrict__ predef, int split_long)
{
    // every lane builds the tables of its block: the builder's 64 counters live in LDS (two local arrays became registers selected by
    // compare chains, ~130 instructions per access -- most of this kernel's time on small blocks)
    __shared__ int16_t T_scratch[64 * 64];
    if (wave_flag && !wave_flag[blockIdx.x]) return; // (launched with 64 lanes per workgroup then: the flags are per 64 slots)
    const uint64_t s = slot_base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // slots [slot_base, n_slots); waves may be partly filled (engine.hip)
    ZdecBlock zb;
    zb.type = 0xFFFFFFFFu; zb.nseq = 0;
    if (s < n_slots) zb = zblocks[s];
    if (split_long && zdec_long_block(zb)) return; // zarc_zdec_seqs_lds has it
    if (s >= n_slots) return;
    if (zb.type != 2 || zb.nseq == 0) return;
    const uint32_t f = zb.frame;
    if (!fast[f]) return;
    const uint8_t *src = frames_base + frame_off[f];
    const uint32_t end = zb.payload + zb.size;
    uint16_t *tab = tables + s * (uint64_t)ZDEC_TABLE_CELLS;
    uint16_t *const tabs[3] = {tab, tab + 1024, tab + 512}; // LL, OF, ML
    const uint16_t *use[3] = {tabs[0], tabs[1], tabs[2]};
    SeqHeader own;
    bool ok = scan_seq_header(src, zb.seq_hdr, end, &own);
    int al[3] = {0, 0, 0};
    for (int t = 0; t < 3 && ok; t++) {
        uint32_t mode, off, len;
        uint64_t owner;
        ok = seq_table_source(t, own, src, s, f, slot_prefix, zblocks, &mode, &off, &len, &owner);
        if (!ok) break;
        if (mode == 0 && predef) { // the predefined distributions: one table for everybody, built once per handle (small blocks use little else)
            use[t] = predef + (t == 0 ? ZDEC_PREDEF_LL : (t == 1 ? ZDEC_PREDEF_OF : ZDEC_PREDEF_ML));
            al[t] = t == 1 ? 5 : 6;
            continue;
        }
        al[t] = build_seq_table(tabs[t], t, mode, src + off, mode == 2 ? len : (mode == 1 ? 1u : 0u), Strided16<64>{&T_scratch[threadIdx.x]});
        if (al[t] < 0) ok = false;
    }
    seq_chain<const uint16_t *>(ok, src, zb, own, end, use[0], use[1], use[2], al[0], al[1], al[2], seqs + seq_index[s], zblocks + s, fast + f);
}

// the three predefined decode tables (RFC 8878 3.1.1.3.2.2), built once per handle: LL at 0 (64 cells), OF at 64 (32), ML at 96 (64)
__global__ void zarc_zdec_predef(uint16_t *__restrict__ out)
{
    const int t = (int)threadIdx.x;
    if (t < 3) (void)build_seq_table(out + (t == 0 ? ZDEC_PREDEF_LL : (t == 1 ? ZDEC_PREDEF_OF : ZDEC_PREDEF_ML)), t, 0u, nullptr, 0u);
}

// Stage 2 with the tables of a wave's 64 blocks SHARED in LDS.  The engine's own frames code the sixteen 64 KiB blocks of a 1 MiB entry with
// one table per type (zge_entropy.hip: zarc_zge_plan; libzstd repeats tables as well): blocks whose tables come from the same
// description (their own, or the same earlier block's through Repeat_Mode; the predefined distributions; an RLE symbol) use one copy.
// A wave's 64 consecutive block slots typically need 4 - 8 table sets instead of 64: they fit LDS (ZDEC_SETS64 tables per type, 23 KiB per
// wave, six waves per CU), and the three lookups per sequence stop being 64-byte lines from 80 000 tables in HBM / MALL (178 GB per
// launch at BASELINE configs[1], 11 x the whole path's algorithmic bytes).  A wave that needs more tables of some type than fit
// sets wave_flag[its index] and leaves: the launch of zarc_zdec_seqs behind this one does its 64 slots the old way.
// literal-length (code) / match-length (64 + code) code -> baseline | extra bits << 20
__device__ __forceinline__ uint32_t seq_code_info(uint32_t i)
{
    uint32_t base, bits;
    if (i < 64) ll_code_info(i < 36 ? i : 0u, base, bits); else ml_code_info(i - 64 < 53 ? i - 64 : 0u, base, bits);
    return base | (bits << 20);
}
#ifndef ZDEC_SETS64
#define ZDEC_SETS64 8 // 64 blocks of the engine's own frames = four groups of sixteen: four sets per type, four to spare (A/B: tools/r4_sets.sh)
#endif
#ifndef ZDEC_SETS32
#define ZDEC_SETS32 5
#endif
template <int LANES, int SETS>
__device__ __forceinline__ void zdec_seqs_shared_body(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off, uint64_t n_slots,
                                                      const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                      const uint64_t *__restrict__ seq_index, uint64_t *__restrict__ seqs,
                                                      uint32_t *__restrict__ fast, uint64_t slot_base, uint32_t *__restrict__ wave_flag)
{
    __shared__ uint16_t T_ll[SETS][512], T_ml[SETS][512], T_of[SETS][256];
    __shared__ int32_t T_al[3][SETS];
    __shared__ uint32_t lead[3][SETS];
    __shared__ uint32_t T_info[128];
    __shared__ int16_t T_scratch[64][3 * SETS]; // entry i of builder b at [i][b]
    const int lane = zd::lane_id();
    for (int i = lane; i < 128; i += LANES) T_info[i] = seq_code_info((uint32_t)i);
    const uint64_t s = slot_base + (uint64_t)blockIdx.x * LANES + (uint64_t)lane;
    ZdecBlock zb;
    zb.type = 0xFFFFFFFFu; zb.nseq = 0; zb.frame = 0; zb.payload = 0; zb.size = 0; zb.seq_hdr = 0; zb.lit_len = 0;
    if (s < n_slots) zb = zblocks[s];
    const uint32_t f = zb.type == 2 ? zb.frame : 0u;
    const bool mine = s < n_slots && zb.type == 2 && zb.nseq != 0 && fast[f] != 0; // a lane with a block to decode
    const uint8_t *src = frames_base + (mine ? frame_off[f] : 0);
    const uint32_t end = zb.payload + zb.size;
    SeqHeader own;
    own.bits_off = 0;
    for (int t = 0; t < 3; t++) { own.mode[t] = 0; own.off[t] = 0; own.len[t] = 0; }
    bool ok = mine && scan_seq_header(src, zb.seq_hdr, end, &own);
    // where each of my three tables is described, and a key that is equal for lanes whose table is the same one
    uint32_t mode[3] = {0, 0, 0}, off[3] = {0, 0, 0}, len[3] = {0, 0, 0}, key[3] = {0, 0, 0}, idx[3] = {0, 0, 0};
    for (int t = 0; t < 3 && ok; t++) {
        uint64_t owner;
        ok = seq_table_source(t, own, src, s, f, slot_prefix, zblocks, &mode[t], &off[t], &len[t], &owner);
        // (slots of one launch lie within 2^30 of each other: engine.hip caps the table scratch at 8 GiB = 3.4 M slots)
        if (ok) key[t] = mode[t] == 0 ? 0xC0000000u : (mode[t] == 1 ? (0x80000000u | src[off[t]]) : (uint32_t)(owner & 0x3FFFFFFFu));
    }
    // table indices: the lanes of one key share a table; the first lane of each key is the one whose description is read
    uint32_t most = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        uint64_t rem = zd::ballot(ok);
        uint32_t n = 0;
        while (rem) { // uniform
            const uint32_t first = (uint32_t)zd::ctz64(rem);
            const uint32_t k = zd::readlane(key[t], first);
            const uint64_t m = zd::ballot(ok && key[t] == k);
            if (ok && key[t] == k) idx[t] = n;
            if (n < (uint32_t)SETS && lane == 0) lead[t][n] = first;
            n++;
            rem &= ~m;
        }
        most = n > most ? n : most;
        if (lane == 0) for (uint32_t i = n; i < (uint32_t)SETS; i++) lead[t][i] = 0xFFFFFFFFu;
    }
    if (most > (uint32_t)SETS) { if (lane == 0) wave_flag[blockIdx.x] = 1; return; } // uniform: zarc_zdec_seqs takes these 64 slots
    zd::wave_sync();
    // build: lane L builds table L % SETS of type L / SETS from its first user's description (3 x SETS <= 64 lanes at once)
    {
        static_assert(3 * SETS <= LANES, "one building lane per table");
        const int bt = lane / SETS, bn = lane % SETS;
        const uint32_t from = bt < 3 ? lead[bt][bn] : 0xFFFFFFFFu;
        const int fl = from == 0xFFFFFFFFu ? lane : (int)from;
        uint32_t bmode = 0, boff = 0, blen = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const uint32_t m_ = zd::shfl(mode[t], fl), o_ = zd::shfl(off[t], fl), l_ = zd::shfl(len[t], fl);
            if (t == bt) { bmode = m_; boff = o_; blen = l_; }
        }
        const uint32_t bf = zd::shfl(f, fl);
        if (from != 0xFFFFFFFFu) {
            uint16_t *tab = bt == 0 ? &T_ll[bn][0] : (bt == 1 ? &T_of[bn][0] : &T_ml[bn][0]);
            const uint8_t *bsrc = frames_base + frame_off[bf];
            T_al[bt][bn] = build_seq_table(tab, bt, bmode, bsrc + boff, bmode == 2 ? blen : (bmode == 1 ? 1u : 0u), Strided16<3 * SETS>{&T_scratch[0][lane]});
        }
    }
    zd::wave_sync();
    if (!mine) return;
    int al_l = 0, al_o = 0, al_m = 0;
    if (ok) { al_l = T_al[0][idx[0]]; al_o = T_al[1][idx[1]]; al_m = T_al[2][idx[2]]; ok = al_l >= 0 && al_o >= 0 && al_m >= 0; }
    const uint32_t i0 = ok ? idx[0] : 0u, i1 = ok ? idx[1] : 0u, i2 = ok ? idx[2] : 0u;
    seq_chain128<ZDEC_LDS_TAB, ZDEC_LDS_INFO>(ok, src, zb, own, end, (ZDEC_LDS_TAB)&T_ll[i0][0], (ZDEC_LDS_TAB)&T_of[i1][0], (ZDEC_LDS_TAB)&T_ml[i2][0],
                                           al_l, al_o, al_m, (ZDEC_LDS_INFO)&T_info[0], seqs + seq_index[s], zblocks + s, fast + f);
}


// Lanes per wave: the chain of a block is ~235 instructions and one bitstream round trip per sequence, and 80 000 blocks are 1 250 full
// waves -- 1.2 per SIMD, nothing to overlap a wait with (SQ_WAIT_ANY 81 % of the wave-cycles).  Narrower workgroups put more waves on a
// SIMD for the same blocks; their share of the tables shrinks with them (16 lanes = the blocks of two or three frames: 3 tables per type,
// 7.5 KiB, 20 waves per CU).
__global__ void __launch_bounds__(64) zarc_zdec_seqs_shared(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off, uint64_t n_slots,
                                                            const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                            const uint64_t *__restrict__ seq_index, uint64_t *__restrict__ seqs,
                                                            uint32_t *__restrict__ fast, uint64_t slot_base, uint32_t *__restrict__ wave_flag)
{
    zdec_seqs_shared_body<64, ZDEC_SETS64>(frames_base, frame_off, n_slots, slot_prefix, zblocks, seq_index, seqs, fast, slot_base, wave_flag);
}
__global__ void __launch_bounds__(32) zarc_zdec_seqs_shared32(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off, uint64_t n_slots,
                                                              const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                              const uint64_t *__restrict__ seq_index, uint64_t *__restrict__ seqs,
                                                              uint32_t *__restrict__ fast, uint64_t slot_base, uint32_t *__restrict__ wave_flag)
{
    zdec_seqs_shared_body<32, ZDEC_SETS32>(frames_base, frame_off, n_slots, slot_prefix, zblocks, seq_index, seqs, fast, slot_base, wave_flag);
}
__global__ void __launch_bounds__(16) zarc_zdec_seqs_shared16(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off, uint64_t n_slots,
                                                              const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                              const uint64_t *__restrict__ seq_index, uint64_t *__restrict__ seqs,
                                                              uint32_t *__restrict__ fast, uint64_t slot_base, uint32_t *__restrict__ wave_flag)
{
    zdec_seqs_shared_body<16, 3>(frames_base, frame_off, n_slots, slot_prefix, zblocks, seq_index, seqs, fast, slot_base, wave_flag);
}

// Blocks with long chains that share no tables (libzstd's frames: a table set per block; the engine's own mid-sized frames, a workgroup's 64
// slots spread over a dozen of them).  With the tables in HBM scratch every step of zarc_zdec_seqs is a round trip to L2 / HBM per lookup, and
// the kernel lasts as long as its longest block (3 000 - 12 000 sequences x 2 - 5 us).  Here a lane keeps ITS OWN three tables in LDS (2.5 KiB;
// 16 lanes per workgroup) and a step is the straight-line code of the shared-table kernel.  Which blocks: every compressed block of at least
// ZDEC_LONG_NSEQ sequences that the shared-table kernel has not done (zdec_long_block; zarc_zdec_seqs leaves exactly those alone).  They are
// listed first, ordered by sequence count in buckets of 1 024 (zarc_zdec_long_count / _fill: the lanes of a workgroup wait for the longest of
// their sixteen chains, and neighbouring blocks differ by a factor of two and more), the longest first.
__device__ __forceinline__ uint32_t zdec_long_bucket(uint32_t nseq) { const uint32_t b = nseq >> 10; return b < ZDEC_LONG_BUCKETS - 1 ? b : (uint32_t)ZDEC_LONG_BUCKETS - 1; }
// counters[0 .. B) blocks per bucket, [B .. 2B) the fill cursors of zarc_zdec_long_fill, [2B] the total (all zero at launch)
__global__ void __launch_bounds__(64) zarc_zdec_long_count(const ZdecBlock *__restrict__ zblocks, uint64_t slot_base, uint64_t n_slots,
                                                           const uint32_t *__restrict__ wave_flag, uint32_t *__restrict__ counters)
{
    const uint64_t s = slot_base + (uint64_t)blockIdx.x * 64 + threadIdx.x;
    bool lng = false;
    uint32_t bk = 0;
    if (s < n_slots && (!wave_flag || wave_flag[blockIdx.x])) { const ZdecBlock zb = zblocks[s]; lng = zdec_long_block(zb); bk = zdec_long_bucket(zb.nseq); }
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)ZDEC_LONG_BUCKETS; b++) { // one atomic per wave and bucket
        const uint64_t m = zd::ballot(lng && bk == b);
        if (m && threadIdx.x == 0) { atomicAdd(counters + b, (uint32_t)__popcll(m)); atomicAdd(counters + 2 * ZDEC_LONG_BUCKETS, (uint32_t)__popcll(m)); }
    }
}
__global__ void __launch_bounds__(64) zarc_zdec_long_fill(const ZdecBlock *__restrict__ zblocks, uint64_t slot_base, uint64_t n_slots,
                                                          const uint32_t *__restrict__ wave_flag, uint32_t *__restrict__ counters, uint32_t *__restrict__ list)
{
    const uint64_t s = slot_base + (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const int lane = (int)threadIdx.x;
    bool lng = false;
    uint32_t bk = 0;
    if (s < n_slots && (!wave_flag || wave_flag[blockIdx.x])) { const ZdecBlock zb = zblocks[s]; lng = zdec_long_block(zb); bk = zdec_long_bucket(zb.nseq); }
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)ZDEC_LONG_BUCKETS; b++) {
        const uint64_t m = zd::ballot(lng && bk == b);
        if (m == 0) continue; // uniform
        uint32_t at = 0;
        if (lane == 0) {
            uint32_t base = 0; // the longer buckets come first
            for (uint32_t h = b + 1; h < (uint32_t)ZDEC_LONG_BUCKETS; h++) base += counters[h];
            at = base + atomicAdd(counters + ZDEC_LONG_BUCKETS + b, (uint32_t)__popcll(m));
        }
        at = zd::uniform(at);
        if (lng && bk == b) list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)(s - slot_base);
    }
}
__global__ void __launch_bounds__(ZDEC_LDS_LANES) zarc_zdec_seqs_lds(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off,
                                                                     uint64_t n_slots, const uint64_t *__restrict__ slot_prefix, ZdecBlock *__restrict__ zblocks,
                                                                     const uint64_t *__restrict__ seq_index, uint64_t *__restrict__ seqs, uint32_t *__restrict__ fast,
                                                                     uint64_t slot_base, const uint32_t *__restrict__ counters, const uint32_t *__restrict__ list)
{
    __shared__ uint16_t T[ZDEC_LDS_LANES][ZDEC_TABLE_CELLS];
    __shared__ uint32_t T_info[128];
    __shared__ int16_t T_scratch[64][ZDEC_LDS_LANES]; // entry i of lane l at [i][l]
    const int l = (int)threadIdx.x;
    const uint32_t total = counters[2 * ZDEC_LONG_BUCKETS];
    if ((uint64_t)blockIdx.x * ZDEC_LDS_LANES >= total) return; // uniform: the grid covers every slot, the list only the long ones
    for (int i = l; i < 128; i += ZDEC_LDS_LANES) T_info[i] = seq_code_info((uint32_t)i);
    zd::wave_sync();
    const uint32_t at = blockIdx.x * ZDEC_LDS_LANES + (uint32_t)l;
    if (at >= total) return;
    const uint64_t s = slot_base + list[at];
    if (s >= n_slots) return; // (not reachable)
    const ZdecBlock zb = zblocks[s];
    const uint32_t f = zb.frame;
    if (!fast[f]) return;
    const uint8_t *src = frames_base + frame_off[f];
    const uint32_t end = zb.payload + zb.size;
    SeqHeader own;
    bool ok = scan_seq_header(src, zb.seq_hdr, end, &own);
    int al[3] = {0, 0, 0};
    uint16_t *const tabs[3] = {&T[l][0], &T[l][1024], &T[l][512]}; // LL, OF, ML
    for (int t = 0; t < 3 && ok; t++) {
        uint32_t mode, off, len;
        uint64_t owner;
        ok = seq_table_source(t, own, src, s, f, slot_prefix, zblocks, &mode, &off, &len, &owner);
        if (!ok) break;
        al[t] = build_seq_table(tabs[t], t, mode, src + off, mode == 2 ? len : (mode == 1 ? 1u : 0u), Strided16<ZDEC_LDS_LANES>{&T_scratch[0][l]});
        if (al[t] < 0) ok = false;
    }
    seq_chain128<ZDEC_LDS_TAB, ZDEC_LDS_INFO>(ok, src, zb, own, end, (ZDEC_LDS_TAB)tabs[0], (ZDEC_LDS_TAB)tabs[1], (ZDEC_LDS_TAB)tabs[2], al[0], al[1], al[2],
                                           (ZDEC_LDS_INFO)&T_info[0], seqs + seq_index[s], zblocks + s, fast + f);
}


// Huffman literals of the fast path.  One wave per ZDEC_LIT_GROUP consecutive block slots.  The wave stages the blocks' tree
// descriptions in LDS, sixteen lanes turn them into weights side by side (the FSE-coded weights are a serial chain per block: done
// one block after the other by lane 0, as until round 4, that chain was 85 % of the kernel's time), the wave builds the sixteen
// decoders, then every lane decodes one stream (4 streams x 16 blocks).
// A full 2^11-cell decode table per block would cap a CU at 160 decoding lanes (160 KiB / 4 KiB x 4 streams), so the table
// is kept in its canonical form (0.3 KiB): cells of the format's table are ordered by weight, then by symbol, and the cells
// of weight w start at a multiple of 2^(w-1), hence for a cell x of weight w the symbol is sorted[adj[w] + (x >> (w-1))] with
// adj[w] = (symbols of lower weight) - (start[w] >> (w-1)); w comes from comparing x with start[] (canon_symbol_r).
struct HufCanon {
    uint8_t sorted[256]; // symbols by (weight, symbol)
    uint16_t start[12];  // first cell of weight w (start[w] for w = 1 .. table log; beyond: 2^log)
    int16_t adj[12];
};
constexpr uint32_t HUF_DESC_MAX = 129; // header byte + at most 127 bytes of FSE-coded weights (or 64 of direct ones)
struct HufBuild {
    union {
        struct {
            alignas(8) uint8_t desc[144]; // the tree description, staged (the bit reader may look 8 bytes past its end)
            int16_t norm[64];
            uint16_t next[64];
            uint16_t wtab[64];        // FSE table for Huffman weights (accuracy <= 6)
        };
        HufCanon canon;               // built from weights[] once the description has been read: the scratch above is dead by then
    };
    uint8_t weights[256];
};
struct LitLds {
    HufBuild blk[ZDEC_LIT_GROUP];
    int32_t bits[ZDEC_LIT_GROUP];                // table log per block, 0 = no decoder
    uint32_t used[ZDEC_LIT_GROUP];               // bytes of the tree description in front of the streams
    int32_t nw[ZDEC_LIT_GROUP];                  // weights read, 0 = none
    uint32_t frame[ZDEC_LIT_GROUP];
};

// The tree description B.desc[0..len) -> B.weights, all of it on the calling lane (its own HufBuild).  Returns the bytes the
// description takes or -1; same rules as huf_read_weights.
__device__ int huf_read_weights_lane(HufBuild &B, uint32_t len, int *nweights)
{
    if (len < 1) return -1;
    const uint32_t hb = B.desc[0];
    if (hb >= 128) {
        const int n = (int)hb - 127, consumed = 1 + (n + 1) / 2;
        if ((uint32_t)consumed > len) return -1;
        for (int i = 0; i < n; i++) {
            const uint8_t byte = B.desc[1 + i / 2];
            B.weights[i] = (i & 1) ? (byte & 15) : (byte >> 4);
        }
        *nweights = n;
        return consumed;
    }
    if (hb == 0 || 1 + hb > len) return -1;
    int nsym = 0, al = 0, cnt = -1;
    const int used = fse_read_desc(B.desc + 1, hb, 6, 63, B.norm, &nsym, &al); // weights are <= 11; 63 bounds B.norm
    BackBits b;
    if (used > 0 && (uint32_t)used < hb && fse_build_dtable(B.wtab, B.norm, nsym, al, B.next) && b.init(B.desc + 1 + used, hb - (uint32_t)used)) {
        uint32_t s1 = b.read(al), s2 = b.read(al);
        if (b.bitpos >= 0) {
            cnt = 0;
            for (;;) {
                if (cnt > 253) { cnt = -1; break; }
                B.weights[cnt++] = (uint8_t)cell_sym(B.wtab[s1]);
                s1 = cell_base(B.wtab[s1], al) + b.read((int)cell_nbits(B.wtab[s1], al));
                if (b.bitpos < 0) { B.weights[cnt++] = (uint8_t)cell_sym(B.wtab[s2]); break; }
                if (cnt > 253) { cnt = -1; break; }
                B.weights[cnt++] = (uint8_t)cell_sym(B.wtab[s2]);
                s2 = cell_base(B.wtab[s2], al) + b.read((int)cell_nbits(B.wtab[s2], al));
                if (b.bitpos < 0) { B.weights[cnt++] = (uint8_t)cell_sym(B.wtab[s1]); break; }
            }
        }
    }
    if (cnt < 1) return -1;
    *nweights = cnt;
    return 1 + (int)hb;
}

// weights[0..n) (LDS, room for one more) -> canonical decoder.  Uniform; same validity rules as huf_build_table.  Returns the table log or 0.
__device__ int huf_build_canon(uint8_t *weights, int n, int lane, HufCanon &C)
{
    uint32_t part = 0;
    bool bad = false;
    for (int i = lane; i < n; i += 64) {
        const uint32_t w = weights[i];
        if (w > 11) bad = true;
        else if (w) part += 1u << (w - 1);
    }
    const uint32_t sum = zd::wave_sum(part);
    if (zd::ballot(bad) != 0 || sum == 0) return 0;
    const int max_bits = zd::hb32(sum) + 1;
    if (max_bits > 11) return 0;
    const uint32_t left = (1u << max_bits) - sum;
    if (left & (left - 1)) return 0;
    const uint32_t last_w = (uint32_t)zd::hb32(left) + 1;
    if (lane == 0) weights[n] = (uint8_t)last_w;
    zd::wave_sync();
    const int nsym = n + 1;
    uint32_t cnt[12], my_w[4];
#pragma unroll
    for (int w = 0; w < 12; w++) cnt[w] = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int sidx = r * 64 + lane;
        const uint32_t w = sidx < nsym ? weights[sidx] : 0u;
        my_w[r] = w;
#pragma unroll
        for (int ww = 1; ww < 12; ww++) cnt[ww] += (uint32_t)__popcll(zd::ballot(w == (uint32_t)ww));
    }
    uint32_t start[13], rb[12], pos = 0, rank = 0;
    start[0] = 0;
#pragma unroll
    for (int ww = 1; ww < 12; ww++) { start[ww] = pos; rb[ww] = rank; pos += cnt[ww] << (ww - 1); rank += cnt[ww]; }
    start[12] = pos;
    if (pos != (1u << max_bits)) return 0;
    // symbols in (weight, symbol) order
    const uint64_t lt = (1ull << lane) - 1;
    uint32_t run[12];
#pragma unroll
    for (int w = 0; w < 12; w++) run[w] = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t w = my_w[r];
#pragma unroll
        for (int ww = 1; ww < 12; ww++) {
            const uint64_t m = zd::ballot(w == (uint32_t)ww);
            if (w == (uint32_t)ww) C.sorted[rb[ww] + run[ww] + (uint32_t)__popcll(m & lt)] = (uint8_t)(r * 64 + lane);
            run[ww] += (uint32_t)__popcll(m);
        }
    }
    if (lane < 12) {
        uint32_t st = 1u << max_bits;
        int32_t ad = 0;
#pragma unroll
        for (int ww = 1; ww < 12; ww++)
            if (lane == ww) { st = ww <= max_bits ? start[ww] : (1u << max_bits); ad = (int32_t)rb[ww] - (int32_t)(start[ww] >> (ww - 1)); }
        C.start[lane] = (uint16_t)st;
        C.adj[lane] = (int16_t)ad;
    }
    zd::wave_sync();
    return max_bits;
}

// The weight of a cell x (x = the next max_bits bits) is 1 + the number of weights ww >= 2 whose first cell start[ww] is <= x (the cells are
// ordered by weight, start[] rises): ten compares against values the lane keeps in registers for the whole stream.  (Until round 4 a
// 128-entry index gave the weight in one LDS access where its bucket lay inside one weight and a loop over start[] in LDS otherwise -- with
// 64 streams in lockstep some lane took the loop at nearly every symbol, and every lane waited for its eleven LDS reads.)
struct CanonStarts { uint32_t st[10]; };
__device__ __forceinline__ void canon_starts(const HufCanon &C, int max_bits, CanonStarts &R)
{
#pragma unroll
    for (int ww = 2; ww <= 11; ww++) R.st[ww - 2] = ww <= max_bits ? (uint32_t)C.start[ww] : 0xFFFFFFFFu;
}
__device__ __forceinline__ uint32_t canon_symbol_r(const HufCanon &C, const CanonStarts &R, int max_bits, uint32_t x, uint32_t *nbits)
{
    uint32_t w = 1;
#pragma unroll
    for (int k = 0; k < 10; k++) w += x >= R.st[k] ? 1u : 0u;
    *nbits = (uint32_t)max_bits + 1u - w;
    return C.sorted[(int32_t)C.adj[w] + (int32_t)(x >> (w - 1))];
}
// (inlined at its one call site: the compiler then knows the decoder lives in LDS and the streams in global memory -- as a function of
// its own it went through flat loads, 18 of them per 8 symbols)
__device__ __forceinline__ bool huf_decode_stream_canon(const HufCanon &C, int max_bits, const uint8_t *src, uint32_t len, uint8_t *out, uint32_t nout)
{
    SeqBits b; // the word below the window is requested ahead: the stream's L2 round trip is off the symbol chain
    if (!b.init(src, len)) return false;
    CanonStarts R;
    canon_starts(C, max_bits, R);
    uint32_t i = 0;
    auto eight = [&]() -> uint64_t {
        uint64_t w = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // refill on a fixed schedule (4 symbols take at most 44 of the >= 57 bits a refill provides): a data-dependent
            // refill would have some lane of the wave waiting for its load at nearly every symbol
            if ((j & 3) == 0) b.refill();
            uint32_t nb;
            const uint32_t sym = canon_symbol_r(C, R, max_bits, (uint32_t)((b.c << b.used) >> (64 - max_bits)), &nb);
            b.used += (int32_t)nb; b.bitpos -= (int32_t)nb;
            w |= (uint64_t)sym << (8 * j);
        }
        return w;
    };
    for (; i + 16 <= nout; i += 16) { // 16-byte stores: every lane writes its own stream, and a line that takes 16 stores to fill leaves L2 half written more often than one that takes 8
        struct { uint64_t a, b; } w2;
        w2.a = eight();
        w2.b = eight();
        __builtin_memcpy(out + i, &w2, 16);
    }
    for (; i + 8 <= nout; i += 8) {
        const uint64_t w = eight();
        __builtin_memcpy(out + i, &w, 8);
    }
    for (; i < nout; i++) {
        uint32_t nb;
        if (b.used + max_bits > 64) b.refill();
        out[i] = (uint8_t)canon_symbol_r(C, R, max_bits, (uint32_t)((b.c << b.used) >> (64 - max_bits)), &nb);
        b.used += (int32_t)nb; b.bitpos -= (int32_t)nb;
    }
    return b.bitpos == 0;
}

__global__ void __launch_bounds__(64) zarc_zdec_literals(const uint8_t *__restrict__ frames_base, const uint64_t *__restrict__ frame_off, uint64_t n_slots,
                                                         const uint64_t *__restrict__ slot_prefix, const ZdecBlock *__restrict__ zblocks,
                                                         const uint64_t *__restrict__ lit_index, uint8_t *__restrict__ lits, uint32_t *__restrict__ fast,
                                                         uint64_t slot_base /* slots [slot_base, n_slots) */)
{
    __shared__ LitLds S;
    const int lane = zd::lane_id();
    const uint64_t s0 = slot_base + (uint64_t)blockIdx.x * ZDEC_LIT_GROUP;
    static_assert(ZDEC_LIT_GROUP * 4 == 64, "four lanes and four streams per block, one wave");
    const int i = lane >> 2;
    const uint32_t k = (uint32_t)lane & 3u;
    const uint64_t s = s0 + (uint64_t)i;
    // ---- phase 1a: the four lanes of a block stage its tree description (Treeless: that of the nearest earlier block of the frame that carries one) ----
    ZdecBlock zb = {};
    bool want = false;
    uint32_t f = 0, dlen = 0;
    const uint8_t *dsrc = frames_base;
    if (s < n_slots) {
        zb = zblocks[s];
        if (zb.type == 2 && zb.lit_type >= 2) { f = zb.frame; want = fast[f] != 0; }
    }
    if (want) {
        uint64_t from = s;
        if (zb.lit_type == 3) {
            want = false;
            const uint64_t first = slot_prefix[f];
            for (uint64_t j = s; j > first;) {
                j--;
                const ZdecBlock pb = zblocks[j];
                if (pb.type == 2 && pb.lit_type == 2) { from = j; want = true; break; }
            }
            if (!want) fast[f] = 0;
        }
        if (want) {
            const ZdecBlock sb = zblocks[from];
            dsrc = frames_base + frame_off[f] + sb.lit_off;
            dlen = sb.lit_comp < HUF_DESC_MAX ? sb.lit_comp : HUF_DESC_MAX;
        }
    }
    {
        uint8_t got[33];
#pragma unroll
        for (uint32_t j = 0; j < 33; j++) { const uint32_t at = k * 33 + j; got[j] = at < dlen ? dsrc[at] : (uint8_t)0; }
#pragma unroll
        for (uint32_t j = 0; j < 33; j++) S.blk[i].desc[k * 33 + j] = got[j];
    }
    zd::wave_sync();
    // ---- phase 1b: one lane per block reads the weights ----
    if (k == 0) {
        int nw = 0, used = 0;
        if (want) {
            used = huf_read_weights_lane(S.blk[i], dlen, &nw);
            if (used < 0) { nw = 0; fast[f] = 0; }
        }
        S.nw[i] = nw;
        S.used[i] = (nw && zb.lit_type == 2) ? (uint32_t)used : 0u;
        S.frame[i] = f;
    }
    zd::wave_sync();
    // ---- phase 1c: the decoders, wave-cooperative, one block after the other ----
    for (int g = 0; g < ZDEC_LIT_GROUP; g++) {
        const int nw = S.nw[g]; // uniform
        int tl = 0;
        if (nw) tl = huf_build_canon(S.blk[g].weights, nw, lane, S.blk[g].canon);
        if (lane == 0) {
            S.bits[g] = tl;
            if (nw && !tl) fast[S.frame[g]] = 0;
        }
    }
    zd::wave_sync();
    // ---- phase 2: one stream per lane ----
    if (s >= n_slots || S.bits[i] == 0) return;
    if (zb.lit_streams == 1 && k != 0) return;
    bool ok = S.used[i] <= zb.lit_comp;
    if (ok) {
        const uint8_t *hp = frames_base + frame_off[f] + zb.lit_off + S.used[i];
        const uint32_t rem = zb.lit_comp - S.used[i];
        uint8_t *dst = lits + lit_index[s];
        const HufCanon &ht = S.blk[i].canon;
        const int tl = S.bits[i];
        uint32_t so = 0, sl = rem, dof = 0, dn = zb.lit_len; // single stream: all of it
        if (zb.lit_streams != 1) {
            if (rem < 6) ok = false;
            else {
                const uint32_t s1 = hp[0] | ((uint32_t)hp[1] << 8), s2 = hp[2] | ((uint32_t)hp[3] << 8), s3 = hp[4] | ((uint32_t)hp[5] << 8);
                const uint32_t per = (zb.lit_len + 3) / 4;
                if (6 + s1 + s2 + s3 > rem || per * 3 > zb.lit_len) ok = false;
                else {
                    const uint32_t s4 = rem - 6 - s1 - s2 - s3;
                    so = 6 + (k == 0 ? 0 : (k == 1 ? s1 : (k == 2 ? s1 + s2 : s1 + s2 + s3)));
                    sl = k == 0 ? s1 : (k == 1 ? s2 : (k == 2 ? s3 : s4));
                    dof = k * per;
                    dn = k < 3 ? per : zb.lit_len - 3 * per;
                }
            }
        }
        if (ok) ok = huf_decode_stream_canon(ht, tl, hp + so, sl, dst + dof, dn);
    }
    if (!ok) fast[f] = 0;
}
