// zarc_amd/csrc/zge_entropy.hip -- encoder stage 2: Huffman literals + FSE sequences -> block bytes (gfx950).
//
// Part of the replacement for `CCtx::compress2` at crates/zarc/src/encode/lowlevel_frames.rs:29-31.
// One wave per block (64 KiB since round 4; blocks are independent here but for the table plan below, and
// the repcode history restarts per block), so tens of thousands of waves are in flight and the serial
// pieces (tree construction, table normalisation, FSE state chains) are latency-hidden by occupancy:
//   - byte histograms with LDS atomics, symbol ranking by counting (256 symbols, 4 per lane)
//   - Huffman code lengths (two-queue merge, limited to 11 bits), canonical codes, weight description
//     (direct or FSE-compressed) on lane 0
//   - literal streams: 64 symbols per step, wave prefix-sum of code lengths, codes OR-ed into an LDS
//     staging window (ds_or), whole bytes flushed to HBM
//   - sequences: literal lengths and repeat-offset codes for 64 sequences at a time (the history after sequence i
//     is a function of few neighbours: two ballots + count-leading-zeros); code histograms in parallel; the three
//     FSE state chains run on lanes 0..2 with the per-symbol constants fetched by all lanes beforehand (one
//     dependent LDS access per step), then all 64 lanes pack their sequence's bit fields the same way
// Bit-identical to oracle/zstd_enc_model.c (encode_literals / encode_sequences).
#include "zarc_device.h"
#include "zarc_kernels.h"

namespace {

#ifdef ZARC_HIPEMU
#define ZGE_CLOCK() 0ull
#else
#define ZGE_CLOCK() ((unsigned long long)__builtin_readcyclecounter())
#endif
#define ENT_PROF(i) do { if (prof && lane == 0) { const unsigned long long now_ = ZGE_CLOCK(); atomicAdd(prof + (i), now_ - tprev); tprev = now_; } } while (0)

constexpr int HUF_MAXBITS = 11;
constexpr uint32_t MIN_HUF_LITERALS = 64;

__constant__ const int16_t E_LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                               2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ const int16_t E_ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                               1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                               1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ const int16_t E_OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                               1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
// Literal-length / match-length codes (RFC 8878 3.1.1.3.2.1.1), and the extra bits of a value, worked out arithmetically: the code
// tables have a regular shape (runs of 2, 4, 8, ... values per code, then one code per power of two), and a table in memory would be a
// vector load per sequence in two passes of the coder.
__device__ __forceinline__ uint32_t ll_code(uint32_t ll)
{
    if (ll < 16) return ll;
    if (ll >= 64) return (uint32_t)zd::hb32(ll) + 19;
    if (ll < 24) return 16 + ((ll - 16) >> 1);   // 16..19: two values each
    if (ll < 32) return 20 + ((ll - 24) >> 2);   // 20, 21: four
    if (ll < 48) return 22 + ((ll - 32) >> 3);   // 22, 23: eight
    return 24;                                   // 48..63
}
__device__ __forceinline__ uint32_t ml_code(uint32_t ml)
{
    const uint32_t b = ml - 3;
    if (b < 32) return b;
    if (b >= 128) return (uint32_t)zd::hb32(b) + 36;
    if (b < 40) return 32 + ((b - 32) >> 1);     // 32..35: two values each
    if (b < 48) return 36 + ((b - 40) >> 2);     // 36, 37: four
    if (b < 64) return 38 + ((b - 48) >> 3);     // 38, 39: eight
    if (b < 96) return 40 + ((b - 64) >> 4);     // 40, 41: sixteen
    return 42;                                   // 96..127
}
// number of extra bits of a code and the extra-bits value of `v` (v minus the code's baseline = its low bits: every run of a code
// starts at a multiple of its length)
__device__ __forceinline__ uint32_t ll_extra(uint32_t ll, uint32_t code, uint32_t *bits)
{
    const uint32_t nb = code < 16 ? 0u : (code >= 25 ? code - 19 : (uint32_t)((0x433221111ull >> (4 * (code - 16))) & 15));
    *bits = nb;
    return ll & ((1u << nb) - 1); // baselines 16, 18, .. 24, 28, 32, 40, 48, 64, 128, ..: all multiples of 2^nb
}
__device__ __forceinline__ uint32_t ml_extra(uint32_t ml, uint32_t code, uint32_t *bits)
{
    const uint32_t nb = code < 32 ? 0u : (code >= 43 ? code - 36 : (uint32_t)((0x54433221111ull >> (4 * (code - 32))) & 15));
    *bits = nb;
    return (ml - 3) & ((1u << nb) - 1); // baselines - 3 = 32, 34, .. 40, 44, 48, 56, 64, 80, 96, 128, ..: multiples of 2^nb
}

__device__ __forceinline__ uint32_t ll_code_bits(uint32_t code) { return code < 16 ? 0u : (code >= 25 ? code - 19 : (uint32_t)((0x433221111ull >> (4 * (code - 16))) & 15)); }
__device__ __forceinline__ uint32_t ml_code_bits(uint32_t code) { return code < 32 ? 0u : (code >= 43 ? code - 36 : (uint32_t)((0x54433221111ull >> (4 * (code - 32))) & 15)); }

// ---- lane-serial bit writer into a byte buffer (LDS or global) ----
struct BitW {
    uint8_t *p; uint32_t pos, cap; uint64_t acc; int nb; bool overflow;
    __device__ void init(uint8_t *ptr, uint32_t c) { p = ptr; pos = 0; cap = c; acc = 0; nb = 0; overflow = false; }
    __device__ void add(uint32_t v, int n)
    {
        if (n == 0) return;
        acc |= (uint64_t)(v & (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1))) << nb;
        nb += n;
        while (nb >= 8) {
            if (pos < cap) p[pos] = (uint8_t)acc; else overflow = true;
            pos++; acc >>= 8; nb -= 8;
        }
    }
    __device__ void flush_partial() { if (nb > 0) { if (pos < cap) p[pos] = (uint8_t)acc; else overflow = true; pos++; nb = 0; acc = 0; } }
    __device__ uint32_t close() { add(1, 1); flush_partial(); return pos; }
};

struct FseCtab { uint16_t *state_tab; int32_t *dnb; int32_t *dfs; int al; };

// The Huffman-construction arrays are dead once the literals section is coded, and the FSE sequence tables are
// only needed after it: the two live in a union, which keeps the footprint under 10 KiB (16 waves per CU).
struct EntLds {
    uint16_t code[256]; // code | len << 11 (codes are at most 11 bits long)
    union { uint32_t stage[224]; uint8_t cellsym[512]; }; // bit-packing window; FSE table construction happens between packings
    int32_t ctrl[32];
    union {
        struct {
            uint32_t count[256];
            uint32_t w[512];
            uint16_t parent[512];
            uint16_t order[256];
            uint8_t depth[512];
            uint8_t len8[256], wt[256];
            uint16_t st_w[64];
            int32_t dnb_w[16], dfs_w[16];
            uint32_t cw[16];
            int16_t wnorm[16];
            uint8_t tmp[192], hdesc[192];
        } h;
        struct {
            uint16_t st[1280];     // state tables: LL at 0, ML at 512, OF at 1024 (one array: the chains index it with the table's start folded into the symbol constant)
            int32_t dnb_ll[36], dfs_ll[36], dnb_ml[53], dfs_ml[53], dnb_of[32], dfs_of[32];
            int16_t norm[3][64];   // (8-byte aligned: once the tables are built, the chains keep the states before their steps here, four to a 64-bit store)
            uint32_t cl[36], co[32], cm[53];
            uint8_t desc[3][80];
            uint64_t pre[3][64];   // LL / OF / ML of the current 64 sequences: {dnb, byte offset of dfs in st[]} of each sequence's symbol
        } s;
    };
};
enum { ST_LL = 0, ST_ML = 512, ST_OF = 1024 };
enum { X_TMP = 0, X_DLEN = 1, X_MODE_L = 2, X_MODE_O = 3, X_MODE_M = 4, X_AL_L = 5, X_AL_O = 6, X_AL_M = 7, X_DL_L = 8, X_DL_O = 9,
       X_DL_M = 10, X_RLE_L = 11, X_RLE_O = 12, X_RLE_M = 13, X_NSYM_L = 14, X_NSYM_O = 15, X_NSYM_M = 16, X_MAXBITS = 17, X_NSYM_LAST = 18,
       X_COST_L = 19, X_COST_O = 20, X_COST_M = 21 /* the own choice's cost in 1/256 bit, description included (< 2^27: 43 690 sequences x 9 bits x 256) */ };

// ---- FSE helpers (lane-serial; same arithmetic as the model) ----
__device__ void fse_build_ctab(FseCtab &t, const int16_t *norm, int nsym, int al, uint8_t *cellsym)
{
    const int T = 1 << al, step = (T >> 1) + (T >> 3) + 3, mask = T - 1;
    int high = T - 1, pos = 0;
    for (int s = 0; s < nsym; s++) if (norm[s] == -1) cellsym[high--] = (uint8_t)s;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            cellsym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    // cumulative starts; dfs[] doubles as the running "next" index, then gets its final value
    int total = 0;
    for (int s = 0; s < nsym; s++) { t.dfs[s] = total; total += norm[s] == -1 ? 1 : norm[s]; }
    // every cell takes the next slot of its symbol: an LDS atomic with return per cell -- LDS executes them in order, so the eight of a
    // trip are in flight together instead of one read-modify-write round trip after the other
    for (int i = 0; i < T; i += 8) {
        int slot[8];
#pragma unroll
        for (int k = 0; k < 8; k++) slot[k] = atomicAdd(&t.dfs[cellsym[i + k]], 1);
#pragma unroll
        for (int k = 0; k < 8; k++) t.state_tab[slot[k]] = (uint16_t)(T + i + k);
    }
    total = 0;
    for (int s = 0; s < nsym; s++) {
        const int n = norm[s];
        if (n == 0) { t.dnb[s] = ((al + 1) << 16) - T; t.dfs[s] = 0; continue; }
        if (n == -1 || n == 1) { t.dnb[s] = (al << 16) - T; t.dfs[s] = total - 1; total += 1; }
        else {
            const int max_bits_out = al - zd::hb32((uint32_t)(n - 1));
            const int min_state_plus = n << max_bits_out;
            t.dnb[s] = (max_bits_out << 16) - min_state_plus;
            t.dfs[s] = total - n;
            total += n;
        }
    }
    t.al = al;
}
// The same table built by the whole wave (uniform call; norm[] / nsym / al as above, `cellsym` >= 2^al bytes of LDS scratch):
//   starts      an exclusive prefix sum of the symbols' cell counts (a "less than one" symbol, norm -1, has one cell)
//   spread      the j-th visited position is (j * step) mod T, positions above `high` (the less-than-one symbols' cells at the top) are
//               skipped: every lane takes T/64 consecutive j, a wave scan of the valid counts gives each valid j its occupant rank,
//               a binary search over the starts the rank's symbol -- instead of T dependent steps on one lane
//   state table every cell takes the next slot of its symbol, 64 cells per round: a cell's slot is its symbol's running slot plus the
//               number of lower lanes of the round that hold the same symbol (six ballots give every lane the mask of its symbol's
//               lanes), so cells in rising order get rising slots, as the serial loop gives them -- computed, not left to the order in
//               which the LDS executes same-address atomics of one instruction (round 3 used ds_add_rtn here; a different order would
//               have produced frames that decode to wrong bytes)
//   per symbol  dnb / dfs, one lane per symbol
__device__ void fse_build_ctab_wave(FseCtab &t, const int16_t *norm, int nsym, int al, uint8_t *cellsym, int lane)
{
    const int T = 1 << al, step = (T >> 1) + (T >> 3) + 3, mask = T - 1;
    // cell counts and their exclusive prefix (nsym <= 64: one symbol per lane)
    const int n_l = lane < nsym ? (int)norm[lane] : 0;
    const uint32_t cells_l = n_l == -1 ? 1u : (uint32_t)n_l;          // all cells of this symbol (state-table slots)
    const uint32_t reg_l = n_l > 0 ? (uint32_t)n_l : 0u;              // its cells in the spread region
    const uint32_t start_all = zd::wave_scan_incl(cells_l) - cells_l;
    const uint32_t start_reg = zd::wave_scan_incl(reg_l) - reg_l;
    const uint64_t lowm = zd::ballot(n_l == -1);
    const int nlow = (int)__popcll(lowm), high = T - 1 - nlow;
    if (n_l == -1) cellsym[T - 1 - (int)__popcll(lowm & ((1ull << lane) - 1))] = (uint8_t)lane; // the first less-than-one symbol takes the top cell
    if (lane < nsym) { t.dfs[lane] = (int32_t)start_all; t.dnb[lane] = (int32_t)start_reg; }     // dnb[] doubles as the spread starts for the search below
    zd::wave_sync();
    {
        const int per = T >= 64 ? T / 64 : 1;
        const int j0 = lane * per;
        uint32_t valid = 0; // bit q: position of j0 + q is in the spread region
        if (j0 < T) for (int q = 0; q < per; q++) if ((((j0 + q) * step) & mask) <= high) valid |= 1u << q;
        const uint32_t cnt = (uint32_t)__popc(valid);
        uint32_t rank = zd::wave_scan_incl(cnt) - cnt;
        for (int q = 0; q < per; q++) {
            if (!((valid >> q) & 1)) continue;
            // the symbol whose spread range holds this rank: the last symbol with start_reg <= rank and a positive count
            int lo = 0, hi = nsym - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((uint32_t)t.dnb[mid] <= rank) lo = mid; else hi = mid - 1; }
            while (norm[lo] <= 0) lo--; // symbols without cells in the region share their successor's start: step back to the owner
            cellsym[((j0 + q) * step) & mask] = (uint8_t)lo;
            rank++;
        }
    }
    zd::wave_sync();
    for (int i0 = 0; i0 < T; i0 += 64) {
        const int i = i0 + lane;
        const bool act = i < T;
        const uint32_t sym = act ? (uint32_t)cellsym[i] : 64u; // (64: matches no symbol; T < 64 only below accuracy 6)
        // lanes of this group that hold the same symbol: six ballots, one per bit of the symbol -- no reliance on the order in which the
        // LDS would execute same-address atomics of one instruction
        uint64_t same = zd::ballot(act);
#pragma unroll
        for (int b = 0; b < 6; b++) { const uint64_t m = zd::ballot((sym >> b) & 1u); same &= ((sym >> b) & 1u) ? m : ~m; }
        const uint32_t below = (uint32_t)__popcll(same & ((1ull << lane) - 1)), all = (uint32_t)__popcll(same);
        int base = 0;
        if (act) base = t.dfs[sym];
        zd::wave_lds_order(); // every lane has read its symbol's running slot before the group's last cell of that symbol moves it on
        if (act) {
            t.state_tab[base + (int)below] = (uint16_t)(T + i);
            if (below + 1 == all) t.dfs[sym] = base + (int)all;
        }
        zd::wave_lds_order(); // one group's 64 cells before the next one's (program order on hardware; the emulator needs the rendezvous)
    }
    zd::wave_sync();
    if (lane < nsym) {
        const int n = n_l, total = (int)start_all;
        if (n == 0) { t.dnb[lane] = ((al + 1) << 16) - T; t.dfs[lane] = 0; }
        else if (n == -1 || n == 1) { t.dnb[lane] = (al << 16) - T; t.dfs[lane] = total - 1; }
        else {
            const int max_bits_out = al - zd::hb32((uint32_t)(n - 1));
            t.dnb[lane] = (max_bits_out << 16) - (n << max_bits_out);
            t.dfs[lane] = total - n;
        }
    }
    t.al = al;
    zd::wave_sync();
}
__device__ __forceinline__ uint32_t fse_init_state(const FseCtab &t, int s)
{
    const int nb = (t.dnb[s] + (1 << 15)) >> 16;
    const int value = (nb << 16) - t.dnb[s];
    return t.state_tab[(value >> nb) + t.dfs[s]];
}
// returns the new state; *bits = value | nb << 16
__device__ __forceinline__ uint32_t fse_step(const FseCtab &t, uint32_t state, int s, uint32_t *bits)
{
    const int nb = (int)((state + (uint32_t)t.dnb[s]) >> 16);
    *bits = (state & ((1u << nb) - 1)) | ((uint32_t)nb << 16);
    return t.state_tab[(int)(state >> nb) + t.dfs[s]];
}
__device__ uint32_t log2_fp8(uint32_t x)
{
    const int h = zd::hb32(x);
    const uint32_t frac = h >= 8 ? (x >> (h - 8)) - 256 : (x << (8 - h)) - 256;
    return (uint32_t)h * 256 + frac;
}
__device__ void fse_normalize(const uint32_t *count, int nsym, uint32_t total, int al, int16_t *norm)
{
    const int T = 1 << al;
    int sum = 0;
    for (int s = 0; s < nsym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        uint64_t v = ((uint64_t)count[s] * (uint64_t)T + total / 2) / total;
        if (v < 1) v = 1;
        norm[s] = (int16_t)v;
        sum += (int)v;
    }
    while (sum != T) {
        int best = -1;
        for (int s = 0; s < nsym; s++) if (norm[s] > 0 && (best < 0 || norm[s] > norm[best])) best = s;
        if (sum > T) {
            int take = sum - T;
            const int room = norm[best] - 1;
            if (take > room) take = room;
            if (take <= 0) break;
            norm[best] = (int16_t)(norm[best] - take);
            sum -= take;
        } else {
            norm[best] = (int16_t)(norm[best] + (T - sum));
            sum = T;
        }
    }
}
__device__ uint32_t fse_write_desc(uint8_t *dst, uint32_t cap, const int16_t *norm, int nsym, int al)
{
    BitW b;
    b.init(dst, cap);
    int remaining = (1 << al) + 1, threshold = 1 << al, nb = al + 1, s = 0;
    b.add((uint32_t)(al - 5), 4);
    while (remaining > 1 && s < nsym) {
        const int count = norm[s++], max = 2 * threshold - 1 - remaining;
        int val = count + 1;
        remaining -= count < 0 ? -count : count;
        if (val >= threshold) val += max;
        if (val < max) b.add((uint32_t)val, nb - 1); else b.add((uint32_t)val, nb);
        if (count == 0) {
            int run = 0;
            while (s + run < nsym && norm[s + run] == 0) run++;
            s += run;
            while (run >= 3) { b.add(3, 2); run -= 3; }
            b.add((uint32_t)run, 2);
        }
        while (remaining < threshold) { nb--; threshold >>= 1; }
    }
    b.flush_partial();
    return b.overflow ? 0 : b.pos;
}
__device__ uint64_t dist_cost(const uint32_t *count, const int16_t *norm, int nsym, int al)
{
    uint64_t c = 0;
    for (int s = 0; s < nsym; s++) {
        if (!count[s]) continue;
        if (norm[s] == 0) return ~0ull;
        const uint32_t n = norm[s] < 0 ? 1u : (uint32_t)norm[s];
        c += (uint64_t)count[s] * ((uint32_t)al * 256 - log2_fp8(n));
    }
    return c;
}
// lane 0: pick predefined / RLE / dynamic table; results go to ctrl[] slots starting at `slot` offsets
__device__ void choose_table(EntLds &L, int which, const uint32_t *count, int maxsym, uint32_t nseq, const int16_t *def, int def_n,
                             int def_al, int max_al)
{
    int distinct = 0, last = 0;
    for (int s = 0; s <= maxsym; s++) if (count[s]) { distinct++; last = s; }
    int16_t *norm = L.s.norm[which];
    L.ctrl[X_COST_L + which] = 0;
    if (distinct == 1) { L.ctrl[X_MODE_L + which] = 1; L.ctrl[X_RLE_L + which] = last; L.ctrl[X_DL_L + which] = 0; return; }
    uint64_t cost_def = ~0ull;
    if (last < def_n) {
        // the predefined distribution lives in constant memory; stage it in norm[] to reuse dist_cost
        for (int s = 0; s < def_n; s++) norm[s] = def[s];
        cost_def = dist_cost(count, norm, def_n, def_al);
    }
    int al = zd::hb32(nseq > 1 ? nseq - 1 : 1) - 2;
    if (al > max_al) al = max_al;
    if (al < 5) al = 5;
    while ((1 << al) < distinct) al++;
    const int nsym = last + 1;
    fse_normalize(count, nsym, nseq, al, norm);
    const uint32_t dl = fse_write_desc(L.s.desc[which], 80, norm, nsym, al);
    const uint64_t cost_dyn = dl ? dist_cost(count, norm, nsym, al) + (uint64_t)dl * 8 * 256 : ~0ull;
    if (cost_def <= cost_dyn) {
        for (int s = 0; s < def_n; s++) norm[s] = def[s];
        L.ctrl[X_MODE_L + which] = 0; L.ctrl[X_NSYM_L + which] = def_n; L.ctrl[X_AL_L + which] = def_al; L.ctrl[X_DL_L + which] = 0;
        L.ctrl[X_COST_L + which] = (int)(uint32_t)cost_def;
    } else {
        L.ctrl[X_MODE_L + which] = 2; L.ctrl[X_NSYM_L + which] = nsym; L.ctrl[X_AL_L + which] = al; L.ctrl[X_DL_L + which] = (int)dl;
        L.ctrl[X_COST_L + which] = (int)(uint32_t)cost_dyn;
    }
}

// ---- Huffman construction (lane 0 for the serial parts) ----
// L.h.count -> L.h.len8 (code lengths).  Uniform; returns number of present symbols.
__device__ int huf_build_lengths(EntLds &L, int lane)
{
    // rank by counting over the PRESENT symbols only: keys count << 8 | symbol, compacted (ascending (count, symbol) = ascending key);
    // key i's rank is the number of smaller keys.  (w[256..] is free until the tree is built.)
    uint32_t present = 0;
    uint32_t *const keys = &L.h.w[256];
    {
        const uint64_t lt = (1ull << lane) - 1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int s = r * 64 + lane;
            const uint32_t c = L.h.count[s];
            L.h.len8[s] = 0;
            const uint64_t m = zd::ballot(c != 0);
            if (c) keys[present + (uint32_t)__popcll(m & lt)] = (c << 8) | (uint32_t)s;
            present += (uint32_t)__popcll(m);
        }
    }
    zd::wave_sync();
    for (uint32_t i = (uint32_t)lane; i < present; i += 64) {
        const uint32_t k = keys[i];
        uint32_t rank = 0;
        for (uint32_t t = 0; t < present; t++) rank += keys[t] < k ? 1u : 0u;
        L.h.order[rank] = (uint16_t)(k & 0xFF);
    }
    zd::wave_sync();
    const int n = (int)present;
    if (n < 2) { if (n == 1 && lane == 0) L.h.len8[L.h.order[0]] = 1; zd::wave_sync(); return n; }
    for (int i = lane; i < n; i += 64) L.h.w[i] = L.h.count[L.h.order[i]];
    zd::wave_sync();
    if (lane == 0) { // the two-queue merge is a chain: one lane
        int leaf = 0, inode = n, next = n;
        while (next < 2 * n - 1) {
            int a, b;
            if (leaf < n && (inode >= next || L.h.w[leaf] <= L.h.w[inode])) a = leaf++; else a = inode++;
            if (leaf < n && (inode >= next || L.h.w[leaf] <= L.h.w[inode])) b = leaf++; else b = inode++;
            L.h.w[next] = L.h.w[a] + L.h.w[b];
            L.h.parent[a] = (uint16_t)next;
            L.h.parent[b] = (uint16_t)next;
            next++;
        }
    }
    zd::wave_sync();
    // leaf depths: every lane walks its leaves up to the root (only the leaves' depths are used; the serial top-down pass over all 2n - 1
    // nodes was a sixth of this function), capped at 63 like the model; histogram of the depths with LDS atomics
    int *const num = (int *)&L.h.w[0]; // the merge weights are dead from here on; a local array would live in scratch (HBM)
    {
        uint32_t dep[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = r * 64 + lane;
            dep[r] = 0;
            if (i < n) { int j = i; uint32_t d = 0; while (j != 2 * n - 2) { j = L.h.parent[j]; d++; } dep[r] = d > 63 ? 63 : d; }
        }
        zd::wave_sync(); // every walk is done before w[] is reused
        for (int k = lane; k < 64; k += 64) num[k] = 0;
        zd::wave_sync();
#pragma unroll
        for (int r = 0; r < 4; r++) if (r * 64 + lane < n) atomicAdd(&num[dep[r]], 1);
        zd::wave_sync();
    }
    if (lane == 0) { // limit to HUF_MAXBITS and repair the Kraft sum: a few steps over at most 64 counters
        for (int k = HUF_MAXBITS + 1; k < 64; k++) { num[HUF_MAXBITS] += num[k]; num[k] = 0; }
        uint32_t total = 0;
        for (int k = 1; k <= HUF_MAXBITS; k++) total += (uint32_t)num[k] << (HUF_MAXBITS - k);
        while (total != (1u << HUF_MAXBITS)) {
            num[HUF_MAXBITS]--;
            for (int k = HUF_MAXBITS - 1; k > 0; k--) if (num[k]) { num[k]--; num[k + 1] += 2; break; }
            total--;
        }
    }
    zd::wave_sync();
    // the most frequent symbols (end of `order`) get the shortest codes: the symbol t places from the top gets the smallest length k with
    // num[1] + .. + num[k] > t
    {
        uint32_t cum[HUF_MAXBITS + 1];
        uint32_t c = 0;
#pragma unroll
        for (int k = 1; k <= HUF_MAXBITS; k++) { c += (uint32_t)num[k]; cum[k] = c; }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = r * 64 + lane;
            if (j < n) {
                const uint32_t t = (uint32_t)(n - 1 - j);
                uint32_t k = 1;
#pragma unroll
                for (int q = 1; q < HUF_MAXBITS; q++) k += cum[q] <= t ? 1u : 0u;
                L.h.len8[L.h.order[j]] = (uint8_t)k;
            }
        }
    }
    zd::wave_sync();
    return n;
}

// Canonical codes in zstd weight order; fills L.code, ctrl[X_MAXBITS], ctrl[X_NSYM_LAST].  Whole wave, four symbols per lane (symbol
// r * 64 + lane): a symbol's code is the first code of its weight plus the number of lower symbols of the same length -- a ballot and a
// population count per length and round instead of three 256-step loops on one lane (they were a third of the stage's fixed cost per
// block).  Same arithmetic as the model: rank_start[w] = sum over lighter weights of count << (weight - 1), code = (rank_start[w] +
// index << (w - 1)) >> (w - 1).
__device__ void huf_assign_codes(EntLds &L, int lane)
{
    uint32_t len[4], idx[4];
    uint32_t maxlen = 0, last = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) { len[r] = L.h.len8[r * 64 + lane]; idx[r] = 0; if (len[r] > maxlen) maxlen = len[r]; if (len[r]) last = (uint32_t)(r * 64 + lane); }
    maxlen = zd::uniform(zd::wave_max(maxlen));
    last = zd::uniform(zd::wave_max(last));
    const uint64_t lt = (1ull << lane) - 1;
    uint32_t *const rank_start = &L.h.w[64]; // LDS (w[] is free after the tree is built)
    uint32_t pos = 0;
    for (uint32_t w = 1; w <= maxlen; w++) { // uniform: weight w = code length maxlen + 1 - w, lightest first
        const uint32_t lv = maxlen + 1 - w;
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint64_t m = zd::ballot(len[r] == lv);
            if (len[r] == lv) idx[r] = cnt + (uint32_t)__popcll(m & lt);
            cnt += (uint32_t)__popcll(m);
        }
        if (lane == 0) rank_start[w] = pos;
        pos += cnt << (w - 1);
    }
    zd::wave_sync();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t code = 0;
        if (len[r]) { const uint32_t w = maxlen + 1 - len[r]; code = ((rank_start[w] >> (w - 1)) + idx[r]) | (len[r] << 11); }
        L.code[r * 64 + lane] = (uint16_t)code;
    }
    if (lane == 0) { L.ctrl[X_MAXBITS] = (int)maxlen; L.ctrl[X_NSYM_LAST] = (int)last; }
    zd::wave_sync();
}

// lane 0: Huffman tree description into L.h.hdesc; returns its length (0 = not representable)
__device__ uint32_t huf_write_desc(EntLds &L)
{
    const int n = L.ctrl[X_NSYM_LAST]; // (the weights L.h.wt[] were filled in by the whole wave)
    if (n > 1) { // (so was the histogram of the weights of symbols 0 .. n-1, L.h.cw[])
        int nsym = 0, distinct = 0, al = 6;
        uint32_t maxc = 0;
        for (int s = 0; s < 13; s++) if (L.h.cw[s]) { nsym = s + 1; distinct++; if (L.h.cw[s] > maxc) maxc = L.h.cw[s]; }
        if (distinct > 1 && maxc > 1) {
            { int lim = zd::hb32((uint32_t)(n - 1)) - 2; if (lim < al) al = lim; if (al < 5) al = 5; }
            while ((1 << al) < distinct) al++;
            fse_normalize(L.h.cw, nsym, (uint32_t)n, al, L.h.wnorm);
            const uint32_t hdr = fse_write_desc(L.h.tmp, 192, L.h.wnorm, nsym, al);
            if (hdr) {
                FseCtab ct = {L.h.st_w, L.h.dnb_w, L.h.dfs_w, 0};
                fse_build_ctab(ct, L.h.wnorm, nsym, al, L.cellsym);
                BitW b;
                b.init(L.h.tmp + hdr, 192 - hdr);
                uint32_t s1, s2, bits;
                int ip = n;
                if (n & 1) {
                    s1 = fse_init_state(ct, L.h.wt[--ip]);
                    s2 = fse_init_state(ct, L.h.wt[--ip]);
                    s1 = fse_step(ct, s1, L.h.wt[--ip], &bits); b.add(bits & 0xFFFF, (int)(bits >> 16));
                } else {
                    s2 = fse_init_state(ct, L.h.wt[--ip]);
                    s1 = fse_init_state(ct, L.h.wt[--ip]);
                }
                while (ip > 0) {
                    s2 = fse_step(ct, s2, L.h.wt[--ip], &bits); b.add(bits & 0xFFFF, (int)(bits >> 16));
                    s1 = fse_step(ct, s1, L.h.wt[--ip], &bits); b.add(bits & 0xFFFF, (int)(bits >> 16));
                }
                b.add(s2, al);
                b.add(s1, al);
                const uint32_t body = b.close();
                if (!b.overflow && hdr + body < 128 && (n > 128 || hdr + body < (uint32_t)(n + 1) / 2)) {
                    L.h.hdesc[0] = (uint8_t)(hdr + body);
                    for (uint32_t i = 0; i < hdr + body; i++) L.h.hdesc[1 + i] = L.h.tmp[i];
                    return 1 + hdr + body;
                }
            }
        }
    }
    if (n > 128) return 0;
    L.h.hdesc[0] = (uint8_t)(127 + n);
    for (int i = 0; i < n; i += 2) L.h.hdesc[1 + i / 2] = (uint8_t)((L.h.wt[i] << 4) | (i + 1 < n ? L.h.wt[i + 1] : 0));
    return (uint32_t)(1 + (n + 1) / 2);
}

// Wave-parallel bit packer: fields are OR-ed into an LDS staging window and whole bytes are flushed to `dst`.
struct WavePacker {
    uint32_t *stage; // LDS, >= 224 words
    uint8_t *dst;
    uint32_t pos;    // bytes flushed so far
    uint32_t carry;  // pending bits (0..7) kept in stage[0]'s low byte
    uint32_t cap;    // bytes available at dst
    bool overflow;
    __device__ void begin(uint32_t *st, uint8_t *d, uint32_t c, int lane)
    {
        stage = st; dst = d; pos = 0; carry = 0; cap = c; overflow = false;
        for (int i = lane; i < 224; i += 64) stage[i] = 0;
        zd::wave_sync();
    }
    // every lane contributes (lo,hi) = up to 96 bits, `nbits` of them, in lane order
    __device__ void put(uint64_t lo, uint32_t hi, uint32_t nbits, int lane)
    {
        const uint32_t incl = zd::wave_scan_incl(nbits);
        const uint32_t total = zd::readlane(incl, 63);
        const uint32_t at = carry + incl - nbits;
        if (nbits) {
            const uint32_t w = at >> 5, sh = at & 31;
            // 96-bit value shifted left by sh (<32) -> up to four 32-bit words
            const uint32_t v0 = (uint32_t)lo, v1 = (uint32_t)(lo >> 32), v2 = hi;
            const uint32_t o0 = v0 << sh;
            const uint32_t o1 = sh ? (v1 << sh) | (v0 >> (32 - sh)) : v1;
            const uint32_t o2 = sh ? (v2 << sh) | (v1 >> (32 - sh)) : v2;
            const uint32_t o3 = sh ? (v2 >> (32 - sh)) : 0u;
            if (o0) atomicOr(&stage[w], o0);
            if (o1) atomicOr(&stage[w + 1], o1);
            if (o2) atomicOr(&stage[w + 2], o2);
            if (o3) atomicOr(&stage[w + 3], o3);
        }
        zd::wave_sync();
        const uint32_t bits = carry + total, full = bits >> 3;
        const uint8_t *sb = (const uint8_t *)stage;
        if (pos + full + 1 > cap) overflow = true; // keep one byte for the final partial byte
        if (!overflow) { // whole words first (no alignment needed for the store), then the last 1..3 bytes
            const uint32_t fw = full >> 2;
            for (uint32_t i = (uint32_t)lane; i < fw; i += 64) { const uint32_t w = stage[i]; __builtin_memcpy(dst + pos + 4 * i, &w, 4); }
            if ((uint32_t)lane < (full & 3)) dst[pos + 4 * fw + (uint32_t)lane] = sb[4 * fw + (uint32_t)lane];
        }
        const uint32_t keep = (bits & 7) ? (uint32_t)sb[full] : 0u;
        zd::wave_sync();
        const uint32_t words = (bits + 31) / 32 + 1;
        for (uint32_t i = (uint32_t)lane; i < words; i += 64) stage[i] = i == 0 ? keep : 0u;
        zd::wave_sync();
        pos += full;
        carry = bits & 7;
    }
    // end mark + final partial byte; returns total bytes
    __device__ uint32_t finish(int lane)
    {
        put(1, 0, lane == 0 ? 1u : 0u, lane);
        if (carry) {
            if (lane == 0 && !overflow) dst[pos] = (uint8_t)stage[0];
            pos += 1;
            carry = 0;
        }
        return pos;
    }
};

// PHASE 0: the whole stage in one pass (every block chooses its sequence tables alone).  PHASE 1 / 2: the same code cut in two where
// the tables are chosen -- pass 1 ends with the block's code histograms and own choices in plans[bi], zarc_zge_plan settles the tables of
// each group of blocks, pass 2 builds them and codes the sequences.  Blocks without sequences to code are finished by pass 1.
template <int PHASE>
__device__ __forceinline__ void zge_entropy_body(EntLds &L, uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *__restrict__ blocks, uint64_t *__restrict__ seq_scratch,
                                                 const uint8_t *__restrict__ lit_scratch, uint8_t *__restrict__ out_scratch,
                                                 unsigned long long *__restrict__ prof /* stage ticks (diagnostics) or null */, ZgePlan *__restrict__ plans)
{
    const int lane = zd::lane_id();
    unsigned long long tprev = ZGE_CLOCK();
    const uint32_t bi = blockIdx.x;
    if (bi >= n_blocks) return;
    ZgeBlock *rec = blocks + bi;
    ZgePlan *const plan = PHASE ? plans + bi : nullptr;
    if (PHASE == 2 && !plan->active) return;
    if (rec->type == 1) { if (PHASE == 1 && lane == 0) plan->active = 0; return; } // RLE block: nothing to code
    const uint32_t nlit = rec->nlit, src_len = rec->src_len;
    uint32_t nseq = PHASE == 2 ? plan->nseq : rec->nseq;
    uint64_t *seq = seq_scratch + (uint64_t)bi * zge_seq_stride(slot_bytes);
    const uint8_t *lit = lit_scratch + (uint64_t)bi * zge_lit_stride(slot_bytes);
    uint8_t *out = out_scratch + (uint64_t)bi * zge_out_stride(slot_bytes);
    const uint32_t out_cap = (uint32_t)zge_out_stride(slot_bytes);
    bool fail = false;

    // ================= literals section =================
    uint32_t lsz = PHASE == 2 ? plan->lsz : 0u;
    if (PHASE != 2) {
        const uint32_t n = nlit;
        const uint32_t raw_hdr = n < 32 ? 1u : (n < 4096 ? 2u : 3u);
        for (int i = lane; i < 256; i += 64) L.h.count[i] = 0;
        zd::wave_sync();
        { // literal buffers start 16-byte aligned: four bytes per load
            const uint32_t n4 = n / 4;
            const uint32_t *l4 = (const uint32_t *)lit;
            for (uint32_t i0 = 0; i0 < n4; i0 += 256) { // four loads in flight per lane
                uint32_t wv[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) { const uint32_t i = i0 + 64 * k + (uint32_t)lane; wv[k] = i < n4 ? l4[i] : 0u; }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (i0 + 64 * k + (uint32_t)lane >= n4) continue;
                    const uint32_t w = wv[k];
                    atomicAdd(&L.h.count[w & 0xFF], 1u); atomicAdd(&L.h.count[(w >> 8) & 0xFF], 1u);
                    atomicAdd(&L.h.count[(w >> 16) & 0xFF], 1u); atomicAdd(&L.h.count[w >> 24], 1u);
                }
            }
            for (uint32_t i = n4 * 4 + (uint32_t)lane; i < n; i += 64) atomicAdd(&L.h.count[lit[i]], 1u);
        }
        zd::wave_sync();
        ENT_PROF(0);
        uint32_t distinct = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) distinct += (uint32_t)__popcll(zd::ballot(L.h.count[r * 64 + lane] != 0));
        // a flat histogram is not worth a Huffman attempt (the rule libzstd's huf_compress uses: "probably not compressible")
        uint32_t largest = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) { const uint32_t c = L.h.count[r * 64 + lane]; largest = c > largest ? c : largest; }
        largest = zd::uniform(zd::wave_max(largest));
        const bool flat = largest <= (n >> 7) + 4;
        int kind = 0; // 0 raw, 1 rle, 2 huffman
        if (n >= 2 && distinct == 1) kind = 1;
        else if (n >= MIN_HUF_LITERALS && distinct >= 2 && !flat) {
            huf_build_lengths(L, lane);
            huf_assign_codes(L, lane);
            uint32_t est_bits = 0; // at most 131 072 literals x 11 bits
            {
                const int max_bits = L.ctrl[X_MAXBITS], nlast = L.ctrl[X_NSYM_LAST];
                if (lane < 16) L.h.cw[lane] = 0;
                zd::wave_sync();
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int sym = r * 64 + lane;
                    const uint32_t l = L.h.len8[sym];
                    const uint32_t wgt = l ? (uint32_t)(max_bits + 1 - (int)l) : 0u;
                    L.h.wt[sym] = (uint8_t)wgt;       // weights for the tree description (symbols below the last present one are read)
                    if (sym < nlast) atomicAdd(&L.h.cw[wgt], 1u); // ... and their histogram (the description codes them with FSE)
                    est_bits += L.h.count[sym] * l;
                }
                est_bits = zd::uniform(zd::wave_sum(est_bits));
            }
            zd::wave_sync();
            if (lane == 0) {
                const uint32_t dlen = huf_write_desc(L);
                const uint64_t est = dlen + ((uint64_t)est_bits + 7) / 8 + (n >= 256 ? 10 : 1);
                L.ctrl[X_DLEN] = (dlen && est + 3 < n) ? (int)dlen : 0;
            }
            zd::wave_sync();
            ENT_PROF(1);
            const uint32_t dlen = (uint32_t)L.ctrl[X_DLEN];
            zd::wave_sync();
            if (dlen) {
                const bool single = n < 256;
                const uint32_t hdr = single ? 3u : (n < 1024 ? 3u : (n < 16384 ? 4u : 5u));
                uint8_t *body = out + hdr;
                for (uint32_t i = (uint32_t)lane; i < dlen; i += 64) body[i] = L.h.hdesc[i];
                uint32_t pos = dlen, ssz[4] = {0, 0, 0, 0};
                const uint32_t nstreams = single ? 1u : 4u, per = single ? n : (n + 3) / 4, jt = pos;
                if (!single) pos += 6;
                WavePacker pk;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (k >= nstreams) break;
                    const uint32_t beg = k * per, cnt = single ? n : (k < 3 ? per : n - 3 * per);
                    pk.begin(L.stage, body + pos, out_cap - hdr - pos, lane);
                    for (uint32_t j0 = 0; j0 < cnt; j0 += 512) { // eight symbols (<= 88 bits) per lane and step: an eighth of the prefix sums / flushes
                        const uint32_t j = j0 + 8 * (uint32_t)lane;
                        uint64_t ca = 0, cb = 0;     // codes of symbols 0..3 and 4..7 of this lane (the stream runs from the last literal down)
                        uint32_t na = 0, nb_ = 0;
                        if (j + 8 <= cnt) {          // the eight bytes in one load (literal buffers are plain memory: no alignment needed)
                            uint64_t w;
                            __builtin_memcpy(&w, lit + beg + cnt - 8 - j, 8);
#pragma unroll
                            for (uint32_t q = 0; q < 4; q++) { const uint32_t e = L.code[(w >> (8 * (7 - q))) & 0xFF]; ca |= (uint64_t)(e & 0x7FF) << na; na += e >> 11; }
#pragma unroll
                            for (uint32_t q = 4; q < 8; q++) { const uint32_t e = L.code[(w >> (8 * (7 - q))) & 0xFF]; cb |= (uint64_t)(e & 0x7FF) << nb_; nb_ += e >> 11; }
                        } else if (j < cnt) {
                            for (uint32_t q = 0; q < 4 && j + q < cnt; q++) { const uint32_t e = L.code[lit[beg + cnt - 1 - j - q]]; ca |= (uint64_t)(e & 0x7FF) << na; na += e >> 11; }
                            for (uint32_t q = 4; q < 8 && j + q < cnt; q++) { const uint32_t e = L.code[lit[beg + cnt - 1 - j - q]]; cb |= (uint64_t)(e & 0x7FF) << nb_; nb_ += e >> 11; }
                        }
                        const uint64_t lo = ca | (na < 64 ? cb << na : 0ull);
                        const uint32_t hi = na ? (uint32_t)(cb >> (64 - na)) : 0u; // na <= 44 and cb < 2^44: at most 24 bits
                        pk.put(lo, hi, na + nb_, lane);
                    }
                    ssz[k] = pk.finish(lane);
                    if (pk.overflow) fail = true;
                    pos += ssz[k];
                }
                if (!single) {
                    if (lane == 0) for (int k = 0; k < 3; k++) { body[jt + 2 * k] = (uint8_t)ssz[k]; body[jt + 2 * k + 1] = (uint8_t)(ssz[k] >> 8); }
                    if (ssz[0] > 65535 || ssz[1] > 65535 || ssz[2] > 65535) fail = true;
                }
                const uint32_t comp = pos;
                if (!fail && comp + hdr < n + raw_hdr) {
                    uint64_t v;
                    if (hdr == 3) v = 2u | ((single ? 0u : 1u) << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << 14);
                    else if (hdr == 4) v = 2u | (2u << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << 18);
                    else v = 2u | (3u << 2) | ((uint64_t)n << 4) | ((uint64_t)comp << 22);
                    if (lane == 0) for (uint32_t i = 0; i < hdr; i++) out[i] = (uint8_t)(v >> (8 * i));
                    lsz = hdr + comp;
                    kind = 2;
                }
            }
        }
        if (kind == 1) {
            if (lane == 0) {
                if (raw_hdr == 1) out[0] = (uint8_t)(1 | (n << 3));
                else if (raw_hdr == 2) { out[0] = (uint8_t)(1 | (1 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); }
                else { out[0] = (uint8_t)(1 | (3 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); out[2] = (uint8_t)(n >> 12); }
                out[raw_hdr] = lit[0];
            }
            lsz = raw_hdr + 1;
        } else if (kind == 0) {
            if (lane == 0) {
                if (raw_hdr == 1) out[0] = (uint8_t)(0 | (n << 3));
                else if (raw_hdr == 2) { out[0] = (uint8_t)(0 | (1 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); }
                else { out[0] = (uint8_t)(0 | (3 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); out[2] = (uint8_t)(n >> 12); }
            }
            // a block without sequences whose literals stay raw cannot beat its own size: it becomes a raw block (copied from the
            // source by the assembly pass), so the literal bytes need not be moved here
            if (nseq != 0) for (uint32_t i = (uint32_t)lane; i < n; i += 64) out[raw_hdr + i] = lit[i];
            lsz = raw_hdr + n;
        }
    }
    if (PHASE != 2) zd::wave_sync_global();
    ENT_PROF(2);

    // ================= sequences pre-pass =================
    // The match finder stores (literal position, match length, offset); here the literal length becomes the
    // difference of neighbouring literal positions and the offset is resolved against the repcode history
    // (RFC 8878 3.1.1.5).  It starts UNKNOWN (0) in every block because blocks are coded independently of their
    // predecessors' type.
    // The history after sequence i is a function of few neighbours, so 64 sequences resolve at once:
    //   r0_i = o_i                                   (every rule leaves the coded offset in front)
    //   r1_i = keep1_i ? r1_{i-1} : o_{i-1}          keep1 = literals present and o_i == r0
    //   r2_i = keep2_i ? r2_{i-1} : r1_{i-1}         keep2 = o_i hits r0/r1 (with literals) or r1 (without)
    // "the value at the last position that did not keep" is a ballot + count-leading-zeros per lane.
    // Before that, pieces of one long match are joined: a sequence without literals that continues its predecessor at the same
    // offset (the finder caps a match at 256 bytes per position and at its tile's overrun window) is added to the predecessor's
    // length, so match lengths reach the format's 131 074 (model: merge_sequences).  64 sequences per round, compacted in place:
    // a head's length is a difference of the round's prefix sums; the last head of a round stays pending in scalar registers
    // because its run may go on in the next round.
    if (PHASE != 2 && nseq > 1) {
        const uint64_t lt = (1ull << lane) - 1;
        uint32_t out = 0;                       // heads so far, including the pending one
        uint32_t pend_lp = 0, pend_ml = 0, pend_o = 0, c_lp = 0, c_o = 0; // pending head; literal position / offset of the previous round's last sequence
        bool have_pend = false;
        uint64_t s_next = (uint32_t)lane < nseq ? seq[lane] : 0; // every pass over the sequences requests its next round before it works on this one
        for (uint32_t base = 0; base < nseq; base += 64) {
            const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
            const bool valid = (uint32_t)lane < cnt;
            const uint64_t s = valid ? s_next : 0;
            if (base + 64 + (uint32_t)lane < nseq) s_next = seq[base + 64 + (uint32_t)lane]; // (this round stores below base + 64 only)
            const uint32_t lp = zge_seq_ll(s), ml = valid ? zge_seq_ml(s) : 0u, o = zge_seq_ofv(s);
            uint32_t plp = zd::shfl_up1(lp), po = zd::shfl_up1(o);
            if (lane == 0) { plp = c_lp; po = c_o; }
            const bool cont = valid && (base + (uint32_t)lane) > 0 && lp == plp && o == po;
            const uint64_t hm = zd::ballot(valid && !cont);
            const uint32_t psum = zd::wave_scan_incl(ml); // inclusive prefix sums of the match lengths
            const uint32_t first = hm ? (uint32_t)zd::ctz64(hm) : cnt; // lanes before the first head continue the pending head
            if (first > 0) pend_ml += zd::readlane(psum, first - 1);
            if (hm) {
                if (have_pend && lane == 0) seq[out - 1] = zge_pack_seq(pend_lp, pend_ml, pend_o);
                const uint32_t nh = (uint32_t)__popcll(hm), last = 63u - (uint32_t)__clzll((long long)hm);
                // my run ends in front of the next head (or with the round)
                const uint64_t above = lane == 63 ? 0ull : (hm >> (lane + 1)) << (lane + 1);
                const uint32_t stop = above ? (uint32_t)zd::ctz64(above) : cnt;  // first lane that is not mine
                const uint32_t run = zd::shfl(psum, (int)stop - 1) - psum + ml;
                const uint32_t rank = (uint32_t)__popcll(hm & lt);
                if (((hm >> lane) & 1) && (uint32_t)lane != last) seq[out + rank] = zge_pack_seq(lp, run, o);
                pend_lp = zd::readlane(lp, last); pend_ml = zd::readlane(run, last); pend_o = zd::readlane(o, last);
                have_pend = true;
                out += nh;
            }
            c_lp = zd::readlane(lp, cnt - 1); c_o = zd::readlane(o, cnt - 1);
        }
        if (have_pend && lane == 0) seq[out - 1] = zge_pack_seq(pend_lp, pend_ml, pend_o);
        nseq = out;
        zd::wave_sync_global(); // the pass below reads what other lanes wrote here
    }
    if (PHASE != 2) {
        uint32_t r0 = 0, r1 = 0, r2 = 0, carry = 0; // wave-uniform history / literal position carried between rounds
        uint64_t s_next = (uint32_t)lane < nseq ? seq[lane] : 0;
        for (uint32_t base = 0; base < nseq; base += 64) {
            const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
            const bool valid = (uint32_t)lane < cnt;
            const uint64_t s = valid ? s_next : 0;
            if (base + 64 + (uint32_t)lane < nseq) s_next = seq[base + 64 + (uint32_t)lane];
            const uint32_t litpos = zge_seq_ll(s), ml = zge_seq_ml(s), o = zge_seq_ofv(s);
            uint32_t prev = zd::shfl_up1(litpos);
            if (lane == 0) prev = carry;
            const uint32_t ll = litpos - prev;
            carry = zd::readlane(litpos, cnt - 1);
            const bool z = ll == 0;
            uint32_t a = zd::shfl_up1(o);            // r0 before this sequence
            if (lane == 0) a = r0;
            const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1); // lanes <= mine
            // r1 after each sequence
            const bool keep1 = !valid || (!z && o == a);
            const uint64_t nk1 = zd::ballot(!keep1) & upto;
            const uint32_t src1 = zd::shfl(a, nk1 ? 63 - __clzll((long long)nk1) : 0);
            const uint32_t r1_after = nk1 ? src1 : r1;
            uint32_t bb = zd::shfl_up1(r1_after);   // r1 before this sequence
            if (lane == 0) bb = r1;
            // r2 after each sequence
            const bool keep2 = !valid || (z ? o == bb : (o == a || o == bb));
            const uint64_t nk2 = zd::ballot(!keep2) & upto;
            const uint32_t src2 = zd::shfl(bb, nk2 ? 63 - __clzll((long long)nk2) : 0);
            const uint32_t r2_after = nk2 ? src2 : r2;
            uint32_t c = zd::shfl_up1(r2_after);    // r2 before this sequence
            if (lane == 0) c = r2;
            uint32_t ofv;
            if (!z) ofv = o == a ? 1u : (o == bb ? 2u : (o == c ? 3u : o + 3));
            else ofv = o == bb ? 1u : (o == c ? 2u : ((a > 1 && o == a - 1) ? 3u : o + 3));
            if (valid) seq[base + (uint32_t)lane] = zge_pack_seq(ll, ml, ofv);
            r0 = zd::readlane(o, cnt - 1);
            r1 = zd::readlane(r1_after, cnt - 1);
            r2 = zd::readlane(r2_after, cnt - 1);
        }
        zd::wave_sync_global(); // the coding passes below read seq[] with a different lane mapping
    }
    ENT_PROF(3);

    // ================= sequences section =================
    uint32_t ssz = 0;
    if (!fail) {
        uint8_t *so = out + lsz;
        const uint32_t scap = out_cap - lsz;
        uint32_t pos = 0;
        if (lane == 0) {
            if (nseq < 128) so[0] = (uint8_t)nseq;
            else if (nseq < 0x7F00) { so[0] = (uint8_t)((nseq >> 8) + 128); so[1] = (uint8_t)nseq; }
            else { so[0] = 255; so[1] = (uint8_t)(nseq - 0x7F00); so[2] = (uint8_t)((nseq - 0x7F00) >> 8); }
        }
        pos = nseq < 128 ? 1u : (nseq < 0x7F00 ? 2u : 3u);
        if (nseq == 0) ssz = pos;
        else {
          if (PHASE != 2) {
            for (int i = lane; i < 36; i += 64) L.s.cl[i] = 0;
            for (int i = lane; i < 32; i += 64) L.s.co[i] = 0;
            for (int i = lane; i < 53; i += 64) L.s.cm[i] = 0;
            zd::wave_sync();
            for (uint32_t i0 = 0; i0 < nseq; i0 += 256) { // four loads in flight per lane
                uint64_t sv[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) { const uint32_t i = i0 + 64 * k + (uint32_t)lane; sv[k] = i < nseq ? seq[i] : 0; }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (i0 + 64 * k + (uint32_t)lane >= nseq) continue;
                    atomicAdd(&L.s.cl[ll_code(zge_seq_ll(sv[k]))], 1u);
                    atomicAdd(&L.s.cm[ml_code(zge_seq_ml(sv[k]))], 1u);
                    atomicAdd(&L.s.co[zd::hb32(zge_seq_ofv(sv[k]))], 1u);
                }
            }
            zd::wave_sync();
            if (lane < 3) {
                // the three tables are independent: lane 0 chooses and builds LL, lane 1 OF, lane 2 ML side by side (the wave issues
                // the longest of the three instead of their sum).  The cell scratch of each build lies in pre[], idle until the chains.
                const int which = lane;
                const uint32_t *cnt = which == 0 ? L.s.cl : (which == 1 ? L.s.co : L.s.cm);
                const int16_t *def = which == 0 ? E_LL_DEFAULT : (which == 1 ? E_OF_DEFAULT : E_ML_DEFAULT);
                choose_table(L, which, cnt, which == 0 ? 35 : (which == 1 ? 31 : 52), nseq, def, which == 0 ? 36 : (which == 1 ? 29 : 53),
                             which == 1 ? 5 : 6, which == 1 ? 8 : 9);
            }
            zd::wave_sync();
          }
          if (PHASE == 1) {
            // the block's plan record: histograms, own choices (zarc_zge_plan may replace them by the group's table), what pass 2 needs
#pragma unroll
            for (int which = 0; which < 3; which++) {
                ZgePlanTable *pt = &plan->t[which];
                const uint32_t *cnt = which == 0 ? L.s.cl : (which == 1 ? L.s.co : L.s.cm);
                const int nc = which == 0 ? 36 : (which == 1 ? 32 : 53);
                pt->count[lane] = lane < nc ? cnt[lane] : 0u;
                pt->norm[lane] = L.s.norm[which][lane];
                pt->desc[lane] = L.s.desc[which][lane];
                if (lane < 16) pt->desc[64 + lane] = L.s.desc[which][64 + lane];
                if (lane == 0) {
                    pt->cost = (uint32_t)L.ctrl[X_COST_L + which];
                    pt->mode = (uint8_t)L.ctrl[X_MODE_L + which]; pt->al = (uint8_t)L.ctrl[X_AL_L + which]; pt->nsym = (uint8_t)L.ctrl[X_NSYM_L + which];
                    pt->dl = (uint8_t)L.ctrl[X_DL_L + which]; pt->rle = (uint8_t)L.ctrl[X_RLE_L + which];
                }
            }
            {
                uint32_t xb = 0;
                if (lane < 36) xb += L.s.cl[lane] * ll_code_bits((uint32_t)lane);
                if (lane < 53) xb += L.s.cm[lane] * ml_code_bits((uint32_t)lane);
                if (lane < 32) xb += L.s.co[lane] * (uint32_t)lane;
                xb = zd::wave_sum(xb);
                if (lane == 0) { plan->active = 1; plan->nseq = nseq; plan->lsz = lsz; plan->extra_bits = xb; plan->guaranteed = 0; }
            }
            return; // pass 2 goes on from here
          }
          if (PHASE == 2) {
            // the tables the plan settled on: normalised counts, description, mode per type -> where pass 0 has them after choose_table
#pragma unroll
            for (int which = 0; which < 3; which++) {
                const ZgePlanTable *pt = &plan->t[which];
                L.s.norm[which][lane] = pt->norm[lane];
                L.s.desc[which][lane] = pt->desc[lane];
                if (lane < 16) L.s.desc[which][64 + lane] = pt->desc[64 + lane];
                if (lane == 0) {
                    L.ctrl[X_MODE_L + which] = pt->mode; L.ctrl[X_AL_L + which] = pt->al; L.ctrl[X_NSYM_L + which] = pt->nsym;
                    L.ctrl[X_DL_L + which] = pt->dl; L.ctrl[X_RLE_L + which] = pt->rle;
                }
            }
            zd::wave_sync();
          }
            // ... and the whole wave builds them, one after the other (round 3: the serial build on three lanes was a sixth of the stage)
#pragma unroll
            for (int which = 0; which < 3; which++) {
                FseCtab tt = which == 0 ? FseCtab{L.s.st + ST_LL, L.s.dnb_ll, L.s.dfs_ll, 0}
                                        : (which == 1 ? FseCtab{L.s.st + ST_OF, L.s.dnb_of, L.s.dfs_of, 0} : FseCtab{L.s.st + ST_ML, L.s.dnb_ml, L.s.dfs_ml, 0});
                uint8_t *cells = (uint8_t *)L.s.pre; // 512 bytes of scratch, idle until the chains
                if (L.ctrl[X_MODE_L + which] != 1) fse_build_ctab_wave(tt, L.s.norm[which], L.ctrl[X_NSYM_L + which], L.ctrl[X_AL_L + which], cells, lane);
#ifdef ZGE_DEBUG_FSE
                if (L.ctrl[X_MODE_L + which] != 1 && lane == 0) {
                    static thread_local uint16_t st2[512]; static thread_local int32_t dnb2[64], dfs2[64]; static thread_local uint8_t cs2[512];
                    FseCtab t2 = {st2, dnb2, dfs2, 0};
                    const int nsym = L.ctrl[X_NSYM_L + which], al = L.ctrl[X_AL_L + which];
                    fse_build_ctab(t2, L.s.norm[which], nsym, al, cs2);
                    bool bad = false;
                    for (int i = 0; i < (1 << al); i++) bad |= st2[i] != tt.state_tab[i];
                    for (int i = 0; i < nsym; i++) bad |= dnb2[i] != tt.dnb[i] || dfs2[i] != tt.dfs[i];
                    if (bad) { for (int i = 0; i < (1 << al); i++) if (st2[i] != tt.state_tab[i]) { printf("st[%d] serial %d wave %d; ", i, st2[i], tt.state_tab[i]); break; }
                        for (int i = 0; i < nsym; i++) if (dnb2[i] != tt.dnb[i] || dfs2[i] != tt.dfs[i]) { printf("sym %d dnb %d/%d dfs %d/%d; ", i, dnb2[i], tt.dnb[i], dfs2[i], tt.dfs[i]); break; }
                        printf("FSE mismatch which %d al %d nsym %d mode %d:", which, al, nsym, L.ctrl[X_MODE_L + which]); for (int i = 0; i < nsym; i++) printf(" %d", L.s.norm[which][i]); printf("\n"); }
                }
#endif
            }
            zd::wave_sync();
            ENT_PROF(4);
            const int mode_l = L.ctrl[X_MODE_L], mode_o = L.ctrl[X_MODE_O], mode_m = L.ctrl[X_MODE_M];
            const uint32_t dl_l = (uint32_t)L.ctrl[X_DL_L], dl_o = (uint32_t)L.ctrl[X_DL_O], dl_m = (uint32_t)L.ctrl[X_DL_M];
            const int al_l = L.ctrl[X_AL_L], al_o = L.ctrl[X_AL_O], al_m = L.ctrl[X_AL_M];
            if (pos + 4 + dl_l + dl_o + dl_m > scap) fail = true;
            if (!fail) {
                if (lane == 0) {
                    so[pos] = (uint8_t)((mode_l << 6) | (mode_o << 4) | (mode_m << 2));
                    uint32_t q = pos + 1;
                    if (mode_l == 1) so[q++] = (uint8_t)L.ctrl[X_RLE_L]; else for (uint32_t i = 0; i < dl_l; i++) so[q++] = L.s.desc[0][i];
                    if (mode_o == 1) so[q++] = (uint8_t)L.ctrl[X_RLE_O]; else for (uint32_t i = 0; i < dl_o; i++) so[q++] = L.s.desc[1][i];
                    if (mode_m == 1) so[q++] = (uint8_t)L.ctrl[X_RLE_M]; else for (uint32_t i = 0; i < dl_m; i++) so[q++] = L.s.desc[2][i];
                }
                pos += 1 + (mode_l == 1 ? 1u : dl_l) + (mode_o == 1 ? 1u : dl_o) + (mode_m == 1 ? 1u : dl_m);
                // FSE state chains: lane 0 = LL, lane 1 = OF, lane 2 = ML; 64 sequences per round, last -> first
                const int my_mode = lane == 0 ? mode_l : (lane == 1 ? mode_o : (lane == 2 ? mode_m : 1));
                uint32_t state = 0;
                WavePacker pk;
                pk.begin(L.stage, so + pos, scap - pos, lane);
                uint64_t s_next = (uint32_t)lane < nseq ? seq[nseq - 1 - (uint32_t)lane] : 0;
                for (uint32_t done = 0; done < nseq; done += 64) {
                    const uint32_t cnt = nseq - done < 64 ? nseq - done : 64;
                    // element e of this round is sequence index (nseq-1-done-e); lane e classifies it (requested a round ahead)
                    uint32_t ll = 0, ml = 3, ofv = 1, llc = 0, mlc = 0, ofc = 0;
                    const uint64_t s = s_next;
                    if (done + 64 + (uint32_t)lane < nseq) s_next = seq[nseq - 1 - done - 64 - (uint32_t)lane];
                    if ((uint32_t)lane < cnt) {
                        ll = zge_seq_ll(s); ml = zge_seq_ml(s); ofv = zge_seq_ofv(s);
                        llc = ll_code(ll); mlc = ml_code(ml); ofc = (uint32_t)zd::hb32(ofv);
                        // per-symbol transition constants fetched by all lanes at once: the serial loop below then has a
                        // single dependent LDS access per step (the state table)
                        L.s.pre[0][lane] = (uint32_t)L.s.dnb_ll[llc] | ((uint64_t)(uint32_t)(2 * (L.s.dfs_ll[llc] + ST_LL)) << 32); // byte offset into st[]
                        L.s.pre[1][lane] = (uint32_t)L.s.dnb_of[ofc] | ((uint64_t)(uint32_t)(2 * (L.s.dfs_of[ofc] + ST_OF)) << 32);
                        L.s.pre[2][lane] = (uint32_t)L.s.dnb_ml[mlc] | ((uint64_t)(uint32_t)(2 * (L.s.dfs_ml[mlc] + ST_ML)) << 32);
                    }
                    zd::wave_sync();
                    if (lane < 3 && my_mode != 1) {
                        // The chain does the bare minimum per step -- remember the state, find the next one (one dependent LDS access);
                        // the bits each step emits follow from (state before, symbol constant) and are worked out by all lanes below.
                        // Four steps per trip: their symbol constants are fetched together and the loop bookkeeping is paid once.
                        uint64_t *const pp = L.s.pre[lane];
                        // the state each step starts from goes to sbp[] (the normalised counts are dead once the tables are built): four
                        // steps' states in ONE 64-bit store -- the stage is bound by LDS instruction issue, and a store per step was a
                        // third of the chain's LDS instructions
                        typedef uint16_t __attribute__((may_alias)) u16a;
                        typedef uint64_t __attribute__((may_alias)) u64a;
                        u16a *const sbp = (u16a *)&L.s.norm[0][0] + 64 * lane;
                        uint32_t e = 0;
                        if (done == 0) { // first symbol coded: the state that needs no bits (fse_init_state)
                            const uint64_t c0 = pp[0];
                            const int32_t d = (int32_t)(uint32_t)c0, f = (int32_t)(uint32_t)(c0 >> 32);
                            const int nb0 = (d + (1 << 15)) >> 16;
                            const int value = (nb0 << 16) - d;
                            state = *(const uint16_t *)((const uint8_t *)L.s.st + (2 * (value >> nb0) + f));
                            e = 1;
                        }
#define CHAIN_NEXT(cur) do { const uint32_t nb_ = (state + (uint32_t)(cur)) >> 16;                                     \
                             state = *(const uint16_t *)((const uint8_t *)L.s.st + ((int)((state >> nb_) << 1) + (int32_t)(uint32_t)((cur) >> 32))); } while (0)
                        for (; (e & 3u) && e < cnt; e++) { const uint64_t c0 = pp[e]; sbp[e] = (uint16_t)state; CHAIN_NEXT(c0); }
                        for (; e + 4 <= cnt; e += 4) {
                            const uint64_t c0 = pp[e], c1 = pp[e + 1], c2 = pp[e + 2], c3 = pp[e + 3];
                            uint64_t four = state;
                            CHAIN_NEXT(c0); four |= (uint64_t)state << 16;
                            CHAIN_NEXT(c1); four |= (uint64_t)state << 32;
                            CHAIN_NEXT(c2); four |= (uint64_t)state << 48;
                            CHAIN_NEXT(c3);
                            *(u64a *)(sbp + e) = four; // states are below 2^11 (table size 2^9 at most, states T .. 2T - 1)
                        }
                        for (; e < cnt; e++) { const uint64_t c0 = pp[e]; sbp[e] = (uint16_t)state; CHAIN_NEXT(c0); }
#undef CHAIN_NEXT
                    }
                    zd::wave_sync();
                    uint64_t lo = 0;
                    uint32_t hi = 0, nb = 0;
                    if ((uint32_t)lane < cnt) {
                        // state bits of the three chains: value | nb << 16 from (state before the step, dnb of the symbol); the very first
                        // symbol coded and tables in RLE mode emit nothing
                        uint32_t bo = 0, bm = 0, bl = 0;
                        const bool first = done + (uint32_t)lane == 0;
                        if (!first) {
                            if (mode_o != 1) { const uint32_t c = (uint32_t)L.s.pre[1][lane], sb = ((const uint16_t __attribute__((may_alias)) *)&L.s.norm[0][0])[1 * 64 + lane], n_ = (sb + c) >> 16; bo = (sb & ((1u << n_) - 1)) | (n_ << 16); }
                            if (mode_m != 1) { const uint32_t c = (uint32_t)L.s.pre[2][lane], sb = ((const uint16_t __attribute__((may_alias)) *)&L.s.norm[0][0])[2 * 64 + lane], n_ = (sb + c) >> 16; bm = (sb & ((1u << n_) - 1)) | (n_ << 16); }
                            if (mode_l != 1) { const uint32_t c = (uint32_t)L.s.pre[0][lane], sb = ((const uint16_t __attribute__((may_alias)) *)&L.s.norm[0][0])[0 * 64 + lane], n_ = (sb + c) >> 16; bl = (sb & ((1u << n_) - 1)) | (n_ << 16); }
                        }
                        // order: OF state bits, ML state bits, LL state bits, LL extra, ML extra, OF extra
                        uint64_t acc = bo & 0xFFFF; uint32_t sh = bo >> 16;
                        acc |= (uint64_t)(bm & 0xFFFF) << sh; sh += bm >> 16;
                        acc |= (uint64_t)(bl & 0xFFFF) << sh; sh += bl >> 16;                     // <= 27 bits
                        uint32_t xb;
                        acc |= (uint64_t)ll_extra(ll, llc, &xb) << sh; sh += xb;                  // <= 43
                        acc |= (uint64_t)ml_extra(ml, mlc, &xb) << sh; sh += xb;                  // <= 59
                        const uint64_t ofx = (uint64_t)(ofv - (1u << ofc));
                        lo = acc | (ofx << sh);
                        hi = sh == 0 ? 0u : (uint32_t)(ofx >> (64 - sh));
                        nb = sh + ofc;
                    }
                    pk.put(lo, hi, nb, lane);
                    zd::wave_sync();
                }
                // flush ML, OF, LL states (in that order)
                {
                    const uint32_t st_l = zd::shfl(state, 0), st_o = zd::shfl(state, 1), st_m = zd::shfl(state, 2);
                    uint64_t lo = 0; uint32_t nb = 0;
                    if (lane == 0) {
                        if (mode_m != 1) { lo |= (uint64_t)(st_m & ((1u << al_m) - 1)) << nb; nb += (uint32_t)al_m; }
                        if (mode_o != 1) { lo |= (uint64_t)(st_o & ((1u << al_o) - 1)) << nb; nb += (uint32_t)al_o; }
                        if (mode_l != 1) { lo |= (uint64_t)(st_l & ((1u << al_l) - 1)) << nb; nb += (uint32_t)al_l; }
                    }
                    pk.put(lo, 0, nb, lane);
                }
                const uint32_t bytes = pk.finish(lane);
                if (pk.overflow || pos + bytes > scap) fail = true;
                ssz = pos + bytes;
            }
        }
    }
    ENT_PROF(5);
    const uint32_t csz = fail ? 0u : lsz + ssz;
    if (lane == 0) {
        if (csz && csz < src_len) { rec->type = 2; rec->out_len = csz; }
        else { rec->type = 0; rec->out_len = src_len; }
        if (PHASE == 1) plan->active = 0; // no sequences to code (or the literals failed): the block record is final
    }
}

} // namespace

__global__ void __launch_bounds__(64, 5) zarc_zge_entropy(uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *__restrict__ blocks, uint64_t *__restrict__ seq_scratch,
                                                       const uint8_t *__restrict__ lit_scratch, uint8_t *__restrict__ out_scratch,
                                                       unsigned long long *__restrict__ prof)
{
    __shared__ EntLds L;
    zge_entropy_body<0>(L, n_blocks, slot_bytes, blocks, seq_scratch, lit_scratch, out_scratch, prof, nullptr);
}
__global__ void __launch_bounds__(64, 5) zarc_zge_entropy_p1(uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *__restrict__ blocks, uint64_t *__restrict__ seq_scratch,
                                                          const uint8_t *__restrict__ lit_scratch, uint8_t *__restrict__ out_scratch,
                                                          unsigned long long *__restrict__ prof, ZgePlan *__restrict__ plans)
{
    __shared__ EntLds L;
    zge_entropy_body<1>(L, n_blocks, slot_bytes, blocks, seq_scratch, lit_scratch, out_scratch, prof, plans);
}
__global__ void __launch_bounds__(64, 5) zarc_zge_entropy_p2(uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *__restrict__ blocks, uint64_t *__restrict__ seq_scratch,
                                                          const uint8_t *__restrict__ lit_scratch, uint8_t *__restrict__ out_scratch,
                                                          unsigned long long *__restrict__ prof, ZgePlan *__restrict__ plans)
{
    __shared__ EntLds L;
    zge_entropy_body<2>(L, n_blocks, slot_bytes, blocks, seq_scratch, lit_scratch, out_scratch, prof, plans);
}

// The table plan (model: zstd_enc_model.c, seq_plan_group).  One wave per group of ZGE_TABLE_GROUP blocks of a frame (the host lists
// the groups' first block slots), one lane per symbol.  Per table type: the sum of the blocks' histograms is
// normalised like a block's own table would be; if coding every block of the group with that one table costs at most 1/64 more bits
// than the blocks' own choices, the first block describes it and the others say Repeat_Mode -- as long as the chain holds: a block
// hands the table on only if an upper bound of its coded size (every state transition at its symbol's larger bit count) is below its
// raw size, i.e. it cannot end up a raw block, whose tables the decoder never sees; the block behind any other describes the table
// again.  Blocks that keep RLE mode for a type break that type's chain the same way.  Costs fit 32 bits: a group holds at most
// 16 x 21 845 sequences at 9 x 256 units each.
__global__ void __launch_bounds__(64) zarc_zge_plan(uint32_t n_blocks, const ZgeBlock *__restrict__ blocks, ZgePlan *__restrict__ plans,
                                                    const uint32_t *__restrict__ group_start /* first block slot of each group of a multi-block frame */)
{
    __shared__ int16_t gnorm[3][64];
    __shared__ uint8_t gdesc[3][80];
    __shared__ uint32_t gdl[3];
    const int lane = zd::lane_id();
    const uint32_t bi = group_start[blockIdx.x];
    if (bi >= n_blocks) return;
    const uint32_t frame = blocks[bi].frame, index0 = blocks[bi].index;
    if (index0 % ZGE_TABLE_GROUP) return;
    uint32_t nb = 1;
    while (nb < ZGE_TABLE_GROUP && bi + nb < n_blocks && blocks[bi + nb].frame == frame && blocks[bi + nb].index == index0 + nb) nb++;
    if (nb < 2) return; // a block on its own keeps its own choices
    bool use_group[3] = {false, false, false};
    uint32_t g_al[3] = {0, 0, 0}, g_nsym[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int max_al = t == 1 ? 8 : 9;
        uint32_t sum = 0, total = 0, cost_own = 0, np = 0;
        for (uint32_t b = 0; b < nb; b++) {
            const ZgePlan *p = plans + bi + b;
            if (!p->active || p->t[t].mode == 1) continue; // uniform
            sum += p->t[t].count[lane];
            total += p->nseq; cost_own += p->t[t].cost; np++;
        }
        if (np < 2) continue;
        const uint64_t present = zd::ballot(sum != 0);
        const int distinct = (int)__popcll(present), last = 63 - (int)__clzll((long long)present);
        int al = zd::hb32(total > 1 ? total - 1 : 1) - 2;
        if (al > max_al) al = max_al;
        if (al < 5) al = 5;
        while ((1 << al) < distinct) al++;
        const uint32_t T = 1u << al;
        // fse_normalize, a symbol per lane: round to nearest (at least 1), then the surplus / deficit goes to the largest entry (lowest
        // symbol on ties), one wave maximum per step
        uint32_t v = 0;
        if (sum) { v = (sum * T + total / 2) / total; if (v < 1) v = 1; } // sum * T < 2^28
        uint32_t vs = zd::uniform(zd::wave_sum(v));
        while (vs != T) {
            const uint32_t key = v ? (v << 6) | (uint32_t)(63 - lane) : 0u;
            const uint32_t best = 63u - (zd::uniform(zd::wave_max(key)) & 63u);
            const uint32_t bv = zd::readlane(v, best);
            if (vs > T) {
                uint32_t take = vs - T;
                const uint32_t room = bv - 1;
                if (take > room) take = room;
                if (take == 0) break;
                if ((uint32_t)lane == best) v -= take;
                vs -= take;
            } else {
                if ((uint32_t)lane == best) v += T - vs;
                vs = T;
            }
        }
        gnorm[t][lane] = (int16_t)v; // zero beyond the last symbol
        zd::wave_sync();
        if (lane == 0) gdl[t] = fse_write_desc(gdesc[t], 80, gnorm[t], last + 1, al);
        zd::wave_sync();
        const uint32_t dl = gdl[t];
        if (!dl) continue;
        const uint32_t cost_group = zd::uniform(zd::wave_sum(sum ? sum * ((uint32_t)al * 256 - log2_fp8(v)) : 0u)) + dl * 8 * 256;
        use_group[t] = cost_group <= cost_own + (cost_own >> 6);
        g_al[t] = (uint32_t)al; g_nsym[t] = (uint32_t)(last + 1);
    }
    if (!use_group[0] && !use_group[1] && !use_group[2]) return; // every block keeps its own choices: nothing to write
    bool have[3] = {false, false, false}; // the decoder is known to hold the group's table of this type
    for (uint32_t b = 0; b < nb; b++) {
        ZgePlan *p = plans + bi + b;
        if (!p->active) continue; // uniform
        uint32_t ub_bits = 1 + p->extra_bits, hdrs = 0;
        bool grp[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            ZgePlanTable *pt = &p->t[t];
            uint32_t mode = pt->mode, al = pt->al, dl = pt->dl;
            int n_l = pt->norm[lane];
            grp[t] = mode != 1 && use_group[t];
            if (grp[t]) {
                mode = have[t] ? 3u : 2u; al = g_al[t]; dl = mode == 2 ? gdl[t] : 0u; n_l = gnorm[t][lane];
                pt->norm[lane] = (int16_t)n_l;
                if (mode == 2) { pt->desc[lane] = gdesc[t][lane]; if (lane < 16) pt->desc[64 + lane] = gdesc[t][64 + lane]; }
                if (lane == 0) { pt->mode = (uint8_t)mode; pt->al = (uint8_t)al; pt->nsym = (uint8_t)g_nsym[t]; pt->dl = (uint8_t)dl; }
            }
            if (mode != 1) {
                const uint32_t cnt = pt->count[lane];
                const uint32_t mb = (n_l == -1 || n_l == 1) ? al : (n_l > 1 ? al - (uint32_t)zd::hb32((uint32_t)(n_l - 1)) : 0u);
                ub_bits += zd::uniform(zd::wave_sum(cnt * mb)) + al;
            }
            hdrs += mode == 1 ? 1u : dl;
        }
        const uint32_t nseq = p->nseq;
        const uint32_t ub = p->lsz + (nseq < 128 ? 1u : (nseq < 0x7F00 ? 2u : 3u)) + 1 + (ub_bits + 7) / 8 + hdrs;
        const bool guaranteed = ub < blocks[bi + b].src_len;
        if (lane == 0) p->guaranteed = guaranteed ? 1u : 0u;
#pragma unroll
        for (int t = 0; t < 3; t++) have[t] = grp[t] && guaranteed;
    }
}
