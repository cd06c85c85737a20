// zarc_amd/csrc/zge_match.hip -- encoder stage 1: LZ77 match finding for gfx950.
//
// Part of the replacement for `CCtx::compress2` at crates/zarc/src/encode/lowlevel_frames.rs:29-31
// (called per entry from Encoder::add_data_frame, crates/zarc/src/encode/content_frame.rs:41).
//
// One 1024-thread workgroup per frame, one frame per CU at a time: the two position hash tables (8-byte
// "long" hash and 5-byte "short" hash, 2^14 u32 entries each = 128 KiB) stay in LDS for the whole frame,
// so matches reach back across all earlier blocks of the frame (window = frame, capped at 2^window_log).
// A block (<= 128 KiB) is swept in tiles of 1024 positions, one position per thread:
//   S0/S1 stage the tile's bytes in LDS (coalesced dword loads), hash every position (64-bit multiplies)
//   S2    ordered lookup + insert by wave 0, 64 positions per step: LDS executes one wave's instructions in
//         order, so a position sees every insert of earlier 64-groups with no waiting between steps
//   S3    every position scores its candidates {long, short, 2 recent offsets}: common prefix (LDS for the
//         tile side, L2/HBM for far sources), backward-extension potential, bit-cost model
//   S4    backward propagation: position t may start the match found at t+k, k bytes earlier
//   S5    one-byte lazy rule -> take flag and successor next[t] for every position
//   S6    the greedy parse IS the path from the entry cursor through next[]: pointer doubling over the tile
//         (10 rounds of jump[t] = jump[jump[t]] with monotone marking) instead of a serial walk
//   S7    ballot/popcount prefix sums place literal bytes and sequences; nothing is serial per sequence
// Output per block: packed (literal position, match length, offset) + literal bytes in HBM scratch; literal
// lengths and repcodes are resolved by the entropy stage.  Deterministic and bit-identical to
// oracle/zstd_enc_model.c (tests/ compare them).
#include "zarc_device.h"
#include "zarc_kernels.h"

namespace {

constexpr int TILE = 1024;
constexpr int THREADS = 1024;
constexpr int TAB_LOG_MAX = 14;
constexpr int CAP_MAX = 256;
constexpr int TB_BYTES = 12 + TILE + CAP_MAX + 24; // 8 bytes before the tile, compare overrun after it

struct MatchLds {
    uint32_t tl[1 << TAB_LOG_MAX];
    uint32_t ts[1 << TAB_LOG_MAX];
    uint32_t a0[TILE], a1[TILE]; // S1: hashes -> S2: candidates (pos+1) -> S3: own match {offset ; len | back<<16 | rep<<24}
    uint32_t b0[TILE], b1[TILE]; // S4: match after backward propagation {offset ; len | rep<<24}
    uint16_t ja[TILE + 2], jb[TILE + 2];
    uint8_t mark[TILE];
    uint32_t tb[(TB_BYTES + 3) / 4];
    uint32_t wsel[16], wlit[16];
    uint32_t ctrl[16];
};
enum { K_POS = 0, K_REP0 = 1, K_REP1 = 2, K_FLAG = 3, K_NEW0 = 4, K_NEW1 = 5 };

__device__ __forceinline__ uint32_t hash_long(uint64_t v, int bits) { return (uint32_t)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - bits)); }
__device__ __forceinline__ uint32_t hash_short(uint64_t v, int bits, int nbytes)
{
    return (uint32_t)(((v << (64 - 8 * nbytes)) * 0x9E3779B185EBCA87ULL) >> (64 - bits));
}

__device__ __forceinline__ int32_t score_of(const ZgeParams &P, uint32_t len, uint32_t off, bool is_rep)
{
    if (is_rep) return (int32_t)(P.lit_cost * (int)len) - P.rep_cost;
    return (int32_t)(P.lit_cost * (int)len) - P.match_cost - zd::hb32(off);
}

// 8 bytes at frame position `pos`: from the staged window [lo, hi) in LDS when fully inside, else from HBM/L2
struct Win {
    const uint8_t *src;  // frame start in global memory
    const uint8_t *lds;  // LDS byte that corresponds to frame position `lo`
    uint64_t lo, hi;
    __device__ __forceinline__ uint64_t ld8(uint64_t pos) const
    {
        if (pos >= lo && pos + 8 <= hi) return zd::load_u64(lds + (pos - lo));
        return zd::load_u64(src + pos);
    }
};

__device__ __forceinline__ uint32_t match_len(const Win &w, uint64_t p, uint64_t q, uint32_t limit)
{
    uint32_t n = 0;
    while (n + 8 <= limit) {
        const uint64_t x = w.ld8(p + n) ^ w.ld8(q + n);
        if (x) return n + (uint32_t)(zd::ctz64(x) >> 3);
        n += 8;
    }
    if (n < limit) {
        const uint64_t x = w.ld8(p + n) ^ w.ld8(q + n);
        uint32_t m = x ? (uint32_t)(zd::ctz64(x) >> 3) : 8u;
        if (m > limit - n) m = limit - n;
        n += m;
    }
    return n;
}

} // namespace

__global__ void __launch_bounds__(1024) zarc_zge_match(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                       const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, uint32_t n_frames,
                                                       const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                                       uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch)
{
    __shared__ MatchLds L;
    const int tid = (int)threadIdx.x, lane = zd::lane_id(), wave = zd::wave_id();
    const uint32_t f = order[blockIdx.x];
    const uint8_t *src = src_base + src_off[f];
    const uint64_t n = src_len[f];
    const uint64_t window = n <= (1ull << P.window_log) ? (n ? n : 1) : (1ull << P.window_log);
    const uint64_t hash_end = n >= 8 ? n - 7 : 0;
    // block records / scratch slots are numbered within the sub-batch: block_prefix is indexed by blockIdx
    const uint64_t first_block = block_prefix[blockIdx.x];
    const uint32_t nblocks = (uint32_t)(block_prefix[blockIdx.x + 1] - first_block);
    const uint32_t cap_max = (uint32_t)(P.cap < CAP_MAX ? P.cap : CAP_MAX);

    for (int i = tid; i < (1 << TAB_LOG_MAX); i += THREADS) { L.tl[i] = 0; L.ts[i] = 0; }
    __syncthreads();

    for (uint32_t b = 0; b < nblocks; b++) {
        const uint64_t bs = (uint64_t)b * ZARC_BLOCK;
        const uint64_t be = bs + ZARC_BLOCK < n ? bs + ZARC_BLOCK : n;
        const uint32_t blen = (uint32_t)(be - bs);
        ZgeBlock *rec = blocks + first_block + b;
        uint64_t *seq_out = seq_scratch + (first_block + b) * (uint64_t)ZARC_MAX_SEQ;
        uint8_t *lit_out = lit_scratch + (first_block + b) * (uint64_t)(ZARC_BLOCK + 64);

        // ---- RLE block detection: every byte equals the first one (8 bytes per load; blocks start 16-byte aligned) ----
        if (tid == 0) { L.ctrl[K_FLAG] = 0; L.ctrl[K_POS] = 0; L.ctrl[K_REP0] = 0; L.ctrl[K_REP1] = 0; }
        __syncthreads();
        {
            bool diff = false;
            const uint8_t first = blen ? src[bs] : 0;
            const uint64_t pat = 0x0101010101010101ull * first;
            const uint64_t *w = (const uint64_t *)(src + bs);
            const uint32_t nw = blen / 8;
            for (uint32_t i = (uint32_t)tid; i < nw; i += THREADS) diff |= w[i] != pat;
            for (uint32_t i = nw * 8 + (uint32_t)tid; i < blen; i += THREADS) diff |= src[bs + i] != first;
            if (diff) L.ctrl[K_FLAG] = 1; // benign race: every writer stores 1
        }
        __syncthreads();
        const bool all_same = L.ctrl[K_FLAG] == 0;
        if (tid == 0) {
            rec->frame = f; rec->index = b; rec->src_len = blen; rec->nseq = 0; rec->nlit = 0;
            rec->type = (all_same && blen >= 2) ? 1u : 2u; rec->out_len = 0; rec->pad = 0;
        }
        if (all_same && blen >= 2) { __syncthreads(); continue; } // nothing is inserted for RLE blocks (same rule as the model)

        uint32_t nseq = 0, lp = 0; // replicated in every thread

        for (uint64_t tile = bs; tile < be; tile += TILE) {
            const uint64_t tend = tile + TILE < be ? tile + TILE : be;
            const uint32_t tcount = (uint32_t)(tend - tile);
            __syncthreads(); // K_POS / K_REP* of the previous tile are final; LDS work arrays are free again
            const uint64_t pos = bs + L.ctrl[K_POS];
            if (pos >= tend) continue; // whole tile already covered by a match: skip it (nothing is inserted)
            const uint32_t erep0 = L.ctrl[K_REP0], erep1 = L.ctrl[K_REP1];

            // ---- S0: tile bytes (8 before .. cap+16 after) -> LDS ----
            Win W;
            {
                const uint64_t lo = tile >= 8 ? tile - 8 : 0;
                uint64_t hi = tend + cap_max + 16;
                if (hi > n + 16) hi = n + 16;             // the arena is padded by ZARC_GPU_PAD
                const uintptr_t a = (uintptr_t)(src + lo);
                const uint32_t mis = (uint32_t)(a & 3);
                const uint32_t *w = (const uint32_t *)(a - mis);
                const int ndw = (int)((hi - lo + mis + 3) / 4);
                for (int i = tid; i < ndw; i += THREADS) L.tb[i] = w[i];
                W.src = src; W.lds = (const uint8_t *)L.tb + mis; W.lo = lo; W.hi = hi;
            }
            __syncthreads();
            // ---- S1: hashes ----
            const uint64_t p = tile + (uint64_t)tid;
            const bool in_tile = (uint32_t)tid < tcount;
            {
                uint32_t h = 0xFFFFFFFFu;
                if (in_tile && p < hash_end) {
                    const uint64_t v = W.ld8(p);
                    h = hash_long(v, P.long_log) | (hash_short(v, P.short_log, P.short_bytes) << 16);
                }
                L.a0[tid] = h;
            }
            __syncthreads();
            // ---- S2: ordered lookup + insert (wave 0); no waits between the 16 steps on hardware ----
            if (wave == 0) {
                uint32_t cl[TILE / 64], cs[TILE / 64];
#pragma unroll
                for (int k = 0; k < TILE / 64; k++) {
                    const uint32_t h = L.a0[k * 64 + lane];
                    const bool act = h != 0xFFFFFFFFu;
                    const uint32_t hl = h & 0xFFFF, hs = (h >> 16) & 0xFFFF;
                    cl[k] = act ? L.tl[hl] : 0u;
                    cs[k] = act ? L.ts[hs] : 0u;
                    zd::wave_lds_order(); // lookups of this 64-group precede its inserts
                    if (act) {
                        const uint32_t v = (uint32_t)(tile + (uint32_t)(k * 64 + lane)) + 1;
                        atomicMax(&L.tl[hl], v);
                        atomicMax(&L.ts[hs], v);
                    }
                    zd::wave_lds_order(); // inserts precede the next group's lookups
                }
#pragma unroll
                for (int k = 0; k < TILE / 64; k++) { L.a0[k * 64 + lane] = cl[k]; L.a1[k * 64 + lane] = cs[k]; }
            }
            __syncthreads();
            // ---- S3: own candidates ----
            {
                uint32_t o = 0, w = 0;
                if (in_tile && !(P.dbg & 1)) {
                    const uint32_t c0 = L.a0[tid], c1 = L.a1[tid];
                    const uint32_t limit = (uint32_t)(be - p), cap = limit < cap_max ? limit : cap_max;
                    uint32_t offs[4];
                    offs[0] = c0 ? (uint32_t)p - (c0 - 1) : 0u;
                    offs[1] = (c1 && c1 != c0) ? (uint32_t)p - (c1 - 1) : 0u;
                    if (P.short_window_log < 32 && offs[1] > (1u << P.short_window_log)) offs[1] = 0;
                    offs[2] = P.rep_search > 0 ? erep0 : 0u;
                    offs[3] = (P.rep_search > 1 && erep1 != erep0) ? erep1 : 0u;
                    uint32_t best_len = 0, best_off = 0;
                    bool best_rep = false;
                    int32_t best_score = -1000000;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t off = offs[k];
                        if (off == 0 || off > p || off > window) continue;
                        const bool is_rep = off == erep0 || off == erep1;
                        const uint32_t len = match_len(W, p, p - off, cap);
                        if (len < (uint32_t)(is_rep ? P.min_rep : P.min_match)) continue;
                        const int32_t sc = score_of(P, len, off, is_rep);
                        if (sc > best_score) { best_score = sc; best_len = len; best_off = off; best_rep = is_rep; }
                    }
                    if (best_len && best_score > 0) {
                        // backward-extension potential: equal bytes just before the match and its source
                        const uint64_t q = p - best_off;
                        uint32_t maxb = (uint32_t)P.back_cap;
                        if (p - bs < maxb) maxb = (uint32_t)(p - bs);
                        if (q < maxb) maxb = (uint32_t)q;
                        uint32_t back = 0;
                        if (maxb) {
                            if (q >= 8) { // then p >= 8 too
                                const uint64_t x = W.ld8(p - 8) ^ W.ld8(q - 8);
                                back = x ? (uint32_t)(__clzll((long long)x) >> 3) : 8u;
                                if (back > maxb) back = maxb;
                            } else {
                                while (back < maxb && src[p - back - 1] == src[q - back - 1]) back++;
                            }
                        }
                        o = best_off;
                        w = best_len | (back << 16) | ((best_rep ? 1u : 0u) << 24);
                    }
                }
                __syncthreads(); // every thread has read its candidates from a0/a1
                L.a0[tid] = o;
                L.a1[tid] = w;
            }
            __syncthreads();
            // ---- S4: backward propagation ----
            {
                uint32_t bo = 0, bw = 0;
                if (in_tile) {
                    const uint32_t mo = L.a0[tid], mw = L.a1[tid];
                    uint32_t blen_ = mw & 0xFFFF, boff = mo;
                    bool brep = (mw >> 24) & 1;
                    int32_t bscore = blen_ ? score_of(P, blen_, boff, brep) : 0;
                    for (uint32_t k = 1; k <= (uint32_t)P.back_cap && (uint32_t)tid + k < tcount; k++) {
                        const uint32_t nw = L.a1[tid + k];
                        const uint32_t nl = nw & 0xFFFF, nbk = (nw >> 16) & 0xFF;
                        if (!nl || nbk < k) continue;
                        const uint32_t no = L.a0[tid + k];
                        const bool nr = (nw >> 24) & 1;
                        const int32_t sc = score_of(P, nl + k, no, nr);
                        if (sc > bscore) { bscore = sc; blen_ = nl + k; boff = no; brep = nr; }
                    }
                    bo = boff;
                    bw = blen_ | ((brep ? 1u : 0u) << 24);
                }
                L.b0[tid] = bo;
                L.b1[tid] = bw;
            }
            __syncthreads();
            // ---- S5: take flag (one-byte lazy lookahead inside the tile) and successor ----
            const uint32_t my_off = L.b0[tid], my_w = L.b1[tid];
            const uint32_t my_len = my_w & 0xFFFF;
            bool take = in_tile && my_len != 0;
            if (take && P.lazy && (uint32_t)tid + 1 < tcount) {
                const uint32_t w2 = L.b1[tid + 1];
                const uint32_t l2 = w2 & 0xFFFF;
                if (l2 && score_of(P, l2, L.b0[tid + 1], (w2 >> 24) & 1) > score_of(P, my_len, my_off, (my_w >> 24) & 1) + P.lazy_delta) take = false;
            }
            const uint32_t nx = take ? (uint32_t)tid + my_len : (uint32_t)tid + 1; // true successor (may leave the tile)
            const uint32_t entry = (uint32_t)((pos > tile ? pos : tile) - tile);
            L.ja[tid] = (uint16_t)(nx < tcount ? nx : tcount);
            L.mark[tid] = (uint32_t)tid == entry ? 1 : 0;
            if (tid == 0) { L.ja[tcount] = (uint16_t)tcount; L.jb[tcount] = (uint16_t)tcount; }
            __syncthreads();
            // ---- S6: parse path by pointer doubling (marking is monotone and only ever marks path nodes) ----
            if (!(P.dbg & 2)) {
#pragma unroll 1
                for (int r = 0; r < 10; r++) {
                    const uint16_t *A = (r & 1) ? L.jb : L.ja;
                    uint16_t *B = (r & 1) ? L.ja : L.jb;
                    if (in_tile) {
                        const uint32_t j = A[tid];
                        if (L.mark[tid] && j < tcount) L.mark[j] = 1;
                        B[tid] = A[j];
                    }
                    __syncthreads();
                }
            }
            // ---- S7: emission ----
            const bool marked = in_tile && L.mark[tid] != 0;
            const bool sel = marked && take, islit = marked && !take;
            const uint64_t msel = zd::ballot(sel), mlit = zd::ballot(islit);
            if (lane == 0) { L.wsel[wave] = (uint32_t)__popcll(msel); L.wlit[wave] = (uint32_t)__popcll(mlit); }
            if (marked && nx >= tcount) L.ctrl[K_POS] = (uint32_t)(tile - bs) + nx; // the last path node: unique writer
            __syncthreads();
            uint32_t sel_before = 0, lit_before = 0, sel_total = 0, lit_total = 0;
#pragma unroll
            for (int wv = 0; wv < THREADS / 64; wv++) {
                const uint32_t s_ = L.wsel[wv], l_ = L.wlit[wv];
                if (wv < wave) { sel_before += s_; lit_before += l_; }
                sel_total += s_;
                lit_total += l_;
            }
            const uint64_t lt = (1ull << lane) - 1;
            const uint32_t my_sel_idx = sel_before + (uint32_t)__popcll(msel & lt);
            const uint32_t my_lit_idx = lit_before + (uint32_t)__popcll(mlit & lt);
            if (sel) {
                // the literal position stands in for the literal length (difference of neighbours, taken in stage 2)
                seq_out[nseq + my_sel_idx] = zge_pack_seq(lp + my_lit_idx, my_len, my_off);
                if (my_sel_idx + 1 == sel_total) L.ctrl[K_NEW0] = my_off;
                if (my_sel_idx + 2 == sel_total) L.ctrl[K_NEW1] = my_off;
            }
            if (islit) lit_out[lp + my_lit_idx] = W.lds[p - W.lo];
            nseq += sel_total;
            lp += lit_total;
            __syncthreads();
            if (tid == 0) { // offset guesses for the next tile: offsets of the last two matches selected so far
                if (sel_total >= 2) { L.ctrl[K_REP0] = L.ctrl[K_NEW0]; L.ctrl[K_REP1] = L.ctrl[K_NEW1]; }
                else if (sel_total == 1) { L.ctrl[K_REP1] = L.ctrl[K_REP0]; L.ctrl[K_REP0] = L.ctrl[K_NEW0]; }
            }
        }
        if (tid == 0) { rec->nseq = nseq; rec->nlit = lp; }
        __syncthreads();
    }
}
