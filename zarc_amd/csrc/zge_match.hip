// zarc_amd/csrc/zge_match.hip -- encoder stage 1: LZ77 match finding for gfx950.
//
// Part of the replacement for `CCtx::compress2` at crates/zarc/src/encode/lowlevel_frames.rs:29-31
// (called per entry from Encoder::add_data_frame, crates/zarc/src/encode/content_frame.rs:41).
//
// One workgroup (8 waves) per frame, one frame per CU at a time: the two position hash tables
// (8-byte "long" hash and 5-byte "short" hash, 2^14 u32 entries each = 128 KiB) live in LDS for the whole
// frame, so matches reach back across all earlier blocks of the frame (window = frame, capped at 2^window_log).
// A block (<=128 KiB) is swept in tiles of 1024 positions:
//   stage A  (wave 0)    ordered lookup+insert, 64 positions at a time: position p sees every insert
//                        of earlier 64-groups (LDS executes one wave's instructions in order)
//   stage B  (all waves) every position scores its candidates {long, short, rep0, rep1}: common-prefix
//                        length (capped) against HBM/L2, backward extension, cost model -> M[p]
//   stage C  (wave 0)    greedy selection with one-byte lazy lookahead, done 64 positions per step with
//                        ballots: the serial walk only visits *selected matches*, never literals;
//                        literal bytes are compacted with a ballot/popcount prefix
// Output per block: packed sequences (ll, ml, offset value) + literal bytes in HBM scratch for stage 2.
// The parse is deterministic and bit-identical to oracle/zstd_enc_model.c (tests/ compare them).
#include "zarc_device.h"
#include "zarc_kernels.h"

namespace {

constexpr int TILE = 1024;
constexpr int THREADS = 512;
constexpr int TAB_LOG_MAX = 14;

struct MatchLds {
    uint32_t tl[1 << TAB_LOG_MAX];
    uint32_t ts[1 << TAB_LOG_MAX];
    uint32_t cand_l[TILE], cand_s[TILE]; // stage A -> B
    uint32_t m_off[TILE], m_w[TILE];      // stage B -> C : offset ; len | back << 8 | (score + 32768) << 16
    uint32_t tb[(TILE + 32) / 4];         // tile bytes (+ slack for 8-byte reads at the last positions)
    uint32_t ctrl[16];
};
enum { K_POS_LO = 0, K_REP0 = 1, K_REP1 = 2, K_FLAG = 3 };

__device__ __forceinline__ uint32_t hash_long(uint64_t v, int bits) { return (uint32_t)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - bits)); }
__device__ __forceinline__ uint32_t hash_short(uint64_t v, int bits, int nbytes)
{
    return (uint32_t)(((v << (64 - 8 * nbytes)) * 0x9E3779B185EBCA87ULL) >> (64 - bits));
}

// common prefix of src[p..] and src[q..] (q < p), at most `limit` bytes, 8 bytes per step
__device__ __forceinline__ uint32_t match_len(const uint8_t *src, uint64_t p, uint64_t q, uint32_t limit)
{
    uint32_t n = 0;
    while (n + 8 <= limit) {
        const uint64_t x = zd::load_u64(src + p + n) ^ zd::load_u64(src + q + n);
        if (x) return n + (uint32_t)(zd::ctz64(x) >> 3);
        n += 8;
    }
    if (n < limit) {
        const uint64_t x = zd::load_u64(src + p + n) ^ zd::load_u64(src + q + n);
        uint32_t m = x ? (uint32_t)(zd::ctz64(x) >> 3) : 8u;
        if (m > limit - n) m = limit - n;
        n += m;
    }
    return n;
}

__device__ __forceinline__ int32_t score_of(const ZgeParams &P, uint32_t len, uint32_t off, bool is_rep)
{
    if (is_rep) return (int32_t)(P.lit_cost * (int)len) - P.rep_cost;
    return (int32_t)(P.lit_cost * (int)len) - P.match_cost - zd::hb32(off);
}

} // namespace

__global__ void __launch_bounds__(512) zarc_zge_match(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
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

    for (int i = tid; i < (1 << TAB_LOG_MAX); i += THREADS) { L.tl[i] = 0; L.ts[i] = 0; }
    __syncthreads();

    for (uint32_t b = 0; b < nblocks; b++) {
        const uint64_t bs = (uint64_t)b * ZARC_BLOCK;
        const uint64_t be = bs + ZARC_BLOCK < n ? bs + ZARC_BLOCK : n;
        const uint32_t blen = (uint32_t)(be - bs);
        ZgeBlock *rec = blocks + first_block + b;
        uint64_t *seq_out = seq_scratch + (first_block + b) * (uint64_t)ZARC_MAX_SEQ;
        uint8_t *lit_out = lit_scratch + (first_block + b) * (uint64_t)(ZARC_BLOCK + 64);

        // ---- RLE block detection: every byte equals the first one ----
        if (tid == 0) L.ctrl[K_FLAG] = 0;
        __syncthreads();
        {
            bool diff = false;
            const uint8_t first = blen ? src[bs] : 0;
            for (uint64_t i = bs + (uint64_t)tid; i < be; i += THREADS) diff |= src[i] != first;
            if (diff) L.ctrl[K_FLAG] = 1; // benign race: all writers store 1
        }
        __syncthreads();
        const bool all_same = L.ctrl[K_FLAG] == 0;
        __syncthreads();
        if (tid == 0) {
            rec->frame = f; rec->index = b; rec->src_len = blen; rec->nseq = 0; rec->nlit = 0;
            rec->type = (all_same && blen >= 2) ? 1u : 2u; rec->out_len = 0; rec->pad = 0;
        }
        if (all_same && blen >= 2) continue; // nothing is inserted for RLE blocks (same rule as the model)

        // per-block parse state, owned by wave 0 (replicated in its lanes); repcodes restart unknown (0)
        uint64_t anchor = bs, pos = bs;
        uint32_t rep0 = 0, rep1 = 0, rep2 = 0, nseq = 0, lp = 0;
        if (tid == 0) { L.ctrl[K_POS_LO] = 0; L.ctrl[K_REP0] = 0; L.ctrl[K_REP1] = 0; }
        __syncthreads();

        for (uint64_t tile = bs; tile < be; tile += TILE) {
            const uint64_t tend = tile + TILE < be ? tile + TILE : be;
            const uint32_t tcount = (uint32_t)(tend - tile);
            // whole tile already covered by a match: skip it (uniform: the parse cursor is published by wave 0)
            const uint64_t cur = bs + L.ctrl[K_POS_LO];
            if (cur >= tend) continue;
            // ---- stage 0: tile bytes -> LDS (aligned dword loads; the arena is padded) ----
            {
                const uint8_t *tp = src + tile;
                const uintptr_t a = (uintptr_t)tp;
                const uint32_t mis = (uint32_t)(a & 3);
                const uint32_t *w = (const uint32_t *)(a - mis);
                // byte k of the tile sits at L.tb byte (k + mis); 16 bytes past the tile are staged for the
                // 8-byte reads of its last positions (the arena is padded by ZARC_GPU_PAD)
                const int ndw = (int)((tcount + mis + 16 + 3) / 4);
                for (int i = tid; i < ndw; i += THREADS) L.tb[i] = w[i];
            }
            __syncthreads();
            const uint8_t *tb = (const uint8_t *)L.tb + (((uintptr_t)(src + tile)) & 3);
            // ---- stage A: ordered lookup + insert (wave 0) ----
            if (wave == 0) {
                for (int k = 0; k < TILE / 64; k++) {
                    const uint32_t idx = (uint32_t)(k * 64 + lane);
                    const uint64_t p = tile + idx;
                    const bool act = idx < tcount && p < hash_end;
                    uint32_t hl = 0, hs = 0, cl = 0, cs = 0;
                    if (act) {
                        const uint64_t v = zd::load_u64(tb + idx);
                        hl = hash_long(v, P.long_log);
                        hs = hash_short(v, P.short_log, P.short_bytes);
                        cl = L.tl[hl];
                        cs = L.ts[hs];
                    }
                    L.cand_l[idx] = cl;
                    L.cand_s[idx] = cs;
                    zd::wave_sync(); // all lookups of this 64-group precede its inserts
                    if (act) {
                        atomicMax(&L.tl[hl], (uint32_t)p + 1);
                        atomicMax(&L.ts[hs], (uint32_t)p + 1);
                    }
                    zd::wave_sync(); // inserts precede the next group's lookups
                }
            }
            __syncthreads();
            // ---- stage B: score candidates (all waves, 2 positions per thread) ----
            const uint32_t erep0 = L.ctrl[K_REP0], erep1 = L.ctrl[K_REP1];
            for (int r = 0; r < TILE / THREADS; r++) {
                const uint32_t idx = (uint32_t)(r * THREADS + tid);
                if (idx >= tcount) continue;
                const uint64_t p = tile + idx;
                const uint32_t limit = (uint32_t)(be - p), cap = limit < (uint32_t)P.cap ? limit : (uint32_t)P.cap;
                uint32_t best_len = 0, best_off = 0;
                int32_t best_score = -1000000;
                const uint32_t c0 = L.cand_l[idx], c1 = L.cand_s[idx];
                if (c0) {
                    const uint64_t off = p - (c0 - 1);
                    if (off != 0 && off <= window) {
                        const uint32_t len = match_len(src, p, p - off, cap);
                        if (len >= (uint32_t)P.min_match) {
                            const int32_t sc = score_of(P, len, (uint32_t)off, false);
                            if (sc > best_score) { best_score = sc; best_len = len; best_off = (uint32_t)off; }
                        }
                    }
                }
                if (c1 && c1 != c0) {
                    const uint64_t off = p - (c1 - 1);
                    if (off != 0 && off <= window && off <= (1ull << P.short_window_log)) {
                        const uint32_t len = match_len(src, p, p - off, cap);
                        if (len >= (uint32_t)P.min_match) {
                            const int32_t sc = score_of(P, len, (uint32_t)off, false);
                            if (sc > best_score) { best_score = sc; best_len = len; best_off = (uint32_t)off; }
                        }
                    }
                }
                for (int k = 0; k < P.rep_search; k++) {
                    const uint32_t off = k == 0 ? erep0 : erep1;
                    if (off == 0 || off > p || off > window) continue;
                    const uint32_t len = match_len(src, p, p - off, cap);
                    if (len < (uint32_t)P.min_rep) continue;
                    const int32_t sc = score_of(P, len, off, true);
                    if (sc > best_score) { best_score = sc; best_len = len; best_off = off; }
                }
                uint32_t w = 0, o = 0;
                if (best_len && best_score > 0) {
                    // backward extension potential: bytes before p equal to bytes before the source
                    uint32_t back = 0;
                    const uint64_t q = p - best_off;
                    uint32_t maxb = (uint32_t)P.back_cap;
                    if (p - bs < maxb) maxb = (uint32_t)(p - bs);
                    if (q < maxb) maxb = (uint32_t)q;
                    while (back < maxb && src[p - back - 1] == src[q - back - 1]) back++;
                    o = best_off;
                    w = best_len | (back << 8) | ((uint32_t)(best_score + 32768) << 16);
                }
                L.m_off[idx] = o;
                L.m_w[idx] = w;
            }
            __syncthreads();
            // ---- stage C: selection (wave 0), 64 positions per step ----
            if (wave == 0) {
                pos = bs + L.ctrl[K_POS_LO];
                for (uint32_t cb = 0; cb < tcount; cb += 64) {
                    const uint64_t chunk = tile + cb;
                    const uint32_t ccount = tcount - cb < 64 ? tcount - cb : 64;
                    const uint32_t idx = cb + (uint32_t)lane;
                    const bool valid = (uint32_t)lane < ccount;
                    const uint32_t mw = valid ? L.m_w[idx] : 0u, moff = valid ? L.m_off[idx] : 0u;
                    const uint32_t len = mw & 255, back = (mw >> 8) & 255;
                    const int32_t score = (int32_t)(mw >> 16) - 32768;
                    const uint32_t len_n = zd::shfl_down(len, 1);
                    const int32_t score_n = (int32_t)zd::shfl_down((uint32_t)score, 1);
                    const bool lazy_skip = P.lazy && len && (uint32_t)lane + 1 < ccount && len_n && score_n > score + P.lazy_delta;
                    const uint64_t take = zd::ballot(len != 0 && !lazy_skip);
                    uint64_t covered = 0; // positions of this chunk inside a selected match
                    uint64_t p = pos > chunk ? pos : chunk;
                    const uint64_t cend = chunk + ccount;
                    if (pos > chunk) { // entered covered by an earlier match
                        const uint64_t cnt = pos - chunk >= 64 ? 64 : pos - chunk;
                        covered = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
                    }
                    while (p < cend) {
                        const uint32_t rel = (uint32_t)(p - chunk);
                        const uint64_t m = take & ~((rel ? (1ull << rel) : 1ull) - 1);
                        if (!m) break;
                        const int qi = zd::ctz64(m);
                        uint64_t q = chunk + (uint64_t)qi;
                        uint32_t mlen = zd::uniform(zd::shfl(len, qi));
                        const uint32_t off = zd::uniform(zd::shfl(moff, qi));
                        uint32_t bk = zd::uniform(zd::shfl(back, qi));
                        if (mlen == (uint32_t)P.cap) {
                            // forward extension, 64 x 8 bytes per step
                            const uint32_t limit = (uint32_t)(be - q);
                            uint32_t done = mlen;
                            for (;;) {
                                const uint32_t o8 = done + (uint32_t)lane * 8;
                                uint32_t mlane = 0; // matched bytes in this lane's 8-byte window
                                if (o8 < limit) {
                                    const uint64_t x = zd::load_u64(src + q + o8) ^ zd::load_u64(src + q - off + o8);
                                    mlane = x ? (uint32_t)(zd::ctz64(x) >> 3) : 8u;
                                    if (mlane > limit - o8) mlane = limit - o8;
                                }
                                const uint64_t full = zd::ballot(mlane == 8);
                                const int firstbad = ~full ? zd::ctz64(~full) : 64;
                                if (firstbad == 64) { done += 512; continue; }
                                done += (uint32_t)firstbad * 8 + zd::uniform(zd::shfl(mlane, firstbad));
                                break;
                            }
                            mlen = done < limit ? done : limit;
                        }
                        // backward extension is confined to this chunk and to pending literals
                        {
                            const uint64_t floor_ = anchor > chunk ? anchor : chunk;
                            if (bk > q - floor_) bk = (uint32_t)(q - floor_);
                        }
                        q -= bk;
                        mlen += bk;
                        const uint32_t ll = (uint32_t)(q - anchor);
                        // offset value against the live repcode history (RFC 8878 3.1.1.5)
                        uint32_t ofv;
                        if (ll > 0) {
                            if (off == rep0) ofv = 1;
                            else if (off == rep1) { ofv = 2; rep1 = rep0; rep0 = off; }
                            else if (off == rep2) { ofv = 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
                            else { ofv = off + 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
                        } else {
                            if (off == rep1) { ofv = 1; rep1 = rep0; rep0 = off; }
                            else if (off == rep2) { ofv = 2; rep2 = rep1; rep1 = rep0; rep0 = off; }
                            else if (rep0 > 1 && off == rep0 - 1) { ofv = 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
                            else { ofv = off + 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
                        }
                        if (lane == 0) seq_out[nseq] = zge_pack_seq(ll, mlen, ofv);
                        nseq++;
                        anchor = q + mlen;
                        p = anchor;
                        // mark [q, anchor) inside this chunk as covered
                        {
                            const uint32_t lo = (uint32_t)(q - chunk);
                            const uint64_t hi64 = anchor - chunk;
                            const uint64_t below_hi = hi64 >= 64 ? ~0ull : ((1ull << hi64) - 1);
                            const uint64_t below_lo = lo ? ((1ull << lo) - 1) : 0ull;
                            covered |= below_hi & ~below_lo;
                        }
                    }
                    if (p > pos) pos = p;
                    // literal bytes of this chunk, compacted in position order
                    {
                        const uint64_t litmask = ~covered & (ccount >= 64 ? ~0ull : ((1ull << ccount) - 1));
                        if (valid && ((litmask >> lane) & 1)) {
                            const uint32_t dst = lp + (uint32_t)__popcll(litmask & ((1ull << lane) - 1));
                            lit_out[dst] = tb[idx];
                        }
                        lp += (uint32_t)__popcll(litmask);
                    }
                }
                if (lane == 0) {
                    L.ctrl[K_POS_LO] = (uint32_t)((pos > be ? be : pos) - bs);
                    L.ctrl[K_REP0] = rep0;
                    L.ctrl[K_REP1] = rep1;
                }
            }
            __syncthreads();
        }
        if (tid == 0) { rec->nseq = nseq; rec->nlit = lp; }
    }
}
