// zarc_amd/csrc/zge_match.hip -- encoder stage 1: LZ77 match finding for gfx950.
//
// Part of the replacement for `CCtx::compress2` at crates/zarc/src/encode/lowlevel_frames.rs:29-31
// (called per entry from Encoder::add_data_frame, crates/zarc/src/encode/content_frame.rs:41).
//
// Persistent 512-thread workgroups, TWO per CU (their waits overlap), take frames from a queue (largest first).  The
// near hash table(s) of a frame -- level 3: one table of 2^15 16-bit entries on the 5-byte hash; level >= 9: an 8-byte "long" and a
// 4-byte "short" hash table of 2^13 u32 entries each, an entry packing (position+1) << 10 | 10 hash check bits -- are 64 KiB of LDS
// and stay there for the whole frame, so matches reach back across all earlier blocks (64 KiB / the 2 MiB table segment).  The kernel is bound by VALU issue and by barrier / L2 waits, not by HBM.
// A block (<= 64 KiB: ZARC_BLOCK) is swept in tiles of 1024 positions, two positions per thread (t and t+512):
//   S0/S1 the tile's window (recent-offset range before it, compare overrun after it) goes to LDS -- the dword of the
//         NEXT tile is requested now and parked in a register; every position is hashed (32-bit multiplies)
//   S2    ordered lookup + insert, one wave per table, 64 positions per step: LDS executes one wave's instructions
//         in order, so a position sees every insert of earlier 64-groups with no waiting between steps
//   S3    every position scores its candidates: one 16-byte request source[-8..8) per table candidate (first compare
//         and backward extension), the two recent-offset guesses are scored out of LDS while those are in flight;
//         comparisons advance 16 bytes per LDS / global round trip; offers for backward propagation are posted with ds_max
//   far   a third (level >= 9: up to eight more) candidate per position comes from far tables in HBM (one slab per workgroup):
//         same entry format, but only every 2^far_step_log-th position is inserted (entries live that much longer) and a tile's
//         lookups see the inserts of EARLIER tiles only -- lookups are issued in S1 and land during S2, inserts are fire-and-forget
//         atomic max after S3, so neither is on the tile's critical path and no order inside a tile is needed
//   S4    backward propagation: position t may start the match found at t+k, k bytes earlier
//   S5    one-byte lazy rule -> take flag and successor next[t] for every position
//   S6    the greedy parse IS the path from the entry cursor through next[]: per 64-position chunk the exit of
//         every position by 6 rounds of shuffle pointer-jumping, then the chunk entries by a short chain
//         through LDS, then each wave marks its chunk's path with v_readlane -- no workgroup barriers
//   S7    ballot/popcount prefix sums (16-lane scan over the chunks) place literal bytes and sequences
//   A tile in which no position found a match takes an all-literals path after S3; after two such tiles in a row the next
//   1, 3, then 7 tiles are not searched at all (cold stretch: bytes go straight to the literals, nothing is inserted).
// Output per block: packed (literal position, match length, offset) + literal bytes in HBM scratch; literal
// lengths and repcodes are resolved by the entropy stage.  Deterministic and bit-identical to
// oracle/zstd_enc_model.c (tests/ compare them).
#include "zarc_device.h"
#include "zarc_kernels.h"

namespace {

#ifdef ZARC_HIPEMU
#define ZGE_CLOCK() 0ull
#else
#define ZGE_CLOCK() ((unsigned long long)__builtin_readcyclecounter())
#endif
// A thread's positions: wave w owns the ADJACENT 64-position chunks 2w and 2w+1 of the tile (the parse walks a pair without
// going through LDS, and the chain over chunk entries hops pair by pair)
#define ZGE_IDX(u) ((uint32_t)((wave * PER + (u)) * 64 + lane))
// stage timing (diagnostics): ticks since the previous mark, accumulated per stage; each mark sits after a barrier
#ifndef ZGE_S2_DUPMASK
#define ZGE_S2_DUPMASK 0 // A/B switch: 1 = a lane whose right neighbour has the same hash does not store (what the 32-bit tables' atomics need)
#endif
#define ZGE_PROF(i) do { if ((dbg & 1024) && tid == 0) { const unsigned long long now_ = ZGE_CLOCK(); L.prof[i] += now_ - tprev; tprev = now_; } } while (0)

constexpr int TILE = 1024;
constexpr int THREADS = 512;
constexpr int WAVES = THREADS / 64;
constexpr int PER = TILE / THREADS; // positions per thread
static_assert(PER == 2, "the kernel is written for two positions per thread");
constexpr int CHUNKS = TILE / 64;
constexpr int TAG_BITS = 10;
constexpr uint32_t TAG_MASK = (1u << TAG_BITS) - 1;
struct U128 { uint64_t lo, hi; };
constexpr int CAP_MAX = 256;
constexpr int REP_BACK_MAX = 256;
constexpr int TB_BYTES = 12 + REP_BACK_MAX + TILE + CAP_MAX + 24; // rep window + 8 bytes before the tile, compare overrun after it

// NEAR16 (the level-3 finder since round 3): ONE near table of 2^TAB_LOG 16-bit entries keyed by the short hash -- an entry is the
// low 16 bits of a position, a candidate lies 1 .. 65536 bytes back, there are no check bits and no long table (2^15 entries in the
// 64 KiB the two 2^13-entry tables took).  Otherwise: long table, then short table, 2^TAB_LOG 32-bit entries each.
template <int TAB_LOG, bool NEAR16> struct MatchLds {
    static constexpr int TAB_WORDS = NEAR16 ? (1 << TAB_LOG) / 2 : 2 << TAB_LOG;
    uint32_t tab[TAB_WORDS + (NEAR16 ? 1 : 0)]; // NEAR16: one more word = slot 2^TAB_LOG, where positions without a hash look up and store (no branch in the chain)
    uint32_t a0[TILE], a1[TILE]; // S1: hashes -> S2: candidates (pos+1) -> S3: a0 = own match (match_pack) -> S4: a1 = final match
    uint32_t ex[TILE];            // S4: best backward offer per position; S6: first position outside its chunk reached from each position
    uint32_t tb[2][(TB_BYTES + 3) / 4]; // the window of a tile lives in buffer (tile / TILE) & 1: the current tile's and the next one's
    uint32_t wcnt[CHUNKS];       // S6: selected matches << 16 | literals of each chunk
    uint32_t wrep[2 * CHUNKS];   // S6 (level >= 9): offset of each chunk's last selected match, and of the last one different from it
    uint32_t ctrl[16];
    unsigned long long prof[15]; // ZARC_GPU_DBG & 1024: shader-clock ticks per stage, workgroup view from thread 0
};
enum { K_POS = 0, K_REP0 = 1, K_REP1 = 2, K_FLAG = 3, K_SLOT = 4, K_ANY = 5, K_CHG = 6 /* .. 9: a round of the live recent-offset pass changed a match */,
       K_CEND = 10 /* block position behind the last selected match of the last searched tile (0xFFFFFFFF: none) */,
       K_CLEN = 11 /* length of the continuation guess at the tile's cursor */,
       K_LONG = 12 /* EXT_ROUND: a candidate of this tile reaches the cap */, K_LONG2 = 13 /* ... and a selected match was cut at it */ };

// Hashes from 32-bit multiplies only (a 64-bit multiply is four quarter-rate VALU ops on gfx950).
// The near and the far tables index with different numbers of top bits of the SAME 32-bit product.
__device__ __forceinline__ uint32_t hash_long32(uint64_t v)
{
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return lo * 0x9E3779B1u + hi * 0x85EBCA77u;
}
// far long-hash table: 12 bytes (the 8 at the position and the next 4): far offsets are expensive to code, so only repeats of some
// length are worth finding there, and a table keyed by 12 bytes is not crowded by the short repeats of text-like data
__device__ __forceinline__ uint32_t hash_far32(uint64_t v, uint32_t w) { return hash_long32(v) + w * 0xC2B2AE3Du; }
__device__ __forceinline__ uint32_t hash_short32(uint64_t v, int nbytes)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (nbytes < 8) hi &= (nbytes > 4) ? ((1u << (8 * (nbytes - 4))) - 1) : 0u;
    if (nbytes < 4) lo &= (1u << (8 * nbytes)) - 1;
    return lo * 0xC2B2AE3Du + hi * 0x27D4EB2Fu;
}

// Bit-cost model (lit_cost 5, match_cost 12 (level >= 9: 10), rep_cost 9: the engine's fixed defaults, so the literal cost is a
// shift-add instead of a quarter-rate multiply; engine.hip: derive_params() sets exactly these, they are not tunable).
constexpr int REP_COST = 9; // (lit_cost: template parameter LITC, 5 below level 9, 6 from there)
// Parameters of the model that the engine never varies (engine.hip: derive_params sets exactly these and checks them before a
// launch): as constants they cost no scalar registers -- the kernel keeps about a hundred uniform values alive and spills them.
constexpr int F_REP_BACK = 256, F_BACK_CAP = 8, F_LAZY_DELTA = 5, F_MIN_REP = 3, F_SEG_LOG = 21; // rep_search 2, short window unlimited
template <int MATCH_COST, int LITC> __device__ __forceinline__ int32_t score_mc(uint32_t len, uint32_t off, bool is_rep)
{
    static_assert(LITC == 5 || LITC == 6, "literal cost");
    const int32_t lits = (int32_t)((len << 2) + (LITC == 6 ? len << 1 : len));
    return is_rep ? lits - REP_COST : lits - MATCH_COST - zd::hb32(off);
}
#define score_of(P_, len_, off_, rep_) score_mc<MATCH_COST, LITC>((len_), (off_), (rep_))

// a match in one LDS word: offset (21 bits) | length (9 bits; WIDE, the finder with the continuation guess: 10 bits) << 21 | recent-offset flag << 30 (WIDE: 31)
template <bool WIDE> __device__ __forceinline__ uint32_t match_pack_w(uint32_t off, uint32_t len, bool rep) { return off | (len << 21) | ((rep ? 1u : 0u) << (WIDE ? 31 : 30)); }
__device__ __forceinline__ uint32_t match_off(uint32_t m) { return m & 0x1FFFFFu; }
template <bool WIDE> __device__ __forceinline__ uint32_t match_len_w(uint32_t m) { return (m >> 21) & (WIDE ? 0x3FFu : 0x1FFu); }
template <bool WIDE> __device__ __forceinline__ bool match_rep_w(uint32_t m) { return WIDE ? (m >> 31) != 0 : ((m >> 30) & 1u) != 0; }
#define match_pack(o_, l_, r_) match_pack_w<(CONT_CAP > 0 || EXT_ROUND)>((o_), (l_), (r_))
#define match_len(m_) match_len_w<(CONT_CAP > 0 || EXT_ROUND)>(m_)
#define match_rep(m_) match_rep_w<(CONT_CAP > 0 || EXT_ROUND)>(m_)
constexpr int LONG_CAP = 960; // cont_cap: a length stays below 2^10

// Common prefix of frame positions p and p - off from byte `from` on, at most `maxlen` bytes in all, by ONE wave in one round trip:
// lane j compares bytes [from + 16 j, from + 16 j + 16) of both sides out of global memory (1 KiB per trip; the per-lane compare
// loops of S3 advance 16 bytes per trip).  Every argument is the same in all lanes.  Loads stay below maxlen + 16 <= block end + 16.
__device__ __forceinline__ uint32_t wave_match_ext(const uint8_t *src, uint32_t p, uint32_t off, uint32_t from, uint32_t maxlen, int lane)
{
    static_assert(LONG_CAP <= 1024, "one trip of 64 x 16 bytes");
    const uint32_t at = from + 16u * (uint32_t)lane;
    const bool act = at < maxlen;
    U128 a{0, 0}, b{0, 0};
    if (act) { __builtin_memcpy(&a, src + (p + at), 16); __builtin_memcpy(&b, src + (p - off + at), 16); }
    const uint64_t xl = a.lo ^ b.lo, xh = a.hi ^ b.hi;
    const uint32_t mis = xl ? (uint32_t)(zd::ctz64(xl) >> 3) : (xh ? 8u + (uint32_t)(zd::ctz64(xh) >> 3) : 16u);
    const uint64_t bad = zd::ballot(act && mis < 16u);
    uint32_t len = maxlen;
    if (bad) {
        const uint32_t first = (uint32_t)zd::ctz64(bad);
        len = from + 16u * first + zd::readlane(mis, first);
        if (len > maxlen) len = maxlen;
    }
    return len;
}

// The window of a tile staged in LDS: frame bytes [lo, hi) = rep_back + 8 bytes before the tile .. cap + 16 after it,
// fetched as whole dwords starting at the aligned address `w` (one dword per thread: TB_BYTES / 4 <= THREADS).
struct StageWin { const uint32_t *w; int ndw; uint32_t wofs; };
static_assert((TB_BYTES + 3) / 4 <= THREADS, "one staged dword per thread");
__device__ __forceinline__ StageWin stage_window(const ZgeParams &P, const uint8_t *src, uint32_t n, uint32_t tile, uint32_t tend, uint32_t cap_max)
{
    (void)P;
    const uint32_t before = (uint32_t)F_REP_BACK + 8;
    const uint32_t lo = tile >= before ? tile - before : 0;
    uint32_t hi = tend + cap_max + 16;
    if (hi > n + 16) hi = n + 16; // the arena is padded by ZARC_GPU_PAD
    const uintptr_t a = (uintptr_t)(src + lo);
    const uint32_t mis = (uint32_t)(a & 3);
    StageWin s;
    s.w = (const uint32_t *)(a - mis);
    s.ndw = (int)((hi - lo + mis + 3) / 4);
    s.wofs = mis - lo;
    return s;
}

// end of the tile that starts at `tile`: tiles never straddle a block
__device__ __forceinline__ uint32_t tile_end(uint32_t tile, uint32_t n)
{
    uint32_t e = tile + TILE;
    const uint32_t be = (tile / ZARC_BLOCK + 1) * ZARC_BLOCK;
    if (e > be) e = be;
    return e > n ? n : e;
}

} // namespace

// The body is compiled twice: the level-3 finder (one 2^15-entry 16-bit table on the 5-byte hash, content-sampled far table, two
// workgroups per CU) and the deep one for level >= 9 (two tagged 2^13-entry tables, 4-byte short hash, cheaper matches, two-way far
// tables on both hashes, literals priced at LITC 6, a second lazy step LAZY2, and REP_PASS: rounds of the live recent-offset pass;
// 219 registers per thread, so one workgroup per CU).
// DIAG: the timing-only switches of ZARC_GPU_DBG are compiled into a separate instantiation, so the product kernels carry none of
// their scalar tests
// F_FAR_LOG / FAR_WAYS / FAR_SHORT / FAR_STEP_LOG / FAR_RES_LOG / FAR_SKIP / FAR_BACK: the far tables (0 ways = none); the engine checks that
// P carries the same values.  Positions p with p mod 2^FAR_STEP_LOG < 2^FAR_RES_LOG are inserted, positions with
// p mod 2^FAR_RES_LOG == 0 are looked up: of 2^FAR_RES_LOG inserted neighbours exactly one lands on a looked-up position whatever the
// offset of the repeat -- the memory requests per tile (what the far tables cost) go down by that factor.
// FAR_CDC > 0: content-defined sampling of the far table (FAR_STEP_LOG / FAR_RES_LOG unused): a position is inserted AND looked up iff
// the FAR_CDC bits of its 12-byte hash just below the bucket and check bits are zero.  Both occurrences of a repeat sample the same
// relative positions, so a repeat is found when it contains one sample: 64 lookups + 64 inserts per 1024-position tile at FAR_CDC 4
// instead of 256 + 128, and every thread asks for its OWN positions (no hand-over between lanes).
template <int TAB_LOG, int SHORT_BYTES, int MATCH_COST, int F_FAR_LOG, int FAR_WAYS, bool FAR_SHORT, int FAR_STEP_LOG, int FAR_RES_LOG, int FAR_SKIP, int FAR_BACK, bool DIAG,
          bool NEAR16, int FAR_CDC, int LITC, int LAZY2, bool REP_PASS, int CONT_CAP, bool EXT_ROUND>
__device__ __forceinline__ void zge_match_body(MatchLds<TAB_LOG, NEAR16> &L, const ZgeParams &P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                               const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, const uint32_t *__restrict__ units, uint32_t n_units,
                                               const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                               uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch, uint32_t *__restrict__ queue,
                                               uint32_t *__restrict__ far_scratch)
{
    constexpr int NFAR = FAR_WAYS * (FAR_SHORT ? 2 : 1); // far candidates per position
    constexpr int NTAB = 1 + NFAR;                       // table candidates per position: near (long table, else short table), far ...
    constexpr int far_shift = 32 - (F_FAR_LOG + TAG_BITS); // far bucket | tag = top bits of the 32-bit hash product
    constexpr uint32_t far_words = (uint32_t)(((size_t)FAR_WAYS << F_FAR_LOG) * (FAR_SHORT ? 2 : 1));
    uint32_t *const far_l = far_scratch + (size_t)blockIdx.x * far_words;  // this workgroup's slab: long-hash table, then short-hash table
    uint32_t *const far_s = far_l + ((size_t)FAR_WAYS << F_FAR_LOG);
    constexpr uint32_t far_smask = (1u << FAR_STEP_LOG) - 1, far_rmask = (1u << FAR_RES_LOG) - 1;
    static_assert(FAR_BACK % 8 == 0 && FAR_BACK >= 8 && FAR_BACK <= 48, "backward extension of far candidates");
    static_assert(!FAR_CDC || (FAR_WAYS == 1 && !FAR_SHORT && far_shift >= FAR_CDC), "content-defined sampling: one way on the 12-byte hash");
    constexpr uint32_t cdc_mask = (1u << FAR_CDC) - 1;
    uint32_t *const Ltl = L.tab, *const Lts = L.tab + (NEAR16 ? 0 : (1 << TAB_LOG)); // long / short table (not NEAR16)
    typedef uint16_t __attribute__((may_alias)) u16a;
    u16a *const Lt16 = (u16a *)L.tab;                                                  // the 16-bit near table (NEAR16)
    const int tid = (int)threadIdx.x, lane = zd::lane_id();
    const uint32_t dbg = DIAG ? (uint32_t)P.dbg : 0u;
    const int wave = (int)zd::uniform((uint32_t)zd::wave_id()); // scalar: chunk bounds and the parse walk stay on the SALU
    const uint64_t lt = (1ull << lane) - 1;
    const uint32_t seg_mask = (1u << F_SEG_LOG) - 1;
    const uint32_t cap_max = (uint32_t)(P.cap < CAP_MAX ? P.cap : CAP_MAX);
    unsigned long long tprev = ZGE_CLOCK();
    if (tid < 15) L.prof[tid] = 0;

    // Persistent workgroups: the grid is what the chip holds at once (two per CU); every workgroup takes the next UNIT from a queue
    // (largest frames first), so slow and fast frames balance across XCDs whatever their order in the batch.  A unit is one 2^seg_log
    // segment (16 blocks) of a frame: the tables restart at every segment anyway, so the segments of a large frame are independent
    // pieces of work -- a 1 GiB entry is 512 units for 512 workgroups, not a 10-second chain for one.  Everything the finder carries
    // from tile to tile (tables, recent-offset guesses, the cold-stretch counters) starts afresh with a unit (model: the same resets at
    // every segment boundary).
    constexpr uint32_t SEG_BLOCKS = (1u << F_SEG_LOG) / ZARC_BLOCK;
    for (;;) {
    if (tid == 0) L.ctrl[K_SLOT] = atomicAdd(queue, 1u);
    zd::lds_barrier();
    const uint32_t unit = L.ctrl[K_SLOT];
    if (unit >= n_units) break; // uniform: every wave leaves once the queue is empty
    const uint32_t slot = units[2 * unit], ub0 = units[2 * unit + 1]; // queue slot of the frame, first block of the segment
    const uint32_t f = order[slot];
    const uint8_t *src = src_base + src_off[f];
    const uint32_t n = (uint32_t)src_len[f]; // the engine rejects entries of 4 GiB or more: positions are 32-bit
    const uint32_t window = n <= (1u << P.window_log) ? (n ? n : 1u) : (1u << P.window_log);
    const uint32_t hash_end = n >= 8 ? n - 7 : 0;
    // frames of at most far_min_frame bytes do without the far table (the near table reaches that far): no slab to clear, no requests
    const bool far_on = NFAR && n > (uint32_t)P.far_min_frame;
    const uint32_t far_end = (far_on && n >= 12) ? n - 11 : 0; // the far tables' long hash reads 12 bytes
    // block records / scratch slots are numbered within the sub-batch: block_prefix is indexed by queue slot
    const uint64_t first_block = block_prefix[slot];
    const uint32_t nblocks = (uint32_t)(block_prefix[slot + 1] - first_block);

    for (int i = tid; i < MatchLds<TAB_LOG, NEAR16>::TAB_WORDS; i += THREADS) L.tab[i] = 0;
    if (NFAR && far_on) { // the slab still holds the previous frame (whose last inserts were waited for in its last tile)
        uint4 *f4 = (uint4 *)far_l;
        for (uint32_t i = (uint32_t)tid; i < far_words / 4; i += THREADS) f4[i] = make_uint4(0, 0, 0, 0);
        zd::wait_vmem(); // the zeros are in L2 before any wave passes the barrier and looks something up
    }
    if (tid == 0) { L.ctrl[K_REP0] = 0; L.ctrl[K_REP1] = 0; } // recent-offset guesses live for the whole frame
    zd::lds_barrier();
    // Cold stretches (incompressible data): `cold` counts the searched tiles in a row in which no position found a match; from the
    // second one on, the next 1, 3, then 7 tiles are not searched at all (all literals, nothing inserted) -- the way libzstd's
    // search step grows while it finds nothing.  Any match in a searched tile ends the stretch.  Replicated in every thread.
    uint32_t cold = 0, skip_left = 0;
    // Far lookups run one tile ahead: while a tile is in S3 the entries of the NEXT tile's positions are requested (hashes straight
    // from 12 bytes of global memory), so their round trip to L2 / HBM is over when that tile starts and its far sources can be
    // requested in S1 already, ahead of S2.  They are requested before this tile's own inserts go out: a lookup sees the inserts
    // of every searched tile except the one directly in front of it (model: far_pending).
    // With FAR_RES_LOG > 0 only every 2^FAR_RES_LOG-th position is looked up: the wave's 128 >> FAR_RES_LOG lookups are made by its
    // first lanes (lane l asks for position wave * 128 + (l << FAR_RES_LOG), one pass instead of two mostly idle ones) and handed to
    // the lanes that own those positions by a shuffle at the top of the next tile.
    constexpr bool FAR_COMPACT = FAR_RES_LOG > 0 && !FAR_CDC;
    constexpr int FNEXT_ROWS = FAR_COMPACT ? 1 : PER;
    uint32_t fnext[FNEXT_ROWS][NFAR ? NFAR : 1];
    uint32_t pf_far_tile = 0xFFFFFFFFu; // the tile fnext[] belongs to
    uint32_t hfn[PER] = {0xFFFFFFFFu, 0xFFFFFFFFu}; // FAR_CDC: bucket | check bits of this thread's positions of that tile (0xFFFFFFFF: not a far position)
    // Tile windows are staged two tiles ahead: at the top of tile T the window of T+1 goes to the other LDS buffer (it was requested
    // during T-1 and sits in a register) and the window of T+2 is requested.  S1 hashes the next tile's positions out of that buffer.
    uint32_t pf_word = 0, pf_tile = 0xFFFFFFFFu;            // this thread's dword of the window of tile `pf_tile`
    uint32_t lds_tile[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};     // the tile whose window each LDS buffer holds (uniform)
#pragma unroll
    for (int u = 0; u < FNEXT_ROWS; u++)
#pragma unroll
        for (int k = 0; k < (NFAR ? NFAR : 1); k++) fnext[u][k] = 0;

    // (frames up to ZARC_SPLIT_MIN stay one unit: their recent-offset guesses run on across segment boundaries, which is worth a few
    // hundred bytes per boundary on periodic data -- nothing for a large frame, a third of a 4 MB periodic buffer's 600 bytes)
    const uint32_t ub1 = (n > ZARC_SPLIT_MIN && ub0 + SEG_BLOCKS < nblocks) ? ub0 + SEG_BLOCKS : nblocks;
    for (uint32_t b = ub0; b < ub1; b++) {
        const uint32_t bs = b * ZARC_BLOCK;
        const uint32_t be = n - bs > ZARC_BLOCK ? bs + ZARC_BLOCK : n;
        const uint32_t blen = (uint32_t)(be - bs);
        ZgeBlock *rec = blocks + first_block + b;
        uint64_t *seq_out = seq_scratch + (first_block + b) * zge_seq_stride((uint32_t)P.slot_bytes);
        uint8_t *lit_out = lit_scratch + (first_block + b) * zge_lit_stride((uint32_t)P.slot_bytes);

        if (b > ub0 && (bs & seg_mask) == 0) { // a frame that is one unit: new 2^seg_log segment (a multiple of the block size), table positions restart
            for (int i = tid; i < MatchLds<TAB_LOG, NEAR16>::TAB_WORDS; i += THREADS) L.tab[i] = 0;
            if (NFAR) {
                uint4 *f4 = (uint4 *)far_l;
                for (uint32_t i = (uint32_t)tid; i < far_words / 4; i += THREADS) f4[i] = make_uint4(0, 0, 0, 0);
                zd::wait_vmem();
                pf_far_tile = 0xFFFFFFFFu; // entries requested ahead came from the old segment's tables
            }
        }
        // ---- RLE block detection: every byte equals the first one (8 bytes per load; blocks start 16-byte aligned).  Nearly
        // every block is cleared by a look at its first KiB; only a block that passes that look is read in full here, so the
        // input is not fetched twice ----
        if (tid == 0) { L.ctrl[K_FLAG] = 0; L.ctrl[K_POS] = 0; if (CONT_CAP) L.ctrl[K_CEND] = 0xFFFFFFFFu; }
        zd::lds_barrier();
        const uint8_t first = blen ? src[bs] : 0;
        const uint64_t pat = 0x0101010101010101ull * first;
        const uint64_t *w = (const uint64_t *)(src + bs);
        if (blen >= 1024 && tid < 128 && w[tid] != pat) L.ctrl[K_FLAG] = 1; // benign race: every writer stores 1
        zd::lds_barrier();
        if (L.ctrl[K_FLAG] == 0) {
            bool diff = false;
            const uint32_t nw = blen / 8;
            for (uint32_t i = (uint32_t)tid; i < nw; i += THREADS) diff |= w[i] != pat;
            for (uint32_t i = nw * 8 + (uint32_t)tid; i < blen; i += THREADS) diff |= src[bs + i] != first;
            if (diff) L.ctrl[K_FLAG] = 1;
        }
        zd::lds_barrier();
        const bool all_same = L.ctrl[K_FLAG] == 0;
        if (tid == 0) {
            rec->frame = f; rec->index = b; rec->src_len = blen; rec->nseq = 0; rec->nlit = 0;
            rec->type = (all_same && blen >= 2) ? 1u : 2u; rec->out_len = 0; rec->pad = 0;
        }
        if (all_same && blen >= 2) { zd::lds_barrier(); continue; } // nothing is inserted for RLE blocks (same rule as the model)

        uint32_t nseq = 0, lp = 0; // replicated in every thread

        for (uint32_t tile = bs; tile < be; tile += TILE) {
            const uint32_t tend = be - tile > TILE ? tile + TILE : be;
            const uint32_t tcount = (uint32_t)(tend - tile);
            const uint32_t segbase = tile & ~seg_mask;
            zd::lds_barrier(); // K_POS / K_REP* of the previous tile are final; LDS work arrays are free again
            ZGE_PROF(0);
            const uint32_t pos = bs + L.ctrl[K_POS];
            // continuation guess: the last selected match of the last searched tile ends exactly at the cursor
            const bool cont = CONT_CAP && L.ctrl[K_CEND] == L.ctrl[K_POS] && pos >= tile;
            if (pos >= tend) { if (NFAR) zd::wait_vmem(); continue; } // whole tile already covered by a match: skip it (nothing is inserted; the next searched tile asks for its far entries in S1: the last inserts are waited for here)
            if (skip_left) { // cold stretch: the tile is not searched -- its bytes go straight from HBM to the literals
                skip_left--;
                const uint32_t start = (uint32_t)((pos > tile ? pos : tile) - tile);
                { // 16 bytes per thread (one wave moves a whole tile) instead of a byte per position
                    const uint32_t cntb = tcount - start, k = (uint32_t)tid * 16;
                    const uint8_t *sp = src + tile + start;
                    uint8_t *dp = lit_out + lp;
                    if (!(dbg & 32)) {
                        if (k + 16 <= cntb) { U128 v; __builtin_memcpy(&v, sp + k, 16); __builtin_memcpy(dp + k, &v, 16); }
                        else for (uint32_t r = k; r < cntb; r++) dp[r] = sp[r];
                    }
                }
                lp += tcount - start;
                if (NFAR) zd::wait_vmem(); // as above: the next searched tile looks its far entries up in S1
                zd::lds_barrier(); // every thread has read K_POS
                if (tid == 0) L.ctrl[K_POS] = (uint32_t)(tend - bs);
                continue;
            }
            const uint32_t erep0 = L.ctrl[K_REP0], erep1 = L.ctrl[K_REP1];
            if (CONT_CAP && cont) { // uniform, rare.  Continuation guess (model: cont_cap): a match cut at its cap goes on at the same offset, however
                // far back its source lies: the wave that owns the cursor's position compares up to CONT_CAP bytes in one trip, here, where
                // few registers are live; the length waits in LDS for S3 (barriers in between), where it ranks behind every other candidate.
                const uint32_t idx_c = pos - tile; // < tcount: a tile the cursor has passed is not searched
                if (wave == (int)(idx_c >> 7)) {
                    const uint32_t limit = (uint32_t)(be - pos);
                    uint32_t clen = 0;
                    if (erep0 != 0 && erep0 <= pos && erep0 <= window) clen = wave_match_ext(src, pos, erep0, 0, limit < (uint32_t)CONT_CAP ? limit : (uint32_t)CONT_CAP, lane);
                    if (lane == 0) L.ctrl[K_CLEN] = clen;
                }
            }
            if (tid == 0) L.ctrl[K_ANY] = 0; // set by any position of this tile that finds a match
            if (REP_PASS && tid < 4) L.ctrl[K_CHG + tid] = 0;
            if (EXT_ROUND && tid == 0) { L.ctrl[K_LONG] = 0; L.ctrl[K_LONG2] = 0; }

            // ---- S0: tile bytes (8 before .. cap+16 after) -> LDS ----
            // frame position `pos` of the staged window [lo, hi) lives at LDS byte tbb[pos + wofs] (u32 arithmetic)
            const uint32_t bsel = (tile / TILE) & 1u;
            const uint8_t *const tbb = (const uint8_t *)L.tb[bsel];
            const uint8_t *const tbn = (const uint8_t *)L.tb[bsel ^ 1u]; // the next tile's window
            const StageWin sw = stage_window(P, src, n, tile, tend, cap_max);
            const uint32_t wofs = sw.wofs;
            const uint32_t ntile = tile + TILE; // may be the first tile of the next block
            uint32_t wofs_n = 0;
            bool staged_now = false; // this tile's own window had to be written just now: S1 must wait for it (uniform)
            {
                // this tile's window: normally staged while the previous tile was worked on
                if (lds_tile[bsel] != tile) {
                    uint32_t v = pf_word;
                    if (pf_tile != tile && tid < sw.ndw) v = sw.w[tid]; // first tile of a frame, or after skipped tiles
                    if (tid < sw.ndw) L.tb[bsel][tid] = v;
                    lds_tile[bsel] = tile;
                    staged_now = true;
                }
                if (ntile < n) {
                    const StageWin nw = stage_window(P, src, n, ntile, tile_end(ntile, n), cap_max);
                    wofs_n = nw.wofs;
                    uint32_t v = pf_word;
                    if (pf_tile != ntile && tid < nw.ndw) v = nw.w[tid];
                    if (tid < nw.ndw) L.tb[bsel ^ 1u][tid] = v;
                    lds_tile[bsel ^ 1u] = ntile;
                    // request the window of the tile after that now: it arrives while this tile is being worked on
                    const uint32_t n2 = ntile + TILE;
                    if (n2 < n) {
                        const StageWin w2 = stage_window(P, src, n, n2, tile_end(n2, n), cap_max);
                        if (tid < w2.ndw) pf_word = w2.w[tid];
                        pf_tile = n2;
                    }
                }
            }
            // In the steady state this tile's window has been in LDS since the previous tile and the one written above (the next
            // tile's) is not read before S3: no barrier between S0 and S1.
            if (staged_now) zd::lds_barrier();
            ZGE_PROF(1);
            // ---- S1: hashes (index << TAG_BITS | tag) ----
            uint64_t p8[PER]; // first 8 bytes at each of this thread's positions
            // far-table entries (requested during the previous tile; only a tile that follows unsearched ones asks here and waits in
            // S3), the check bits they must carry, and -- entries at hand -- the far candidates' sources, requested now: ahead of S2
            uint32_t fe[PER][NFAR ? NFAR : 1], ftag[PER][FAR_SHORT ? 2 : 1];
            // FAR_CDC: bucket | check bits of a position that IS a far position (else 0xFFFFFFFF) wait in a1[] -- free until S4 with one near
            // table -- for S3 (a tile that asked for its entries late) and for the inserts after S3: two registers less across S3
            uint32_t foffs[PER][NFAR ? NFAR : 1];
            uint64_t qf[PER][NFAR ? NFAR : 1]; // far candidates: source[0 .. 8) only (registers); a far winner fetches the 8 bytes in front later
            const bool far_ahead = NFAR && pf_far_tile == tile; // uniform
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t p = tile + idx;
                uint32_t hl = 0xFFFFFFFFu, hs = 0xFFFFFFFFu;
                p8[u] = 0;
#pragma unroll
                for (int k = 0; k < (NFAR ? NFAR : 1); k++) {
                    if (FAR_COMPACT) { // from the lane that asked for this position (only looked-up positions read a meaningful value)
                        const uint32_t got = zd::shfl(fnext[0][k], (int)(((uint32_t)u * 64u + (uint32_t)lane) >> FAR_RES_LOG));
                        fe[u][k] = (far_ahead && !(idx & far_rmask)) ? got : 0u;
                    } else fe[u][k] = far_ahead ? fnext[FAR_COMPACT ? 0 : u][k] : 0u; // (FAR_CDC: a position that is no far position got 0)
                    foffs[u][k] = 0; qf[u][k] = 0;
                }
                ftag[u][0] = 0xFFFFFFFFu; // matches no entry
                if (FAR_SHORT) ftag[u][1] = 0xFFFFFFFFu;
                uint32_t hfc = 0xFFFFFFFFu;
                if (idx < tcount) {
                    p8[u] = zd::load_u64(tbb + (uint32_t)(p + wofs));
                    if (p < hash_end && !(dbg & 64)) {
                        const uint32_t h32s = hash_short32(p8[u], SHORT_BYTES);
                        const uint32_t h32l = NEAR16 ? 0u : hash_long32(p8[u]);
                        hl = NEAR16 ? 0xFFFFFFFFu : h32l >> (32 - (TAB_LOG + TAG_BITS));
                        hs = NEAR16 ? h32s >> (32 - TAB_LOG) : h32s >> (32 - (TAB_LOG + TAG_BITS));
                        if (FAR_CDC) {
                            // Is this a far position?  Known since the previous tile's S3 (hfn) when the entries were requested there;
                            // a tile that follows unsearched ones hashes its 12 bytes here and asks now (the entries are waited for in S3).
                            if (p < far_end && !(dbg & 2048)) {
                                if (far_ahead) hfc = hfn[u];
                                else {
                                    const uint32_t w8 = zd::load_u32(tbb + (uint32_t)(p + 8 + wofs));
                                    const uint32_t h32f = hash_far32(p8[u], w8);
                                    if (((h32f >> (far_shift - FAR_CDC)) & cdc_mask) == 0 && !((uint32_t)p8[u] == (uint32_t)(p8[u] >> 32) && (uint32_t)p8[u] == w8)) hfc = h32f >> far_shift;
                                }
                            }
                            if (hfc != 0xFFFFFFFFu) {
                                if (!far_ahead) fe[u][0] = zd::load_l2_u32(far_l + (size_t)(hfc >> TAG_BITS));
                                else {
                                    const uint32_t e = fe[u][0];
                                    uint32_t o = (e && (e & TAG_MASK) == (hfc & TAG_MASK) && !(dbg & 4096)) ? p - (segbase + (e >> TAG_BITS) - 1) : 0u;
                                    if (o + 8 > p || o > window || (dbg & 1)) o = 0;
                                    foffs[u][0] = o;
                                    if (o) qf[u][0] = zd::load_u64(src + (p - o));
                                }
                            }
                        } else
                        if (NFAR && p < far_end && !(p & far_rmask) && !(dbg & 2048)) {
                            const uint32_t hf = (h32l + zd::load_u32(tbb + (uint32_t)(p + 8 + wofs)) * 0xC2B2AE3Du) >> far_shift, hg = h32s >> far_shift;
                            ftag[u][0] = hf & TAG_MASK;
                            if (FAR_SHORT) ftag[u][1] = hg & TAG_MASK;
                            if (!far_ahead) {
                                if (FAR_WAYS == 2) { // both ways of a bucket in one 8-byte request
                                    const uint64_t el = zd::load_l2_u64(far_l + (size_t)(hf >> TAG_BITS) * 2);
                                    fe[u][0] = (uint32_t)el; fe[u][1] = (uint32_t)(el >> 32);
                                    if (FAR_SHORT) { const uint64_t es = zd::load_l2_u64(far_s + (size_t)(hg >> TAG_BITS) * 2); fe[u][2] = (uint32_t)es; fe[u][3] = (uint32_t)(es >> 32); }
                                } else {
#pragma unroll
                                for (int w = 0; w < FAR_WAYS; w++) {
                                    fe[u][w] = zd::load_l2_u32(far_l + ((size_t)(hf >> TAG_BITS) * FAR_WAYS + w));
                                    if (FAR_SHORT) fe[u][FAR_WAYS + w] = zd::load_l2_u32(far_s + ((size_t)(hg >> TAG_BITS) * FAR_WAYS + w));
                                }
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < NFAR; k++) {
                                    const uint32_t e = fe[u][k];
                                    uint32_t o = (e && (e & TAG_MASK) == ftag[u][k / (FAR_WAYS ? FAR_WAYS : 1)] && !(dbg & 4096)) ? p - (segbase + (e >> TAG_BITS) - 1) : 0u;
#pragma unroll
                                    for (int j = 0; j < k; j++) if (o == foffs[u][j]) o = 0;
                                    if (o + 8 > p || o > window || (dbg & 1)) o = 0;
                                    foffs[u][k] = o;
                                    if (o) qf[u][k] = zd::load_u64(src + (p - o));
                                }
                            }
                        }
                    }
                }
                L.a0[idx] = NEAR16 ? (hs == 0xFFFFFFFFu ? (1u << TAB_LOG) : hs) : hl; // NEAR16: the one table's hash goes through a0 (no hash: the spare slot), a1 is free until S4
                if (!NEAR16) L.a1[idx] = hs;
                else if (FAR_CDC) L.a1[idx] = hfc;
                L.ex[idx] = 0; // S4 offers start empty (the previous tile's walk is over)
            }
            zd::lds_barrier();
            ZGE_PROF(2);
            // ---- S2: ordered lookup + insert (wave 0); no waits between the steps on hardware ----
            // The two tables are independent: wave 0 owns the long table, wave 1 the short one.
            if (NEAR16) {
              if (wave == 0 && !(dbg & 4)) {
                // One table of 16-bit entries (the low 16 bits of the position), one wave, and -- as with the 32-bit tables -- not a
                // single wait inside the chain: 16 x (lookup, store) go to the LDS back to back, which executes them in order.  There is
                // no 16-bit LDS atomic; the stores are plain ds_write_b16, and when several lanes of a group have the same bucket the
                // value of the HIGHEST lane stays.  That is how gfx950's LDS resolves same-address stores of one instruction (measured:
                // tools/micro/lds_write_order.hip, every one of 143 326 contested slots in random / periodic / all-equal / same-dword
                // patterns; tests/test_gpu_parity.py runs it on every box), and it is what the model's ascending loop does.  Groups are
                // 64-aligned, so a group's positions never wrap in 16 bits.  Were a chip to resolve such stores differently the frames
                // would still be valid (a candidate is only ever a position to compare with), they would differ from the model's.
                zd::wave_priority<3>(); // the other seven waves wait for this one: go ahead of the co-resident workgroup
                uint32_t h[CHUNKS], e[CHUNKS];
#if ZGE_S2_DUPMASK
                uint64_t dupmask = 0; // bit k: the next lane of group k has the same hash, so its store supersedes mine
#endif
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) {
                    h[k] = L.a0[k * 64 + lane];
#if ZGE_S2_DUPMASK
                    const uint32_t hn = lane < 63 ? L.a0[k * 64 + lane + 1] : 0xFFFFFFFFu;
                    if (hn == h[k]) dupmask |= 1ull << k;
#endif
                }
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) {
                    e[k] = (uint32_t)Lt16[h[k]];
                    zd::wave_lds_order(); // lookups of this 64-group precede its stores
#if ZGE_S2_DUPMASK
                    if (!((dupmask >> k) & 1))
#endif
                    Lt16[h[k]] = (uint16_t)((uint32_t)tile + (uint32_t)(k * 64 + lane));
                    zd::wave_lds_order(); // stores precede the next group's lookups
                }
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) L.a0[k * 64 + lane] = e[k]; // the entries as they are: S3 turns them into distances
                zd::wave_priority<0>();
              }
            } else
            if (wave < 2 && !(dbg & 4)) {
                zd::wave_priority<3>(); // the other six waves wait for these two: go ahead of the co-resident workgroup
                uint32_t *tab = wave == 0 ? Ltl : Lts;
                uint32_t *hc = wave == 0 ? L.a0 : L.a1; // hashes in, candidates out
                uint32_t h[CHUNKS], e[CHUNKS];
                uint64_t dupmask = 0; // bit k: the next lane of group k has the same hash, so its insert supersedes mine
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) {
                    h[k] = hc[k * 64 + lane];
                    const uint32_t hn = lane < 63 ? hc[k * 64 + lane + 1] : 0xFFFFFFFFu;
                    if (hn == h[k]) dupmask |= 1ull << k;
                }
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) {
                    const bool act = h[k] != 0xFFFFFFFFu;
                    e[k] = act ? tab[h[k] >> TAG_BITS] : 0u;
                    zd::wave_lds_order(); // lookups of this 64-group precede its inserts
                    // runs of equal hashes (zero runs, repeated bytes) would serialise 64 atomics on one LDS word: only the
                    // last lane of a run inserts -- ds_max keeps the highest position anyway, so the table ends up identical
                    if (act && !((dupmask >> k) & 1)) {
                        const uint32_t code = ((uint32_t)(tile - segbase) + (uint32_t)(k * 64 + lane) + 1) << TAG_BITS;
                        atomicMax(&tab[h[k] >> TAG_BITS], code | (h[k] & TAG_MASK));
                    }
                    zd::wave_lds_order(); // inserts precede the next group's lookups
                }
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) // table entry with its check bits cancelled: a hit has zero low bits and is non-zero
                    hc[k * 64 + lane] = e[k] ^ (h[k] & TAG_MASK);
                zd::wave_priority<0>();
            }
            // far inserts of the PREVIOUS searched tile (and everything else this wave has in flight: this tile's far sources, needed next)
            // are complete before any wave asks for the next tile's far entries in S3
            if (NFAR) zd::wait_vmem();
            zd::lds_barrier();
            ZGE_PROF(3);
            // ---- S3: own candidates {long, short, 2 recent offsets}.  The kernel is bound by VALU issue, so the loads carry no
            // address arithmetic: the tile side and the recent-offset sources (always inside the staged window, see the
            // idx + rep_back rule) are LDS reads at a lane offset; hash candidates are global loads `frame base (SGPR) + position`.
            // Phase A requests the first 8 source bytes of every candidate of BOTH positions, phase B scores them and loads more
            // only for candidates that match 8 bytes. ----
            uint32_t near_e[PER] = {0, 0}; // NEAR16: the near table's entries of this thread's positions (S2 left them in a0)
            if (FAR_CDC) {
                // the NEXT tile's far entries, content-defined: every thread hashes its own two positions of that tile out of its window
                // in the other LDS buffer (written in S0, two barriers ago) and asks for the entry where the position is a far position;
                // the bucket | check bits stay in a register until that tile's inserts.  The requests have S3 to come back.
                // (all LDS reads of this stage -- the next tile's 12 bytes and the near table's entries of both positions -- go out together:
                // under their conditions each would be a round trip of its own)
                uint64_t vnx[PER];
                uint32_t wnx[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t pn = ntile + ZGE_IDX(u);
                    const uint32_t at = pn < far_end ? (uint32_t)(pn + wofs_n) : 0u; // no next position: any address inside the buffer
                    vnx[u] = zd::load_u64(tbn + at);
                    wnx[u] = zd::load_u32(tbn + at + 8);
                    if (NEAR16) near_e[u] = L.a0[ZGE_IDX(u)];
                }
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t pn = ntile + ZGE_IDX(u);
                    hfn[u] = 0xFFFFFFFFu;
                    fnext[u][0] = 0;
                    if (pn < far_end && !(dbg & (2048 | 32768))) {
                        const uint64_t vn = vnx[u];
                        const uint32_t wn = wnx[u];
                        const uint32_t h32f = hash_far32(vn, wn);
                        // (windows of period 1, 2 or 4 -- three equal words: runs, zero padding -- are no far positions: every position of such a stretch would hit one bucket)
                        if (((h32f >> (far_shift - FAR_CDC)) & cdc_mask) == 0 && !((uint32_t)vn == (uint32_t)(vn >> 32) && (uint32_t)vn == wn)) {
                            hfn[u] = h32f >> far_shift;
                            fnext[u][0] = zd::load_l2_u32(far_l + (size_t)(hfn[u] >> TAG_BITS));
                        }
                    }
                }
                pf_far_tile = ntile;
            } else
            if (NFAR) {
                // the NEXT tile's far entries: hashes out of its window in the other LDS buffer; requested here, behind two barriers since S0 wrote that buffer; the requests have S3 to come back
#pragma unroll
                for (int u = 0; u < FNEXT_ROWS; u++) {
                    const uint32_t idxn = FAR_COMPACT ? (uint32_t)wave * 128u + ((uint32_t)lane << FAR_RES_LOG) : ZGE_IDX(u);
                    const uint32_t pn = ntile + idxn;
                    const bool mine = (!FAR_COMPACT || (uint32_t)lane < (128u >> FAR_RES_LOG)) && !((dbg & 16384) && wave < 2) && !(dbg & 32768);
                    if (mine && pn < far_end && !(pn & far_rmask) && !(dbg & 2048)) {
                        const uint64_t v = zd::load_u64(tbn + (uint32_t)(pn + wofs_n));
                        const uint32_t hf = hash_far32(v, zd::load_u32(tbn + (uint32_t)(pn + 8 + wofs_n))) >> far_shift, hg = hash_short32(v, SHORT_BYTES) >> far_shift;
                        if (FAR_WAYS == 2) { // both ways of a bucket in one 8-byte request
                            const uint64_t el = zd::load_l2_u64(far_l + (size_t)(hf >> TAG_BITS) * 2);
                            fnext[u][0] = (uint32_t)el; fnext[u][1] = (uint32_t)(el >> 32);
                            if (FAR_SHORT) { const uint64_t es = zd::load_l2_u64(far_s + (size_t)(hg >> TAG_BITS) * 2); fnext[u][2] = (uint32_t)es; fnext[u][3] = (uint32_t)(es >> 32); }
                        } else {
#pragma unroll
                        for (int w = 0; w < FAR_WAYS; w++) {
                            fnext[u][w] = zd::load_l2_u32(far_l + ((size_t)(hf >> TAG_BITS) * FAR_WAYS + w));
                            if (FAR_SHORT) fnext[u][FAR_WAYS + w] = zd::load_l2_u32(far_s + ((size_t)(hg >> TAG_BITS) * FAR_WAYS + w));
                        }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < NFAR; k++) fnext[u][k] = 0;
                    }
                }
                pf_far_tile = ntile;
            }
            uint32_t mo[PER], mw[PER];
            uint32_t offs[PER][NTAB];
            U128 q16[PER];                    // near candidate: source[-8 .. 8)
            // NEAR16, runs: inside a repeat every position's 5-byte hash leads to the same distance, and then its match at that distance is
            // its left neighbour's minus one byte.  A position whose (validated) near distance equals its left neighbour's -- same
            // 64-position chunk, i.e. the lane below -- is a FOLLOWER: it requests no source and compares nothing, its near length is
            // handed down from the head of its run (the nearest lane below that is no follower) after the heads have compared.  What S3
            // costs is source requests per CU; in compressible data most positions are followers.
            bool fol[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t p = tile + idx;
                uint32_t cn = 0;
                fol[u] = false;
                if (NEAR16) { // S2 left the table entry: the candidate lies d = 1 .. 65536 bytes back, d = (p - entry) mod 2^16 (0 -> 65536)
                    const uint32_t d = ((p - (FAR_CDC ? near_e[u] : L.a0[idx]) - 1u) & 0xFFFFu) + 1u;
                    cn = (idx < tcount && p < hash_end && d <= p && !(dbg & (5 | 64))) ? p - d + 1 : 0u;
                } else
                if (idx < tcount && !(dbg & 5)) {
                    // candidate position (+1): a table hit whose check bits agreed.  One near candidate: the long table's when it has a
                    // hit, the short table's only otherwise (with a long-hash hit at hand the short-hash candidate changes 0.003 % of the
                    // output and would cost a source fetch per position)
                    const uint32_t w0 = L.a0[idx], w1 = L.a1[idx];
                    const uint32_t wn = (w0 && !(w0 & TAG_MASK)) ? w0 : w1;
                    cn = (wn && !(wn & TAG_MASK)) ? segbase + (wn >> TAG_BITS) : 0u;
                }
                uint32_t on = cn ? p - (cn - 1) : 0u;
                // the source needs 8 bytes in front of it (frame positions 0..7 are not used as sources)
                if (on + 8 > p || on > window || idx >= tcount || (dbg & 1)) on = 0;
                offs[u][0] = on;
                if (NEAR16) { const uint32_t below = zd::shfl_up1(on); fol[u] = lane > 0 && on != 0 && on == below && !(dbg & 16); }
                // one 16-byte request: source[-8 .. 0) for the backward extension, source[0 .. 8)
                q16[u] = U128{0, 0};
                if (on && !fol[u]) __builtin_memcpy(&q16[u], src + (p - on - 8), 16);
                if (NFAR && !far_ahead) { // entries asked for in S1 of this very tile: their sources can only be requested now
#pragma unroll
                    for (int k = 0; k < NFAR; k++) {
                        const uint32_t e = fe[u][k];
                        const uint32_t want = FAR_CDC ? (L.a1[idx] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (L.a1[idx] & TAG_MASK)) : ftag[u][k / (FAR_WAYS ? FAR_WAYS : 1)];
                        uint32_t o = (e && (e & TAG_MASK) == want && !(dbg & 4096)) ? p - (segbase + (e >> TAG_BITS) - 1) : 0u;
#pragma unroll
                        for (int j = 0; j < k; j++) if (o == foffs[u][j]) o = 0;
                        if (o + 8 > p || o > window || idx >= tcount || (dbg & 1)) o = 0;
                        foffs[u][k] = o;
                        if (o) qf[u][k] = zd::load_u64(src + (p - o));
                    }
                }
#pragma unroll
                // the near table has offered it already -- unless this position is a follower: its handed-down near length is the head's
                // minus the distance, and a head whose candidate was a hash collision at the very same distance hands down nothing
                for (int k = 0; k < NFAR; k++) offs[u][1 + k] = (foffs[u][k] == on && !fol[u]) ? 0u : foffs[u][k];
            }
            ZGE_PROF(9);
            // while those loads are in flight: the two recent-offset guesses of both positions.  Both sides are inside the staged
            // window (guesses are limited to idx + rep_back), so this is LDS-only work.  rres = length | (1 << 9 if the second
            // guess won), 0 = none; equal lengths keep the first.
            uint32_t rres[PER];
            {
                const bool use0 = erep0 != 0 && erep0 <= window;
                const bool use1 = erep1 != 0 && erep1 != erep0 && erep1 <= window;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    const uint32_t p = tile + idx;
                    const uint32_t limit = idx < tcount ? (uint32_t)(be - p) : 0u, cap = limit < cap_max ? limit : cap_max;
                    uint32_t res = 0;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t off = k == 0 ? erep0 : erep1;
                        if (!(k == 0 ? use0 : use1) || off > idx + (uint32_t)F_REP_BACK || off > p || idx >= tcount || (dbg & 1)) continue;
                        uint64_t x = zd::load_u64(tbb + (uint32_t)(p - off + wofs)) ^ p8[u];
                        uint32_t len = 0;
                        while (!x && len + 8 < cap) { // 16 bytes per LDS round trip (the second half stays inside the staged window)
                            const uint64_t a0 = zd::load_u64(tbb + (uint32_t)(p + len + 8 + wofs)), b0 = zd::load_u64(tbb + (uint32_t)(p - off + len + 8 + wofs));
                            const uint64_t a1 = zd::load_u64(tbb + (uint32_t)(p + len + 16 + wofs)), b1 = zd::load_u64(tbb + (uint32_t)(p - off + len + 16 + wofs));
                            len += 8;
                            x = a0 ^ b0;
                            if (!x && len + 8 < cap) { len += 8; x = a1 ^ b1; }
                        }
                        len += x ? (uint32_t)(zd::ctz64(x) >> 3) : 8u;
                        if (len > cap) len = cap;
                        if (len >= (uint32_t)F_MIN_REP && len > (res & 0x1FFu)) res = len | ((uint32_t)k << 9);
                    }
                    rres[u] = res;
                }
            }
            ZGE_PROF(10);
            // common prefix of position p (first 8 bytes p8v) with the source `off` bytes back whose first 8 bytes are `first8`, at most `cap`:
            // 8 bytes per step, 16 per global round trip; reads past `cap` stay inside the staged window / the padded arena
            auto prefix_len = [&](uint32_t p, uint64_t p8v, uint32_t cap, uint32_t off, uint64_t first8) -> uint32_t {
                uint64_t x = first8 ^ p8v;
                uint32_t len = 0;
                while (!x && len + 8 < cap) {
                    U128 sv;
                    __builtin_memcpy(&sv, src + (p - off + len + 8), 16);
                    const uint64_t a0 = zd::load_u64(tbb + (uint32_t)(p + len + 8 + wofs)), a1 = zd::load_u64(tbb + (uint32_t)(p + len + 16 + wofs));
                    len += 8;
                    x = a0 ^ sv.lo;
                    if (!x && len + 8 < cap) { len += 8; x = a1 ^ sv.hi; }
                }
                len += x ? (uint32_t)(zd::ctz64(x) >> 3) : 8u;
                return len > cap ? cap : len;
            };
            // FAR_MERGED (the level-3 finder): a position in 16 has a far candidate, so running its comparison inside the per-position
            // code below means two instances per wave executed for a lane or two each.  Instead the near candidate and the recent-offset
            // guesses are settled per position first, and ONE instance afterwards takes every lane's far candidate (of whichever of its
            // two positions has one; a lane with two goes round again).  The model's order is near, far, guesses with ties to the earlier:
            // merged last, the far candidate therefore wins a tie against a guess and loses one against the near candidate.
            constexpr bool FAR_MERGED = NEAR16 && FAR_CDC && NFAR == 1;
            uint32_t b_len[PER], b_off[PER], b_flags[PER], near_acc[PER]; // flags: 1 rep, 2 from a guess, 4 follower's near match, 8 far
            int32_t b_score[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t p = tile + idx;
                mo[u] = 0; mw[u] = 0;
                const uint32_t limit = idx < tcount ? (uint32_t)(be - p) : 0u, cap = limit < cap_max ? limit : cap_max;
                uint32_t best_len = 0, best_off = 0, best_flags = 0;
                int32_t best_score = -1000000;
                uint32_t near_len = 0; // NEAR16: the near candidate's length -- compared by the heads, handed down to the followers
                if (NEAR16) {
                    if (offs[u][0] && !fol[u]) near_len = prefix_len(p, p8[u], cap, offs[u][0], q16[u].hi);
                    const uint64_t heads = zd::ballot(!fol[u]);                                  // lane 0 is never a follower
                    const uint32_t hl = 63u - (uint32_t)__clzll((long long)(heads & (~0ull >> (63 - lane)))); // the nearest head at or below this lane
                    const uint32_t handed = zd::shfl(near_len, (int)hl);
                    if (fol[u]) { const uint32_t j = (uint32_t)lane - hl; near_len = handed > j ? handed - j : 0u; }
                }
                near_acc[u] = 0;
#pragma unroll
                for (int k = 0; k < NTAB; k++) {
                    if (FAR_MERGED && k >= 1) break;
                    const uint32_t off = offs[u][k];
                    if (!off) continue;
                    if (FAR_SKIP && k >= 1 && best_len >= (uint32_t)FAR_SKIP) continue; // with that many bytes in hand a far candidate is not looked at: it would be the tile's longest compare
                    const bool is_rep = off == erep0 || off == erep1;
                    const uint32_t len = (NEAR16 && k < 1) ? near_len : prefix_len(p, p8[u], cap, off, k < 1 ? q16[u].hi : qf[u][k < 1 ? 0 : k - 1]);
                    if (len < (uint32_t)(is_rep ? F_MIN_REP : P.min_match)) continue;
                    const int32_t sc = score_of(P, len, off, is_rep);
                    if (sc > best_score) { best_score = sc; best_len = len; best_off = off; best_flags = (is_rep ? 1u : 0u) | ((NEAR16 && k < 1 && fol[u]) ? 4u : 0u) | (k >= 1 ? 8u : 0u); if (k < 1) near_acc[u] = len; }
                }
                { // the recent-offset guesses rank after the table candidates (ties keep the earlier candidate)
                    const uint32_t r = rres[u];
                    if (r) {
                        const uint32_t len = r & 0x1FFu;
                        const int32_t sc = score_of(P, len, 1, true);
                        if (sc > best_score) { best_score = sc; best_len = len; best_off = (r >> 9) ? erep1 : erep0; best_flags = 1u | 2u; }
                    }
                }
                b_len[u] = best_len; b_off[u] = best_off; b_flags[u] = best_flags; b_score[u] = best_score;
            }
            if (FAR_MERGED) {
                uint32_t pend = (offs[0][1] ? 1u : 0u) | (offs[1][1] ? 2u : 0u);
                while (zd::ballot(pend != 0)) { // uniform; usually one round, rarely two
                    if (pend) {
                        const int sel = (pend & 1u) ? 0 : 1;
                        pend &= sel ? ~2u : ~1u;
                        const uint32_t idx = sel ? ZGE_IDX(1) : ZGE_IDX(0);
                        const uint32_t p = tile + idx;
                        const uint32_t limit = (uint32_t)(be - p), cap = limit < cap_max ? limit : cap_max; // (a far candidate exists only for idx < tcount)
                        const uint32_t off = sel ? offs[1][1] : offs[0][1];
                        const uint64_t pv = sel ? p8[1] : p8[0], first8 = sel ? qf[1][0] : qf[0][0];
                        const uint32_t nacc = sel ? near_acc[1] : near_acc[0];
                        if (!(FAR_SKIP && nacc >= (uint32_t)FAR_SKIP)) {
                            const bool is_rep = off == erep0 || off == erep1;
                            const uint32_t len = prefix_len(p, pv, cap, off, first8);
                            if (len >= (uint32_t)(is_rep ? F_MIN_REP : P.min_match)) {
                                const int32_t sc = score_of(P, len, off, is_rep);
                                const int32_t cur = sel ? b_score[1] : b_score[0];
                                const uint32_t cfl = sel ? b_flags[1] : b_flags[0];
                                if ((cfl & 2u) ? sc >= cur : sc > cur) {
                                    const uint32_t nf = (is_rep ? 1u : 0u) | 8u;
                                    if (sel) { b_score[1] = sc; b_len[1] = len; b_off[1] = off; b_flags[1] = nf; }
                                    else { b_score[0] = sc; b_len[0] = len; b_off[0] = off; b_flags[0] = nf; }
                                }
                            }
                        }
                    }
                }
            }
            if (CONT_CAP && cont) { // uniform.  The continuation guess (model: cont_cap; compared at the top of the tile): ranks last, recent-offset cost, no backward extension
                const uint32_t idx_c = pos - tile;
                if (wave == (int)(idx_c >> 7) && (uint32_t)lane == (idx_c & 63u)) {
                    const uint32_t clen = L.ctrl[K_CLEN];
                    if (clen >= (uint32_t)F_MIN_REP) {
                        const int32_t sc = score_of(P, clen, erep0, true);
                        if (idx_c & 64u) { if (sc > b_score[1]) { b_score[1] = sc; b_len[1] = clen; b_off[1] = erep0; b_flags[1] = 1u | 4u; } }
                        else { if (sc > b_score[0]) { b_score[0] = sc; b_len[0] = clen; b_off[0] = erep0; b_flags[0] = 1u | 4u; } }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t idx = ZGE_IDX(u);
                const uint32_t p = tile + idx;
                const uint32_t best_len = b_len[u], best_off = b_off[u];
                const int32_t best_score = b_score[u];
                const bool best_rep = b_flags[u] & 1u, from_guess = b_flags[u] & 2u, best_fol = b_flags[u] & 4u, best_far = b_flags[u] & 8u;
                if (best_len && best_score > 0) {
                    // backward-extension potential: equal bytes just before the match and its source (none next to the frame start).  Far
                    // candidates, found some way into a repeat, may go back FAR_BACK bytes.
                    const uint32_t q = p - best_off;
                    const uint32_t bcap = (NFAR && best_far) ? (uint32_t)FAR_BACK : (uint32_t)F_BACK_CAP;
                    uint32_t maxb = bcap;
                    if (p - bs < maxb) maxb = (uint32_t)(p - bs);
                    uint32_t back = 0;
                    if (maxb && q >= bcap && !best_fol) { // then p - bcap and (for a guess) q - 8 are inside the staged window
                        uint64_t best_before = q16[u].lo; // the 8 bytes in front of the near candidate's source came with its first request
                        if (from_guess) best_before = zd::load_u64(tbb + (uint32_t)(q - 8 + wofs));
                        if (NFAR && best_far) best_before = zd::load_u64(src + (q - 8));
                        const uint64_t x = zd::load_u64(tbb + (uint32_t)(p - 8 + wofs)) ^ best_before;
                        back = x ? (uint32_t)(__clzll((long long)x) >> 3) : 8u;
                        if (NFAR && FAR_BACK > 8 && best_far) { // further back 8 bytes at a time while everything so far was equal
#pragma unroll
                            for (uint32_t j = 16; j <= (uint32_t)FAR_BACK; j += 8) {
                                if (back == j - 8 && maxb > j - 8) {
                                    const uint64_t x2 = zd::load_u64(tbb + (uint32_t)(p - j + wofs)) ^ zd::load_u64(src + (q - j));
                                    back += x2 ? (uint32_t)(__clzll((long long)x2) >> 3) : 8u;
                                }
                            }
                        }
                        if (back > maxb) back = maxb;
                    }
                    mo[u] = best_off;
                    mw[u] = best_len | (back << 16) | ((best_rep ? 1u : 0u) << 24);
                }
            }
            ZGE_PROF(11);
            // own match -> a0 (each thread overwrites only the candidate slots it has just read itself: no barrier).  A match
            // fits one word: offsets stay below 2^21 (table positions restart every 2^seg_log <= 2^21 bytes, recent-offset
            // guesses are shorter still) and lengths below 2^10 (the continuation guess: LONG_CAP 960; every other candidate cap 256 + FAR_BACK).
#pragma unroll
            for (int u = 0; u < PER; u++) L.a0[ZGE_IDX(u)] = match_pack(mo[u], mw[u] & 0xFFFF, (mw[u] >> 24) & 1);
            if (mo[0] | mo[1]) L.ctrl[K_ANY] = 1; // benign race: every writer stores 1
            if (EXT_ROUND) { // a candidate that may reach the cap once adopted (length + backward extension): the tile looks at its selected matches after the parse
                const uint32_t r0 = (mw[0] & 0xFFFFu) + ((mw[0] >> 16) & 0xFFu), r1 = (mw[1] & 0xFFFFu) + ((mw[1] >> 16) & 0xFFu);
                if ((r0 > r1 ? r0 : r1) >= cap_max) L.ctrl[K_LONG] = 1;
            }
            uint32_t fo[PER], fw[PER]; // final match of each position: its own, or the one it adopted from a position behind it
            uint64_t msel[PER], mlit[PER];
            // S4 - S6 (zge_parse_round.h): once per tile; with REP_PASS once more after every round of the live recent-offset pass
            {
#define ZGE_FIRST 1
#include "zge_parse_round.h"
#undef ZGE_FIRST
            }
            if (EXT_ROUND && L.ctrl[K_LONG]) { // uniform, rare (model: rep_pass 1 with ext_cap, live_reps 0).  A SELECTED match that was cut at the cap goes on at
                // its offset, LONG_CAP bytes at most -- one trip of the whole wave per such match -- and the tile is propagated and parsed
                // once more: a long repeat MiB back, found at one sampled far position, stays one match instead of pieces that each have
                // to be found again (the GPU code objects: 1.21 -> 1.045 of libzstd -3).  Tiles without a candidate of cap length never come
                // here; of those that do, the ones without such a SELECTED match leave after one barrier.
                uint64_t extm[PER];
                bool any = false;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    const uint32_t flen = fw[u] & 0xFFFFu, lim = idx < tcount ? (uint32_t)(be - tile) - idx : 0u;
                    extm[u] = msel[u] & zd::ballot(flen >= cap_max && lim > flen);
                    any |= extm[u] != 0;
                }
                if (any && lane == 0) L.ctrl[K_LONG2] = 1;
                zd::lds_barrier();
                if (L.ctrl[K_LONG2]) {
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    const uint32_t p = tile + idx;
                    const uint32_t flen = fw[u] & 0xFFFFu;
                    uint64_t ext = extm[u];
                    while (ext) { // uniform
                        const uint32_t who = (uint32_t)zd::ctz64(ext);
                        ext &= ext - 1;
                        const uint32_t p_w = zd::readlane(p, who), len_w = zd::readlane(flen, who), off_w = zd::readlane(fo[u], who);
                        const uint32_t lim_w = (uint32_t)(be - p_w), maxl = lim_w < (uint32_t)LONG_CAP ? lim_w : (uint32_t)LONG_CAP;
                        if (maxl > len_w) {
                            const uint32_t e = wave_match_ext(src, p_w, off_w, len_w, maxl, lane);
                            if ((uint32_t)lane == who && e > len_w) { mo[u] = fo[u]; mw[u] = e | (fw[u] & (1u << 24)); }
                        }
                    }
                    L.a0[idx] = match_pack(mo[u], mw[u] & 0xFFFF, (mw[u] >> 24) & 1);
                    L.ex[idx] = 0;
                }
                zd::lds_barrier(); // own matches are in a0, the offers start empty
                {
#define ZGE_FIRST 0
#include "zge_parse_round.h"
#undef ZGE_FIRST
                }
                }
            }
            if (REP_PASS) for (uint32_t it = 0; it < (uint32_t)P.rep_pass; it++) {
            // ---- live recent offsets (level >= 9; model: matchfind_block, rep_pass).  libzstd's lazy parsers try the offsets of the
            // matches they took last at every position; the guesses of S3 are the offsets the PREVIOUS tile ended with.  With a parse
            // of the tile in hand, every position tries the last two different offsets of the selected matches in front of it as two
            // more candidates at recent-offset cost; then the tile is propagated and parsed again.  The walk left, per chunk, the
            // offset of its last selected match and the last one different from it: the state (live, live1) at a chunk's entry is a
            // scalar chain over the chunks before it, inside the chunk one scalar step per selected match. ----
            {
                const uint32_t wr = L.wrep[lane & 31];
                uint32_t sl = erep0, sl1 = erep1; // uniform
                for (int c = 0; c < wave * PER; c++) {
                    const uint32_t cl = zd::readlane(wr, (uint32_t)(2 * c)), cd = zd::readlane(wr, (uint32_t)(2 * c + 1));
                    if (cl) { if (cd) { sl1 = cd; sl = cl; } else if (cl != sl) { sl1 = sl; sl = cl; } }
                }
                uint32_t lv[PER], lv1[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    lv[u] = sl; lv1[u] = sl1;
                    uint64_t rem = msel[u];
                    while (rem) { // the selected matches of this chunk in order: lanes above one see its offset
                        const uint32_t q = (uint32_t)zd::ctz64(rem);
                        rem &= rem - 1;
                        const uint32_t o = zd::readlane(fo[u], q);
                        if (o != sl) { sl1 = sl; sl = o; }
                        if ((uint32_t)lane > q) { lv[u] = sl; lv1[u] = sl1; }
                    }
                }
                // only positions on the path (literals, selected matches) and the position after one are tried: a position inside a selected
                // match would compare the rest of that very match at its own offset
                uint64_t onpath[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) { const uint64_t mk = msel[u] | mlit[u]; onpath[u] = mk | (mk << 1); }
                U128 qa[PER][2];
                bool ok[PER][2];
                bool changed = false;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    const uint32_t p = tile + idx;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t lo = k ? lv1[u] : lv[u];
                        ok[u][k] = idx < tcount && lo != 0 && lo + 8 <= p && lo <= window && !(k && lv1[u] == lv[u]) && ((onpath[u] >> lane) & 1);
                        qa[u][k] = U128{0, 0};
                        if (ok[u][k]) __builtin_memcpy(&qa[u][k], src + (p - lo - 8), 16);
                    }
                }
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t idx = ZGE_IDX(u);
                    const uint32_t p = tile + idx;
                    const uint32_t limit = idx < tcount ? (uint32_t)(be - p) : 0u, cap = limit < cap_max ? limit : cap_max;
                    uint32_t olen = mw[u] & 0xFFFF;
                    int32_t cur = olen ? score_of(P, olen, mo[u], (mw[u] >> 24) & 1) : 0;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (!ok[u][k]) continue;
                        const uint32_t lo = k ? lv1[u] : lv[u];
                        const uint32_t len = prefix_len(p, p8[u], cap, lo, qa[u][k].hi);
                        if (len < (uint32_t)F_MIN_REP) continue;
                        const int32_t sc = score_of(P, len, lo, true);
                        if (sc > cur) {
                            cur = sc;
                            uint32_t maxb = (uint32_t)F_BACK_CAP;
                            if (p - bs < maxb) maxb = (uint32_t)(p - bs);
                            const uint64_t x = zd::load_u64(tbb + (uint32_t)(p - 8 + wofs)) ^ qa[u][k].lo;
                            uint32_t back = x ? (uint32_t)(__clzll((long long)x) >> 3) : 8u;
                            if (back > maxb) back = maxb;
                            mo[u] = lo;
                            mw[u] = len | (back << 16) | (1u << 24);
                            changed = true;
                        }
                    }
                    if (CONT_CAP) {
                        // a SELECTED match that was cut at the cap goes on at its offset, CONT_CAP bytes at most (model: the live pass): the pieces
                        // of a long repeat stay one match even where the next piece's position would not have found the offset again.  One trip of
                        // the whole wave per such match; none on most data.
                        const uint32_t flen = fw[u] & 0xFFFFu;
                        uint64_t ext = msel[u] & zd::ballot(flen >= cap_max);
                        while (ext) { // uniform
                            const uint32_t who = (uint32_t)zd::ctz64(ext);
                            ext &= ext - 1;
                            const uint32_t p_w = zd::readlane(p, who), len_w = zd::readlane(flen, who), off_w = zd::readlane(fo[u], who);
                            const uint32_t lim_w = (uint32_t)(be - p_w), maxl = lim_w < (uint32_t)CONT_CAP ? lim_w : (uint32_t)CONT_CAP;
                            if (maxl > len_w) {
                                const uint32_t e = wave_match_ext(src, p_w, off_w, len_w, maxl, lane);
                                if ((uint32_t)lane == who && e > len_w) { mo[u] = fo[u]; mw[u] = e | (fw[u] & (1u << 24)); changed = true; }
                            }
                        }
                    }
                    L.a0[idx] = match_pack(mo[u], mw[u] & 0xFFFF, (mw[u] >> 24) & 1);
                    L.ex[idx] = 0;
                }
                if (changed) L.ctrl[K_CHG + it] = 1; // benign race: every writer stores 1
            }
            zd::lds_barrier(); // own matches are in a0, the offers start empty
            ZGE_PROF(14); // (diagnostics: the live recent-offset pass)
            if (L.ctrl[K_CHG + it] == 0) break; // no position took a live offset: another parse (and every further round) would repeat the last one
            {
#define ZGE_FIRST 0
#include "zge_parse_round.h"
#undef ZGE_FIRST
            }
            }
            // (The far inserts of this tile, issued after S3, must be in L2 before the next lookups go out -- those of the tile after the
            // next, in the next tile's S3.  Every wave waits for its own just before the barrier in front of that S3, i.e. while the table
            // wave is in S2: two thirds of a tile later instead of a third, and in time the other waves spend at that barrier anyway.)
            // counts of the chunks before mine: a 16-lane scan of the packed per-chunk counts (every wave repeats it)
            uint32_t sel_total, lit_total, sel_before[PER], lit_before[PER];
            {
                static_assert(CHUNKS == 16, "the chunk counts are scanned inside one 16-lane row");
                const uint32_t cv = zd::row_scan_incl(lane < CHUNKS ? L.wcnt[lane] : 0u);
                const uint32_t tot = zd::readlane(cv, CHUNKS - 1);
                sel_total = tot >> 16;
                lit_total = tot & 0xFFFFu;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t incl = zd::readlane(cv, (uint32_t)(wave * PER + u));
                    sel_before[u] = (incl >> 16) - (uint32_t)__popcll(msel[u]);
                    lit_before[u] = (incl & 0xFFFFu) - (uint32_t)__popcll(mlit[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t my_sel_idx = sel_before[u] + (uint32_t)__popcll(msel[u] & lt);
                const uint32_t my_lit_idx = lit_before[u] + (uint32_t)__popcll(mlit[u] & lt);
                if (dbg & 32) continue;
                if ((msel[u] >> lane) & 1) {
                    // the literal position stands in for the literal length (difference of neighbours, taken in stage 2)
                    seq_out[nseq + my_sel_idx] = zge_pack_seq(lp + my_lit_idx, fw[u] & 0xFFFF, fo[u]);
                    // offset guesses for the next tile: the offsets of the last two matches selected so far (every thread
                    // took its copy of the old ones at the top of the tile, so they can be replaced in place)
                    if (my_sel_idx + 1 == sel_total) { L.ctrl[K_REP0] = fo[u]; if (sel_total == 1) L.ctrl[K_REP1] = erep0; if (CONT_CAP) L.ctrl[K_CEND] = (uint32_t)(tile - bs) + ZGE_IDX(u) + (fw[u] & 0xFFFF); }
                    if (my_sel_idx + 2 == sel_total) L.ctrl[K_REP1] = fo[u];
                }
                if ((mlit[u] >> lane) & 1) lit_out[lp + my_lit_idx] = (uint8_t)p8[u];
            }
            nseq += sel_total;
            lp += lit_total;
            ZGE_PROF(8);
        }
        if (tid == 0) { rec->nseq = nseq; rec->nlit = lp; }
        if (NFAR) zd::wait_vmem(); // the block's last far inserts: the slab may be cleared next (new segment, next frame)
        zd::lds_barrier();
    }
    } // next frame from the queue
    if ((dbg & 1024) && tid < 15) atomicAdd((unsigned long long *)(queue + 2) + tid, L.prof[tid]);
}
#undef score_of
#undef match_pack
#undef match_len
#undef match_rep

__global__ void __launch_bounds__(512, 4) zarc_zge_match(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                      const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, const uint32_t *__restrict__ units, uint32_t n_units,
                                                      const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                                      uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch, uint32_t *__restrict__ queue,
                                                      uint32_t *__restrict__ far_scratch)
{
    __shared__ MatchLds<15, true> L;
    zge_match_body<15, 5, 12, 16, 1, false, 5, 2, 64, 48, false, true, 4, 5, 0, false, 0, true>(L, P, src_base, src_off, src_len, order, units, n_units, block_prefix, blocks, seq_scratch, lit_scratch, queue, far_scratch);
}

// The fast finder (level 1 and the negative levels): the level-3 finder's 16-bit near table, follower runs and recent-offset guesses,
// and nothing behind them -- no far table (no slab, no requests), no lazy step (P.lazy 0), no extension round.
__global__ void __launch_bounds__(512, 4) zarc_zge_match_fast(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                           const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, const uint32_t *__restrict__ units, uint32_t n_units,
                                                           const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                                           uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch, uint32_t *__restrict__ queue,
                                                           uint32_t *__restrict__ far_scratch)
{
    __shared__ MatchLds<15, true> L;
    zge_match_body<15, 5, 12, 0, 0, false, 0, 0, 0, 8, false, true, 0, 5, 0, false, 0, false>(L, P, src_base, src_off, src_len, order, units, n_units, block_prefix, blocks, seq_scratch, lit_scratch, queue, far_scratch);
}

#ifdef ZARC_GPU_DIAG
// the same kernel with the ZARC_GPU_DBG switches (stage clocks, timing-only ablations): only in the diagnostic build of the
// library (make DIAG=1 -> libzarc_gpu_diag.so, used by tools/); the product library has no such code
__global__ void __launch_bounds__(512, 4) zarc_zge_match_diag(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                           const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, const uint32_t *__restrict__ units, uint32_t n_units,
                                                           const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                                           uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch, uint32_t *__restrict__ queue,
                                                           uint32_t *__restrict__ far_scratch)
{
    __shared__ MatchLds<15, true> L;
    zge_match_body<15, 5, 12, 16, 1, false, 5, 2, 64, 48, true, true, 4, 5, 0, false, 0, true>(L, P, src_base, src_off, src_len, order, units, n_units, block_prefix, blocks, seq_scratch, lit_scratch, queue, far_scratch);
}
#endif

__global__ void __launch_bounds__(512, 2) zarc_zge_match_deep(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                           const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, const uint32_t *__restrict__ units, uint32_t n_units,
                                                           const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                                           uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch, uint32_t *__restrict__ queue,
                                                           uint32_t *__restrict__ far_scratch)
{
    __shared__ MatchLds<13, false> L;
    zge_match_body<13, 4, 10, 16, 2, true, 1, 0, 0, 48, false, false, 0, 6, 5, true, LONG_CAP, false>(L, P, src_base, src_off, src_len, order, units, n_units, block_prefix, blocks, seq_scratch, lit_scratch, queue, far_scratch);
}

#ifdef ZARC_GPU_DIAG
__global__ void __launch_bounds__(512, 2) zarc_zge_match_deep_diag(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                           const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, const uint32_t *__restrict__ units, uint32_t n_units,
                                                           const uint64_t *__restrict__ block_prefix, ZgeBlock *__restrict__ blocks,
                                                           uint64_t *__restrict__ seq_scratch, uint8_t *__restrict__ lit_scratch, uint32_t *__restrict__ queue,
                                                           uint32_t *__restrict__ far_scratch)
{
    __shared__ MatchLds<13, false> L;
    zge_match_body<13, 4, 10, 16, 2, true, 1, 0, 0, 48, true, false, 0, 6, 5, true, LONG_CAP, false>(L, P, src_base, src_off, src_len, order, units, n_units, block_prefix, blocks, seq_scratch, lit_scratch, queue, far_scratch);
}
#endif
