/*
 * oracle/zge_model.h -- TEST INFRASTRUCTURE ONLY (see oracle.h, zstd_enc_model.c).
 * Parameters and records of the engine encoder's CPU model.
 */
#ifndef ZGE_MODEL_H
#define ZGE_MODEL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZGE_BLOCK (64 * 1024) /* zarc_kernels.h: ZARC_BLOCK -- the encoder's blocks (round 4: 64 KiB; the format allows 128) */
#define ZGE_SPLIT_MIN ((size_t)4 << 20) /* zarc_kernels.h: ZARC_SPLIT_MIN */
#define ZGE_MIN_HUF_LITERALS 64

typedef struct {
    int level;        /* informational; selects the defaults                                   */
    int checksum;     /* ZSTD_c_checksumFlag (crates/zarc-cli/src/pack.rs:227 always sets it)   */
    int window_log;   /* frames larger than 2^window_log are not single-segment                */
    int long_log, short_log, short_bytes;
    int tile, sub, cap;
    int min_match, min_rep, rep_search, back_cap, lazy, lazy_delta;
    int lit_cost, match_cost, rep_cost;
    int short_window_log; /* reach of the short-hash table (16 when its entries are u16) */
    int rep_back; /* recent-offset guesses must start at most this many bytes before the tile */
    int tag_bits, seg_log; /* hash check bits kept in each table entry; tables restart every 2^seg_log bytes */
    /* Far tables (HBM in the engine): same entry format as the near ones, but a tile's lookups see the inserts of all EARLIER tiles
     * only, and only every 2^far_step_log-th position is inserted (so an entry lives 2^far_step_log times longer; any repeat of
     * 8 + 2^far_step_log - 1 bytes still contains an inserted position).  Way w of a bucket keeps the most recent position from
     * tiles with (tile index mod far_ways) == w: candidates of different ages without any insertion order inside a tile. */
    int far_log;      /* 2^far_log buckets per far table; 0 = no far tables */
    int far_ways;     /* 1, 2 or 4 */
    int far_step_log; /* positions p with (p mod 2^far_step_log) < 2^far_res_log are inserted ...                              */
    int far_res_log;  /* ... and positions with p mod 2^far_res_log == 0 are looked up: of 2^far_res_log inserted neighbours
                         exactly one lands on a looked-up position, whatever the offset of the repeat                          */
    int far_short;    /* 1 = a second far table keyed by the short hash */
    int far_skip;     /* a far candidate is not compared once a candidate of this many bytes is in hand (0 = always compared)        */
    int far_back;     /* backward-extension cap of far candidates (they are found up to 2^far_step_log + 2^far_res_log - 2
                         positions into a repeat; with content-defined sampling 2^far_cdc_log positions on average) */
    /* Round 3, the level-3 finder: */
    int near16;       /* 1 = ONE near table of 2^short_log 16-bit entries keyed by the short hash (no long table, no check bits): an
                         entry is the low 16 bits of the position, so a candidate lies 1 .. 65536 bytes back (a stale entry is just a
                         candidate that fails its compare); the far table covers everything beyond                                     */
    int far_cdc_log;  /* > 0: content-defined sampling of the far table: a position is inserted AND looked up iff the far_cdc_log
                         bits of its 12-byte hash just below the bucket and check bits are zero (far_step_log / far_res_log unused) and its three
                         4-byte words are not all equal (no runs / periods 1, 2, 4: those are the near table's business): both
                         occurrences of a repeat sample the same relative positions, so a repeat is found if it contains one sample */
    int far_min_frame; /* frames of at most this many bytes do without the far table (the 16-bit near table reaches 64 KiB): the engine
                          then has no 256 KiB slab to clear per frame -- what a batch of small entries spent most of its time on */
    /* Round 3, the level >= 9 finder: */
    int rep_pass;     /* rounds of the live recent-offset pass per tile (zstd_enc_model.c: matchfind_block), 0 = none */
    int lazy2_delta;  /* > 0: a selected match also steps aside for one two bytes ahead that scores more than this much higher */
    /* Round 3, long matches: */
    int far_cap;      /* > cap: far candidates are compared over this many bytes (a match word holds lengths below 1024: 960 + far_back) */
    int cont_cap;     /* > 0: continuation guess over this many bytes at the parse cursor of a tile (zstd_enc_model.c: matchfind_block) */
    int ext_cap;      /* > 0: in a round of rep_pass, a SELECTED match of `cap` bytes or more goes on at its offset, up to this many bytes */
    int live_reps;    /* 1 = the rounds of rep_pass try the live recent offsets (level >= 9); 0 = they only continue long matches */
    /* Round 4: */
    int seq_repeat;   /* 1 = groups of eight blocks share one FSE table per sequence-code type (first block describes it, the others say
                         Repeat_Mode) when that costs at most 1/64 more bits than the blocks' own tables (zstd_enc_model.c: seq_plan_group);
                         the engine always does */
} zge_params;

typedef struct { uint32_t ll, ml, off, ofv; } zge_seq;

typedef struct {
    uint64_t seqs, rep_seqs, match_bytes, lit_bytes, lit_section, seq_section;
    uint32_t blk_raw, blk_rle, blk_comp, lit_raw, lit_rle, lit_huf, seq_mode[4];
} zge_stats;

size_t zge_bound(size_t n);
void zge_default_params(zge_params *P, int level);
int zge_encode_frame(const zge_params *P, const void *src, size_t n, void *dst, size_t cap,
                     size_t *out_len, zge_stats *st);
uint32_t zge_ll_code(uint32_t ll);
uint32_t zge_ml_code(uint32_t ml);

#ifdef __cplusplus
}
#endif
#endif
