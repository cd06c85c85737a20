// zarc_amd/csrc/zarc_kernels.h -- kernel entry points and the records they exchange through HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// per-frame status values; numerically identical to ZARC_GPU_FRAME_* in include/zarc_gpu.h
enum {
    ZARC_FRAME_OK = 0, ZARC_FRAME_CORRUPT = 1, ZARC_FRAME_CHECKSUM = 2, ZARC_FRAME_DIGEST = 3,
    ZARC_FRAME_DSTSIZE = 4, ZARC_FRAME_BAD_MAGIC = 5, ZARC_FRAME_UNSUPPORTED = 6, ZARC_FRAME_SRCSIZE = 7
};

constexpr uint32_t ZARC_BLOCK_MAX = 128 * 1024;      // Zstandard Block_Maximum_Size: what the decoder must take, what store-mode frames use
// The ENCODER's blocks are 64 KiB (round 4).  A block is the unit of everything that is serial in the format -- one Huffman table for its
// literals, one FSE state chain for its sequences -- so halving it doubles the decoder's parallelism (sequence and literal chains half as
// long) and lets the literal statistics follow the data (libzstd 1.5 splits blocks where they change: its frame of a libtorch slice has
// 112 blocks for 2 MiB).  Model, ours / libzstd -3, 128 -> 64 KiB: ELF tables 1.015 -> 0.999, package.json 1.033 -> 1.022, machine code
// 1.039 -> 1.033, the libtorch string tables 1.067 -> 1.049 (inside the contract), text +0.02 .. 0.1 %, GPU code objects 1.024 -> 1.038;
// 32 KiB: text +0.3 %, code objects 1.07.  The sequence tables are shared by groups of 16 blocks (1 MiB, as before).
constexpr uint32_t ZARC_BLOCK = 64 * 1024;
constexpr uint32_t ZARC_MAX_SEQ = ZARC_BLOCK / 3 + 8; // sequences per block (every match >= 3 bytes)

// ---- encoder tuning (mirrors oracle/zge_model.h zge_params; plain ints so the struct can be passed by value)
struct ZgeParams {
    int level, checksum, window_log, long_log, short_log, short_bytes, tile, sub, cap, min_match, min_rep, rep_search,
        back_cap, lazy, lazy_delta, lit_cost, match_cost, rep_cost, short_window_log, rep_back, tag_bits, seg_log,
        far_log, far_ways, far_step_log, far_res_log, far_short, far_skip, far_back, near16, far_cdc_log, far_min_frame, rep_pass, lazy2_delta, far_cap, cont_cap, ext_cap, live_reps, slot_bytes, dbg;
};
// Encoder scratch of one block slot (sequences, literals, coded block): sized by the largest block of the SUB-BATCH (slot_bytes <=
// ZARC_BLOCK, a multiple of 16) -- a batch of a million 1 KiB entries must not reserve 600 KiB per entry.
__host__ __device__ inline uint64_t zge_seq_stride(uint32_t slot_bytes) { return slot_bytes / 3 + 8; }   // sequences (8 bytes each)
__host__ __device__ inline uint64_t zge_lit_stride(uint32_t slot_bytes) { return (uint64_t)slot_bytes + 64; }
__host__ __device__ inline uint64_t zge_out_stride(uint32_t slot_bytes) { return (uint64_t)slot_bytes + 1024; }
// words of far-table scratch one match-finder workgroup needs (HBM): 2^far_log buckets x ways, once or twice
__host__ __device__ inline size_t zge_far_words(const ZgeParams &P) { return P.far_log ? (((size_t)P.far_ways << P.far_log) * (P.far_short ? 2 : 1)) : 0; }

// Per-block record written by the match finder and completed by the entropy coder.
struct ZgeBlock {
    uint32_t frame;      // entry index
    uint32_t index;      // block index inside the frame
    uint32_t src_len;    // uncompressed bytes in this block
    uint32_t nseq, nlit; // match finder output
    uint32_t type;       // 0 raw, 1 rle, 2 compressed (set by the entropy coder; rle by the match finder)
    uint32_t out_len;    // bytes of block content (without the 3-byte header)
    uint32_t pad;
};
// Sequence-table plan of one block (entropy stage, multi-block frames): written by pass 1 (code histograms, the block's own best table
// per type), settled per group of ZGE_TABLE_GROUP blocks by zarc_zge_plan (one shared table per type where that is cheap: the first
// block describes it, the others say Repeat_Mode), read by pass 2 (model: zstd_enc_model.c, seq_plan / seq_plan_group).
constexpr uint32_t ZGE_TABLE_GROUP = 16;
struct ZgePlanTable {
    int16_t norm[64];   // normalised counts of the table to code with (pass 1: the block's own choice; zarc_zge_plan: the group's)
    uint8_t desc[80];   // its FSE description (mode 2)
    uint32_t count[64]; // histogram of the block's codes of this type (zero beyond the largest code)
    uint32_t cost;      // the own choice's cost in 1/256 bit, description included (modes 0 and 2)
    uint8_t mode, al, nsym, dl, rle, pad[3]; // Symbol_Compression_Mode 0..3, accuracy log, symbols in norm[], description bytes, RLE symbol
};
struct ZgePlan {
    uint32_t active;     // 1 = pass 2 codes this block's sequences section (0: RLE / raw / no sequences: the block record is final)
    uint32_t nseq;       // sequences after joining
    uint32_t lsz;        // bytes of the literals section in front
    uint32_t extra_bits; // bits of all extra-bits fields (for the size bound)
    uint32_t guaranteed; // zarc_zge_plan: an upper bound of the coded size is below the raw size
    uint32_t pad[3];
    ZgePlanTable t[3];   // LL, OF, ML
};
static_assert(sizeof(ZgePlanTable) == 476 && sizeof(ZgePlan) == 32 + 3 * 476, "plan record layout");
// one sequence: ofv (28 bits) | ll << 28 (18 bits) | ml << 46 (18 bits)
__host__ __device__ inline uint64_t zge_pack_seq(uint32_t ll, uint32_t ml, uint32_t ofv) { return (uint64_t)ofv | ((uint64_t)ll << 28) | ((uint64_t)ml << 46); }
__host__ __device__ inline uint32_t zge_seq_ofv(uint64_t s) { return (uint32_t)(s & 0xFFFFFFFu); }
__host__ __device__ inline uint32_t zge_seq_ll(uint64_t s) { return (uint32_t)((s >> 28) & 0x3FFFFu); }
__host__ __device__ inline uint32_t zge_seq_ml(uint64_t s) { return (uint32_t)((s >> 46) & 0x3FFFFu); }

// Decoder: one slot per block of a frame (the slots of frame f start at slot_prefix[f]); filled by zarc_zdec_scan.
struct ZdecBlock {
    uint32_t frame;    // frame index
    uint32_t type;     // 0 raw, 1 RLE, 2 compressed; 0xFFFFFFFF = slot not used
    uint32_t payload;  // frame offset of the block content (after the 3-byte block header)
    uint32_t size;     // Block_Size field
    uint32_t nseq;     // compressed blocks: Number_of_Sequences
    uint32_t seq_hdr;  // compressed blocks with sequences: frame offset of the Symbol_Compression_Modes byte
    uint32_t state;    // set to 1 by zarc_zdec_seqs once the block's sequences are in the sequence scratch
    uint32_t rep[3];   // repeat-offset history after the block, symbolic (see ZDEC_REP_REF): absolute, or "history slot at block start minus delta"
    uint32_t lit_type; // compressed blocks: Literals_Block_Type (0 raw, 1 RLE, 2 Huffman, 3 treeless)
    uint32_t lit_len;  // regenerated size of the literals
    uint32_t lit_off;  // Huffman literals: frame offset of the section body (tree description, then the streams)
    uint32_t lit_comp; // Huffman literals: bytes of that body
    uint32_t lit_streams; // 1 or 4
    uint32_t pad[3];   // [0] reach: bytes in front of the block its matches read at most (ZDEC_REACH_UNKNOWN: anywhere); [1] bytes the block regenerates
};
static_assert(sizeof(ZdecBlock) == 72, "block slot layout");
constexpr int ZDEC_LDS_LANES = 16;  // active lanes (= block slots) per wave of zarc_zdec_seqs_lds: 16 x 2.5 KiB of tables in LDS
constexpr uint32_t ZDEC_LONG_NSEQ = 768; // a block with at least this many sequences is zarc_zdec_seqs_lds's when the shared-table kernel has not done it
constexpr int ZDEC_LONG_BUCKETS = 16;    // ... listed by sequence count in buckets of 1 024 (the last one: 15 360 and more)
constexpr int ZDEC_LIT_GROUP = 16;  // block slots per wave of zarc_zdec_literals (4 Huffman streams each)
// Fast-path sequences are stored with zge_pack_seq(); the offset field is already resolved against the repeat-offset history
// as far as the block alone allows: bit 17 of the literal-length field set = the offset is (history slot at block start) - delta,
// the offset field then holds slot | delta << 2; otherwise the offset field is the absolute offset.
constexpr uint32_t ZARC_SPLIT_MIN = 4u << 20; // encoder: frames larger than this are searched segment by segment (2 MiB) by different workgroups
constexpr uint32_t ZDEC_LL_REF = 1u << 17;
constexpr uint32_t ZDEC_MAX_DELTA = 1u << 20; // "history slot minus delta": repeated `first history entry - 1` codes add up (libzstd -9 .. -19 on records-like data)
// A run of blocks of one frame that reads nothing in front of its first block: the unit of work of the fast frame pass (engine.hip).
struct ZdecPiece {
    uint32_t frame, first, count; // frame index; first block; number of blocks (a whole frame: first 0, count 0xFFFFFFFF)
    uint32_t rep[3];              // repeat-offset history in front of the first block
    uint64_t out_start, out_len;  // where the piece's output starts inside the frame's, and how many bytes its blocks regenerate
};
constexpr uint32_t ZDEC_REACH_UNKNOWN = 0xFFFFFFFFu; // ZdecBlock.pad[0]: the block's matches may reach anywhere in front of it
constexpr uint32_t ZDEC_REP_REF = 0x80000000u; // same idea for ZdecBlock::rep[]: REF | slot | delta << 2
constexpr int ZDEC_TABLE_CELLS = 1280; // per block slot: LL 512 + ML 512 + OF 256 FSE decode cells (u16)

// ---- kernels -----------------------------------------------------------------------------------
__global__ void zarc_blake3_chunks(const uint8_t *base, const uint64_t *off, const uint64_t *len, const uint64_t *chunk_prefix,
                                   uint32_t n_entries, uint64_t total_chunks, uint32_t *cvs, uint32_t *digests);
__global__ void zarc_blake3_tree(const uint64_t *chunk_prefix, uint32_t n_entries, uint32_t *cvs, uint32_t *tmp, uint32_t *digests);
__global__ void zarc_xxh64(const uint8_t *base, const uint64_t *off, const uint64_t *len, uint32_t n_entries, uint64_t *out);
__global__ void zarc_zstd_decode(const uint8_t *frames_base, const uint64_t *frame_off, const uint64_t *frame_len, uint8_t *dst_base,
                                 const uint64_t *dst_off, const uint64_t *raw_len, const uint32_t *order, uint32_t n_frames,
                                 uint8_t *lit_scratch, int32_t *status, uint32_t *stored_checksum, int dbg, uint32_t *queue,
                                 const uint32_t *fast /* per frame: handled by zarc_zstd_frames; null = take every frame */);
// frame pass of the decoder fast path: frames whose sequences and literals were decoded ahead (fast[f] != 0)
__global__ void zarc_zstd_frames(const uint8_t *frames_base, const uint64_t *frame_off, const uint64_t *frame_len, uint8_t *dst_base,
                                 const uint64_t *dst_off, const uint64_t *raw_len, const ZdecPiece *pieces, uint32_t n_listed /* pieces[] holds this many */,
                                 uint32_t first_unlisted /* behind them: frames first_unlisted .. as one piece each */, uint32_t n_pieces, int32_t *status,
                                 uint32_t *stored_checksum, int dbg, uint32_t *queue, const uint32_t *fast, const uint64_t *slot_prefix,
                                 const ZdecBlock *zblocks, const uint64_t *seq_index, const uint64_t *seqs, const uint64_t *lit_index,
                                 const uint8_t *lits);
__global__ void zarc_gather(const uint8_t *src_base, const uint64_t *src_off, const uint64_t *len, const uint64_t *dense_off, uint32_t n, uint8_t *dst);
__global__ void zarc_zge_store(const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len, uint32_t n_frames, uint8_t *dst_base,
                               const uint64_t *dst_off, uint64_t *dst_len);
// exclusive prefix sum of in[i * stride] (at least `floor` each) -> out[0 .. n] and *total; one workgroup of 1024 threads
__global__ void zarc_scan_u32(const uint32_t *in, uint32_t stride, uint32_t floor_, uint64_t n, uint64_t *out, uint64_t *total);
// decoder fast path, stage 1: one LANE per frame walks the block headers (no payload is touched) -> block slots, nseq[], fast[]
__global__ void zarc_zdec_count(const uint8_t *frames_base, const uint64_t *frame_off, const uint64_t *frame_len, const uint64_t *raw_len, uint32_t n_frames,
                                uint32_t *nblocks);
__global__ void zarc_zdec_scan(const uint8_t *frames_base, const uint64_t *frame_off, const uint64_t *frame_len, const uint64_t *raw_len,
                               uint32_t n_frames, const uint64_t *slot_prefix, ZdecBlock *zblocks, uint32_t *counts /* per slot: nseq, Huffman literal bytes */,
                               uint32_t *fast);
// Huffman literals of the fast path: one wave per ZDEC_LIT_GROUP block slots (tables in LDS, one stream per lane) -> lits[]
__global__ void zarc_zdec_literals(const uint8_t *frames_base, const uint64_t *frame_off, uint64_t n_slots, const uint64_t *slot_prefix,
                                   const ZdecBlock *zblocks, const uint64_t *lit_index, uint8_t *lits, uint32_t *fast, uint64_t slot_base);
// stage 2, every lane's own tables in LDS (one wave of ZDEC_LDS_LANES active lanes per workgroup): the compressed blocks of at least
// ZDEC_LONG_NSEQ sequences that zarc_zdec_seqs_shared has not done, from a list ordered by sequence count (zarc_zdec_long_count / _fill);
// zarc_zdec_seqs (split_long) leaves exactly those alone
__global__ void zarc_zdec_long_count(const ZdecBlock *zblocks, uint64_t slot_base, uint64_t n_slots, const uint32_t *wave_flag, uint32_t *counters);
__global__ void zarc_zdec_long_fill(const ZdecBlock *zblocks, uint64_t slot_base, uint64_t n_slots, const uint32_t *wave_flag, uint32_t *counters,
                                    uint32_t *list);
__global__ void zarc_zdec_seqs_lds(const uint8_t *frames_base, const uint64_t *frame_off, uint64_t n_slots, const uint64_t *slot_prefix,
                                   ZdecBlock *zblocks, const uint64_t *seq_index, uint64_t *seqs, uint32_t *fast, uint64_t slot_base,
                                   const uint32_t *counters, const uint32_t *list);
// stage 2: one LANE per block slot entropy-decodes the block's sequences (FSE tables in HBM scratch) into seqs[]; with wave_flag only the
// 64-slot waves zarc_zdec_seqs_shared turned down
__global__ void zarc_zdec_seqs(const uint8_t *frames_base, const uint64_t *frame_off, uint64_t n_slots, const uint64_t *slot_prefix,
                               ZdecBlock *zblocks, const uint64_t *seq_index, uint64_t *seqs, uint16_t *tables, uint32_t *fast, uint64_t slot_base,
                               const uint32_t *wave_flag /* may be null */, const uint16_t *predef /* zarc_zdec_predef's tables, or null */,
                               int split_long /* 1: blocks with a long chain are zarc_zdec_seqs_lds's */);
constexpr int ZDEC_PREDEF_LL = 0, ZDEC_PREDEF_OF = 64, ZDEC_PREDEF_ML = 96, ZDEC_PREDEF_CELLS = 160;
__global__ void zarc_zdec_predef(uint16_t *out);
// stage 2 with the tables shared by a wave's 64 blocks in LDS (Repeat_Mode / equal descriptions); sets wave_flag[wave] where they do not fit
__global__ void zarc_zdec_seqs_shared(const uint8_t *frames_base, const uint64_t *frame_off, uint64_t n_slots, const uint64_t *slot_prefix,
                                      ZdecBlock *zblocks, const uint64_t *seq_index, uint64_t *seqs, uint32_t *fast, uint64_t slot_base,
                                      uint32_t *wave_flag);
// the same with 32 / 16 block slots per workgroup (a flag per workgroup; the launch of zarc_zdec_seqs behind it uses the same width)
__global__ void zarc_zdec_seqs_shared32(const uint8_t *frames_base, const uint64_t *frame_off, uint64_t n_slots, const uint64_t *slot_prefix,
                                        ZdecBlock *zblocks, const uint64_t *seq_index, uint64_t *seqs, uint32_t *fast, uint64_t slot_base,
                                        uint32_t *wave_flag);
__global__ void zarc_zdec_seqs_shared16(const uint8_t *frames_base, const uint64_t *frame_off, uint64_t n_slots, const uint64_t *slot_prefix,
                                        ZdecBlock *zblocks, const uint64_t *seq_index, uint64_t *seqs, uint32_t *fast, uint64_t slot_base,
                                        uint32_t *wave_flag);
// status[i]: keeps decode errors; else CHECKSUM if the stored XXH64 differs; else DIGEST if expect differs
__global__ void zarc_unpack_verdict(uint32_t n, const uint64_t *xxh, const uint32_t *stored_checksum, const uint32_t *digests,
                                    const uint32_t *expect /* may be null */, int32_t *status);
__global__ void zarc_corpus_fill(uint8_t *base, const uint64_t *off, const uint64_t *len, uint32_t n, uint64_t first_index, int kind);

// encoder
__global__ void zarc_zge_match(ZgeParams P, const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len,
                               const uint32_t *order, const uint32_t *units /* (queue slot, first block) per 2 MiB segment */, uint32_t n_units, const uint64_t *block_prefix, ZgeBlock *blocks,
                               uint64_t *seq_scratch, uint8_t *lit_scratch, uint32_t *queue, uint32_t *far_scratch);
// the fast finder (level 1 and below): the same near table, no far table, no lazy step, no extension round
__global__ void zarc_zge_match_fast(ZgeParams P, const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len,
                                    const uint32_t *order, const uint32_t *units, uint32_t n_units, const uint64_t *block_prefix, ZgeBlock *blocks,
                                    uint64_t *seq_scratch, uint8_t *lit_scratch, uint32_t *queue, uint32_t *far_scratch);
// the same with the ZARC_GPU_DBG switches (diagnostic build only)
__global__ void zarc_zge_match_diag(ZgeParams P, const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len,
                               const uint32_t *order, const uint32_t *units /* (queue slot, first block) per 2 MiB segment */, uint32_t n_units, const uint64_t *block_prefix, ZgeBlock *blocks,
                               uint64_t *seq_scratch, uint8_t *lit_scratch, uint32_t *queue, uint32_t *far_scratch);
// the deep finder (level >= 9): same arguments; two tagged tables, 4-byte short hash, two-way far tables, live recent-offset rounds
__global__ void zarc_zge_match_deep(ZgeParams P, const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len,
                                    const uint32_t *order, const uint32_t *units /* (queue slot, first block) per 2 MiB segment */, uint32_t n_units, const uint64_t *block_prefix, ZgeBlock *blocks,
                                    uint64_t *seq_scratch, uint8_t *lit_scratch, uint32_t *queue, uint32_t *far_scratch);
#ifdef ZARC_GPU_DIAG
__global__ void zarc_zge_match_deep_diag(ZgeParams P, const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len,
                                    const uint32_t *order, const uint32_t *units /* (queue slot, first block) per 2 MiB segment */, uint32_t n_units, const uint64_t *block_prefix, ZgeBlock *blocks,
                                    uint64_t *seq_scratch, uint8_t *lit_scratch, uint32_t *queue, uint32_t *far_scratch);
#endif
// entropy stage, everything in one pass (sub-batches without a multi-block frame: every block chooses its tables alone)
__global__ void zarc_zge_entropy(uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *blocks, uint64_t *seq_scratch, const uint8_t *lit_scratch,
                                 uint8_t *out_scratch, unsigned long long *prof);
// ... or in two passes around the table plan: literals + sequence pre-pass + histograms + own choices -> plans[]; the plan; sequences
__global__ void zarc_zge_entropy_p1(uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *blocks, uint64_t *seq_scratch, const uint8_t *lit_scratch,
                                    uint8_t *out_scratch, unsigned long long *prof, ZgePlan *plans);
__global__ void zarc_zge_plan(uint32_t n_blocks, const ZgeBlock *blocks, ZgePlan *plans, const uint32_t *group_start);
__global__ void zarc_zge_entropy_p2(uint32_t n_blocks, uint32_t slot_bytes, ZgeBlock *blocks, uint64_t *seq_scratch, const uint8_t *lit_scratch,
                                    uint8_t *out_scratch, unsigned long long *prof, ZgePlan *plans);
__global__ void zarc_zge_assemble(ZgeParams P, const uint8_t *src_base, const uint64_t *src_off, const uint64_t *src_len,
                                  const uint32_t *order, uint32_t n_frames, const uint64_t *block_prefix, const ZgeBlock *blocks, const uint8_t *out_scratch,
                                  const uint64_t *xxh, uint8_t *dst_base, const uint64_t *dst_off, uint64_t *dst_len);
