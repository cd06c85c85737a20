/*
 * zarc_gpu.h -- C ABI of the MI355X (gfx950) content engine for the Zarc archive format.
 *
 * This is the drop-in boundary for the ONE hot path of passcod/zarc: the per-entry content pipeline
 * (BLAKE3 content digest + Zstandard frame encode/decode + XXH64 frame checksum).  Every entry point
 * names the reference interface it replaces (paths relative to the reference tree):
 *
 *   pack   : Encoder::add_data_frame            crates/zarc/src/encode/content_frame.rs:20-60
 *            Encoder::write_compressed_frame    crates/zarc/src/encode/lowlevel_frames.rs:19-39
 *   unpack : Decoder::read_content_frame        crates/zarc/src/decode/frame_iterator.rs:14-27
 *            FrameIterator::{next,digest,verify} crates/zarc/src/decode/frame_iterator.rs:38-104
 *            ZstdFrameIterator::decompress_step crates/zarc/src/decode/zstd_iterator.rs:88-153
 *   digest : DigestType::verify_data            crates/zarc/src/integrity.rs:107-117
 *   ctx    : CCtx::try_create / init(0) / set_parameter / reset     crates/zarc/src/encode.rs:58-97
 *            DCtx::try_create                   crates/zarc/src/decode/zstd_iterator.rs:29
 *   errors : map_zstd_error / error::zstd       crates/zarc/src/lib.rs:27-30, decode/error.rs:35-38
 *
 * The reference processes one entry per call on one CPU thread; the engine takes a *batch* of entries
 * (frames are independent by construction: a fresh session per frame, content_frame.rs:37-39, and a
 * fresh DCtx per frame, zstd_iterator.rs:28-29) and runs them concurrently on one GPU.  Host plumbing
 * (dedup on digest, running offsets, directory, trailer) stays with the caller -- see INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; 0 = success, negative = ZARC_GPU_E_*; nothing throws or
 * aborts across this boundary; the callee never frees caller memory.  A handle is single-owner (one
 * host thread at a time), several handles may coexist.  There is NO CPU fallback: without a usable
 * HIP device zarc_gpu_create fails with ZARC_GPU_E_DEVICE.
 */
#ifndef ZARC_GPU_H
#define ZARC_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZARC_GPU_ABI_VERSION 2 /* 2: round 3's entry points (device_count, *_dedup, FRAME_DUPLICATE, PX_ZERO_COPY) + round 4's (warnings, levels) */
#define ZARC_GPU_DIGEST_LEN 32  /* DigestType::digest_len(), crates/zarc/src/integrity.rs:100-104 */
#define ZARC_GPU_ALIGN 16       /* device-resident entries / outputs must start 16-byte aligned   */
#define ZARC_GPU_PAD 64         /* readable slack required after the last byte of a device arena   */

typedef struct zarc_gpu zarc_gpu_t;

/* call-level errors (return values) */
enum {
    ZARC_GPU_OK = 0,
    ZARC_GPU_E_DEVICE = -1,      /* HIP error or no device; text via zarc_gpu_last_error()          */
    ZARC_GPU_E_NOMEM = -2,       /* "failed allocating zstd context" analogue (encode.rs:61)        */
    ZARC_GPU_E_PARAM = -3,       /* bad argument / parameter out of bounds                          */
    ZARC_GPU_E_UNSUPPORTED = -4, /* parameter accepted by libzstd but not by this engine            */
    ZARC_GPU_E_DSTSIZE = -5      /* dst_cap smaller than the sum of zarc_gpu_bound() of the entries */
};

/* per-frame status values (status[i]); names follow ZSTD_getErrorName where one exists */
enum {
    ZARC_GPU_FRAME_OK = 0,
    ZARC_GPU_FRAME_CORRUPT = 1,        /* "Data corruption detected"                                */
    ZARC_GPU_FRAME_CHECKSUM = 2,       /* "Restored data doesn't match checksum" (XXH64 mismatch)    */
    ZARC_GPU_FRAME_DIGEST = 3,         /* BLAKE3 != expected: REPORTED, not fatal (unpack.rs:118-120)*/
    ZARC_GPU_FRAME_DSTSIZE = 4,        /* "Destination buffer is too small"                          */
    ZARC_GPU_FRAME_BAD_MAGIC = 5,      /* "Unknown frame descriptor"                                 */
    ZARC_GPU_FRAME_UNSUPPORTED = 6,    /* dictionary id / window beyond the engine limit             */
    ZARC_GPU_FRAME_SRCSIZE = 7,        /* "Src size is incorrect" (frame shorter/longer than given)  */
    ZARC_GPU_FRAME_DUPLICATE = 8       /* pack, hash-first entry points only: content already known to the caller, nothing was compressed
                                          ("frame already exists, skipping", content_frame.rs:30-33); not an error                       */
};

/* Parameter ids are libzstd's ZSTD_cParameter values, which is what zstd_safe::CParameter maps to and
 * what `Encoder::set_zstd_parameter` (encode.rs:84-89) and `--zstd PARAM=VALUE` (zarc-cli/src/pack.rs:
 * 86-217) carry. */
enum {
    ZARC_GPU_P_COMPRESSION_LEVEL = 100,
    ZARC_GPU_P_WINDOW_LOG = 101,
    ZARC_GPU_P_HASH_LOG = 102,
    ZARC_GPU_P_CHAIN_LOG = 103,
    ZARC_GPU_P_SEARCH_LOG = 104,
    ZARC_GPU_P_MIN_MATCH = 105,
    ZARC_GPU_P_TARGET_LENGTH = 106,
    ZARC_GPU_P_STRATEGY = 107,
    ZARC_GPU_P_ENABLE_LDM = 160,
    ZARC_GPU_P_LDM_HASH_LOG = 161,
    ZARC_GPU_P_LDM_MIN_MATCH = 162,
    ZARC_GPU_P_LDM_BUCKET_SIZE_LOG = 163,
    ZARC_GPU_P_LDM_HASH_RATE_LOG = 164,
    ZARC_GPU_P_CONTENT_SIZE_FLAG = 200,
    ZARC_GPU_P_CHECKSUM_FLAG = 201,
    ZARC_GPU_P_DICT_ID_FLAG = 202,
    /* Engine tuning, not libzstd ids: how a batch is cut up, never what bytes come out (frames are identical for every value).
     * The library reads NO environment variable; these are the only switches. */
    ZARC_GPU_PX_SCRATCH_MB = 9001,   /* scratch budget in MiB (0 = encoder: up to 64 GiB / 45 % of free HBM; decoder: what the device has): batches beyond it run as sub-batches */
    ZARC_GPU_PX_STAGE_CHUNK = 9002,  /* host-pointer entry points: content bytes per staged chunk (0 = 2 GiB pack / 4 GiB unpack; >= 4096)         */
    ZARC_GPU_PX_STAGE_THREAD = 9003, /* 1 (default) = a helper thread moves neighbouring chunks over PCIe while the kernels run                     */
    ZARC_GPU_PX_COPY_THREADS = 9004, /* host threads that fill / drain the pinned staging ring (default 8)                                          */
    ZARC_GPU_PX_DEC_GROUPS = 9005,   /* unpack: frames are dealt by descending size into this many groups whose stages overlap (1..4; 0 = by the
                                        batch: 2 when its largest frame has 4 MiB and more and four times the mean size, else 1)                                                    */
    ZARC_GPU_PX_ZERO_COPY = 9006     /* host-pointer entry points: when every buffer of a chunk is page-locked memory the device can reach (hipHostMalloc /
                                        hipHostRegister by the caller, same HIP runtime) AND the buffers form runs -- contiguous in the caller's memory and in
                                        batch order -- of at least this many KiB on average, the DMA engines move them directly and the staging ring with its
                                        host memcpy pass is skipped.  Default 4096 (a DMA per small scattered buffer is slower than the ring); 0 = always stage.
                                        Ordinary (pageable) buffers are staged either way. */
};
/* What the engine does with the libzstd ids (pack.rs:86-217 forwards them all):
 *   CompressionLevel  -131072..22 accepted; four finders (zarc_gpu_level_finder says which one a level runs):
 *                     <= 1 (level 1 and the negative levels): the FAST finder -- one LDS table of 2^15 16-bit entries on a 5-byte hash,
 *                     candidates up to 64 KiB back, recent-offset guesses; no far table, no lazy step;
 *                     2..8 (0 = default = 3): the level-3 finder -- the same table plus a 2^16-bucket far table in HBM on a 12-byte hash,
 *                     content-sampled one position in 16, one-byte lazy evaluation, long matches continued in an extension round;
 *                     9..14: the deep finder -- two tagged LDS tables, 4-byte short hash, 2-way far tables on both hashes, a parse that
 *                     tries the live repeat offsets in two more rounds per tile and looks two bytes ahead;
 *                     15..22: the deep finder with four such rounds.  Levels inside one tier give identical frames.
 *   WindowLog         honoured for the frame header / the farthest offset (10..27; default 21, level >= 9: 22).
 *   MinMatch          4..7 honoured (3 is raised to 4); default 5, level >= 9: 4.
 *   HashLog, ChainLog, SearchLog, TargetLength, Strategy
 *                     accepted inside libzstd's bounds (6..30, 6..30, 1..30, 0..131072, 1..9; 0 = default), returned by
 *                     zarc_gpu_get_params, and ADVISORY: table sizes and the search are fixed by the kernels, so the frames are the
 *                     level's frames whatever these say (libzstd would search harder or less hard; the frames are valid either way).
 *   ContentSizeFlag   only 1.  ChecksumFlag honoured.  DictIdFlag accepted (zarc has no dictionaries).
 *   EnableLongDistanceMatching, LdmHashLog, LdmMinMatch, LdmBucketSizeLog, LdmHashRateLog (160-164)
 *                     accepted inside libzstd's bounds and ADVISORY as well: the far tables of the finders are the engine's long-distance
 *                     matcher at every level above 1, whatever these say.
 *   NbWorkers / JobSize / OverlapLog (400-402), experimental ids: ZARC_GPU_E_UNSUPPORTED.
 * zarc_gpu_parameter_advisory(id) tells a caller (the CLI prints a warning) that an id is accepted but changes nothing. */

typedef struct {
    int level;             /* 0 => default 3 (encode.rs:62 init(0))                                  */
    int checksum_flag;     /* the CLI always sets 1 (zarc-cli/src/pack.rs:227)                       */
    int content_size_flag; /* 1: frame content size is always written (one-shot compress2 behaviour) */
    int window_log;        /* 0 => level default (21 at level 3)                                     */
    int hash_log, chain_log, search_log, min_match, target_length, strategy; /* 0 => engine default  */
    int compress;          /* Encoder::enable_compression (encode.rs:95-97); 0 => raw-block frames   */
} zarc_gpu_params;

/* ---- context -------------------------------------------------------------------------------- */
/* CCtx::try_create + init(0) / DCtx::try_create.  device = HIP device ordinal. */
int zarc_gpu_create(zarc_gpu_t **out, int device);
/* Number of usable HIP devices (0 when there is none or the runtime fails): what `--gpus N` is checked against.  Frames are
 * independent, so a caller that wants G devices opens G handles and deals its batch itself. */
int zarc_gpu_device_count(void);
void zarc_gpu_destroy(zarc_gpu_t *h);
/* Sticky across batches, like Encoder::set_zstd_parameter.  Unknown ids -> ZARC_GPU_E_PARAM; ids libzstd
 * knows but the engine ignores (LDM, NbWorkers, ...) -> ZARC_GPU_E_UNSUPPORTED. */
int zarc_gpu_set_parameter(zarc_gpu_t *h, int param_id, int value);
void zarc_gpu_get_params(const zarc_gpu_t *h, zarc_gpu_params *out);
/* Encoder::enable_compression */
void zarc_gpu_enable_compression(zarc_gpu_t *h, int compress);
/* Worst-case frame size for an n-byte entry (all raw blocks): n + 3*max(1,ceil(n/65536)) + 18,
 * rounded up to ZARC_GPU_ALIGN.  The reference's Vec capacity rule (lowlevel_frames.rs:21) is smaller
 * than libzstd's own bound; this one never fails. */
size_t zarc_gpu_bound(size_t n);
const char *zarc_gpu_error_name(int code);         /* call-level codes (negative) */
const char *zarc_gpu_frame_status_name(int status); /* per-frame status (>= 0)     */
const char *zarc_gpu_last_error(const zarc_gpu_t *h);
int zarc_gpu_abi_version(void);
/* The level whose finder `level` runs: 1 (level <= 1), 3 (2..8 and 0), 9 (9..14) or 15 (15..22).  `zarc pack --level 19` warns that it packs
 * with the level-15 finder (zarc-cli/src/pack.rs:24-33 accepts -131072..22 and libzstd has a parameter set for each). */
int zarc_gpu_level_finder(int level);
/* 1 = the parameter id is accepted and remembered but ADVISORY (search-effort and long-distance-matching hints, pack.rs:86-217): the
 * frames are the level's frames whatever its value.  0 = honoured, or not accepted at all. */
int zarc_gpu_parameter_advisory(int id);

/* ---- pack: BLAKE3 + Zstandard frame encode (+ XXH64) ------------------------------------------- */
/* Host-memory form (mirrors add_data_frame's `&[u8]` input).  For every entry i the engine writes one
 * complete Zstandard frame at dst + dst_off[i] of dst_len[i] bytes (the value the caller adds to
 * Encoder.offset and stores as Frame.length, content_frame.rs:45-57) and digest[i] (Frame.digest).
 * dst_off[] are slot starts (slot i has room for zarc_gpu_bound(src_len[i])); frames are concatenable in
 * index order.  Dedup (content_frame.rs:30-33) is the caller's: compare digest[], drop later duplicates. */
int zarc_gpu_pack_batch(zarc_gpu_t *h, size_t n, const void *const *src, const size_t *src_len,
                        void *dst, size_t dst_cap, size_t *dst_off, size_t *dst_len,
                        uint8_t (*digest)[ZARC_GPU_DIGEST_LEN], int *status);

/* Device-memory form: d_src_base/d_dst are device pointers, the small per-entry arrays stay on the
 * host.  src_off[i] must be multiples of ZARC_GPU_ALIGN and the arena must extend ZARC_GPU_PAD bytes
 * past the last entry.  Outputs as above (dst_off/dst_len/digest/status are host arrays). */
int zarc_gpu_pack_batch_device(zarc_gpu_t *h, size_t n, const void *d_src_base, const uint64_t *src_off,
                               const uint64_t *src_len, void *d_dst, size_t dst_cap, uint64_t *dst_off,
                               uint64_t *dst_len, uint8_t *digest /* n*32 */, int *status);

/* Hash first, like the reference (content_frame.rs:26-33 hashes, looks the digest up and returns before compressing known content):
 * every entry is digested, then `known(ctx, digest, i)` is called once per entry in index order on the calling thread; a nonzero
 * return skips the entry (status[i] = ZARC_GPU_FRAME_DUPLICATE, dst_len[i] = 0, digest[i] set, nothing compressed).  A callback that
 * returns 0 should remember the digest so that a later copy inside the same batch is skipped too: first occurrence wins.  known == NULL
 * is zarc_gpu_pack_batch.  dst_off[i] is meaningful for compressed entries only.  The digest pass costs about 0.5 ms per GiB where
 * the match finder costs 20: on content that repeats (the reference's own benchmark tree is half duplicates) the saving is the
 * duplicates' whole share. */
typedef int (*zarc_gpu_known_fn)(void *ctx, const uint8_t digest[ZARC_GPU_DIGEST_LEN], size_t index);
int zarc_gpu_pack_batch_dedup(zarc_gpu_t *h, size_t n, const void *const *src, const size_t *src_len,
                              void *dst, size_t dst_cap, size_t *dst_off, size_t *dst_len,
                              uint8_t (*digest)[ZARC_GPU_DIGEST_LEN], int *status, zarc_gpu_known_fn known, void *ctx);
int zarc_gpu_pack_batch_device_dedup(zarc_gpu_t *h, size_t n, const void *d_src_base, const uint64_t *src_off,
                                     const uint64_t *src_len, void *d_dst, size_t dst_cap, uint64_t *dst_off,
                                     uint64_t *dst_len, uint8_t *digest /* n*32 */, int *status,
                                     zarc_gpu_known_fn known, void *ctx);

/* ---- unpack: Zstandard frame decode (+ XXH64 verify) + BLAKE3 verify --------------------------- */
/* frame[i]/frame_len[i] = Frame.offset/.length slice of the archive, raw_len[i] = Frame.uncompressed
 * (crates/zarc/src/directory/frame.rs:17-31), dst[i] = buffer of raw_len[i] bytes.  expect may be NULL;
 * when given, a mismatch sets status[i] = ZARC_GPU_FRAME_DIGEST but the bytes are still delivered
 * (unpack.rs:118-120).  digest[i] always receives the BLAKE3 of what was decoded. */
int zarc_gpu_unpack_batch(zarc_gpu_t *h, size_t n, const void *const *frame, const size_t *frame_len,
                          const size_t *raw_len, void *const *dst,
                          const uint8_t (*expect)[ZARC_GPU_DIGEST_LEN],
                          uint8_t (*digest)[ZARC_GPU_DIGEST_LEN], int *status);

int zarc_gpu_unpack_batch_device(zarc_gpu_t *h, size_t n, const void *d_frames_base, const uint64_t *frame_off,
                                 const uint64_t *frame_len, void *d_dst_base, const uint64_t *dst_off,
                                 const uint64_t *raw_len, const uint8_t *expect /* n*32 or NULL */,
                                 uint8_t *digest /* n*32 */, int *status);

/* ---- digest only (DigestType::verify_data, integrity.rs:107-117) ------------------------------- */
int zarc_gpu_blake3_batch(zarc_gpu_t *h, size_t n, const void *const *src, const size_t *len,
                          uint8_t (*digest)[ZARC_GPU_DIGEST_LEN]);
int zarc_gpu_blake3_batch_device(zarc_gpu_t *h, size_t n, const void *d_base, const uint64_t *off,
                                 const uint64_t *len, uint8_t *digest /* n*32 */);
/* XXH64(seed 0) of each entry (what libzstd appends/verifies as the frame checksum). */
int zarc_gpu_xxh64_batch_device(zarc_gpu_t *h, size_t n, const void *d_base, const uint64_t *off,
                                const uint64_t *len, uint64_t *out);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------- */
/* Device time of the kernels of the most recent batch call, measured with HIP events on the engine's
 * own stream.  `which` selects a kernel; returns milliseconds (<0 if not run). */
enum {
    ZARC_GPU_T_BLAKE3 = 0,    /* chunk + tree kernels                  */
    ZARC_GPU_T_XXH64 = 1,
    ZARC_GPU_T_MATCH = 2,     /* encoder: match finder                 */
    ZARC_GPU_T_ENTROPY = 3,   /* encoder: Huffman/FSE block coder      */
    ZARC_GPU_T_ASSEMBLE = 4,  /* encoder: frame assembly               */
    ZARC_GPU_T_DECODE = 5,    /* decoder                               */
    ZARC_GPU_T_TOTAL = 6,     /* pack: setup + match + entropy + assembly (the digest and the checksum run on side streams beside them and
                                 have their own entries); unpack / digest-only calls: first launch .. last launch                       */
    ZARC_GPU_T_DEC_SEQS = 7,  /* decoder stage 2: sequence entropy decoding (zarc_zdec_seqs)             */
    ZARC_GPU_T_DEC_LITS = 8,  /* decoder stage 2: Huffman literals (zarc_zdec_literals, side stream)      */
    ZARC_GPU_T_DEC_FRAMES = 9,/* decoder frame pass (zarc_zstd_frames + the inline decoder for the rest)  */
    ZARC_GPU_T_COUNT = 10
};
float zarc_gpu_last_kernel_ms(const zarc_gpu_t *h, int which);
/* Fill a device buffer with entries of the synthetic corpus; entry i of the
 * call is corpus entry first_index+i, kind = index mod 4 when kind < 0. */
int zarc_gpu_corpus_fill_device(zarc_gpu_t *h, size_t n, void *d_base, const uint64_t *off,
                                const uint64_t *len, uint64_t first_index, int kind);
/* Thin wrappers so that a ctypes-only caller can manage device memory without torch. */
int zarc_gpu_device_malloc(zarc_gpu_t *h, void **d_ptr, size_t bytes);
int zarc_gpu_device_free(zarc_gpu_t *h, void *d_ptr);
int zarc_gpu_memcpy_h2d(zarc_gpu_t *h, void *d_dst, const void *src, size_t bytes);
int zarc_gpu_memcpy_d2h(zarc_gpu_t *h, void *dst, const void *d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
