/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the arithmetic on zarc's per-entry content pipeline
 * (SURVEY.md section 8).  The reference (passcod/zarc, Rust) delegates this arithmetic to
 * third-party crates that are NOT vendored under /root/reference:
 *     blake3 1.5.0                 (Cargo.lock:191-192)
 *     zstd-sys 2.0.9+zstd.1.5.5    (Cargo.lock:2480-2481)   [libzstd 1.5.5, includes XXH64]
 * so these files restate the *published* algorithms (BLAKE3 paper, xxHash spec, RFC 8878) and are
 * anchored on the reference's call sites:
 *     blake3::hash(content)                   crates/zarc/src/encode/content_frame.rs:26
 *     DigestType::verify_data                 crates/zarc/src/integrity.rs:107-117
 *     blake3::Hasher::{update,finalize}       crates/zarc/src/decode/frame_iterator.rs:54,99,77
 *     CCtx::compress2 (+ XXH64 checksum)      crates/zarc/src/encode/lowlevel_frames.rs:29-31
 *     DCtx::decompress_stream                 crates/zarc/src/decode/zstd_iterator.rs:104-107
 *
 * PARITY PINNING: the reference holds no golden vectors for this path (its only test is
 * crates/zarc-cli/src/args.rs:86-90).  The oracle is instead pinned against
 *   - BLAKE3: the published known-answer vectors (tests/golden/blake3_kat.json),
 *   - XXH64 : python-xxhash / libxxhash on this image (tests/golden/xxh64_kat.json),
 *   - zstd  : frames produced by real libzstd builds (1.4.8 / 1.4.9 / 1.5.7 on this image) decoded
 *             by oracle_zstd_decode and compared with the original bytes
 *             (tests/golden/zstd_frames/), and frames produced by the engine decoded by libzstd.
 * The pinned crate versions themselves cannot run here (no Rust toolchain): with respect to
 * *those exact binaries* parity is unpinned; with respect to the algorithms they implement it is
 * pinned as above.
 *
 * Nothing under zarc_amd/ (the product) may include, link or call this code.
 */
#ifndef ZARC_ORACLE_H
#define ZARC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BLAKE3 (default hash mode, 32-byte output) ------------------------------------------- */
typedef struct {
    uint32_t cv_stack[54 * 8]; /* subtree chaining values, one per set bit of chunks-so-far      */
    int      cv_stack_len;
    uint64_t chunk_counter;    /* index of the chunk currently being filled                       */
    uint32_t chunk_cv[8];      /* chaining value inside the current chunk                         */
    uint8_t  block[64];        /* partial block buffer                                            */
    int      block_len;        /* bytes in `block`                                                */
    int      blocks_done;      /* blocks compressed in current chunk (0..15)                      */
} oracle_blake3_hasher;

void oracle_blake3_init(oracle_blake3_hasher *h);
void oracle_blake3_update(oracle_blake3_hasher *h, const void *data, size_t len);
void oracle_blake3_finalize(const oracle_blake3_hasher *h, uint8_t out[32]);
/* one-shot, mirrors blake3::hash(&[u8]) at content_frame.rs:26 */
void oracle_blake3(const void *data, size_t len, uint8_t out[32]);

/* ---- XXH64 (seed as given; zstd uses seed 0 and keeps the low 32 bits) ---------------------- */
uint64_t oracle_xxh64(const void *data, size_t len, uint64_t seed);

/* ---- Zstandard frame decoder (RFC 8878) ----------------------------------------------------- */
enum {
    ORACLE_ZSTD_OK = 0,
    ORACLE_ZSTD_E_TRUNCATED = -1,  /* input ended early                                            */
    ORACLE_ZSTD_E_MAGIC = -2,      /* not a zstd frame                                             */
    ORACLE_ZSTD_E_CORRUPT = -3,    /* malformed block / table / bitstream                          */
    ORACLE_ZSTD_E_DSTSIZE = -4,    /* output does not fit                                          */
    ORACLE_ZSTD_E_CHECKSUM = -5,   /* XXH64 content checksum mismatch                              */
    ORACLE_ZSTD_E_UNSUPPORTED = -6 /* dictionary id present                                        */
};
/* Decodes exactly one frame starting at src.  On success returns ORACLE_ZSTD_OK and sets
 * *dst_len (bytes produced) and *consumed (bytes of src the frame occupied). */
int oracle_zstd_decode_frame(const void *src, size_t src_len, void *dst, size_t dst_cap,
                             size_t *dst_len, size_t *consumed);
/* Reads the frame header only: returns content size (or -1 if absent) via *fcs, window size via
 * *window, header length via return value (>0) or a negative error. */
int oracle_zstd_frame_header(const void *src, size_t src_len, int64_t *fcs, uint64_t *window,
                             int *has_checksum, int *single_segment);
/* Frame statistics (test/bench aid): number of blocks by type. */
int oracle_zstd_frame_stats(const void *src, size_t src_len, int counts[3] /*raw,rle,compressed*/);

#ifdef __cplusplus
}
#endif
#endif
