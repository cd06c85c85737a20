// zarc_amd/csrc/zge_assemble.hip -- encoder stage 3 (frame assembly), unpack verdicts, corpus fill.
//
// Frame layout written here (RFC 8878 3.1.1; field order as crates/ozarc/src/framing.rs:106-278):
//   magic 28 B5 2F FD | descriptor | [window byte] | frame content size | blocks ... | [XXH64 low 32, LE]
// Same header rules as libzstd's one-shot compress2 at crates/zarc/src/encode/lowlevel_frames.rs:29-31:
// Single_Segment whenever the content fits the window, FCS in its smallest form, no dictionary id,
// checksum when ChecksumFlag is set (crates/zarc-cli/src/pack.rs:227).
#include "zarc_device.h"
#include "zarc_kernels.h"
#include "corpus.h"

// workgroup-cooperative copy, 16 bytes per thread and step (source and destination need no alignment)
__device__ __forceinline__ void group_copy(uint8_t *__restrict__ d, const uint8_t *__restrict__ s, uint32_t n, int tid, int nthreads)
{
    struct B16 { uint64_t a, b; };
    const uint32_t n16 = n / 16;
    for (uint32_t i = (uint32_t)tid; i < n16; i += (uint32_t)nthreads) { B16 v; __builtin_memcpy(&v, s + 16 * (uint64_t)i, 16); __builtin_memcpy(d + 16 * (uint64_t)i, &v, 16); }
    for (uint32_t i = n16 * 16 + (uint32_t)tid; i < n; i += (uint32_t)nthreads) d[i] = s[i];
}

// One workgroup per frame: block payloads are gathered from the per-block scratch slots (or from the
// source for raw / RLE blocks) into one contiguous frame.
__global__ void __launch_bounds__(256) zarc_zge_assemble(ZgeParams P, const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                         const uint64_t *__restrict__ src_len, const uint32_t *__restrict__ order, uint32_t n_frames,
                                                         const uint64_t *__restrict__ block_prefix,
                                                         const ZgeBlock *__restrict__ blocks, const uint8_t *__restrict__ out_scratch,
                                                         const uint64_t *__restrict__ xxh, uint8_t *__restrict__ dst_base,
                                                         const uint64_t *__restrict__ dst_off, uint64_t *__restrict__ dst_len)
{
    if (blockIdx.x >= n_frames) return;
    const uint32_t f = order[blockIdx.x]; // block_prefix / scratch slots are indexed by position in the sub-batch
    const int tid = (int)threadIdx.x;
    const uint8_t *src = src_base + src_off[f];
    const uint64_t n = src_len[f];
    uint8_t *dst = dst_base + dst_off[f];
    uint64_t pos = 0;
    {
        const int wlog = P.window_log;
        const bool single = n <= (1ull << wlog);
        const uint32_t fcs_flag = n < 256 ? 0u : (n < 65536 + 256 ? 1u : (n <= 0xFFFFFFFFull ? 2u : 3u));
        const uint32_t fcs_bytes = fcs_flag == 0 ? (single ? 1u : 0u) : (1u << fcs_flag);
        const uint64_t v = fcs_flag == 1 ? n - 256 : n;
        if (tid == 0) {
            dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
            dst[4] = (uint8_t)((fcs_flag << 6) | ((single ? 1u : 0u) << 5) | ((P.checksum ? 1u : 0u) << 2));
            uint32_t q = 5;
            if (!single) dst[q++] = (uint8_t)((wlog - 10) << 3);
            for (uint32_t i = 0; i < fcs_bytes; i++) dst[q++] = (uint8_t)(v >> (8 * i));
        }
        pos = 5 + (single ? 0u : 1u) + fcs_bytes;
    }
    const uint64_t first = block_prefix[blockIdx.x];
    const uint32_t nblocks = (uint32_t)(block_prefix[blockIdx.x + 1] - first);
    if (n == 0) { // a single empty raw block
        if (tid == 0) { dst[pos] = 1; dst[pos + 1] = 0; dst[pos + 2] = 0; }
        pos += 3;
    } else {
        for (uint32_t b = 0; b < nblocks; b++) {
            const ZgeBlock rec = blocks[first + b];
            const uint32_t last = b + 1 == nblocks ? 1u : 0u;
            const uint32_t size_field = rec.type == 2 ? rec.out_len : rec.src_len;
            const uint32_t hdr = last | (rec.type << 1) | (size_field << 3);
            if (tid == 0) { dst[pos] = (uint8_t)hdr; dst[pos + 1] = (uint8_t)(hdr >> 8); dst[pos + 2] = (uint8_t)(hdr >> 16); }
            pos += 3;
            const uint8_t *from;
            uint32_t cnt;
            if (rec.type == 2) { from = out_scratch + (first + b) * zge_out_stride((uint32_t)P.slot_bytes); cnt = rec.out_len; }
            else if (rec.type == 1) { from = src + (uint64_t)b * ZARC_BLOCK; cnt = 1; }
            else { from = src + (uint64_t)b * ZARC_BLOCK; cnt = rec.src_len; }
            group_copy(dst + pos, from, cnt, tid, (int)blockDim.x);
            pos += cnt;
        }
    }
    if (P.checksum) {
        const uint32_t x = (uint32_t)xxh[f];
        if (tid == 0) { dst[pos] = (uint8_t)x; dst[pos + 1] = (uint8_t)(x >> 8); dst[pos + 2] = (uint8_t)(x >> 16); dst[pos + 3] = (uint8_t)(x >> 24); }
        pos += 4;
    }
    if (tid == 0) dst_len[f] = pos;
}

// Dense copy of n scattered byte ranges (frames in their worst-case slots -> back to back), so that the host-pointer
// entry points move only real bytes over PCIe.  One workgroup per range.
__global__ void __launch_bounds__(256) zarc_gather(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ len,
                                                   const uint64_t *__restrict__ dense_off, uint32_t n, uint8_t *__restrict__ dst)
{
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint8_t *s = src_base + src_off[i];
    uint8_t *d = dst + dense_off[i];
    const uint64_t l = len[i];
    for (uint64_t at = 0; at < l; at += 0x40000000ull) group_copy(d + at, s + at, (uint32_t)(l - at > 0x40000000ull ? 0x40000000ull : l - at), (int)threadIdx.x, (int)blockDim.x);
}

// Store mode (Encoder::enable_compression(false), crates/zarc/src/encode.rs:95-97 -> write_uncompressed_frame,
// encode/lowlevel_frames.rs:47-84): the content goes into Raw blocks.  Like the reference's frame this one carries an
// 8-byte Frame_Content_Size, no Single_Segment flag and no checksum -- but it also carries the Window_Descriptor that
// flag combination requires (the reference omits it, which makes its stored frames undecodable: SURVEY.md quirk 2), and
// the blocks are the format's 128 KiB instead of 65 535 bytes.  Window 128 KiB = the smallest that allows such blocks.
//   28 B5 2F FD | C0 | 38 | content size (8 bytes LE) | { 3-byte block header, <= 131072 raw bytes }*
__global__ void __launch_bounds__(256) zarc_zge_store(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
                                                      const uint64_t *__restrict__ src_len, uint32_t n_frames, uint8_t *__restrict__ dst_base,
                                                      const uint64_t *__restrict__ dst_off, uint64_t *__restrict__ dst_len)
{
    const uint32_t f = blockIdx.x;
    if (f >= n_frames) return;
    const int tid = (int)threadIdx.x;
    const uint8_t *src = src_base + src_off[f];
    const uint64_t n = src_len[f];
    uint8_t *dst = dst_base + dst_off[f];
    if (tid == 0) {
        dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
        dst[4] = 0xC0; // Frame_Content_Size flag 3 (8 bytes), not single segment, no checksum, no dictionary id
        dst[5] = 0x38; // Window_Descriptor: exponent 7, mantissa 0 -> 128 KiB
        for (int i = 0; i < 8; i++) dst[6 + i] = (uint8_t)(n >> (8 * i));
    }
    uint64_t pos = 14;
    const uint64_t nblocks = n == 0 ? 1 : (n + ZARC_BLOCK_MAX - 1) / ZARC_BLOCK_MAX; // store mode: the format's largest blocks
    for (uint64_t b = 0; b < nblocks; b++) {
        const uint64_t at = b * ZARC_BLOCK_MAX;
        const uint32_t cnt = (uint32_t)(n - at > ZARC_BLOCK_MAX ? ZARC_BLOCK_MAX : n - at);
        const uint32_t hdr = (b + 1 == nblocks ? 1u : 0u) | (cnt << 3); // type 0 = Raw
        if (tid == 0) { dst[pos] = (uint8_t)hdr; dst[pos + 1] = (uint8_t)(hdr >> 8); dst[pos + 2] = (uint8_t)(hdr >> 16); }
        pos += 3;
        group_copy(dst + pos, src + at, cnt, tid, (int)blockDim.x);
        pos += cnt;
    }
    if (tid == 0) dst_len[f] = pos;
}

// status[i] keeps a decode error; otherwise CHECKSUM when the stored XXH64 differs (what libzstd reports as
// "Restored data doesn't match checksum"); otherwise DIGEST when the BLAKE3 differs from `expect` -- which the
// reference only logs (crates/zarc-cli/src/unpack.rs:118-120), so the bytes are delivered either way.
__global__ void zarc_unpack_verdict(uint32_t n, const uint64_t *__restrict__ xxh, const uint32_t *__restrict__ stored_checksum,
                                    const uint32_t *__restrict__ digests, const uint32_t *__restrict__ expect, int32_t *__restrict__ status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (status[i] != ZARC_FRAME_OK) return;
    if (stored_checksum[2 * i] && (uint32_t)xxh[i] != stored_checksum[2 * i + 1]) { status[i] = ZARC_FRAME_CHECKSUM; return; }
    if (expect) {
        uint32_t diff = 0; // constant-time compare, like integrity.rs:17-22
        for (int w = 0; w < 8; w++) diff |= digests[(uint64_t)i * 8 + w] ^ expect[(uint64_t)i * 8 + w];
        if (diff) status[i] = ZARC_FRAME_DIGEST;
    }
}

// Synthetic corpus: one lane generates one entry serially from its own counter stream.
__global__ void zarc_corpus_fill(uint8_t *__restrict__ base, const uint64_t *__restrict__ off, const uint64_t *__restrict__ len, uint32_t n,
                                 uint64_t first_index, int kind)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zarc_corpus_entry(base + off[i], (size_t)len[i], first_index + i, kind);
}
