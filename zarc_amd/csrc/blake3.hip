// zarc_amd/csrc/blake3.hip -- batched BLAKE3-256 content digests for gfx950.
//
// Replaces the reference's `blake3::hash(content)` (crates/zarc/src/encode/content_frame.rs:26) and the
// incremental hasher used on unpack (crates/zarc/src/decode/frame_iterator.rs:54,99,77;
// crates/zarc/src/integrity.rs:107-117) for a whole batch of entries at once.
//
// Two kernels:
//   blake3_chunks : one lane per 1 KiB chunk (16 blocks x 7 rounds, all u32 add/xor/rotate -> VALU;
//                   no MFMA: this is integer work).  Writes one 32-byte chaining value per chunk.
//                   Single-chunk entries are finished here (ROOT flag) and produce the digest.
//   blake3_tree   : one workgroup per entry; merges chaining values level by level (adjacent pairs,
//                   odd node carried up == BLAKE3's left-full tree), ROOT flag on the last merge.
// Algorithmic traffic: N bytes read + 32 bytes per chunk of CVs + 32 bytes of digest per entry.
#include "zarc_device.h"
#include "zarc_kernels.h"

namespace {

constexpr uint32_t B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8;
constexpr uint32_t IV0 = 0x6A09E667u, IV1 = 0xBB67AE85u, IV2 = 0x3C6EF372u, IV3 = 0xA54FF53Au,
                   IV4 = 0x510E527Fu, IV5 = 0x9B05688Cu, IV6 = 0x1F83D9ABu, IV7 = 0x5BE0CD19u;

#define B3_G(a, b, c, d, x, y)          \
    a = a + b + x; d = zd::rotr32(d ^ a, 16); \
    c = c + d;     b = zd::rotr32(b ^ c, 12); \
    a = a + b + y; d = zd::rotr32(d ^ a, 8);  \
    c = c + d;     b = zd::rotr32(b ^ c, 7);

// one round with message word indices fixed at compile time (the permutation is folded into the
// indices, so no register moves are needed between rounds)
#define B3_ROUND(m, i0, i1, i2, i3, i4, i5, i6, i7, i8, i9, i10, i11, i12, i13, i14, i15) \
    B3_G(v0, v4, v8, v12, m[i0], m[i1])   B3_G(v1, v5, v9, v13, m[i2], m[i3])             \
    B3_G(v2, v6, v10, v14, m[i4], m[i5])  B3_G(v3, v7, v11, v15, m[i6], m[i7])            \
    B3_G(v0, v5, v10, v15, m[i8], m[i9])  B3_G(v1, v6, v11, v12, m[i10], m[i11])          \
    B3_G(v2, v7, v8, v13, m[i12], m[i13]) B3_G(v3, v4, v9, v14, m[i14], m[i15])

// cv <- compress(cv, m, counter, block_len, flags), first 8 output words
__device__ __forceinline__ void b3_compress(uint32_t cv[8], const uint32_t m[16], uint32_t counter_lo,
                                            uint32_t counter_hi, uint32_t block_len, uint32_t flags)
{
    uint32_t v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
    uint32_t v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = counter_lo, v13 = counter_hi, v14 = block_len, v15 = flags;
    B3_ROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B3_ROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
    B3_ROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
    B3_ROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
    B3_ROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
    B3_ROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
    B3_ROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
    cv[0] = v0 ^ v8;  cv[1] = v1 ^ v9;  cv[2] = v2 ^ v10; cv[3] = v3 ^ v11;
    cv[4] = v4 ^ v12; cv[5] = v5 ^ v13; cv[6] = v6 ^ v14; cv[7] = v7 ^ v15;
}

__device__ __forceinline__ void b3_iv(uint32_t cv[8])
{
    cv[0] = IV0; cv[1] = IV1; cv[2] = IV2; cv[3] = IV3; cv[4] = IV4; cv[5] = IV5; cv[6] = IV6; cv[7] = IV7;
}

} // namespace

// chunk_prefix[i] = number of chunks of entries 0..i-1 (n+1 values); an entry of len L has
// max(1, ceil(L/1024)) chunks.  cvs holds 8 words per chunk, digests 8 words per entry.
__global__ void __launch_bounds__(256) zarc_blake3_chunks(const uint8_t *__restrict__ base, const uint64_t *__restrict__ off,
                                                          const uint64_t *__restrict__ len, const uint64_t *__restrict__ chunk_prefix,
                                                          uint32_t n_entries, uint64_t total_chunks, uint32_t *__restrict__ cvs,
                                                          uint32_t *__restrict__ digests)
{
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_chunks) return;
    // entry = last index with chunk_prefix[entry] <= g
    uint32_t lo = 0, hi = n_entries;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (chunk_prefix[mid] <= g) lo = mid; else hi = mid;
    }
    const uint32_t e = lo;
    const uint64_t chunk = g - chunk_prefix[e];
    const uint64_t elen = len[e];
    const uint64_t nchunks = chunk_prefix[e + 1] - chunk_prefix[e];
    const uint8_t *p = base + off[e] + chunk * 1024;
    const uint64_t rem = elen - chunk * 1024;
    const uint32_t clen = rem > 1024 ? 1024u : (uint32_t)rem;          // bytes in this chunk (0 only for an empty entry)
    const uint32_t nblocks = clen == 0 ? 1u : (clen + 63) / 64;
    const uint32_t root = nchunks == 1 ? B3_ROOT : 0u;
    const bool aligned = (((uintptr_t)p) & 15) == 0;
    uint32_t cv[8];
    b3_iv(cv);
    for (uint32_t b = 0; b < nblocks; b++) {
        uint32_t m[16];
        const uint32_t boff = b * 64;
        const uint32_t blen = clen - boff >= 64 ? 64u : clen - boff;
        if (blen == 64 && aligned) {
            const uint4 *q = (const uint4 *)(p + boff);
            uint4 a = q[0], bb = q[1], c = q[2], d = q[3];
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = bb.x; m[5] = bb.y; m[6] = bb.z; m[7] = bb.w;
            m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
        } else {
#pragma unroll
            for (int w = 0; w < 16; w++) {
                uint32_t x = 0;
                const uint32_t wo = (uint32_t)w * 4;
                if (wo + 4 <= blen) {
                    x = zd::load_u32(p + boff + wo);
                } else if (wo < blen) {
                    for (uint32_t k = 0; k < blen - wo; k++) x |= (uint32_t)p[boff + wo + k] << (8 * k);
                }
                m[w] = x;
            }
        }
        uint32_t flags = (b == 0 ? B3_CHUNK_START : 0u) | (b == nblocks - 1 ? (B3_CHUNK_END | root) : 0u);
        b3_compress(cv, m, (uint32_t)chunk, (uint32_t)(chunk >> 32), blen, flags);
    }
    uint32_t *out = nchunks == 1 ? digests + (uint64_t)e * 8 : cvs + g * 8;
    uint4 *o4 = (uint4 *)out;
    o4[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    o4[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// One workgroup per entry with more than one chunk.  cvs is reduced in place, level by level;
// `tmp` is a second buffer of the same shape (ping-pong).
__global__ void __launch_bounds__(256) zarc_blake3_tree(const uint64_t *__restrict__ chunk_prefix, uint32_t n_entries,
                                                        uint32_t *__restrict__ cvs, uint32_t *__restrict__ tmp,
                                                        uint32_t *__restrict__ digests)
{
    for (uint32_t e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const uint64_t first = chunk_prefix[e];
        uint64_t m = chunk_prefix[e + 1] - first;
        if (m <= 1) continue; // finished by the chunk kernel
        uint32_t *src = cvs + first * 8, *dst = tmp + first * 8;
        while (m > 1) {
            const uint64_t pairs = m >> 1;
            const bool last = m == 2;
            for (uint64_t i = threadIdx.x; i < pairs; i += blockDim.x) {
                const uint4 *q = (const uint4 *)(src + i * 16);
                uint4 a = q[0], b = q[1], c = q[2], d = q[3];
                uint32_t msg[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
                uint32_t cv[8];
                b3_iv(cv);
                b3_compress(cv, msg, 0, 0, 64, B3_PARENT | (last ? B3_ROOT : 0u));
                uint4 *o = (uint4 *)(last ? digests + (uint64_t)e * 8 : dst + i * 8);
                o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
            }
            if ((m & 1) && threadIdx.x == 0) { // odd node is carried up unchanged
                const uint4 *q = (const uint4 *)(src + (m - 1) * 8);
                uint4 *o = (uint4 *)(dst + pairs * 8);
                o[0] = q[0];
                o[1] = q[1];
            }
            __syncthreads(); // level hand-off through global memory, workgroup scope
            m = pairs + (m & 1);
            uint32_t *t = src; src = dst; dst = t;
        }
    }
}
