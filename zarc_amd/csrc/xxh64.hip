// zarc_amd/csrc/xxh64.hip -- batched XXH64 (seed 0) content checksums for gfx950.
//
// libzstd computes this inside ZSTD_compress2 / ZSTD_decompressStream whenever ChecksumFlag is on, which
// the reference CLI always sets (crates/zarc-cli/src/pack.rs:227); the low 32 bits trail each frame
// (crates/ozarc/src/framing.rs:118-125).  The stripe loop is a serial chain per accumulator, so the
// parallelism is 4 accumulators x many entries: four adjacent lanes own one entry (16 entries per
// wave); each lane streams 8 of every 32 bytes, so a group reads 32 contiguous bytes per step.
#include "zarc_device.h"
#include "zarc_kernels.h"

namespace {
constexpr uint64_t XP1 = 0x9E3779B185EBCA87ULL, XP2 = 0xC2B2AE3D27D4EB4FULL, XP3 = 0x165667B19E3779F9ULL,
                   XP4 = 0x85EBCA77C2B2AE63ULL, XP5 = 0x27D4EB2F165667C5ULL;
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t x) { return zd::rotl64(acc + x * XP2, 31) * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t h, uint64_t v) { return (h ^ xround(0, v)) * XP1 + XP4; }
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    uint32_t lo = zd::shfl((uint32_t)v, src), hi = zd::shfl((uint32_t)(v >> 32), src);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
} // namespace

// out[e] = XXH64(base+off[e], len[e], seed 0).  Entries must start 8-byte aligned.
__global__ void __launch_bounds__(256) zarc_xxh64(const uint8_t *__restrict__ base, const uint64_t *__restrict__ off,
                                                  const uint64_t *__restrict__ len, uint32_t n_entries, uint64_t *__restrict__ out)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = gid >> 2, k = gid & 3;
    const int lane = zd::lane_id(), gl = lane & ~3; // first lane of this group
    const bool valid = e < n_entries;
    const uint8_t *p = valid ? base + off[e] : base;
    const uint64_t n = valid ? len[e] : 0;
    const uint64_t stripes = n >> 5;
    uint64_t acc = k == 0 ? XP1 + XP2 : (k == 1 ? XP2 : (k == 2 ? 0 : 0 - XP1));
    const uint64_t *q = (const uint64_t *)p + k;
    uint64_t s = 0;
    // The loads do not depend on the accumulator chain.  One entry is one chain however many entries the batch has (a 16 MiB entry
    // is half a million rounds for its four lanes), so the chain must never wait for memory: 8 stripes are consumed while the
    // next 16 are in flight (three register sets in rotation).
    if (stripes >= 16) {
        uint64_t x[8], y[8], z[8];
#define XLOAD(v, at) _Pragma("unroll") for (int i = 0; i < 8; i++) v[i] = q[(at) + 4 * i]
#define XROUNDS(v) _Pragma("unroll") for (int i = 0; i < 8; i++) acc = xround(acc, v[i])
        XLOAD(x, 0); XLOAD(y, 32);
        for (; s + 40 <= stripes; s += 24) { // the three sets change roles in place: no register copies that would wait for the newest loads
            XLOAD(z, 64); XROUNDS(x);
            XLOAD(x, 96); XROUNDS(y);
            XLOAD(y, 128); XROUNDS(z);
            q += 96;
        }
        XROUNDS(x); XROUNDS(y);
#undef XLOAD
#undef XROUNDS
        s += 16; q += 64;
    }
    for (; s < stripes; s++) { acc = xround(acc, q[0]); q += 4; }
    // combine the four accumulators of the group (all lanes run the shuffles; lane k==0 finishes)
    uint64_t v1 = shfl64(acc, gl), v2 = shfl64(acc, gl + 1), v3 = shfl64(acc, gl + 2), v4 = shfl64(acc, gl + 3);
    if (!valid || k != 0) return;
    uint64_t h;
    if (n >= 32) {
        h = zd::rotl64(v1, 1) + zd::rotl64(v2, 7) + zd::rotl64(v3, 12) + zd::rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = XP5;
    }
    h += n;
    const uint8_t *t = p + (stripes << 5), *end = p + n;
    while (t + 8 <= end) {
        h ^= xround(0, *(const uint64_t *)t);
        h = zd::rotl64(h, 27) * XP1 + XP4;
        t += 8;
    }
    if (t + 4 <= end) {
        h ^= (uint64_t)(*(const uint32_t *)t) * XP1;
        h = zd::rotl64(h, 23) * XP2 + XP3;
        t += 4;
    }
    while (t < end) {
        h ^= (uint64_t)(*t) * XP5;
        h = zd::rotl64(h, 11) * XP1;
        t++;
    }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    out[e] = h;
}
