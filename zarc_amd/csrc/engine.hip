// zarc_amd/csrc/engine.hip -- host side of the C ABI declared in include/zarc_gpu.h.
//
// One handle owns its HIP streams on one device (the engine stream plus two side streams for kernels and copies
// that may overlap it) and grow-only workspaces in HBM.  A batch call uploads the small per-entry descriptors,
// launches the kernel chain, times each stage with HIP events recorded on the stream the stage runs on, and copies
// back only per-entry results (lengths, digests, status).  The host-pointer entry points add chunked staging through
// a pinned ring with a few copy threads.  There is no CPU implementation of the codec behind this file: every
// data-path byte is hashed, matched, coded and decoded by the gfx950 kernels.
#include "../../include/zarc_gpu.h"
#include "zarc_kernels.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); } // (zarc_gpu_destroy deletes the handle: every scratch buffer goes with it)
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = n + n / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

} // namespace

struct zarc_gpu {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // side stream: independent stage-2 kernels of the decoder run next to each other
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream3 = nullptr;
    hipEvent_t ev_fork3 = nullptr, ev_join3 = nullptr;
    // pinned staging ring of the host-pointer entry points (allocated on first use)
    static constexpr int PIN_SLOTS = 8; // slots 0..3: the ring of the way in (H2D); 4..7: the way out (D2H, two alternate) -- the two directions run at once
    static constexpr size_t PIN_PIECE = (size_t)32 << 20;
    uint8_t *pin[PIN_SLOTS] = {};
    hipEvent_t pin_ev[PIN_SLOTS] = {};
    DevBuf d_dense, d_goff, d_glen, d_gdense; // gather of the frames of a chunk before they cross PCIe
    zarc_gpu_params params{};
    std::string last_error;
    std::mutex err_mu; // last_error is also written by the staging helper thread of the host-pointer entry points
    hipStream_t stream_stage = nullptr; // PCIe staging of the host-pointer entry points: its copies never queue behind side kernels
    hipStream_t stream_stage_out = nullptr; // the way out has its own stream (and helper thread): PCIe carries both directions at once
    // descriptors
    DevBuf d_units; // encoder: (queue slot, first block) of every 2 MiB segment
    DevBuf d_pieces; // decoder: ZdecPiece per unit of work of the frame pass
    DevBuf d_off, d_len, d_chunk_prefix, d_block_prefix, d_order, d_dst_off, d_dst_len, d_raw_len, d_frame_off, d_frame_len;
    // hashing
    DevBuf d_cvs, d_cvs_tmp, d_digests, d_xxh, d_expect;
    // encoder
    DevBuf d_blocks, d_seq, d_lit, d_out, d_far, d_plan, d_groups;
    // decoder
    DevBuf d_declit, d_status, d_stored_ck;
    DevBuf d_slot_prefix, d_zblocks, d_nseq, d_fast, d_seqidx, d_seqs, d_ztables, d_litidx, d_lits, d_seqflag, d_totals, d_predef, d_longlist, d_longcnt; // decoder fast path (sequences decoded ahead)
    DevBuf d_queue; // frame queues of the persistent kernels (one u32 each)
    int num_cus = 1;
    int deep_per_cu = 0; // workgroups of zarc_zge_match_deep a CU holds (0: not asked yet)
    uint64_t blake3_total_chunks = 0;
    hipEvent_t ev_b3[2] = {}; // digest kernels on the side stream (pack)
    // staging arenas for the host-pointer entry points
    DevBuf d_arena_in, d_arena_out;
    hipEvent_t ev[16] = {};
    // decoder groups (zarc_gpu_unpack_batch_device): a pair of streams per group; events: 0/1 sequences, 2/3 literals, 4/5 frame passes, 6/7 checksum,
    // 8/9 digest, 10 literals done (join), 11 both stages ahead done (the host waits for it when it cuts frames into pieces)
    static constexpr int DEC_GROUPS = 4;
    hipStream_t gs[2 * DEC_GROUPS] = {};
    hipEvent_t ev_g[DEC_GROUPS][12] = {};
    int dec_groups = 0;        // ZARC_GPU_PX_DEC_GROUPS: 0 = by the batch's shape
    uint64_t dec_split_above = 0; // unpack: a batch of this many content bytes ran out of device memory; such batches are split at once
    float ms[ZARC_GPU_T_COUNT];
    size_t scratch_budget = 0; // 0 = derive from free memory (ZARC_GPU_PX_SCRATCH_MB)
    uint64_t stage_chunk = 0;  // 0 = default per entry point (ZARC_GPU_PX_STAGE_CHUNK)
    int stage_thread = 1;      // ZARC_GPU_PX_STAGE_THREAD
    int copy_threads = 8;      // ZARC_GPU_PX_COPY_THREADS
    uint8_t *meta_pin = nullptr; // page-locked arena for descriptor uploads (meta_take)
    size_t meta_cap = 0, meta_used = 0;
    int ldm[5] = {0, 0, 0, 0, 0}; // EnableLongDistanceMatching, LdmHashLog, LdmMinMatch, LdmBucketSizeLog, LdmHashRateLog: remembered, advisory
    int zero_copy = 4096;      // ZARC_GPU_PX_ZERO_COPY: page-locked caller memory in runs of this many KiB on average is read / written by the DMA engines directly (0 = never)
};

namespace {

#define ZHIP(call)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            std::lock_guard<std::mutex> g_(h->err_mu);                                              \
            h->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return e_ == hipErrorOutOfMemory ? ZARC_GPU_E_NOMEM : ZARC_GPU_E_DEVICE;                \
        }                                                                                           \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline void set_error(zarc_gpu *h, const char *msg) { std::lock_guard<std::mutex> g(h->err_mu); h->last_error = msg; }

// Timing ablations and kernel-steering switches exist only in the diagnostic build (make DIAG=1 -> libzarc_gpu_diag.so, used by
// tools/): the product library never reads them, so no environment variable can change the bytes it produces.
#ifdef ZARC_GPU_DIAG
inline int diag_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
inline double diag_env_f(const char *name, double dflt) { const char *e = getenv(name); return e ? atof(e) : dflt; }
#else
inline int diag_env(const char *, int dflt) { return dflt; }
inline double diag_env_f(const char *, double dflt) { return dflt; }
#endif

ZgeParams derive_params(const zarc_gpu_params &p)
{
    ZgeParams z{};
    int level = p.level == 0 ? 3 : p.level;
    z.level = level;
    z.checksum = p.checksum_flag ? 1 : 0;
    z.window_log = p.window_log ? p.window_log : (level >= 9 ? 22 : 21);
    if (z.window_log < 10) z.window_log = 10;
    if (z.window_log > 27) z.window_log = 27;
    // level >= 9 selects the deep finder (zarc_zge_match_deep): two tagged near tables, a 4-byte short hash, 4-byte matches, a lower
    // match cost, two-way far tables on both hashes, and a parse that tries the live recent offsets (two rounds per tile) with a second
    // lazy step -- within 6 % of libzstd -9 on every real-data item of the test set but one
    const bool deep = level >= 9;
    // levels below 9 (round 3): ONE near table of 2^15 16-bit entries on the 5-byte hash (candidates 1 .. 65536 bytes back), no long table
    z.near16 = deep ? 0 : 1;
    z.long_log = 13; z.short_log = deep ? 13 : 15; z.short_bytes = deep ? 4 : 5; z.tag_bits = 10; z.seg_log = 21; z.rep_back = 256;
    z.tile = 1024; z.sub = 64; z.cap = diag_env("ZARC_GPU_CAP", 256);
    z.min_match = p.min_match >= 4 && p.min_match <= 7 ? p.min_match : (deep ? 4 : 5);
    z.min_rep = 3; z.rep_search = 2; z.back_cap = 8;
    z.lazy = level >= 2 ? 1 : 0; z.lazy_delta = 5; z.lazy2_delta = deep ? 5 : 0;
    // rounds after a tile's first parse: level >= 9 two, with the live recent offsets; below, one that only continues selected matches cut at
    // the cap (ext_cap) and that the kernel runs in tiles that have such a match
    z.rep_pass = deep ? 2 : 1; z.live_reps = deep ? 1 : 0; z.ext_cap = 960;
    z.far_cap = 0; z.cont_cap = deep ? 960 : 0; // level >= 9: continuation guess at a tile's cursor (zge_match.hip: LONG_CAP); far_cap is the model's only
    z.lit_cost = deep ? 6 : 5; z.match_cost = deep ? 10 : 12; z.rep_cost = 9;
    z.short_window_log = 30;
    // far tables in HBM (zge_match.hip).  Level 3: 2^16 buckets, one way on the 12-byte hash, content-defined sampling: the positions
    // whose hash has four given bits zero (one in 16) are inserted and looked up; level >= 9: 2^16 buckets, two ways on both hashes,
    // every 2nd position inserted, all looked up
    z.far_log = 16; z.far_ways = deep ? 2 : 1; z.far_step_log = deep ? 1 : 5; z.far_res_log = deep ? 0 : 2; z.far_short = deep ? 1 : 0;
    z.far_cdc_log = deep ? 0 : 4;
    z.far_min_frame = deep ? 0 : 65536; // smaller frames do without the far table: the near table reaches 64 KiB
    z.far_back = 48; z.far_skip = deep ? 0 : 64;
    if (deep && level >= 15) z.rep_pass = 4; // levels 15 .. 22: two more rounds of the live recent-offset pass (a run-time count: the same kernel)
    if (level <= 1) { // level 1 and the negative levels: the fast finder (zarc_zge_match_fast) -- the near table alone, no lazy step, no extension round
        z.far_log = 0; z.rep_pass = 0; z.lazy = 0;
    }
    z.dbg = diag_env("ZARC_GPU_DBG", 0); // timing-only ablations (outputs invalid when set): diagnostic build only
    return z;
}

inline hipError_t create_low_priority_stream(hipStream_t *s)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreate(s);
    return hipStreamCreateWithPriority(s, hipStreamDefault, least);
}

// indices 0 .. n-1 by descending len[], equal lengths in index order (what std::stable_sort gives, in O(n): a batch of a million small
// entries spent 80 ms per call in the comparison sort).  Lengths are below 2^32 (larger entries are refused before).
inline std::vector<uint32_t> order_by_size_desc(const uint64_t *len, size_t n)
{
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    if (n < 4096) { std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return len[a] > len[b]; }); return order; }
    // (inverted length << 32 | index) pairs, sorted by the upper half: least significant digit first, 11 bits a pass, the pairs walked in
    // memory order (descending length = ascending inverted key; equal keys keep their order)
    std::vector<uint64_t> a(n), b(n);
    for (size_t i = 0; i < n; i++) a[i] = (uint64_t)(~(uint32_t)len[i]) << 32 | (uint32_t)i;
    for (int pass = 0; pass < 3; pass++) {
        const int shift = 32 + 11 * pass;
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; i++) count[(a[i] >> shift & 2047u) + 1]++;
        for (int d = 0; d < 2048; d++) count[d + 1] += count[d];
        for (size_t i = 0; i < n; i++) b[count[a[i] >> shift & 2047u]++] = a[i];
        a.swap(b);
    }
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)a[i];
    return order;
}

// The decoder's order: descending content size in steps of 4 KiB (what the groups, the piece lists of large frames and the XXH64 lanes
// want), and among frames of one size step the better compressed first.  The lane-per-block and lane-per-stream kernels of stage 2 work
// on consecutive frames, and a wave lasts as long as its longest lane: neighbours with the same share of matches and literals have
// chains of a length (a batch that alternates text, records and incompressible entries kept half of every wave's lanes waiting).
inline std::vector<uint32_t> order_for_decode(const uint64_t *raw_len, const uint64_t *frame_len, size_t n)
{
    auto key = [&](size_t i) -> uint32_t {
        const uint64_t r = raw_len[i], c = frame_len[i];
        const uint32_t dens = r ? (uint32_t)std::min<uint64_t>(4095, c * 4096 / r) : 4095u; // compressed share, 12 bits
        return (~(uint32_t)(r >> 12) & 0xFFFFFu) << 12 | dens;                               // ascending: larger first, denser matches first
    };
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    if (n < 4096) { std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); }); return order; }
    std::vector<uint64_t> a(n), b(n);
    for (size_t i = 0; i < n; i++) a[i] = (uint64_t)key(i) << 32 | (uint32_t)i;
    for (int pass = 0; pass < 3; pass++) {
        const int shift = 32 + 11 * pass;
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; i++) count[(a[i] >> shift & 2047u) + 1]++;
        for (int d = 0; d < 2048; d++) count[d + 1] += count[d];
        for (size_t i = 0; i < n; i++) b[count[a[i] >> shift & 2047u]++] = a[i];
        a.swap(b);
    }
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)a[i];
    return order;
}

inline uint64_t chunks_of(uint64_t len) { return len == 0 ? 1 : (len + 1023) / 1024; }
inline uint64_t blocks_of(uint64_t len) { return len == 0 ? 1 : (len + ZARC_BLOCK - 1) / ZARC_BLOCK; }

struct Timer {
    zarc_gpu *h;
    int next = 0;
    hipError_t mark(int *idx) { *idx = next; return hipEventRecord(h->ev[next++], h->stream); }
};

// Per-entry descriptor arrays travel through a page-locked arena of the handle: an asynchronous copy out of ordinary memory is staged
// by the runtime at a fraction of the link rate, and a call over half a million entries spent 27 ms in five such copies.  The arena is
// reset where a call begins (every batch call ends with its stream synchronised) and grows to the largest call's needs.
void *meta_take(zarc_gpu *h, size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes < 65536) return nullptr; // small arrays: the runtime's own staging is as good
    if (h->meta_used + bytes > h->meta_cap) {
        if (h->meta_used) return nullptr; // part of it is in flight: this array goes the ordinary way, the next call starts with more room
        if (h->meta_pin) { (void)hipHostFree(h->meta_pin); h->meta_pin = nullptr; h->meta_cap = 0; }
        const size_t want = std::max<size_t>(bytes * 20, (size_t)16 << 20); // the first array of a call has 8 bytes per entry; an unpack moves about 130
        if (hipHostMalloc((void **)&h->meta_pin, want, 0) != hipSuccess) { (void)hipGetLastError(); h->meta_pin = nullptr; return nullptr; }
        h->meta_cap = want;
    }
    void *p = h->meta_pin + h->meta_used;
    h->meta_used += bytes;
    return p;
}
int upload_bytes(zarc_gpu *h, DevBuf &b, const void *src, size_t bytes)
{
    ZHIP(b.reserve(std::max<size_t>(bytes, 8)));
    if (!bytes) return 0;
    if (void *pin = meta_take(h, bytes)) { memcpy(pin, src, bytes); src = pin; }
    ZHIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
}
int upload_u64(zarc_gpu *h, DevBuf &b, const uint64_t *src, size_t n) { return upload_bytes(h, b, src, n * 8); }
int upload_u32(zarc_gpu *h, DevBuf &b, const uint32_t *src, size_t n) { return upload_bytes(h, b, src, n * 4); }

// BLAKE3 of n entries described by device arrays d_off/d_len (already uploaded); result in h->d_digests
// `side` (pack): the two kernels go to the low-priority side stream behind an event of the engine stream -- the caller launches the
// match finder first: the side stream waits for an event recorded BEHIND that launch, so the digest kernels start when the match finder
// has finished and run beside the entropy stage (low priority) instead of standing in front of the persistent workgroups
int run_blake3(zarc_gpu *h, size_t n, const uint8_t *d_base, const uint64_t *off, const uint64_t *len, const uint64_t *d_off, const uint64_t *d_len,
               bool prepare_only = false, bool launch_only = false, hipStream_t stream = nullptr)
{
    (void)off;
    if (!stream) stream = h->stream;
    if (launch_only) goto launch;
    {
    std::vector<uint64_t> prefix(n + 1);
    prefix[0] = 0;
    for (size_t i = 0; i < n; i++) prefix[i + 1] = prefix[i] + chunks_of(len[i]);
    const uint64_t total = prefix[n];
    int rc = upload_u64(h, h->d_chunk_prefix, prefix.data(), n + 1);
    if (rc) return rc;
    ZHIP(h->d_cvs.reserve(total * 32));
    ZHIP(h->d_cvs_tmp.reserve(total * 32));
    ZHIP(h->d_digests.reserve(std::max<size_t>(n, 1) * 32));
    h->blake3_total_chunks = total;
    }
    if (prepare_only) return 0;
launch:
    {
    const uint64_t total = h->blake3_total_chunks;
    const uint32_t tpb = 256;
    const uint64_t grid = (total + tpb - 1) / tpb;
    hipLaunchKernelGGL(zarc_blake3_chunks, dim3((unsigned)grid), dim3(tpb), 0, stream, d_base, d_off, d_len, h->d_chunk_prefix.as<uint64_t>(),
                       (uint32_t)n, total, h->d_cvs.as<uint32_t>(), h->d_digests.as<uint32_t>());
    const unsigned tgrid = (unsigned)std::min<size_t>(n, 65535);
    hipLaunchKernelGGL(zarc_blake3_tree, dim3(tgrid), dim3(256), 0, stream, h->d_chunk_prefix.as<uint64_t>(), (uint32_t)n,
                       h->d_cvs.as<uint32_t>(), h->d_cvs_tmp.as<uint32_t>(), h->d_digests.as<uint32_t>());
    ZHIP(hipGetLastError());
    }
    return 0;
}

int run_xxh64(zarc_gpu *h, size_t n, const uint8_t *d_base, const uint64_t *d_off, const uint64_t *d_len, hipStream_t stream = nullptr)
{
    if (!stream) stream = h->stream;
    ZHIP(h->d_xxh.reserve(std::max<size_t>(n, 1) * 8));
    const uint32_t tpb = 256;
    const uint64_t threads = (uint64_t)n * 4;
    hipLaunchKernelGGL(zarc_xxh64, dim3((unsigned)((threads + tpb - 1) / tpb)), dim3(tpb), 0, stream, d_base, d_off, d_len, (uint32_t)n,
                       h->d_xxh.as<uint64_t>());
    ZHIP(hipGetLastError());
    return 0;
}

float elapsed(zarc_gpu *h, int a, int b)
{
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, h->ev[a], h->ev[b]) != hipSuccess) return -1.f;
    return ms;
}

int check_common(zarc_gpu *h, size_t n)
{
    if (!h) return ZARC_GPU_E_PARAM;
    if (n > 0x7FFFFFFFu) { set_error(h, "batch too large"); return ZARC_GPU_E_PARAM; }
    ZHIP(hipSetDevice(h->device));
    for (int i = 0; i < ZARC_GPU_T_COUNT; i++) h->ms[i] = -1.f;
    h->meta_used = 0; // every batch call ends with its stream synchronised: what the arena held has been copied
    return 0;
}

} // namespace

extern "C" {

int zarc_gpu_abi_version(void) { return ZARC_GPU_ABI_VERSION; }

int zarc_gpu_device_count(void)
{
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess && count > 0 ? count : 0;
}

int zarc_gpu_create(zarc_gpu_t **out, int device)
{
    if (!out) return ZARC_GPU_E_PARAM;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZARC_GPU_E_DEVICE;
    zarc_gpu *h = new (std::nothrow) zarc_gpu();
    if (!h) return ZARC_GPU_E_NOMEM;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess || hipStreamCreate(&h->stream2) != hipSuccess || create_low_priority_stream(&h->stream3) != hipSuccess || hipStreamCreate(&h->stream_stage) != hipSuccess || hipStreamCreate(&h->stream_stage_out) != hipSuccess ||
        hipEventCreate(&h->ev_fork3) != hipSuccess || hipEventCreate(&h->ev_join3) != hipSuccess ||
        hipEventCreate(&h->ev_fork) != hipSuccess || hipEventCreate(&h->ev_join) != hipSuccess) { delete h; return ZARC_GPU_E_DEVICE; }
    for (auto &e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete h; return ZARC_GPU_E_DEVICE; }
    for (int i = 0; i < 2 * zarc_gpu::DEC_GROUPS; i++) { // the first group holds the largest frames, whose chains are the critical path: its streams go first
        int least = 0, greatest = 0;
        const bool prio = i < 2 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least;
        if ((prio ? hipStreamCreateWithPriority(&h->gs[i], hipStreamDefault, greatest) : hipStreamCreate(&h->gs[i])) != hipSuccess) { delete h; return ZARC_GPU_E_DEVICE; }
    }
    for (auto &row : h->ev_g)
        for (auto &e : row)
            if (hipEventCreate(&e) != hipSuccess) { delete h; return ZARC_GPU_E_DEVICE; }
    for (auto &e : h->ev_b3)
        if (hipEventCreate(&e) != hipSuccess) { delete h; return ZARC_GPU_E_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    h->params.level = 0;          // CCtx::init(0): default level (crates/zarc/src/encode.rs:62)
    h->params.checksum_flag = 0;  // libzstd default; the zarc CLI switches it on (pack.rs:227)
    h->params.content_size_flag = 1;
    h->params.compress = 1;
    for (int i = 0; i < ZARC_GPU_T_COUNT; i++) h->ms[i] = -1.f;
    *out = h;
    return ZARC_GPU_OK;
}

void zarc_gpu_destroy(zarc_gpu_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    DevBuf *all[] = {&h->d_off, &h->d_len, &h->d_chunk_prefix, &h->d_block_prefix, &h->d_order, &h->d_dst_off, &h->d_dst_len, &h->d_raw_len,
                     &h->d_frame_off, &h->d_frame_len, &h->d_cvs, &h->d_cvs_tmp, &h->d_digests, &h->d_xxh, &h->d_expect, &h->d_blocks, &h->d_seq,
                     &h->d_lit, &h->d_out, &h->d_far, &h->d_declit, &h->d_status, &h->d_stored_ck, &h->d_arena_in, &h->d_arena_out, &h->d_queue,
                     &h->d_units, &h->d_pieces, &h->d_slot_prefix, &h->d_zblocks, &h->d_nseq, &h->d_fast, &h->d_seqidx, &h->d_seqs, &h->d_ztables, &h->d_litidx, &h->d_lits};
    for (DevBuf *b : all) b->release();
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    for (auto &row : h->ev_g) for (auto &e : row) if (e) (void)hipEventDestroy(e);
    for (auto &st : h->gs) if (st) (void)hipStreamDestroy(st);
    for (auto &e : h->ev_b3) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream3) (void)hipStreamDestroy(h->stream3);
    if (h->stream_stage) (void)hipStreamDestroy(h->stream_stage);
    if (h->stream_stage_out) (void)hipStreamDestroy(h->stream_stage_out);
    if (h->ev_fork3) (void)hipEventDestroy(h->ev_fork3);
    if (h->ev_join3) (void)hipEventDestroy(h->ev_join3);
    for (int i = 0; i < zarc_gpu::PIN_SLOTS; i++) { if (h->pin[i]) (void)hipHostFree(h->pin[i]); if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]); }
    h->d_dense.release(); h->d_goff.release(); h->d_glen.release(); h->d_gdense.release();
    if (h->meta_pin) (void)hipHostFree(h->meta_pin);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

int zarc_gpu_set_parameter(zarc_gpu_t *h, int id, int value)
{
    if (!h) return ZARC_GPU_E_PARAM;
    switch (id) {
    case ZARC_GPU_P_COMPRESSION_LEVEL:
        if (value < -131072 || value > 22) return ZARC_GPU_E_PARAM; // same bounds as --level (zarc-cli/src/pack.rs:28-33) / libzstd
        h->params.level = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_WINDOW_LOG:
        if (value != 0 && (value < 10 || value > 27)) return ZARC_GPU_E_PARAM;
        h->params.window_log = value; return ZARC_GPU_OK;
    // Search-effort hints: libzstd accepts them inside its bounds and the reference forwards whatever the user gives (pack.rs:86-217), so
    // `zarc pack --zstd Strategy=btopt` has to work.  They are range-checked as libzstd does (ZSTD_cParam_getBounds), remembered (get_params
    // returns them) and ADVISORY: table sizes, search depth and strategy are fixed by the kernels, the frames are the level's frames.
    case ZARC_GPU_P_HASH_LOG: if (value != 0 && (value < 6 || value > 30)) return ZARC_GPU_E_PARAM; h->params.hash_log = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_CHAIN_LOG: if (value != 0 && (value < 6 || value > 30)) return ZARC_GPU_E_PARAM; h->params.chain_log = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_SEARCH_LOG: if (value != 0 && (value < 1 || value > 30)) return ZARC_GPU_E_PARAM; h->params.search_log = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_TARGET_LENGTH: if (value < 0 || value > 131072) return ZARC_GPU_E_PARAM; h->params.target_length = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_STRATEGY: if (value < 0 || value > 9) return ZARC_GPU_E_PARAM; h->params.strategy = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_MIN_MATCH:
        if (value != 0 && (value < 3 || value > 7)) return ZARC_GPU_E_PARAM;
        h->params.min_match = value; return ZARC_GPU_OK;
    case ZARC_GPU_PX_SCRATCH_MB:
        if (value < 0) return ZARC_GPU_E_PARAM;
        h->scratch_budget = (size_t)value << 20; return ZARC_GPU_OK;
    case ZARC_GPU_PX_STAGE_CHUNK:
        if (value != 0 && value < 4096) return ZARC_GPU_E_PARAM;
        h->stage_chunk = (uint64_t)value; return ZARC_GPU_OK;
    case ZARC_GPU_PX_STAGE_THREAD: h->stage_thread = value ? 1 : 0; return ZARC_GPU_OK;
    case ZARC_GPU_PX_COPY_THREADS:
        if (value < 1 || value > 64) return ZARC_GPU_E_PARAM;
        h->copy_threads = value; return ZARC_GPU_OK;
    case ZARC_GPU_PX_DEC_GROUPS:
        if (value < 0 || value > zarc_gpu::DEC_GROUPS) return ZARC_GPU_E_PARAM;
        h->dec_groups = value; return ZARC_GPU_OK;
    case ZARC_GPU_PX_ZERO_COPY: if (value < 0 || value > (1 << 20)) return ZARC_GPU_E_PARAM; h->zero_copy = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_CONTENT_SIZE_FLAG:
        if (value != 1) return ZARC_GPU_E_UNSUPPORTED; // frames always carry their content size
        return ZARC_GPU_OK;
    case ZARC_GPU_P_CHECKSUM_FLAG: h->params.checksum_flag = value ? 1 : 0; return ZARC_GPU_OK;
    case ZARC_GPU_P_DICT_ID_FLAG: return ZARC_GPU_OK; // no dictionaries in zarc
    // Long-distance matching (pack.rs:89-109 forwards these; libzstd accepts them single-threaded): accepted inside libzstd's bounds and
    // ADVISORY like the search-effort hints -- the finders' far tables in HBM ARE a long-distance matcher, at every level above 1
    case ZARC_GPU_P_ENABLE_LDM: if (value < 0 || value > 2) return ZARC_GPU_E_PARAM; h->ldm[0] = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_LDM_HASH_LOG: if (value != 0 && (value < 6 || value > 30)) return ZARC_GPU_E_PARAM; h->ldm[1] = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_LDM_MIN_MATCH: if (value != 0 && (value < 4 || value > 4096)) return ZARC_GPU_E_PARAM; h->ldm[2] = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_LDM_BUCKET_SIZE_LOG: if (value != 0 && (value < 1 || value > 8)) return ZARC_GPU_E_PARAM; h->ldm[3] = value; return ZARC_GPU_OK;
    case ZARC_GPU_P_LDM_HASH_RATE_LOG: if (value < 0 || value > 25) return ZARC_GPU_E_PARAM; h->ldm[4] = value; return ZARC_GPU_OK;
    default:
        // nbWorkers/jobSize/overlapLog (400-402) and the experimental ids
        if ((id >= 400 && id <= 402) || (id >= 500 && id <= 1020)) return ZARC_GPU_E_UNSUPPORTED;
        return ZARC_GPU_E_PARAM;
    }
}

void zarc_gpu_get_params(const zarc_gpu_t *h, zarc_gpu_params *out) { if (h && out) *out = h->params; }

int zarc_gpu_level_finder(int level)
{
    if (level == 0) level = 3; // encode.rs:62: init(0) = the default level
    return level <= 1 ? 1 : (level < 9 ? 3 : (level < 15 ? 9 : 15));
}

int zarc_gpu_parameter_advisory(int id)
{
    switch (id) {
    case ZARC_GPU_P_HASH_LOG: case ZARC_GPU_P_CHAIN_LOG: case ZARC_GPU_P_SEARCH_LOG: case ZARC_GPU_P_TARGET_LENGTH: case ZARC_GPU_P_STRATEGY:
    case ZARC_GPU_P_ENABLE_LDM: case ZARC_GPU_P_LDM_HASH_LOG: case ZARC_GPU_P_LDM_MIN_MATCH: case ZARC_GPU_P_LDM_BUCKET_SIZE_LOG: case ZARC_GPU_P_LDM_HASH_RATE_LOG:
        return 1;
    default: return 0;
    }
}
void zarc_gpu_enable_compression(zarc_gpu_t *h, int compress) { if (h) h->params.compress = compress ? 1 : 0; }

size_t zarc_gpu_bound(size_t n)
{
    size_t blocks = (n + ZARC_BLOCK - 1) / ZARC_BLOCK;
    if (blocks == 0) blocks = 1;
    return align_up(n + 3 * blocks + 18, ZARC_GPU_ALIGN);
}

const char *zarc_gpu_error_name(int code)
{
    switch (code) {
    case ZARC_GPU_OK: return "No error detected";
    case ZARC_GPU_E_DEVICE: return "HIP device error";
    case ZARC_GPU_E_NOMEM: return "Allocation error : not enough memory";
    case ZARC_GPU_E_PARAM: return "Parameter is out of bound";
    case ZARC_GPU_E_UNSUPPORTED: return "Unsupported parameter";
    case ZARC_GPU_E_DSTSIZE: return "Destination buffer is too small";
    default: return "Unspecified error code";
    }
}
const char *zarc_gpu_frame_status_name(int s)
{
    switch (s) {
    case ZARC_GPU_FRAME_OK: return "No error detected";
    case ZARC_GPU_FRAME_CORRUPT: return "Data corruption detected";
    case ZARC_GPU_FRAME_CHECKSUM: return "Restored data doesn't match checksum";
    case ZARC_GPU_FRAME_DIGEST: return "BLAKE3 digest mismatch";
    case ZARC_GPU_FRAME_DSTSIZE: return "Destination buffer is too small";
    case ZARC_GPU_FRAME_BAD_MAGIC: return "Unknown frame descriptor";
    case ZARC_GPU_FRAME_UNSUPPORTED: return "Unsupported frame parameter";
    case ZARC_GPU_FRAME_SRCSIZE: return "Src size is incorrect";
    case ZARC_GPU_FRAME_DUPLICATE: return "frame already exists, skipping";
    default: return "Unspecified error code";
    }
}
const char *zarc_gpu_last_error(const zarc_gpu_t *h) { return h ? h->last_error.c_str() : "null handle"; }
float zarc_gpu_last_kernel_ms(const zarc_gpu_t *h, int which) { return (h && which >= 0 && which < ZARC_GPU_T_COUNT) ? h->ms[which] : -1.f; }

// ---------------------------------------------------------------------------------------------------
int zarc_gpu_blake3_batch_device(zarc_gpu_t *h, size_t n, const void *d_base, const uint64_t *off, const uint64_t *len, uint8_t *digest)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!d_base || !off || !len || !digest) return ZARC_GPU_E_PARAM;
    for (size_t i = 0; i < n; i++) if (off[i] % ZARC_GPU_ALIGN) { set_error(h, "entry offset not 16-byte aligned"); return ZARC_GPU_E_PARAM; }
    if ((rc = upload_u64(h, h->d_off, off, n))) return rc;
    if ((rc = upload_u64(h, h->d_len, len, n))) return rc;
    Timer t{h};
    int a, b;
    ZHIP(t.mark(&a));
    if ((rc = run_blake3(h, n, (const uint8_t *)d_base, off, len, h->d_off.as<uint64_t>(), h->d_len.as<uint64_t>()))) return rc;
    ZHIP(t.mark(&b));
    ZHIP(hipMemcpyAsync(digest, h->d_digests.p, n * 32, hipMemcpyDeviceToHost, h->stream));
    ZHIP(hipStreamSynchronize(h->stream));
    h->ms[ZARC_GPU_T_BLAKE3] = h->ms[ZARC_GPU_T_TOTAL] = elapsed(h, a, b);
    return ZARC_GPU_OK;
}

int zarc_gpu_xxh64_batch_device(zarc_gpu_t *h, size_t n, const void *d_base, const uint64_t *off, const uint64_t *len, uint64_t *out)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!d_base || !off || !len || !out) return ZARC_GPU_E_PARAM;
    for (size_t i = 0; i < n; i++) if (off[i] % ZARC_GPU_ALIGN) { set_error(h, "entry offset not 16-byte aligned"); return ZARC_GPU_E_PARAM; }
    if ((rc = upload_u64(h, h->d_off, off, n))) return rc;
    if ((rc = upload_u64(h, h->d_len, len, n))) return rc;
    Timer t{h};
    int a, b;
    ZHIP(t.mark(&a));
    if ((rc = run_xxh64(h, n, (const uint8_t *)d_base, h->d_off.as<uint64_t>(), h->d_len.as<uint64_t>()))) return rc;
    ZHIP(t.mark(&b));
    ZHIP(hipMemcpyAsync(out, h->d_xxh.p, n * 8, hipMemcpyDeviceToHost, h->stream));
    ZHIP(hipStreamSynchronize(h->stream));
    h->ms[ZARC_GPU_T_XXH64] = h->ms[ZARC_GPU_T_TOTAL] = elapsed(h, a, b);
    return ZARC_GPU_OK;
}

} // extern "C"
namespace {
// have_digests: the caller has hashed the entries already (hash-first dedup): no digest kernels, `digest` is not written
int pack_device_impl(zarc_gpu_t *h, size_t n, const void *d_src_base, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                     size_t dst_cap, uint64_t *dst_off, uint64_t *dst_len, uint8_t *digest, int *status, bool have_digests)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!d_src_base || !src_off || !src_len || !d_dst || !dst_off || !dst_len || (!digest && !have_digests)) return ZARC_GPU_E_PARAM;
    ZgeParams P = derive_params(h->params); // (slot_bytes is set per sub-batch below)
#ifdef ZARC_GPU_DIAG
    // host-side phase clock of this call (ZARC_GPU_DBG & 512): where a call over a million entries spends its time outside the kernels
    double hp[8] = {0};
    auto hnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double hlast = hnow();
#define HOST_PHASE(i) do { const double n_ = hnow(); hp[i] += n_ - hlast; hlast = n_; } while (0)
#else
#define HOST_PHASE(i) do { } while (0)
#endif
    // the match finder has these compiled in (zge_match.hip: F_*)
    const bool dp = P.level >= 9; // the deep finder (zarc_zge_match_deep)
    const bool fp = P.level <= 1; // the fast finder (zarc_zge_match_fast): the level-3 finder's near table without everything behind it
    if (P.rep_back != 256 || P.back_cap != 8 || P.lazy_delta != 5 || P.min_rep != 3 || P.rep_search != 2 || P.seg_log != 21 || P.short_window_log < 30 ||
        P.lit_cost != (dp ? 6 : 5) || P.rep_cost != 9 || P.tag_bits != 10 || P.far_log != (fp ? 0 : 16) || P.match_cost != (dp ? 10 : 12) || P.far_ways != (dp ? 2 : 1) ||
        P.far_step_log != (dp ? 1 : 5) || P.far_res_log != (dp ? 0 : 2) || (P.far_short != 0) != dp || P.long_log != 13 ||
        P.near16 != (dp ? 0 : 1) || P.short_log != (dp ? 13 : 15) || P.far_cdc_log != (dp ? 0 : 4) || P.lazy2_delta != (dp ? 5 : 0) ||
        P.rep_pass != (dp ? (P.level >= 15 ? 4 : 2) : (fp ? 0 : 1)) || P.live_reps != (dp ? 1 : 0) || P.ext_cap != 960 || P.far_cap != 0 || P.cont_cap != (dp ? 960 : 0) ||
        P.far_back != 48 || P.far_skip != (dp ? 0 : 64) || (fp && P.lazy)) { set_error(h, "internal: encoder parameters differ from the compiled-in ones"); return ZARC_GPU_E_PARAM; }
    uint64_t need = 0;
    for (size_t i = 0; i < n; i++) {
        if (src_off[i] % ZARC_GPU_ALIGN) { set_error(h, "entry offset not 16-byte aligned"); return ZARC_GPU_E_PARAM; }
        if (src_len[i] >= 0xFFFFFFF0ull) { set_error(h, "entries of 4 GiB or more are not supported"); return ZARC_GPU_E_UNSUPPORTED; }
        dst_off[i] = need;
        need += zarc_gpu_bound((size_t)src_len[i]);
    }
    if (need > dst_cap) return ZARC_GPU_E_DSTSIZE;
    if ((rc = upload_u64(h, h->d_off, src_off, n))) return rc;
    if ((rc = upload_u64(h, h->d_len, src_len, n))) return rc;
    if ((rc = upload_u64(h, h->d_dst_off, dst_off, n))) return rc;
    ZHIP(h->d_dst_len.reserve(n * 8));
    const uint8_t *base = (const uint8_t *)d_src_base;
    const uint64_t *d_off = h->d_off.as<uint64_t>(), *d_len = h->d_len.as<uint64_t>();
    Timer t{h};
    int e0, e1, e2;
    ZHIP(t.mark(&e0));
    if (h->params.compress) {
        // the frame checksum is a chain of 64-bit multiplies per entry (few waves, latency-bound): it runs on a side stream from the
        // start and is only waited for by the first frame assembly
        ZHIP(hipEventRecord(h->ev_fork, h->stream));
        ZHIP(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        ZHIP(hipEventRecord(h->ev[14], h->stream2));
        if ((rc = run_xxh64(h, n, base, d_off, d_len, h->stream2))) return rc;
        ZHIP(hipEventRecord(h->ev[15], h->stream2));
        ZHIP(hipEventRecord(h->ev_join, h->stream2));
    }
    // the digest: in store mode right here; otherwise its kernels are queued on the low-priority side stream right behind the first
    // match-finder launch (below): they start when it ends, run beside the entropy stage, and are joined before the digests travel back
    if (!have_digests && (rc = run_blake3(h, n, base, src_off, src_len, d_off, d_len, /*prepare_only=*/h->params.compress != 0))) return rc;
    ZHIP(t.mark(&e1));
    bool digest_queued = have_digests; // nothing to queue
    if (!h->params.compress) {
        // store mode (Encoder::enable_compression(false)): raw-block frames, no checksum (lowlevel_frames.rs:47-84)
        hipLaunchKernelGGL(zarc_zge_store, dim3((unsigned)n), dim3(256), 0, h->stream, base, d_off, d_len, (uint32_t)n, (uint8_t *)d_dst,
                           h->d_dst_off.as<uint64_t>(), h->d_dst_len.as<uint64_t>());
        ZHIP(hipGetLastError());
        ZHIP(t.mark(&e2));
        ZHIP(hipMemcpyAsync(dst_len, h->d_dst_len.p, n * 8, hipMemcpyDeviceToHost, h->stream));
        if (!have_digests) ZHIP(hipMemcpyAsync(digest, h->d_digests.p, n * 32, hipMemcpyDeviceToHost, h->stream));
        ZHIP(hipStreamSynchronize(h->stream));
        if (status) for (size_t i = 0; i < n; i++) status[i] = ZARC_GPU_FRAME_OK;
        h->ms[ZARC_GPU_T_BLAKE3] = elapsed(h, e0, e1);
        h->ms[ZARC_GPU_T_XXH64] = 0;
        h->ms[ZARC_GPU_T_MATCH] = 0;
        h->ms[ZARC_GPU_T_ENTROPY] = 0;
        h->ms[ZARC_GPU_T_ASSEMBLE] = elapsed(h, e1, e2);
        h->ms[ZARC_GPU_T_TOTAL] = h->ms[ZARC_GPU_T_BLAKE3] + h->ms[ZARC_GPU_T_ASSEMBLE];
        return ZARC_GPU_OK;
    }
    ZHIP(t.mark(&e2));
    bool xxh_joined = false;

    // ---- encoder: frames sorted by size (largest first), processed in sub-batches that fit the scratch budget ----
    HOST_PHASE(0); // checks, descriptor uploads, checksum + digest set-up
    const std::vector<uint32_t> order = order_by_size_desc(src_len, n);
    HOST_PHASE(1); // size ordering
    auto per_block_of = [](uint32_t slot) { return (size_t)(zge_seq_stride(slot) * 8 + zge_lit_stride(slot) + zge_out_stride(slot) + sizeof(ZgeBlock) + sizeof(ZgePlan)); };
    size_t budget = h->scratch_budget;
    if (!budget) {
        // up to 64 GiB of scratch (BASELINE configs[1] needs 46 GiB to run as ONE launch per kernel), at most 45 % of what is free
        size_t free_b = 0, total_b = 0;
        budget = (size_t)64 << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 100 * 45 < budget) budget = free_b / 100 * 45;
        if (budget < ((size_t)1 << 30)) budget = (size_t)1 << 30;
    }
    float ms_match = 0, ms_ent = 0, ms_asm = 0;
    size_t start = 0;
    std::vector<uint64_t> bp;
    while (start < n) {
        // Scratch slots are sized by the largest block of the sub-batch (frames come in descending size): slot_bytes.  A sub-batch also
        // ends where the frames have become four times smaller than its slots, so that a batch of a few large and a million small
        // entries does not give every small one a large slot.
        const uint64_t first_len = src_len[order[start]];
        const uint32_t slot = (uint32_t)std::min<uint64_t>(ZARC_BLOCK, std::max<uint64_t>(align_up((size_t)first_len, 16), 1024));
        P.slot_bytes = (int)slot;
        const size_t max_blocks = std::max<size_t>(budget / per_block_of(slot), 1);
        size_t end = start, nb = 0;
        while (end < n) {
            const size_t b = (size_t)blocks_of(src_len[order[end]]);
            if (end > start && nb + b > max_blocks) break;
            if (end > start + 4096 && slot > 1024 && src_len[order[end]] * 4 <= slot) break;
            nb += b;
            end++;
        }
        const size_t m = end - start;
        bp.assign(m + 1, 0);
        for (size_t j = 0; j < m; j++) bp[j + 1] = bp[j] + blocks_of(src_len[order[start + j]]);
        if ((rc = upload_u64(h, h->d_block_prefix, bp.data(), m + 1))) return rc;
        if ((rc = upload_u32(h, h->d_order, order.data() + start, m))) return rc;
        ZHIP(h->d_blocks.reserve(nb * sizeof(ZgeBlock)));
        ZHIP(h->d_seq.reserve(nb * (size_t)zge_seq_stride(slot) * 8));
        ZHIP(h->d_lit.reserve(nb * (size_t)zge_lit_stride(slot)));
        ZHIP(h->d_out.reserve(nb * (size_t)zge_out_stride(slot)));
        ZHIP(h->d_queue.reserve(256));
        int a, b, c, d;
        ZHIP(t.mark(&a));
        ZHIP(hipMemsetAsync(h->d_queue.p, 0, 256, h->stream));
        const bool deep = P.level >= 9;
        // the match finder's units of work: the 2^seg_log segments (16 blocks) of every frame, largest frames first
        std::vector<uint32_t> units;
        {
            const uint32_t seg_blocks = (1u << P.seg_log) / ZARC_BLOCK;
            // longest unit first (the queue is a longest-processing-time-first schedule): an unsplit frame of 8 MiB is a longer piece of
            // work than a 2 MiB segment of a split one and has to start before them; the segments of split frames are interleaved frame
            // by frame (equal size: the stable sort keeps that order)
            struct U { uint64_t bytes; uint32_t slot, b0; };
            std::vector<U> us;
            size_t n_split = 0;
            uint64_t max_blocks = 0;
            while (n_split < m && src_len[order[start + n_split]] > ZARC_SPLIT_MIN) { max_blocks = std::max<uint64_t>(max_blocks, bp[n_split + 1] - bp[n_split]); n_split++; }
            for (uint64_t b0 = 0; b0 < max_blocks; b0 += seg_blocks)
                for (size_t j = 0; j < n_split; j++)
                    if (b0 < bp[j + 1] - bp[j]) {
                        const uint64_t len = src_len[order[start + j]], at = b0 * (uint64_t)ZARC_BLOCK;
                        us.push_back(U{std::min<uint64_t>(len - at, (uint64_t)seg_blocks * ZARC_BLOCK), (uint32_t)j, (uint32_t)b0});
                    }
            if (n_split == 0) { // every frame is one unit (zge_match.hip) and the list is in `order` already: longest first
                units.resize(2 * m);
                for (size_t j = 0; j < m; j++) { units[2 * j] = (uint32_t)j; units[2 * j + 1] = 0u; }
            } else {
                for (size_t j = n_split; j < m; j++) us.push_back(U{src_len[order[start + j]], (uint32_t)j, 0u});
                std::stable_sort(us.begin(), us.end(), [](const U &x, const U &y) { return x.bytes > y.bytes; });
                units.reserve(us.size() * 2);
                for (const U &u : us) { units.push_back(u.slot); units.push_back(u.b0); }
            }
        }
        const size_t n_units = units.size() / 2;
        if ((rc = upload_u32(h, h->d_units, units.data(), units.size()))) return rc;
        HOST_PHASE(2); // sub-batch lists + uploads
        auto match_kernel = deep ? zarc_zge_match_deep : (P.level <= 1 ? zarc_zge_match_fast : zarc_zge_match);
#ifdef ZARC_GPU_DIAG
        if (P.dbg) match_kernel = deep ? zarc_zge_match_deep_diag : zarc_zge_match_diag;
#endif
        // persistent workgroups: as many as the chip holds at once -- two of 80 KiB of LDS and 128 registers per thread on a CU; the
        // deep finder's register count decides whether it is one or two (asked once)
        if (deep && h->deep_per_cu == 0) {
            int per_cu = 0;
            ZHIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, match_kernel, 512, 0));
            h->deep_per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
        }
        const size_t match_grid = std::min<size_t>(n_units, (size_t)h->num_cus * (size_t)(deep ? h->deep_per_cu : 2));
        ZHIP(h->d_far.reserve(match_grid * zge_far_words(P) * 4 + 16)); // one far-table slab per resident workgroup (cleared by the kernel per frame)
        // The digest kernels (low-priority stream, queued behind the FIRST sub-batch's match launch) must be off the chip before another
        // match launch: measured on the configs[4] shape, a second launch that found the last 4 ms of them still running took 326 ms
        // instead of 209 (kernel trace in gpurun_out/trace_dpp.txt) -- the persistent workgroups keep whatever uneven placement the
        // launch moment gave them.
        if (digest_queued && !have_digests) ZHIP(hipStreamWaitEvent(h->stream, h->ev_join3, 0));
        hipLaunchKernelGGL(match_kernel, dim3((unsigned)match_grid), dim3(512), 0, h->stream, P, base, d_off, d_len, h->d_order.as<uint32_t>(), h->d_units.as<uint32_t>(), (uint32_t)n_units,
                           h->d_block_prefix.as<uint64_t>(), h->d_blocks.as<ZgeBlock>(), h->d_seq.as<uint64_t>(), h->d_lit.as<uint8_t>(),
                           h->d_queue.as<uint32_t>(), h->d_far.as<uint32_t>());
        ZHIP(hipGetLastError());
        if (!digest_queued) {
            ZHIP(hipEventRecord(h->ev_fork3, h->stream)); // orders the side stream behind the descriptor uploads (and this launch)
            ZHIP(hipStreamWaitEvent(h->stream3, h->ev_fork3, 0));
            ZHIP(hipEventRecord(h->ev_b3[0], h->stream3));
            if ((rc = run_blake3(h, n, base, src_off, src_len, d_off, d_len, false, /*launch_only=*/true, h->stream3))) return rc;
            ZHIP(hipEventRecord(h->ev_b3[1], h->stream3));
            ZHIP(hipEventRecord(h->ev_join3, h->stream3));
            digest_queued = true;
        }
        ZHIP(t.mark(&b));
        {
            unsigned long long *const eprof = (P.dbg & 1024) ? (unsigned long long *)((char *)h->d_queue.p + 128) : (unsigned long long *)nullptr;
            if (bp[1] - bp[0] > 1) {
                // the sub-batch holds frames of several blocks (the largest comes first): the entropy stage runs in two passes around the
                // table plan, which lets the blocks of a group share their sequence tables (zge_entropy.hip: zarc_zge_plan).  (Pass 2 on
                // its own is bound by LDS instruction issue -- 18.7 ms where its share of the one-pass kernel was 13.7 -- and neither running
                // pass 1 of one half of the blocks beside pass 2 of the other nor moving the digest kernels beside pass 2 gave any of that
                // back: tools/r4_ab5.sh, r4_ab7.sh, EXPERIMENTS.md.)
                ZHIP(h->d_plan.reserve(nb * sizeof(ZgePlan)));
                std::vector<uint32_t> groups; // first block slot of every group of a frame with more than one block (frames come largest first)
                for (size_t j = 0; j < m && bp[j + 1] - bp[j] > 1; j++)
                    for (uint64_t g0 = bp[j]; g0 < bp[j + 1]; g0 += ZGE_TABLE_GROUP) groups.push_back((uint32_t)g0);
                if ((rc = upload_u32(h, h->d_groups, groups.data(), groups.size()))) return rc;
                hipLaunchKernelGGL(zarc_zge_entropy_p1, dim3((unsigned)nb), dim3(64), 0, h->stream, (uint32_t)nb, slot, h->d_blocks.as<ZgeBlock>(),
                                   h->d_seq.as<uint64_t>(), h->d_lit.as<uint8_t>(), h->d_out.as<uint8_t>(), eprof, h->d_plan.as<ZgePlan>());
                hipLaunchKernelGGL(zarc_zge_plan, dim3((unsigned)groups.size()), dim3(64), 0, h->stream, (uint32_t)nb, h->d_blocks.as<ZgeBlock>(), h->d_plan.as<ZgePlan>(),
                                   h->d_groups.as<uint32_t>());
                hipLaunchKernelGGL(zarc_zge_entropy_p2, dim3((unsigned)nb), dim3(64), 0, h->stream, (uint32_t)nb, slot, h->d_blocks.as<ZgeBlock>(),
                                   h->d_seq.as<uint64_t>(), h->d_lit.as<uint8_t>(), h->d_out.as<uint8_t>(), eprof, h->d_plan.as<ZgePlan>());
            } else
                hipLaunchKernelGGL(zarc_zge_entropy, dim3((unsigned)nb), dim3(64), 0, h->stream, (uint32_t)nb, slot, h->d_blocks.as<ZgeBlock>(),
                                   h->d_seq.as<uint64_t>(), h->d_lit.as<uint8_t>(), h->d_out.as<uint8_t>(), eprof);
        }
        ZHIP(hipGetLastError());
        ZHIP(t.mark(&c));
        if (!xxh_joined) { ZHIP(hipStreamWaitEvent(h->stream, h->ev_join, 0)); xxh_joined = true; }
        hipLaunchKernelGGL(zarc_zge_assemble, dim3((unsigned)m), dim3(256), 0, h->stream, P, base, d_off, d_len, h->d_order.as<uint32_t>(), (uint32_t)m,
                           h->d_block_prefix.as<uint64_t>(), h->d_blocks.as<ZgeBlock>(), h->d_out.as<uint8_t>(), h->d_xxh.as<uint64_t>(),
                           (uint8_t *)d_dst, h->d_dst_off.as<uint64_t>(), h->d_dst_len.as<uint64_t>());
        ZHIP(hipGetLastError());
        ZHIP(t.mark(&d));
        HOST_PHASE(3); // launches
        ZHIP(hipStreamSynchronize(h->stream)); // the scratch is reused by the next sub-batch; also bounds the event pool
        HOST_PHASE(4); // waiting for the kernels
        if (P.dbg & 1024) { // stage timing of the match finder (diagnostics)
            unsigned long long prof[15];
            ZHIP(hipMemcpy(prof, (const char *)h->d_queue.p + 8, sizeof prof, hipMemcpyDeviceToHost));
            unsigned long long tot = 0;
            for (int i = 0; i < 15; i++) tot += prof[i];
            fprintf(stderr, "zge_match stage ticks (%% of %llu):", tot);
            for (int i = 0; i < 15; i++) fprintf(stderr, " %d:%.1f", i, tot ? 100.0 * (double)prof[i] / (double)tot : 0.0);
            fprintf(stderr, "\n");
            ZHIP(hipMemcpy(prof, (const char *)h->d_queue.p + 128, 6 * 8, hipMemcpyDeviceToHost));
            tot = 0;
            for (int i = 0; i < 6; i++) tot += prof[i];
            fprintf(stderr, "zge_entropy stage ticks (%% of %llu): hist %.1f huf-build %.1f lit-encode %.1f seq-prepass %.1f seq-tables %.1f seq-encode %.1f\n", tot,
                    100.0 * prof[0] / tot, 100.0 * prof[1] / tot, 100.0 * prof[2] / tot, 100.0 * prof[3] / tot, 100.0 * prof[4] / tot, 100.0 * prof[5] / tot);
        }
        ms_match += elapsed(h, a, b);
        ms_ent += elapsed(h, b, c);
        ms_asm += elapsed(h, c, d);
        t.next = 3;
        start = end;
    }
    if (!have_digests) ZHIP(hipStreamWaitEvent(h->stream, h->ev_join3, 0)); // the digests are done
    ZHIP(hipMemcpyAsync(dst_len, h->d_dst_len.p, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (!have_digests) ZHIP(hipMemcpyAsync(digest, h->d_digests.p, n * 32, hipMemcpyDeviceToHost, h->stream));
    ZHIP(hipStreamSynchronize(h->stream));
    if (status) for (size_t i = 0; i < n; i++) status[i] = ZARC_GPU_FRAME_OK;
    HOST_PHASE(5); // results back
#ifdef ZARC_GPU_DIAG
    if (P.dbg & 512) fprintf(stderr, "pack host phases (ms): setup %.2f order %.2f lists %.2f launch %.2f wait %.2f results %.2f\n", hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]);
#endif
    if (have_digests) h->ms[ZARC_GPU_T_BLAKE3] = 0;
    else { float ms = -1.f; if (hipEventElapsedTime(&ms, h->ev_b3[0], h->ev_b3[1]) == hipSuccess) h->ms[ZARC_GPU_T_BLAKE3] = ms; } // side stream: queue wait + kernels
    h->ms[ZARC_GPU_T_XXH64] = elapsed(h, 14, 15); // side stream: overlaps the match finder, not part of the total
    h->ms[ZARC_GPU_T_MATCH] = ms_match;
    h->ms[ZARC_GPU_T_ENTROPY] = ms_ent;
    h->ms[ZARC_GPU_T_ASSEMBLE] = ms_asm;
    h->ms[ZARC_GPU_T_TOTAL] = elapsed(h, e0, e1) + ms_match + ms_ent + ms_asm; // the digest and the checksum run beside these
    return ZARC_GPU_OK;
}

// Hash first (content_frame.rs:26-33): digest every entry, ask the caller which digests it has a frame for already, compress the rest.
// `known(ctx, digest, i)` is called once per entry in index order on the calling thread; nonzero = skip this entry
// (status ZARC_GPU_FRAME_DUPLICATE, dst_len 0).  A caller that returns 0 is expected to remember the digest, so that a later copy
// inside the same batch is skipped as well (first occurrence wins).
int pack_device_dedup_impl(zarc_gpu_t *h, size_t n, const void *d_src_base, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                           size_t dst_cap, uint64_t *dst_off, uint64_t *dst_len, uint8_t *digest, int *status, zarc_gpu_known_fn known, void *ctx)
{
    if (!known) return pack_device_impl(h, n, d_src_base, src_off, src_len, d_dst, dst_cap, dst_off, dst_len, digest, status, false);
    if (!status) return ZARC_GPU_E_PARAM;
    int rc = zarc_gpu_blake3_batch_device(h, n, d_src_base, src_off, src_len, digest);
    if (rc) return rc;
    const float ms_b3 = h->ms[ZARC_GPU_T_BLAKE3];
    std::vector<size_t> keep;
    for (size_t i = 0; i < n; i++) {
        dst_off[i] = 0; dst_len[i] = 0;
        if (known(ctx, digest + i * 32, i)) status[i] = ZARC_GPU_FRAME_DUPLICATE;
        else keep.push_back(i);
    }
    if (keep.empty()) { for (int t = 0; t < ZARC_GPU_T_COUNT; t++) if (t != ZARC_GPU_T_BLAKE3 && t != ZARC_GPU_T_TOTAL) h->ms[t] = 0; return ZARC_GPU_OK; }
    const size_t m = keep.size();
    std::vector<uint64_t> off(m), len(m), doff(m), dlen(m);
    std::vector<int> st(m);
    for (size_t j = 0; j < m; j++) { off[j] = src_off[keep[j]]; len[j] = src_len[keep[j]]; }
    rc = pack_device_impl(h, m, d_src_base, off.data(), len.data(), d_dst, dst_cap, doff.data(), dlen.data(), nullptr, st.data(), true);
    if (rc) return rc;
    for (size_t j = 0; j < m; j++) { dst_off[keep[j]] = doff[j]; dst_len[keep[j]] = dlen[j]; status[keep[j]] = st[j]; }
    h->ms[ZARC_GPU_T_BLAKE3] = ms_b3;
    h->ms[ZARC_GPU_T_TOTAL] += ms_b3;
    return ZARC_GPU_OK;
}
} // namespace
extern "C" {

int zarc_gpu_pack_batch_device(zarc_gpu_t *h, size_t n, const void *d_src_base, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                               size_t dst_cap, uint64_t *dst_off, uint64_t *dst_len, uint8_t *digest, int *status)
{
    return pack_device_impl(h, n, d_src_base, src_off, src_len, d_dst, dst_cap, dst_off, dst_len, digest, status, false);
}

int zarc_gpu_pack_batch_device_dedup(zarc_gpu_t *h, size_t n, const void *d_src_base, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                                     size_t dst_cap, uint64_t *dst_off, uint64_t *dst_len, uint8_t *digest, int *status, zarc_gpu_known_fn known, void *ctx)
{
    if (n && (!digest || !dst_off || !dst_len)) return ZARC_GPU_E_PARAM;
    return pack_device_dedup_impl(h, n, d_src_base, src_off, src_len, d_dst, dst_cap, dst_off, dst_len, digest, status, known, ctx);
}

// The decoder's stages (sequences + literals ahead, frame pass, checksum + digest) of ONE frame are a chain, and a large frame's chain
// is long whatever runs beside it (one wave walks its blocks; XXH64 is serial by construction).  Frames are therefore taken in
// descending size and dealt into up to DEC_GROUPS groups of equal bytes, each group with its own pair of streams: the frame pass and
// the hashes of the large frames start as soon as THEIR sequences and literals are decoded and run beside the earlier stages of the
// groups behind them.  Per-frame arrays are uploaded in that order (group = index range = slot range); results go back in the caller's.
} // extern "C"
namespace {
constexpr int UNPACK_SPLIT = -1000; // internal: the decoder's scratch for this batch exceeds the budget, run it in two parts

int unpack_device_once(zarc_gpu_t *h, size_t n, const void *d_frames_base, const uint64_t *frame_off_in, const uint64_t *frame_len_in,
                       void *d_dst_base, const uint64_t *dst_off_in, const uint64_t *raw_len_in, const uint8_t *expect_in, uint8_t *digest,
                       int *status)
{
    int rc = 0;
    uint64_t total_raw = 0;
    for (size_t i = 0; i < n; i++) {
        if (dst_off_in[i] % ZARC_GPU_ALIGN) { set_error(h, "output offset not 16-byte aligned"); return ZARC_GPU_E_PARAM; }
        if (frame_len_in[i] >= 0xFFFFFFF0ull || raw_len_in[i] >= 0xFFFFFFF0ull) { set_error(h, "frames of 4 GiB or more are not supported"); return ZARC_GPU_E_UNSUPPORTED; }
        total_raw += raw_len_in[i];
    }
    const std::vector<uint32_t> order = diag_env("ZARC_GPU_DEC_DENSITY", 1) ? order_for_decode(raw_len_in, frame_len_in, n) : order_by_size_desc(raw_len_in, n);
    std::vector<uint64_t> frame_off(n), frame_len(n), dst_off(n), raw_len(n);
    for (size_t i = 0; i < n; i++) { const uint32_t f = order[i]; frame_off[i] = frame_off_in[f]; frame_len[i] = frame_len_in[f]; dst_off[i] = dst_off_in[f]; raw_len[i] = raw_len_in[f]; }
    if ((rc = upload_u64(h, h->d_frame_off, frame_off.data(), n))) return rc;
    if ((rc = upload_u64(h, h->d_frame_len, frame_len.data(), n))) return rc;
    if ((rc = upload_u64(h, h->d_dst_off, dst_off.data(), n))) return rc;
    if ((rc = upload_u64(h, h->d_raw_len, raw_len.data(), n))) return rc;
    { std::vector<uint32_t> ident(n); std::iota(ident.begin(), ident.end(), 0u); if ((rc = upload_u32(h, h->d_order, ident.data(), n))) return rc; } // the queues walk the (sorted) indices
    const size_t dec_grid = std::min<size_t>(n, (size_t)h->num_cus * 16); // 4 waves per SIMD (launch bounds of the frame kernels: 128 VGPRs, no spills)
    ZHIP(h->d_declit.reserve(dec_grid * (size_t)(ZARC_BLOCK_MAX + 64)));
    ZHIP(h->d_queue.reserve(256));
    ZHIP(h->d_status.reserve(n * 4));
    ZHIP(h->d_stored_ck.reserve(n * 8));
    std::vector<uint8_t> expect;
    if (expect_in) { // reordered straight into the page-locked descriptor arena when it has room (32 bytes per frame: the largest array of the call)
        uint8_t *ex = (uint8_t *)meta_take(h, n * 32);
        if (!ex) { expect.resize(n * 32); ex = expect.data(); }
        for (size_t i = 0; i < n; i++) memcpy(ex + i * 32, expect_in + (size_t)order[i] * 32, 32);
        ZHIP(h->d_expect.reserve(n * 32));
        ZHIP(hipMemcpyAsync(h->d_expect.p, ex, n * 32, hipMemcpyHostToDevice, h->stream));
    }
    Timer t{h};
    int e0, e1, e3;
    ZHIP(t.mark(&e0));
    // ---- fast path: block scan (lane per frame), then sequence entropy decoding with one lane per block ----
    bool fastpath = diag_env("ZARC_GPU_DEC_FAST", 1) != 0;
    // slots of the fast path: one per block, counted on the device first (a frame the count turns down keeps one slot and goes to the
    // general decoder)
    // Lean sizing (round 4): a batch without large frames needs nothing on the host but three totals -- no frame is cut into pieces, there
    // is one group -- so the slot / sequence / literal offsets are prefix sums made on the device (zarc_scan_u32) and stay there.  With
    // large frames (4 MiB and more) the host needs the per-block summaries anyway and takes the arrays as before.
    size_t n_large = 0; // frames of 4 MiB and more (sorted order: a prefix): the ones the host may cut into pieces
    while (n_large < n && raw_len[n_large] >= ((uint64_t)4 << 20)) n_large++;
    bool lean = fastpath && (n_large == 0 || n_large * 64 <= n) && diag_env("ZARC_GPU_DEC_LEAN", 1) != 0;
    uint64_t lean_totals[3] = {0, 0, 0};
    std::vector<uint64_t> slot_prefix(n + 1, 0); // lean: only the entries the host needs are filled in (the large frames', the groups' bounds, [n])
    if (fastpath) {
        ZHIP(h->d_fast.reserve(n * 4));
        hipLaunchKernelGGL(zarc_zdec_count, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, (const uint8_t *)d_frames_base, h->d_frame_off.as<uint64_t>(),
                           h->d_frame_len.as<uint64_t>(), h->d_raw_len.as<uint64_t>(), (uint32_t)n, h->d_fast.as<uint32_t>());
        ZHIP(hipGetLastError());
        if (lean) {
            ZHIP(h->d_slot_prefix.reserve((n + 1) * 8));
            ZHIP(h->d_totals.reserve(64));
            hipLaunchKernelGGL(zarc_scan_u32, dim3(1), dim3(1024), 0, h->stream, h->d_fast.as<uint32_t>(), 1u, 1u, (uint64_t)n, h->d_slot_prefix.as<uint64_t>(),
                               h->d_totals.as<uint64_t>());
            ZHIP(hipGetLastError());
            ZHIP(hipMemcpyAsync(lean_totals, h->d_totals.p, 8, hipMemcpyDeviceToHost, h->stream));
            if (n_large) ZHIP(hipMemcpyAsync(slot_prefix.data(), h->d_slot_prefix.p, (n_large + 1) * 8, hipMemcpyDeviceToHost, h->stream));
            ZHIP(hipStreamSynchronize(h->stream));
            slot_prefix[n] = lean_totals[0];
        } else {
            std::vector<uint32_t> nblk(n);
            ZHIP(hipMemcpyAsync(nblk.data(), h->d_fast.p, n * 4, hipMemcpyDeviceToHost, h->stream));
            ZHIP(hipStreamSynchronize(h->stream));
            for (size_t i = 0; i < n; i++) slot_prefix[i + 1] = slot_prefix[i] + std::max<uint32_t>(nblk[i], 1u);
        }
    } else
        for (size_t i = 0; i < n; i++) slot_prefix[i + 1] = slot_prefix[i] + 1;
    const size_t nslots = (size_t)slot_prefix[n];
    if (nslots * (size_t)ZDEC_TABLE_CELLS * 2 > ((size_t)8 << 30)) { fastpath = false; if (lean) { lean = false; for (size_t i = 0; i < n; i++) slot_prefix[i + 1] = i + 1; } } // table scratch out of proportion (millions of tiny frames)
    // groups: [gf[g], gf[g+1]) in sorted order, equal shares of the bytes.  One group (the stages simply follow each other) unless the
    // batch holds frames large enough for their chains to matter.
    int groups = h->dec_groups;
    // Measured (tools/ab_groups.sh): overlapped stages slow each other down (they all live on memory latency), so equal-sized frames are
    // best served by one group (configs[1]: 64 / 82 / 117 ms with 1 / 2 / 4 groups); a batch whose largest frames are several times the
    // mean gains from two (configs[4] shape: 227 / 203 / 285 ms) -- the large half's frame pass and XXH64 chains run beside the small
    // half's sequence stage instead of after it.
    if (groups <= 0) groups = (fastpath && n >= 16 && raw_len[0] >= ((uint64_t)4 << 20) && raw_len[0] * (uint64_t)n >= 4 * total_raw) ? 2 : 1;
    if (!fastpath) groups = 1;
    if (groups > zarc_gpu::DEC_GROUPS) groups = zarc_gpu::DEC_GROUPS;
    if ((size_t)groups > n) groups = (int)n;
    std::vector<size_t> gf(groups + 1, n);
    gf[0] = 0;
    { uint64_t acc = 0; int g = 1;
      for (size_t i = 0; i < n && g < groups; i++) {
          acc += raw_len[i];
          if (acc * (uint64_t)groups >= total_raw * (uint64_t)g && i + 1 + (size_t)(groups - g) <= n) gf[g++] = i + 1; // every group keeps a frame
      }
      for (; g < groups; g++) gf[g] = std::max(gf[g - 1] + 1, n - (size_t)(groups - g)); }
    if (lean && fastpath && groups > 1) { // the slot at which every group starts (lean: the prefix sums live on the device)
        for (int g = 1; g < groups; g++) ZHIP(hipMemcpyAsync(&slot_prefix[gf[g]], h->d_slot_prefix.as<uint64_t>() + gf[g], 8, hipMemcpyDeviceToHost, h->stream));
        ZHIP(hipStreamSynchronize(h->stream));
    }
    // digest descriptors: per group, chunk prefixes rebased to the group's first chunk (the kernels take sub-ranges by pointer)
    std::vector<uint64_t> cprefix(n + (size_t)groups), gchunk0(groups + 1, 0);
    { uint64_t c = 0;
      for (int g = 0; g < groups; g++) {
          gchunk0[g] = c;
          uint64_t r = 0;
          for (size_t i = gf[g]; i < gf[g + 1]; i++) { cprefix[i + (size_t)g] = r; r += chunks_of(raw_len[i]); }
          cprefix[gf[g + 1] + (size_t)g] = r;
          c += r;
      }
      gchunk0[groups] = c; }
    if ((rc = upload_u64(h, h->d_chunk_prefix, cprefix.data(), cprefix.size()))) return rc;
    ZHIP(h->d_cvs.reserve(gchunk0[groups] * 32));
    ZHIP(h->d_cvs_tmp.reserve(gchunk0[groups] * 32));
    ZHIP(h->d_digests.reserve(n * 32));
    ZHIP(h->d_xxh.reserve(n * 8));
    std::vector<uint64_t> seqidx, litidx;
    uint64_t total_seqs = 0, total_lits = 0;
    uint64_t g_seq_at[zarc_gpu::DEC_GROUPS + 1] = {}, g_lit_at[zarc_gpu::DEC_GROUPS + 1] = {}; // lean: the groups' first sequence / literal
    if (fastpath) {
        if (!lean && (rc = upload_u64(h, h->d_slot_prefix, slot_prefix.data(), n + 1))) return rc;
        ZHIP(h->d_zblocks.reserve(nslots * sizeof(ZdecBlock)));
        ZHIP(h->d_nseq.reserve(nslots * 8));
        ZHIP(h->d_fast.reserve(n * 4));
        ZHIP(h->d_seqidx.reserve((nslots + 1) * 8));
        ZHIP(h->d_litidx.reserve((nslots + 1) * 8));
        ZHIP(h->d_ztables.reserve(nslots * (size_t)ZDEC_TABLE_CELLS * 2));
        if (!h->d_predef.p) { // the predefined tables: once per handle
            ZHIP(h->d_predef.reserve(ZDEC_PREDEF_CELLS * 2));
            hipLaunchKernelGGL(zarc_zdec_predef, dim3(1), dim3(64), 0, h->stream, h->d_predef.as<uint16_t>());
            ZHIP(hipGetLastError());
        }
        ZHIP(h->d_seqflag.reserve((nslots / 16 + (size_t)zarc_gpu::DEC_GROUPS + 2) * 4));
        ZHIP(h->d_longlist.reserve((nslots + 64) * 4));                                   // zarc_zdec_seqs_lds: the blocks with long chains, per group
        ZHIP(h->d_longcnt.reserve((size_t)zarc_gpu::DEC_GROUPS * 64 * 4));                // ... and their counters (2 x ZDEC_LONG_BUCKETS + 1 words per group)
        ZHIP(hipMemsetAsync(h->d_zblocks.p, 0xFF, nslots * sizeof(ZdecBlock), h->stream));
        ZHIP(hipMemsetAsync(h->d_nseq.p, 0, nslots * 8, h->stream));
        hipLaunchKernelGGL(zarc_zdec_scan, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, (const uint8_t *)d_frames_base, h->d_frame_off.as<uint64_t>(),
                           h->d_frame_len.as<uint64_t>(), h->d_raw_len.as<uint64_t>(), (uint32_t)n, h->d_slot_prefix.as<uint64_t>(), h->d_zblocks.as<ZdecBlock>(),
                           h->d_nseq.as<uint32_t>(), h->d_fast.as<uint32_t>());
        ZHIP(hipGetLastError());
        if (lean) { // the sequence / literal scratch is sized exactly: sums of the blocks' counts, made where the counts are
            hipLaunchKernelGGL(zarc_scan_u32, dim3(1), dim3(1024), 0, h->stream, h->d_nseq.as<uint32_t>(), 2u, 0u, (uint64_t)nslots, h->d_seqidx.as<uint64_t>(),
                               h->d_totals.as<uint64_t>() + 1);
            hipLaunchKernelGGL(zarc_scan_u32, dim3(1), dim3(1024), 0, h->stream, h->d_nseq.as<uint32_t>() + 1, 2u, 0u, (uint64_t)nslots, h->d_litidx.as<uint64_t>(),
                               h->d_totals.as<uint64_t>() + 2);
            ZHIP(hipGetLastError());
            ZHIP(hipMemcpyAsync(lean_totals + 1, h->d_totals.as<uint64_t>() + 1, 16, hipMemcpyDeviceToHost, h->stream));
            for (int g = 1; g < groups; g++) { // where every group's sequences / literals start
                ZHIP(hipMemcpyAsync(&g_seq_at[g], h->d_seqidx.as<uint64_t>() + slot_prefix[gf[g]], 8, hipMemcpyDeviceToHost, h->stream));
                ZHIP(hipMemcpyAsync(&g_lit_at[g], h->d_litidx.as<uint64_t>() + slot_prefix[gf[g]], 8, hipMemcpyDeviceToHost, h->stream));
            }
            ZHIP(hipStreamSynchronize(h->stream));
            total_seqs = lean_totals[1]; total_lits = lean_totals[2];
            g_seq_at[groups] = total_seqs; g_lit_at[groups] = total_lits;
        } else {
            std::vector<uint32_t> counts(nslots * 2);
            ZHIP(hipMemcpyAsync(counts.data(), h->d_nseq.p, nslots * 8, hipMemcpyDeviceToHost, h->stream));
            ZHIP(hipStreamSynchronize(h->stream)); // the sequence / literal scratch is sized exactly: sums of the blocks' counts
            seqidx.resize(nslots + 1); litidx.resize(nslots + 1);
            uint64_t total = 0, lit_total = 0;
            for (size_t i = 0; i < nslots; i++) { seqidx[i] = total; total += counts[2 * i]; litidx[i] = lit_total; lit_total += counts[2 * i + 1]; }
            seqidx[nslots] = total; litidx[nslots] = lit_total;
            total_seqs = total; total_lits = lit_total;
        }
        if (h->scratch_budget && n > 1 && total_seqs * 8 + total_lits + nslots * (uint64_t)(ZDEC_TABLE_CELLS * 2 + sizeof(ZdecBlock) + 32) > h->scratch_budget) return UNPACK_SPLIT;
        if (!lean) {
            if ((rc = upload_u64(h, h->d_seqidx, seqidx.data(), nslots))) return rc;
            if ((rc = upload_u64(h, h->d_litidx, litidx.data(), nslots))) return rc;
        }
        ZHIP(h->d_seqs.reserve(std::max<uint64_t>(total_seqs, 1) * 8));
        ZHIP(h->d_lits.reserve(std::max<uint64_t>(total_lits, 1) + 64));
    }
    ZHIP(hipMemsetAsync(h->d_queue.p, 0, 256, h->stream)); // two queues per group: the fast frame pass and the general decoder each walk the group's frames
    ZHIP(hipMemsetAsync(h->d_status.p, 0, n * 4, h->stream)); // the pieces of a frame raise its status with atomicMax
    ZHIP(hipEventRecord(h->ev_fork, h->stream)); // descriptors are in place
    const int dec_dbg = diag_env("ZARC_GPU_DBG_DEC", 0);
    const bool side = diag_env("ZARC_GPU_DEC_SIDE", 1) != 0;
    int seq_lanes = diag_env("ZARC_GPU_SEQ_LANES", 64);
    if (seq_lanes != 8 && seq_lanes != 16 && seq_lanes != 32 && seq_lanes != 64) seq_lanes = 64;
    bool have_seq_t[zarc_gpu::DEC_GROUPS] = {}, have_lit_t[zarc_gpu::DEC_GROUPS] = {};
    std::vector<uint32_t> nblk_of(lean ? n_large : n); // blocks per frame (sorted order; lean: of the large frames only)
    for (size_t i = 0; i < nblk_of.size(); i++) nblk_of[i] = (uint32_t)(slot_prefix[i + 1] - slot_prefix[i]);
    std::vector<ZdecPiece> pieces[zarc_gpu::DEC_GROUPS];
    std::vector<ZdecBlock> hblocks;
    std::vector<uint32_t> hfast;
    ZHIP(h->d_pieces.reserve(lean ? (size_t)(slot_prefix[n_large] / 8 + n_large + 16) * sizeof(ZdecPiece) : (nslots / 8 + n + 16) * sizeof(ZdecPiece)));
    size_t piece_base = 0;
    // phase 0 queues the stages ahead of every group, phase 1 the frame passes and hashes: in between the host may have to look at what
    // stage 2 found out about a group's blocks (pieces, below), and the next group's stages ahead must be running by then
    for (int phase = 0; phase < 2; phase++)
    for (int g = 0; g < groups; g++) {
        // The stages ahead (sequences, literals) of all groups follow each other on the engine's two streams -- several sequence kernels side
        // by side were measured and only slow each other down (they live on MALL / fabric latency) -- while the frame pass and the hashes
        // of a group run on the group's own pair of streams beside the next group's stages ahead.
        hipStream_t pa = h->stream, pb = h->stream2;
        hipStream_t sa = groups == 1 ? h->stream : h->gs[2 * g], sb = groups == 1 ? h->stream2 : h->gs[2 * g + 1];
        hipEvent_t *ev = h->ev_g[g];
        const size_t f0 = gf[g], f1 = gf[g + 1], ng = f1 - f0;
        if (phase == 0 && g == 0) ZHIP(hipStreamWaitEvent(pb, h->ev_fork, 0));
        if (phase == 0 && fastpath) {
            hipStream_t sa_post = sa, sb_post = sb;
            sa = pa; sb = pb; // this block launches the stages ahead
            const uint64_t s0 = slot_prefix[f0], s1 = slot_prefix[f1];
            const uint64_t g_seqs = lean ? g_seq_at[g + 1] - g_seq_at[g] : seqidx[s1] - seqidx[s0], g_lits = lean ? g_lit_at[g + 1] - g_lit_at[g] : litidx[s1] - litidx[s0];
            // the literal and the sequence kernels are independent and neither fills the chip: they run side by side
            hipStream_t sl = side ? sb : sa;
            // the shared-table sequence kernel needs 23 KiB of LDS per wave: launched first it gets its place on every CU at once and the
            // literal waves (12.5 KiB each) fill what is left; behind them it would wait for LDS (diagnostic switch: ZARC_GPU_LIT_FIRST)
            const bool lit_first = diag_env("ZARC_GPU_LIT_FIRST", 0) != 0;
            auto launch_literals = [&]() -> int {
            if (g_lits) {
                ZHIP(hipEventRecord(ev[2], sl));
                hipLaunchKernelGGL(zarc_zdec_literals, dim3((unsigned)((s1 - s0 + ZDEC_LIT_GROUP - 1) / ZDEC_LIT_GROUP)), dim3(64), 0, sl, (const uint8_t *)d_frames_base,
                                   h->d_frame_off.as<uint64_t>(), s1, h->d_slot_prefix.as<uint64_t>(), h->d_zblocks.as<ZdecBlock>(), h->d_litidx.as<uint64_t>(),
                                   h->d_lits.as<uint8_t>(), h->d_fast.as<uint32_t>(), s0);
                ZHIP(hipGetLastError());
                ZHIP(hipEventRecord(ev[3], sl));
                have_lit_t[g] = true;
            }
            return ZARC_GPU_OK; };
            if (lit_first && (rc = launch_literals())) return rc;
            if (g_seqs) {
                ZHIP(hipEventRecord(ev[0], sa));
                {
                    // Waves whose 64 blocks share their tables (the engine's own frames: one table set per group of sixteen 64 KiB blocks; libzstd's
                    // Repeat_Mode blocks) decode with the tables in LDS; the others raise their flag and are done by the launches behind it: blocks
                    // with a long chain by zarc_zdec_seqs_lds (every lane's own tables in LDS, on a side stream, from a list ordered by sequence count),
                    // the rest by zarc_zdec_seqs with a table set per block in HBM scratch.
                    // (a batch of small frames -- fewer than four blocks per frame on average -- has nothing to share: a workgroup's 64 blocks
                    // would belong to a dozen frames with a dozen table sets, and every workgroup would hand its blocks on after looking)
                    const bool shared = diag_env("ZARC_GPU_SEQ_SHARED", 1) != 0 && (s1 - s0) >= 4 * (uint64_t)ng;
                    // block slots per workgroup of the shared-table kernel.  Round 4, 128 KiB blocks and a table per 8: 64 -> 27.3 ms, 32 -> 23.7, 16 -> 27.3
                    // (tools/r4_ab4.sh); with the straight-line step, 64 KiB blocks and a table per 16: 64 lanes with 8 sets 13.1 ms alone, 32 lanes with
                    // 5 sets 14.9, and side by side with the literals the smaller LDS share is what counts (tools/r4_sets.sh)
                    int sw = diag_env("ZARC_GPU_SEQ_WIDTH", 64);
                    if (sw != 16 && sw != 32 && sw != 64) sw = 64;
                    if (!shared) sw = seq_lanes;
                    const size_t waves = (size_t)((s1 - s0 + (uint64_t)sw - 1) / (uint64_t)sw);
                    const bool split_long = diag_env("ZARC_GPU_SEQ_LONG", 1) != 0 && (shared ? sw == 64 : sw % ZDEC_LDS_LANES == 0);
                    uint32_t *flags = nullptr;
                    if (shared) {
                        flags = h->d_seqflag.as<uint32_t>() + (size_t)(s0 / 16) + (size_t)g; // (a group's slots need not start at a multiple of the width)
                        ZHIP(hipMemsetAsync(flags, 0, waves * 4, sa));
                        auto kern = sw == 64 ? zarc_zdec_seqs_shared : (sw == 32 ? zarc_zdec_seqs_shared32 : zarc_zdec_seqs_shared16);
                        hipLaunchKernelGGL(kern, dim3((unsigned)waves), dim3(sw), 0, sa, (const uint8_t *)d_frames_base, h->d_frame_off.as<uint64_t>(),
                                           s1, h->d_slot_prefix.as<uint64_t>(), h->d_zblocks.as<ZdecBlock>(), h->d_seqidx.as<uint64_t>(), h->d_seqs.as<uint64_t>(),
                                           h->d_fast.as<uint32_t>(), s0, flags);
                    }
                    if (split_long) {
                        static_assert(2 * ZDEC_LONG_BUCKETS + 1 <= 64, "counter words per group");
                        uint32_t *cnt = h->d_longcnt.as<uint32_t>() + (size_t)g * 64, *list = h->d_longlist.as<uint32_t>() + (size_t)s0;
                        const unsigned w64 = (unsigned)((s1 - s0 + 63) / 64);
                        ZHIP(hipEventRecord(h->ev_fork3, sa));
                        ZHIP(hipStreamWaitEvent(h->stream3, h->ev_fork3, 0));
                        ZHIP(hipMemsetAsync(cnt, 0, 64 * 4, h->stream3));
                        hipLaunchKernelGGL(zarc_zdec_long_count, dim3(w64), dim3(64), 0, h->stream3, (const ZdecBlock *)h->d_zblocks.as<ZdecBlock>(), s0, s1,
                                           (const uint32_t *)flags, cnt);
                        hipLaunchKernelGGL(zarc_zdec_long_fill, dim3(w64), dim3(64), 0, h->stream3, (const ZdecBlock *)h->d_zblocks.as<ZdecBlock>(), s0, s1,
                                           (const uint32_t *)flags, cnt, list);
                        hipLaunchKernelGGL(zarc_zdec_seqs_lds, dim3((unsigned)((s1 - s0 + ZDEC_LDS_LANES - 1) / ZDEC_LDS_LANES)), dim3(ZDEC_LDS_LANES), 0, h->stream3,
                                           (const uint8_t *)d_frames_base, h->d_frame_off.as<uint64_t>(), s1, h->d_slot_prefix.as<uint64_t>(),
                                           h->d_zblocks.as<ZdecBlock>(), h->d_seqidx.as<uint64_t>(), h->d_seqs.as<uint64_t>(), h->d_fast.as<uint32_t>(), s0,
                                           (const uint32_t *)cnt, (const uint32_t *)list);
                        ZHIP(hipGetLastError());
                        ZHIP(hipEventRecord(h->ev_join3, h->stream3));
                    }
                    hipLaunchKernelGGL(zarc_zdec_seqs, dim3((unsigned)waves), dim3(sw), 0, sa, (const uint8_t *)d_frames_base,
                                       h->d_frame_off.as<uint64_t>(), s1, h->d_slot_prefix.as<uint64_t>(), h->d_zblocks.as<ZdecBlock>(), h->d_seqidx.as<uint64_t>(),
                                       h->d_seqs.as<uint64_t>(), h->d_ztables.as<uint16_t>(), h->d_fast.as<uint32_t>(), s0, (const uint32_t *)flags,
                                       h->d_predef.as<uint16_t>(), split_long ? 1 : 0);
                    if (split_long) ZHIP(hipStreamWaitEvent(sa, h->ev_join3, 0));
                }
                ZHIP(hipGetLastError());
                ZHIP(hipEventRecord(ev[1], sa));
                have_seq_t[g] = true;
            }
            if (!lit_first && (rc = launch_literals())) return rc;
            sa = sa_post; sb = sb_post;
            if (groups > 1) { ZHIP(hipEventRecord(ev[1], pa)); ZHIP(hipStreamWaitEvent(sa, ev[1], 0)); } // (re-recorded when the group has no sequences: it still orders the group behind the descriptors)
            if (g_lits && side) { ZHIP(hipEventRecord(ev[10], pb)); ZHIP(hipStreamWaitEvent(sa, ev[10], 0)); }
            ZHIP(hipEventRecord(ev[11], sa));
        }
        if (phase == 0) continue;
        // ---- pieces: the units of work of the frame pass.  A frame is a chain for one wave (its matches read what its earlier blocks
        // wrote) -- unless stage 2's summaries show that from some block on nothing in front of that block is read: the engine's own
        // frames are made of independent 2 MiB segments (zge_match.hip), libzstd's multi-threaded ones of independent jobs.  The host
        // cuts such frames (4 MiB and more) at those blocks, gives every piece its output position and repeat-offset history, and the
        // pieces decode side by side.
        const size_t fl = lean ? std::max(f0, std::min(f1, n_large)) : f1; // lean: frames [f0, fl) get a piece list, [fl, f1) are one piece each (no list)
        if (fastpath) {
            const uint64_t s0 = slot_prefix[f0], s1 = slot_prefix[fl];
            bool large = false;
            for (size_t i = f0; i < fl && !large; i++) large = nblk_of[i] >= 32;
            if (large) {
                ZHIP(hipEventSynchronize(ev[11]));
                hblocks.resize(s1 - s0); hfast.resize(fl - f0);
                ZHIP(hipMemcpyAsync(hblocks.data(), h->d_zblocks.as<ZdecBlock>() + s0, (s1 - s0) * sizeof(ZdecBlock), hipMemcpyDeviceToHost, sa));
                ZHIP(hipMemcpyAsync(hfast.data(), h->d_fast.as<uint32_t>() + f0, (fl - f0) * 4, hipMemcpyDeviceToHost, sa));
                ZHIP(hipStreamSynchronize(sa));
            }
            std::vector<ZdecPiece> &pc = pieces[g];
            std::vector<uint64_t> start, need;
            for (size_t i = f0; i < fl; i++) {
                const uint32_t nb = nblk_of[i];
                ZdecPiece whole{(uint32_t)i, 0u, 0xFFFFFFFFu, {1u, 4u, 8u}, 0ull, raw_len[i]};
                if (!large || nb < 32 || !hfast[i - f0]) { pc.push_back(whole); continue; }
                const ZdecBlock *zb = hblocks.data() + (slot_prefix[i] - s0);
                // where every block's output starts, and the lowest position it reads
                start.assign(nb + 1, 0); need.assign(nb + 1, ~0ull);
                bool usable = true;
                for (uint32_t j = 0; j < nb && usable; j++) {
                    if (zb[j].type > 2 || (zb[j].type == 2 && zb[j].nseq && zb[j].state != 1)) usable = false;
                    start[j + 1] = start[j] + zb[j].pad[1];
                    need[j] = (zb[j].pad[0] == ZDEC_REACH_UNKNOWN || zb[j].pad[0] > start[j]) ? 0 : start[j] - zb[j].pad[0];
                }
                if (!usable || start[nb] != raw_len[i]) { pc.push_back(whole); continue; }
                for (uint32_t j = nb; j-- > 0;) need[j] = std::min(need[j], need[j + 1]); // lowest position read by block j or any later one
                uint32_t r[3] = {1u, 4u, 8u}, first = 0, rfirst[3] = {1u, 4u, 8u};
                bool rep_ok = true;
                for (uint32_t j = 0; j <= nb; j++) {
                    const bool cut = j == nb || (j > first && j - first >= 8 && nb - j >= 8 && need[j] >= start[j] && rep_ok);
                    if (cut) {
                        pc.push_back(ZdecPiece{(uint32_t)i, first, j == nb ? 0xFFFFFFFFu : j - first, {rfirst[0], rfirst[1], rfirst[2]}, start[first], start[j] - start[first]});
                        first = j; rfirst[0] = r[0]; rfirst[1] = r[1]; rfirst[2] = r[2];
                    }
                    if (j == nb) break;
                    if (zb[j].type == 2 && zb[j].nseq) { // history after the block, from its symbolic summary
                        uint32_t nr[3];
                        for (int k = 0; k < 3; k++) {
                            const uint32_t e = zb[j].rep[k];
                            if (e & ZDEC_REP_REF) {
                                const uint32_t slot = e & 3, delta = (e & ~ZDEC_REP_REF) >> 2, hv = slot < 3 ? r[slot] : 0u;
                                if (hv <= delta) { rep_ok = false; nr[k] = 1; } else nr[k] = hv - delta; // (no cut behind a history the frame pass would reject)
                            }
                            else nr[k] = e;
                        }
                        r[0] = nr[0]; r[1] = nr[1]; r[2] = nr[2];
                    }
                }
            }
            std::stable_sort(pc.begin(), pc.end(), [](const ZdecPiece &x, const ZdecPiece &y) { return x.out_len > y.out_len; }); // longest first
            if (!pc.empty()) ZHIP(hipMemcpyAsync(h->d_pieces.as<ZdecPiece>() + piece_base, pc.data(), pc.size() * sizeof(ZdecPiece), hipMemcpyHostToDevice, sa));
        }
        ZHIP(hipEventRecord(ev[4], sa));
        const size_t grid_g = std::max<size_t>(1, std::min<size_t>(ng, dec_grid / (size_t)groups));
        if (fastpath) {
            const size_t listed = pieces[g].size(), np = listed + (f1 - fl); // behind the list: frames [fl, f1), one piece each, made up by the kernel
            hipLaunchKernelGGL(zarc_zstd_frames, dim3((unsigned)std::min<size_t>(np, (size_t)h->num_cus * 16)), dim3(64), 0, sa, (const uint8_t *)d_frames_base, h->d_frame_off.as<uint64_t>(),
                               h->d_frame_len.as<uint64_t>(), (uint8_t *)d_dst_base, h->d_dst_off.as<uint64_t>(), h->d_raw_len.as<uint64_t>(),
                               h->d_pieces.as<ZdecPiece>() + piece_base, (uint32_t)listed, (uint32_t)fl, (uint32_t)np, h->d_status.as<int32_t>(), h->d_stored_ck.as<uint32_t>(), dec_dbg, h->d_queue.as<uint32_t>() + 2 * g,
                               h->d_fast.as<uint32_t>(), h->d_slot_prefix.as<uint64_t>(), h->d_zblocks.as<ZdecBlock>(), h->d_seqidx.as<uint64_t>(),
                               h->d_seqs.as<uint64_t>(), h->d_litidx.as<uint64_t>(), h->d_lits.as<uint8_t>());
            ZHIP(hipGetLastError());
        }
        // frames the fast path turned down (or all of them when it is off) are decoded inline; with nothing to do every wave leaves at once.
        // Each group's launch has its own slice of the per-wave literal scratch.
        hipLaunchKernelGGL(zarc_zstd_decode, dim3((unsigned)grid_g), dim3(64), (size_t)diag_env("ZARC_GPU_DEC_PADLDS", 0), sa,
                           (const uint8_t *)d_frames_base, h->d_frame_off.as<uint64_t>(), h->d_frame_len.as<uint64_t>(), (uint8_t *)d_dst_base,
                           h->d_dst_off.as<uint64_t>(), h->d_raw_len.as<uint64_t>(), h->d_order.as<uint32_t>() + f0, (uint32_t)ng,
                           h->d_declit.as<uint8_t>() + (size_t)g * (dec_grid / (size_t)groups) * (size_t)(ZARC_BLOCK_MAX + 64),
                           h->d_status.as<int32_t>(), h->d_stored_ck.as<uint32_t>(), dec_dbg, h->d_queue.as<uint32_t>() + 2 * g + 1,
                           fastpath ? h->d_fast.as<uint32_t>() : (const uint32_t *)nullptr);
        ZHIP(hipGetLastError());
        ZHIP(hipEventRecord(ev[5], sa));
        // verification passes over the decoded bytes (K2 + XXH64 inside libzstd in the reference): checksum (side stream) and digest
        // next to each other
        ZHIP(hipStreamWaitEvent(sb, ev[5], 0));
        ZHIP(hipEventRecord(ev[6], sb));
        hipLaunchKernelGGL(zarc_xxh64, dim3((unsigned)((ng * 4 + 255) / 256)), dim3(256), 0, sb, (const uint8_t *)d_dst_base, h->d_dst_off.as<uint64_t>() + f0,
                           h->d_raw_len.as<uint64_t>() + f0, (uint32_t)ng, h->d_xxh.as<uint64_t>() + f0);
        ZHIP(hipGetLastError());
        ZHIP(hipEventRecord(ev[7], sb));
        ZHIP(hipEventRecord(ev[8], sa));
        {
            const uint64_t chunks = gchunk0[g + 1] - gchunk0[g];
            hipLaunchKernelGGL(zarc_blake3_chunks, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, sa, (const uint8_t *)d_dst_base, h->d_dst_off.as<uint64_t>() + f0,
                               h->d_raw_len.as<uint64_t>() + f0, h->d_chunk_prefix.as<uint64_t>() + f0 + (size_t)g, (uint32_t)ng, chunks,
                               h->d_cvs.as<uint32_t>() + gchunk0[g] * 8, h->d_digests.as<uint32_t>() + f0 * 8);
            hipLaunchKernelGGL(zarc_blake3_tree, dim3((unsigned)std::min<size_t>(ng, 65535)), dim3(256), 0, sa, h->d_chunk_prefix.as<uint64_t>() + f0 + (size_t)g, (uint32_t)ng,
                               h->d_cvs.as<uint32_t>() + gchunk0[g] * 8, h->d_cvs_tmp.as<uint32_t>() + gchunk0[g] * 8, h->d_digests.as<uint32_t>() + f0 * 8);
            ZHIP(hipGetLastError());
        }
        ZHIP(hipEventRecord(ev[9], sa));
        piece_base += pieces[g].size();
    }
    // join: the decode time ends with the last group's frame pass, the whole call with the last hash
    for (int g = 0; g < groups; g++) if (groups > 1) ZHIP(hipStreamWaitEvent(h->stream, h->ev_g[g][5], 0));
    ZHIP(t.mark(&e1));
    for (int g = 0; g < groups; g++) {
        ZHIP(hipStreamWaitEvent(h->stream, h->ev_g[g][7], 0));
        if (groups > 1) ZHIP(hipStreamWaitEvent(h->stream, h->ev_g[g][9], 0));
    }
    ZHIP(t.mark(&e3));
    hipLaunchKernelGGL(zarc_unpack_verdict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (uint32_t)n, h->d_xxh.as<uint64_t>(),
                       h->d_stored_ck.as<uint32_t>(), h->d_digests.as<uint32_t>(), expect_in ? h->d_expect.as<uint32_t>() : (const uint32_t *)nullptr,
                       h->d_status.as<int32_t>());
    ZHIP(hipGetLastError());
    std::vector<int32_t> st_v;
    std::vector<uint8_t> dg_v;
    int32_t *st = (int32_t *)meta_take(h, n * 4); // results come back through the page-locked arena too, then go to the caller's order
    uint8_t *dg = (uint8_t *)meta_take(h, n * 32);
    if (!st) { st_v.resize(n); st = st_v.data(); }
    if (!dg) { dg_v.resize(n * 32); dg = dg_v.data(); }
    ZHIP(hipMemcpyAsync(st, h->d_status.p, n * 4, hipMemcpyDeviceToHost, h->stream));
    ZHIP(hipMemcpyAsync(dg, h->d_digests.p, n * 32, hipMemcpyDeviceToHost, h->stream));
    ZHIP(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < n; i++) { status[order[i]] = st[i]; memcpy(digest + (size_t)order[i] * 32, dg + i * 32, 32); }
    if (fastpath && diag_env("ZARC_GPU_DEC_STATS", 0)) { // diagnostics: how many frames had their sequences decoded ahead
        std::vector<uint32_t> fl(n);
        ZHIP(hipMemcpy(fl.data(), h->d_fast.p, n * 4, hipMemcpyDeviceToHost));
        size_t nf = 0;
        for (uint32_t v : fl) nf += v != 0;
        size_t np = 0;
        for (int g = 0; g < groups; g++) np += pieces[g].size();
        fprintf(stderr, "zstd_decode: %zu of %zu frames on the fast path, %d group(s), %zu piece(s)\n", nf, n, groups, np);
        if (h->d_seqflag.p) {
            std::vector<uint32_t> wf(nslots / 16 + 1);
            ZHIP(hipMemcpy(wf.data(), h->d_seqflag.p, wf.size() * 4, hipMemcpyDeviceToHost));
            size_t nw = 0;
            for (uint32_t v : wf) nw += v != 0;
            fprintf(stderr, "zdec_seqs: %zu of %zu 64-slot waves workgroups turned down by the shared-table kernel (of at most that many; one group: exact)\n", nw, wf.size());
        }
    }
    h->ms[ZARC_GPU_T_DECODE] = elapsed(h, e0, e1);
    // per-kernel times: sums over the groups (with several groups the kernels of different groups overlap, so the sums exceed T_DECODE)
    float t_seq = 0, t_lit = 0, t_frm = 0, t_xxh = 0, t_b3 = 0;
    for (int g = 0; g < groups; g++) {
        float ms = 0;
        hipEvent_t *ev = h->ev_g[g];
        if (have_seq_t[g] && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) t_seq += ms;
        if (have_lit_t[g] && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) t_lit += ms;
        if (hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess) t_frm += ms;
        if (hipEventElapsedTime(&ms, ev[6], ev[7]) == hipSuccess) t_xxh += ms;
        if (hipEventElapsedTime(&ms, ev[8], ev[9]) == hipSuccess) t_b3 += ms;
    }
    h->ms[ZARC_GPU_T_DEC_SEQS] = t_seq; h->ms[ZARC_GPU_T_DEC_LITS] = t_lit; h->ms[ZARC_GPU_T_DEC_FRAMES] = t_frm;
    h->ms[ZARC_GPU_T_XXH64] = t_xxh; // side stream: overlaps the digest pass
    h->ms[ZARC_GPU_T_BLAKE3] = t_b3;
    h->ms[ZARC_GPU_T_TOTAL] = elapsed(h, e0, e3);
    return ZARC_GPU_OK;
}

// A batch whose decoder scratch (sequences and literals decoded ahead, tables) does not fit -- the budget of ZARC_GPU_PX_SCRATCH_MB, or
// the device itself -- is unpacked in two halves, each of which may split again; the scratch is reused between them.  Frames are
// independent, so nothing changes but the time.
int unpack_device_split(zarc_gpu_t *h, size_t n, const void *d_frames_base, const uint64_t *frame_off, const uint64_t *frame_len,
                        void *d_dst_base, const uint64_t *dst_off, const uint64_t *raw_len, const uint8_t *expect, uint8_t *digest, int *status)
{
    uint64_t total = 0, acc = 0;
    for (size_t i = 0; i < n; i++) total += raw_len[i];
    int rc = UNPACK_SPLIT;
    if (!(n >= 2 && h->dec_split_above && total >= h->dec_split_above)) // (a batch of this size ran out of device memory before: do not try again)
        rc = unpack_device_once(h, n, d_frames_base, frame_off, frame_len, d_dst_base, dst_off, raw_len, expect, digest, status);
    if ((rc != UNPACK_SPLIT && rc != ZARC_GPU_E_NOMEM) || n < 2) return rc == UNPACK_SPLIT ? ZARC_GPU_E_NOMEM : rc;
    (void)hipStreamSynchronize(h->stream);
    if (rc == ZARC_GPU_E_NOMEM) {
        DevBuf *big[] = {&h->d_seqs, &h->d_lits, &h->d_ztables, &h->d_zblocks, &h->d_cvs, &h->d_cvs_tmp, &h->d_seqidx, &h->d_litidx, &h->d_nseq};
        for (DevBuf *b : big) b->release();
        if (!h->dec_split_above || total < h->dec_split_above) h->dec_split_above = total;
    }
    size_t k = 0;
    while (k + 1 < n && (acc + raw_len[k]) * 2 <= total) acc += raw_len[k++];
    if (k == 0) k = 1;
    float ms[ZARC_GPU_T_COUNT];
    const size_t part[3] = {0, k, n};
    for (int i = 0; i < ZARC_GPU_T_COUNT; i++) ms[i] = 0;
    for (int p = 0; p < 2; p++) {
        const size_t a = part[p], m = part[p + 1] - a;
        rc = unpack_device_split(h, m, d_frames_base, frame_off + a, frame_len + a, d_dst_base, dst_off + a, raw_len + a, expect ? expect + a * 32 : nullptr,
                                 digest + a * 32, status + a);
        if (rc) return rc;
        for (int i = 0; i < ZARC_GPU_T_COUNT; i++) ms[i] += h->ms[i];
    }
    for (int i = 0; i < ZARC_GPU_T_COUNT; i++) h->ms[i] = ms[i];
    return ZARC_GPU_OK;
}
} // namespace
extern "C" {

int zarc_gpu_unpack_batch_device(zarc_gpu_t *h, size_t n, const void *d_frames_base, const uint64_t *frame_off, const uint64_t *frame_len,
                                 void *d_dst_base, const uint64_t *dst_off, const uint64_t *raw_len, const uint8_t *expect, uint8_t *digest,
                                 int *status)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!d_frames_base || !frame_off || !frame_len || !d_dst_base || !dst_off || !raw_len || !digest || !status) return ZARC_GPU_E_PARAM;
    return unpack_device_split(h, n, d_frames_base, frame_off, frame_len, d_dst_base, dst_off, raw_len, expect, digest, status);
}

// ---- host-memory entry points: stage through engine-owned arenas -----------------------------------
// ---- host-pointer entry points: chunked, with the PCIe copies of neighbouring chunks overlapped -----------------------
// The caller's buffers are ordinary pageable memory, so hipMemcpyAsync blocks the calling thread while it stages the
// bytes.  A helper thread therefore moves chunk c+1 in and chunk c-1 out (side stream) while this thread runs the
// kernels of chunk c (engine stream); arenas are double-buffered, which also bounds the staging memory of huge batches.
// Chunks are large (2 GiB of content for pack, 4 GiB for unpack): the kernels need thousands of frames in flight to fill the
// chip (a 1 MiB frame is a ~10 ms serial chain for one workgroup / wave), measured: 256 MiB chunks halve the throughput.
// SURVEY.md 8 row f4.
} // extern "C"
namespace {

// A byte range of the caller's memory and where it sits in a flat device range
struct Seg { uint8_t *host; uint64_t dev; uint64_t len; };

int pin_ring(zarc_gpu *h)
{
    for (int i = 0; i < zarc_gpu::PIN_SLOTS; i++) {
        if (!h->pin[i] && hipHostMalloc((void **)&h->pin[i], zarc_gpu::PIN_PIECE, 0) != hipSuccess) { h->pin[i] = nullptr; return ZARC_GPU_E_NOMEM; }
        if (!h->pin_ev[i] && hipEventCreate(&h->pin_ev[i]) != hipSuccess) return ZARC_GPU_E_DEVICE;
    }
    return 0;
}

// copy the parts of `segs` (sorted by .dev, non-overlapping) that fall into device range [lo, hi) between the caller's memory
// and `pinned` (which mirrors [lo, hi)), with several threads: the single-thread memcpy rate is far below PCIe
void piece_copy(const std::vector<Seg> &segs, size_t first_seg, uint64_t lo, uint64_t hi, uint8_t *pinned, bool to_pinned, unsigned want)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 2 ? 1 : (nt > want ? want : nt);
    if (hi - lo < ((uint64_t)1 << 20)) nt = 1;
    auto work = [&](uint64_t a, uint64_t b) {
        for (size_t k = first_seg; k < segs.size() && segs[k].dev < b; k++) {
            const uint64_t s0 = std::max(segs[k].dev, a), s1 = std::min(segs[k].dev + segs[k].len, b);
            if (s0 >= s1) continue;
            if (to_pinned) memcpy(pinned + (s0 - lo), segs[k].host + (s0 - segs[k].dev), s1 - s0);
            else memcpy(segs[k].host + (s0 - segs[k].dev), pinned + (s0 - lo), s1 - s0);
        }
    };
    if (nt == 1) { work(lo, hi); return; }
    std::vector<std::thread> th;
    const uint64_t step = (hi - lo + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t a = lo + t * step, b = std::min(hi, a + step);
        if (a < b) th.emplace_back(work, a, b);
    }
    for (auto &x : th) x.join();
}

// Is every byte of every segment page-locked host memory the device can reach (hipHostMalloc / hipHostRegister)?  Then the DMA engines
// can read and write the caller's buffers themselves and the staging ring -- one more pass of host memcpy over every byte, which is
// what bounds the pageable path at about half the link rate -- is skipped.  One attribute query per segment boundary: segments that
// follow each other in host memory share the answer of the byte between them.
bool segs_pinned(const std::vector<Seg> &segs, uint64_t min_run)
{
    if (segs.empty()) return false;
    { // A DMA per small scattered buffer loses to one pass through the ring (measured: 8192 frames of 0.5 MiB each into separate slots,
      // 26 GiB/s direct against 31 staged): go direct only when the runs that are contiguous on both sides average 4 MiB
        uint64_t bytes = 0, runs = 0;
        for (size_t k = 0; k < segs.size(); k++) {
            bytes += segs[k].len;
            if (k == 0 || segs[k].host != segs[k - 1].host + segs[k - 1].len || segs[k].dev != segs[k - 1].dev + segs[k - 1].len) runs++;
        }
        if (bytes / runs < min_run) return false;
    }
    auto pinned = [](const uint8_t *p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; } // ordinary memory: an error, which must not stay behind as the "last error"
        return a.type == hipMemoryTypeHost;
    };
    const uint8_t *verified_end = nullptr; // one past a byte known to be pinned
    for (const Seg &sg : segs) {
        if (sg.host != verified_end && !pinned(sg.host)) return false;
        if (!pinned(sg.host + sg.len - 1)) return false;
        verified_end = sg.host + sg.len;
    }
    return true;
}

// pinned caller memory <-> device range at `dev_base`: one asynchronous copy per run of segments that are contiguous on both sides
int direct_copy(zarc_gpu *h, hipStream_t stream, const std::vector<Seg> &segs, uint8_t *dev_base, bool to_device)
{
    size_t k = 0;
    while (k < segs.size()) {
        size_t e = k + 1;
        uint64_t len = segs[k].len;
        while (e < segs.size() && segs[e].host == segs[k].host + len && segs[e].dev == segs[k].dev + len && len + segs[e].len < ((uint64_t)1 << 31)) { len += segs[e].len; e++; }
        // A run may span two page-locked allocations that happen to be adjacent, or a buffer with an unpinned hole (segs_pinned looks at
        // the first and the last byte of a segment): the runtime may refuse such a copy.  That is no device error: the caller falls
        // back to the staging ring, which redoes the whole range (copies already queued here move the same bytes).
        const hipError_t e_ = to_device ? hipMemcpyAsync(dev_base + segs[k].dev, segs[k].host, len, hipMemcpyHostToDevice, stream)
                                        : hipMemcpyAsync(segs[k].host, dev_base + segs[k].dev, len, hipMemcpyDeviceToHost, stream);
        if (e_ != hipSuccess) { (void)hipGetLastError(); (void)h; return 1; }
        k = e;
    }
    return 0;
}

// caller memory -> device range [0, total) at `dev_base`, through the pinned ring on `stream` (or straight from pinned caller memory)
int staged_h2d(zarc_gpu *h, hipStream_t stream, const std::vector<Seg> &segs, uint8_t *dev_base, uint64_t total)
{
    if (h->zero_copy && segs_pinned(segs, (uint64_t)h->zero_copy << 10) && direct_copy(h, stream, segs, dev_base, true) == 0) return 0;
    int rc = pin_ring(h);
    if (rc) return rc;
    size_t first = 0;
    int slot = 0;
    for (uint64_t lo = 0; lo < total; lo += zarc_gpu::PIN_PIECE, slot = (slot + 1) % 4) { // slots 0..3
        const uint64_t hi = std::min<uint64_t>(total, lo + zarc_gpu::PIN_PIECE);
        while (first < segs.size() && segs[first].dev + segs[first].len <= lo) first++;
        ZHIP(hipEventSynchronize(h->pin_ev[slot])); // the transfer that last used this slot is over
        piece_copy(segs, first, lo, hi, h->pin[slot], true, (unsigned)h->copy_threads);
        ZHIP(hipMemcpyAsync(dev_base + lo, h->pin[slot], hi - lo, hipMemcpyHostToDevice, stream));
        ZHIP(hipEventRecord(h->pin_ev[slot], stream));
    }
    return 0;
}

// device range [0, total) at `dev_base` -> caller memory; the scatter of piece p overlaps the transfer of piece p+1
int staged_d2h(zarc_gpu *h, hipStream_t stream, const std::vector<Seg> &segs, const uint8_t *dev_base, uint64_t total)
{
    if (h->zero_copy && segs_pinned(segs, (uint64_t)h->zero_copy << 10)) { // the callers synchronise the stream before they hand the buffers back
        if (direct_copy(h, stream, segs, (uint8_t *)dev_base, false) == 0) {
            ZHIP(hipStreamSynchronize(stream));
            return 0;
        }
    }
    int rc = pin_ring(h);
    if (rc) return rc;
    struct Piece { uint64_t lo, hi; int slot; size_t first; };
    size_t first = 0;
    int slot = 4;
    bool have_prev = false;
    Piece prev{};
    for (uint64_t lo = 0; lo < total; lo += zarc_gpu::PIN_PIECE, slot = 4 + ((slot - 3) & 1)) { // slots 4 and 5 alternate
        const uint64_t hi = std::min<uint64_t>(total, lo + zarc_gpu::PIN_PIECE);
        while (first < segs.size() && segs[first].dev + segs[first].len <= lo) first++;
        ZHIP(hipMemcpyAsync(h->pin[slot], dev_base + lo, hi - lo, hipMemcpyDeviceToHost, stream));
        ZHIP(hipEventRecord(h->pin_ev[slot], stream));
        if (have_prev) { ZHIP(hipEventSynchronize(h->pin_ev[prev.slot])); piece_copy(segs, prev.first, prev.lo, prev.hi, h->pin[prev.slot], false, (unsigned)h->copy_threads); }
        prev = Piece{lo, hi, slot, first};
        have_prev = true;
    }
    if (have_prev) { ZHIP(hipEventSynchronize(h->pin_ev[prev.slot])); piece_copy(segs, prev.first, prev.lo, prev.hi, h->pin[prev.slot], false, (unsigned)h->copy_threads); }
    return 0;
}

struct Chunk { size_t i0, i1; uint64_t in_bytes, out_bytes; };

// cut [0, n) into chunks of about STAGE_CHUNK content bytes (at least one entry each)
std::vector<Chunk> make_chunks(size_t n, const std::vector<uint64_t> &in_sz, const std::vector<uint64_t> &out_sz, const std::vector<uint64_t> &weight,
                               uint64_t STAGE_CHUNK, bool ramp = false)
{
    // ramp: the first chunk's way in and the last chunk's way out overlap nothing, so a batch of several chunks starts and ends with a
    // quarter-sized one (the kernels lose efficiency on it, the pipeline fills and drains four times sooner)
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += weight[i];
    ramp = ramp && total >= 3 * STAGE_CHUNK;
    const uint64_t small = STAGE_CHUNK / 4;
    std::vector<Chunk> cs;
    size_t i = 0;
    uint64_t left = total;
    while (i < n) {
        Chunk c{i, i, 0, 0};
        uint64_t w = 0, target = STAGE_CHUNK;
        if (ramp) {
            if (cs.empty()) target = small;
            else if (left <= small + small / 2) target = left;                       // the last, small chunk
            else if (left <= STAGE_CHUNK + small) target = left - small;             // the one before it leaves a quarter behind
        }
        while (c.i1 < n && (c.i1 == c.i0 || w + weight[c.i1] <= target)) { w += weight[c.i1]; c.in_bytes += in_sz[c.i1]; c.out_bytes += out_sz[c.i1]; c.i1++; }
        cs.push_back(c);
        i = c.i1;
        left -= w;
    }
    return cs;
}

} // namespace
extern "C" {

int zarc_gpu_blake3_batch(zarc_gpu_t *h, size_t n, const void *const *src, const size_t *len, uint8_t (*digest)[ZARC_GPU_DIGEST_LEN])
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!src || !len || !digest) return ZARC_GPU_E_PARAM;
    std::vector<uint64_t> off(n), l64(n);
    std::vector<Seg> segs;
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (len[i] && !src[i]) return ZARC_GPU_E_PARAM;
        off[i] = total; l64[i] = len[i];
        if (len[i]) segs.push_back(Seg{(uint8_t *)src[i], total, len[i]});
        total += align_up(len[i], ZARC_GPU_ALIGN);
    }
    ZHIP(h->d_arena_in.reserve(total + ZARC_GPU_PAD + 256));
    if ((rc = staged_h2d(h, h->stream, segs, h->d_arena_in.as<uint8_t>(), total))) return rc;
    return zarc_gpu_blake3_batch_device(h, n, h->d_arena_in.p, off.data(), l64.data(), (uint8_t *)digest);
}

int zarc_gpu_pack_batch(zarc_gpu_t *h, size_t n, const void *const *src, const size_t *src_len, void *dst, size_t dst_cap, size_t *dst_off,
                        size_t *dst_len, uint8_t (*digest)[ZARC_GPU_DIGEST_LEN], int *status)
{
    return zarc_gpu_pack_batch_dedup(h, n, src, src_len, dst, dst_cap, dst_off, dst_len, digest, status, nullptr, nullptr);
}

int zarc_gpu_pack_batch_dedup(zarc_gpu_t *h, size_t n, const void *const *src, const size_t *src_len, void *dst, size_t dst_cap, size_t *dst_off,
                              size_t *dst_len, uint8_t (*digest)[ZARC_GPU_DIGEST_LEN], int *status, zarc_gpu_known_fn known, void *ctx)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!src || !src_len || !dst || !dst_off || !dst_len || !digest || (known && !status)) return ZARC_GPU_E_PARAM;
    std::vector<uint64_t> in_sz(n), out_sz(n), l64(n);
    uint64_t need = 0;
    for (size_t i = 0; i < n; i++) {
        if (src_len[i] && !src[i]) return ZARC_GPU_E_PARAM;
        if ((uint64_t)src_len[i] >= 0xFFFFFFF0ull) { set_error(h, "entries of 4 GiB or more are not supported"); return ZARC_GPU_E_UNSUPPORTED; } // before any size arithmetic or staging
        l64[i] = src_len[i];
        in_sz[i] = align_up(src_len[i], ZARC_GPU_ALIGN);
        out_sz[i] = zarc_gpu_bound(src_len[i]);
        dst_off[i] = (size_t)need; // the caller's buffer uses the same slot layout as the device arena
        need += out_sz[i];
    }
    if (need > dst_cap) return ZARC_GPU_E_DSTSIZE;
    const std::vector<Chunk> cs = make_chunks(n, in_sz, out_sz, in_sz, h->stage_chunk ? h->stage_chunk : (uint64_t)2 << 30, /*ramp=*/true); // measured: 1-2 GiB chunks pack fastest (27 vs 23 GiB/s at 4 GiB)
    uint64_t max_in = 0, max_out = 0;
    for (const Chunk &c : cs) { max_in = std::max(max_in, c.in_bytes); max_out = std::max(max_out, c.out_bytes); }
    const uint64_t in_half = align_up(max_in + ZARC_GPU_PAD + 256, 256), out_half = align_up(max_out + ZARC_GPU_PAD, 256);
    ZHIP(h->d_arena_in.reserve(2 * in_half));
    ZHIP(h->d_arena_out.reserve(2 * out_half));
    uint8_t *const ain = h->d_arena_in.as<uint8_t>(), *const aout = h->d_arena_out.as<uint8_t>();
    const int device = h->device;
    hipStream_t side = h->stream_stage, side_out = h->stream_stage_out;
    std::vector<uint64_t> doff(n), dlen(n);
    auto copy_in = [&](size_t c) -> int { // entries of chunk c -> input half c & 1
        std::vector<Seg> segs;
        uint64_t at = 0;
        for (size_t i = cs[c].i0; i < cs[c].i1; i++) { if (src_len[i]) segs.push_back(Seg{(uint8_t *)src[i], at, src_len[i]}); at += in_sz[i]; }
        return staged_h2d(h, side, segs, ain + (c & 1) * in_half, at);
    };
    auto copy_out = [&](size_t c) -> int { // frames of chunk c (lengths known): packed back to back on the device, then to the caller's slots
        const size_t m = cs[c].i1 - cs[c].i0, i0 = cs[c].i0;
        std::vector<uint64_t> dense(m);
        std::vector<Seg> segs;
        uint64_t at = 0;
        for (size_t k = 0; k < m; k++) { dense[k] = at; if (dlen[i0 + k]) segs.push_back(Seg{(uint8_t *)dst + dst_off[i0 + k], at, dlen[i0 + k]}); at += dlen[i0 + k]; }
        if (h->d_dense.reserve(at + 256) != hipSuccess || h->d_goff.reserve(m * 8) != hipSuccess || h->d_glen.reserve(m * 8) != hipSuccess ||
            h->d_gdense.reserve(m * 8) != hipSuccess) return ZARC_GPU_E_NOMEM;
        if (hipMemcpyAsync(h->d_goff.p, doff.data() + i0, m * 8, hipMemcpyHostToDevice, side_out) != hipSuccess ||
            hipMemcpyAsync(h->d_glen.p, dlen.data() + i0, m * 8, hipMemcpyHostToDevice, side_out) != hipSuccess ||
            hipMemcpyAsync(h->d_gdense.p, dense.data(), m * 8, hipMemcpyHostToDevice, side_out) != hipSuccess) return ZARC_GPU_E_DEVICE;
        hipLaunchKernelGGL(zarc_gather, dim3((unsigned)m), dim3(256), 0, side_out, aout + (c & 1) * out_half, h->d_goff.as<uint64_t>(), h->d_glen.as<uint64_t>(),
                           h->d_gdense.as<uint64_t>(), (uint32_t)m, h->d_dense.as<uint8_t>());
        if (hipGetLastError() != hipSuccess) return ZARC_GPU_E_DEVICE;
        const int r = staged_d2h(h, side_out, segs, h->d_dense.as<uint8_t>(), at);
        if (hipStreamSynchronize(side_out) != hipSuccess) return ZARC_GPU_E_DEVICE; // `dense` is read by an async copy above
        return r;
    };
    float sum[ZARC_GPU_T_COUNT] = {};
    if ((rc = copy_in(0))) return rc;
    ZHIP(hipStreamSynchronize(side));
    for (size_t c = 0; c < cs.size(); c++) {
        // chunk c+1 comes in and chunk c-1 goes out while chunk c is worked on: one helper thread and one stream per direction
        int helper_rc = 0, helper_rc_out = 0;
        auto move_in = [&] {
            (void)hipSetDevice(device);
            if (c + 1 < cs.size()) helper_rc = copy_in(c + 1);
            if (hipStreamSynchronize(side) != hipSuccess) helper_rc = ZARC_GPU_E_DEVICE;
        };
        auto move_out = [&] {
            (void)hipSetDevice(device);
            if (c > 0) helper_rc_out = copy_out(c - 1);
            if (hipStreamSynchronize(side_out) != hipSuccess) helper_rc_out = ZARC_GPU_E_DEVICE;
        };
        std::thread helper, helper_out;
        if (h->stage_thread) { helper = std::thread(move_in); helper_out = std::thread(move_out); } else { move_in(); move_out(); }
        const size_t m = cs[c].i1 - cs[c].i0, i0 = cs[c].i0;
        std::vector<uint64_t> off(m);
        uint64_t at = 0;
        for (size_t k = 0; k < m; k++) { off[k] = at; at += in_sz[i0 + k]; }
        // hash first when the caller can tell known content (the chunk is resident: the digest pass costs 0.5 ms per GiB): the callback sees
        // indices of the whole batch
        struct Shift { zarc_gpu_known_fn fn; void *ctx; size_t base; } shift{known, ctx, i0};
        auto shifted = [](void *c, const uint8_t *d, size_t i) -> int { Shift *s = (Shift *)c; return s->fn(s->ctx, d, s->base + i); };
        rc = pack_device_dedup_impl(h, m, ain + (c & 1) * in_half, off.data(), l64.data() + i0, aout + (c & 1) * out_half, cs[c].out_bytes, doff.data() + i0,
                                    dlen.data() + i0, (uint8_t *)digest[i0], status ? status + i0 : nullptr, known ? (zarc_gpu_known_fn)shifted : nullptr, &shift);
        if (helper.joinable()) helper.join();
        if (helper_out.joinable()) helper_out.join();
        if (rc) return rc;
        if (helper_rc) return helper_rc;
        if (helper_rc_out) return helper_rc_out;
        for (size_t k = 0; k < m; k++) dst_len[i0 + k] = (size_t)dlen[i0 + k];
        for (int t = 0; t < ZARC_GPU_T_COUNT; t++) sum[t] += h->ms[t] > 0 ? h->ms[t] : 0;
    }
    if ((rc = copy_out(cs.size() - 1))) return rc;
    ZHIP(hipStreamSynchronize(side_out));
    for (int t = 0; t < ZARC_GPU_T_COUNT; t++) h->ms[t] = sum[t];
    return ZARC_GPU_OK;
}

int zarc_gpu_unpack_batch(zarc_gpu_t *h, size_t n, const void *const *frame, const size_t *frame_len, const size_t *raw_len, void *const *dst,
                          const uint8_t (*expect)[ZARC_GPU_DIGEST_LEN], uint8_t (*digest)[ZARC_GPU_DIGEST_LEN], int *status)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!frame || !frame_len || !raw_len || !dst || !digest || !status) return ZARC_GPU_E_PARAM;
    std::vector<uint64_t> in_sz(n), out_sz(n), weight(n), flen(n), rlen(n);
    for (size_t i = 0; i < n; i++) {
        if ((frame_len[i] && !frame[i]) || (raw_len[i] && !dst[i])) return ZARC_GPU_E_PARAM;
        if ((uint64_t)frame_len[i] >= 0xFFFFFFF0ull || (uint64_t)raw_len[i] >= 0xFFFFFFF0ull) { set_error(h, "frames of 4 GiB or more are not supported"); return ZARC_GPU_E_UNSUPPORTED; } // before any size arithmetic or staging
        flen[i] = frame_len[i]; rlen[i] = raw_len[i];
        in_sz[i] = align_up(frame_len[i], ZARC_GPU_ALIGN);
        out_sz[i] = align_up(raw_len[i], ZARC_GPU_ALIGN);
        weight[i] = std::max(in_sz[i], out_sz[i]);
    }
    const std::vector<Chunk> cs = make_chunks(n, in_sz, out_sz, weight, h->stage_chunk ? h->stage_chunk : (uint64_t)4 << 30); // the decoder's lane-per-block stage wants many frames at once
    uint64_t max_in = 0, max_out = 0;
    for (const Chunk &c : cs) { max_in = std::max(max_in, c.in_bytes); max_out = std::max(max_out, c.out_bytes); }
    const uint64_t in_half = align_up(max_in + ZARC_GPU_PAD + 256, 256), out_half = align_up(max_out + ZARC_GPU_PAD + 256, 256);
    ZHIP(h->d_arena_in.reserve(2 * in_half));
    ZHIP(h->d_arena_out.reserve(2 * out_half));
    uint8_t *const ain = h->d_arena_in.as<uint8_t>(), *const aout = h->d_arena_out.as<uint8_t>();
    const int device = h->device;
    hipStream_t side = h->stream_stage, side_out = h->stream_stage_out;
    auto copy_in = [&](size_t c) -> int {
        std::vector<Seg> segs;
        uint64_t at = 0;
        for (size_t i = cs[c].i0; i < cs[c].i1; i++) { if (frame_len[i]) segs.push_back(Seg{(uint8_t *)frame[i], at, frame_len[i]}); at += in_sz[i]; }
        return staged_h2d(h, side, segs, ain + (c & 1) * in_half, at);
    };
    auto copy_out = [&](size_t c) -> int {
        std::vector<Seg> segs;
        uint64_t at = 0;
        for (size_t i = cs[c].i0; i < cs[c].i1; i++) {
            // like the reference, bytes are delivered unless the frame itself failed to decode
            const bool decoded = status[i] == ZARC_GPU_FRAME_OK || status[i] == ZARC_GPU_FRAME_DIGEST || status[i] == ZARC_GPU_FRAME_CHECKSUM;
            if (decoded && raw_len[i]) segs.push_back(Seg{(uint8_t *)dst[i], at, raw_len[i]});
            at += out_sz[i];
        }
        return staged_d2h(h, side_out, segs, aout + (c & 1) * out_half, at);
    };
    float sum[ZARC_GPU_T_COUNT] = {};
    if ((rc = copy_in(0))) return rc;
    ZHIP(hipStreamSynchronize(side));
    for (size_t c = 0; c < cs.size(); c++) {
        // chunk c+1 comes in and chunk c-1 goes out while chunk c is worked on: one helper thread and one stream per direction
        int helper_rc = 0, helper_rc_out = 0;
        auto move_in = [&] {
            (void)hipSetDevice(device);
            if (c + 1 < cs.size()) helper_rc = copy_in(c + 1);
            if (hipStreamSynchronize(side) != hipSuccess) helper_rc = ZARC_GPU_E_DEVICE;
        };
        auto move_out = [&] {
            (void)hipSetDevice(device);
            if (c > 0) helper_rc_out = copy_out(c - 1);
            if (hipStreamSynchronize(side_out) != hipSuccess) helper_rc_out = ZARC_GPU_E_DEVICE;
        };
        std::thread helper, helper_out;
        if (h->stage_thread) { helper = std::thread(move_in); helper_out = std::thread(move_out); } else { move_in(); move_out(); }
        const size_t m = cs[c].i1 - cs[c].i0, i0 = cs[c].i0;
        std::vector<uint64_t> foff(m), doff(m);
        uint64_t fa = 0, da = 0;
        for (size_t k = 0; k < m; k++) { foff[k] = fa; fa += in_sz[i0 + k]; doff[k] = da; da += out_sz[i0 + k]; }
        rc = zarc_gpu_unpack_batch_device(h, m, ain + (c & 1) * in_half, foff.data(), flen.data() + i0, aout + (c & 1) * out_half, doff.data(), rlen.data() + i0,
                                          expect ? (const uint8_t *)expect[i0] : nullptr, (uint8_t *)digest[i0], status + i0);
        if (helper.joinable()) helper.join();
        if (helper_out.joinable()) helper_out.join();
        if (rc) return rc;
        if (helper_rc) return helper_rc;
        if (helper_rc_out) return helper_rc_out;
        for (int t = 0; t < ZARC_GPU_T_COUNT; t++) sum[t] += h->ms[t] > 0 ? h->ms[t] : 0;
    }
    if ((rc = copy_out(cs.size() - 1))) return rc;
    ZHIP(hipStreamSynchronize(side_out));
    for (int t = 0; t < ZARC_GPU_T_COUNT; t++) h->ms[t] = sum[t];
    return ZARC_GPU_OK;
}

// ---- bench / test utilities ---------------------------------------------------------------------------
int zarc_gpu_corpus_fill_device(zarc_gpu_t *h, size_t n, void *d_base, const uint64_t *off, const uint64_t *len, uint64_t first_index, int kind)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (n == 0) return ZARC_GPU_OK;
    if (!d_base || !off || !len) return ZARC_GPU_E_PARAM;
    if ((rc = upload_u64(h, h->d_off, off, n))) return rc;
    if ((rc = upload_u64(h, h->d_len, len, n))) return rc;
    hipLaunchKernelGGL(zarc_corpus_fill, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, (uint8_t *)d_base, h->d_off.as<uint64_t>(),
                       h->d_len.as<uint64_t>(), (uint32_t)n, first_index, kind);
    ZHIP(hipGetLastError());
    ZHIP(hipStreamSynchronize(h->stream));
    return ZARC_GPU_OK;
}
int zarc_gpu_device_malloc(zarc_gpu_t *h, void **d_ptr, size_t bytes)
{
    if (!h || !d_ptr) return ZARC_GPU_E_PARAM;
    ZHIP(hipSetDevice(h->device));
    ZHIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return ZARC_GPU_OK;
}
int zarc_gpu_device_free(zarc_gpu_t *h, void *d_ptr)
{
    if (!h) return ZARC_GPU_E_PARAM;
    ZHIP(hipSetDevice(h->device));
    ZHIP(hipFree(d_ptr));
    return ZARC_GPU_OK;
}
int zarc_gpu_memcpy_h2d(zarc_gpu_t *h, void *d_dst, const void *src, size_t bytes)
{
    if (!h) return ZARC_GPU_E_PARAM;
    ZHIP(hipSetDevice(h->device));
    ZHIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return ZARC_GPU_OK;
}
int zarc_gpu_memcpy_d2h(zarc_gpu_t *h, void *dst, const void *d_src, size_t bytes)
{
    if (!h) return ZARC_GPU_E_PARAM;
    ZHIP(hipSetDevice(h->device));
    ZHIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ZARC_GPU_OK;
}

} // extern "C"
