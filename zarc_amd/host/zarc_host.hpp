// zarc_amd/host/zarc_host.hpp -- host-side mirror of the reference's library API for the content path,
// written above the C ABI (include/zarc_gpu.h).  The reference is Rust (no rustc/cargo on the build image), so
// this is the C++ stand-in for what a patched `crates/zarc` would do around the FFI calls; names, argument
// meaning and error behaviour follow the reference:
//
//   zarc::Encoder            crates/zarc/src/encode.rs:27-97 (struct + new/set_zstd_parameter/enable_compression)
//   Encoder::add_data_frame  crates/zarc/src/encode/content_frame.rs:20-60 (hash + compress -> dedup -> Frame record)
//   zarc::Frame              crates/zarc/src/directory/frame.rs:10-32
//   zarc::Digest             crates/zarc/src/integrity.rs:14-36 (constant-time equality :17-22)
//   zarc::FrameReader        crates/zarc/src/decode/frame_iterator.rs:14-104 (read_content_frame + verify)
//
// What differs, on purpose: the engine is batched, so `add_data_frames` takes many entries at once (frames
// are independent: a fresh session per frame, content_frame.rs:37-39).  Call order is preserved: frame order
// = call order, first occurrence of a digest wins, later duplicates write nothing (content_frame.rs:30-33),
// offsets are running sums starting at 12 (encode.rs:65,75).  Unlike the reference, writes use write-all
// semantics (the reference's `writer.write` can short-write, lowlevel_frames.rs:38 -- SURVEY quirk 1).
// The archive directory / trailer (`add_file_entry`, `finalise`, `open`) live in zarc_container.hpp.
//
// Several GPUs: frames are independent, so an Encoder constructed with G devices deals the entries of a
// batch to G engine handles (`shard_assign`: index mod G when all sizes are equal, largest-first onto the least loaded device
// otherwise), packs the shares concurrently (one host thread per handle, no collective, nothing crosses between devices) and
// then does in ORIGINAL index order exactly what the single-device path does: running offsets and first-wins dedup.  The
// archive is byte-identical to the single-device one.
#pragma once
#include "../../include/zarc_gpu.h"
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace zarc {

// header.rs:35-40: skippable-frame magic 0x184D2A50, length 4, ZARC_MAGIC 65 AA DC, version 1
static const uint8_t FILE_MAGIC[12] = {0x50, 0x2A, 0x4D, 0x18, 0x04, 0x00, 0x00, 0x00, 0x65, 0xAA, 0xDC, 0x01};

struct Digest {
    std::array<uint8_t, 32> bytes{};
    // constant-time equality (integrity.rs:17-22 uses subtle::ConstantTimeEq)
    bool operator==(const Digest &o) const
    {
        uint8_t d = 0;
        for (size_t i = 0; i < 32; i++) d |= (uint8_t)(bytes[i] ^ o.bytes[i]);
        return d == 0;
    }
    bool operator<(const Digest &o) const { return std::memcmp(bytes.data(), o.bytes.data(), 32) < 0; } // map key only
};

struct Frame { // directory/frame.rs:12-32
    uint16_t edition = 1;
    uint64_t offset = 0;
    Digest digest;
    uint64_t length = 0;       // compressed bytes in the archive
    uint64_t uncompressed = 0;
};

struct Error : std::runtime_error { // lib.rs:27-30: io::Error::other(ZSTD_getErrorName(code))
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

class Engine { // CCtx/DCtx analogue: one per Encoder / reader (encode.rs:60-62, zstd_iterator.rs:29)
  public:
    explicit Engine(int device = 0)
    {
        int rc = zarc_gpu_create(&h_, device);
        if (rc != ZARC_GPU_OK) throw Error(rc, "failed allocating zstd context"); // encode.rs:61 / ErrorKind::ZstdInit
    }
    ~Engine() { zarc_gpu_destroy(h_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    zarc_gpu_t *get() const { return h_; }
    void check(int rc) const
    {
        if (rc != ZARC_GPU_OK) throw Error(rc, std::string(zarc_gpu_error_name(rc)) + ": " + zarc_gpu_last_error(h_));
    }

  private:
    zarc_gpu_t *h_ = nullptr;
};

// Which device packs which entry.  Equal sizes: entry i goes to device i mod G.  Mixed sizes (BASELINE
// configs[4]): largest first onto the least loaded device by bytes (ties: the lower device), each share then back in index
// order.  Deterministic; zarc_amd/shard.py is the same function for bench.py and the tests.
inline std::vector<std::vector<size_t>> shard_assign(const size_t *len, size_t n, size_t g)
{
    std::vector<std::vector<size_t>> out(g ? g : 1);
    if (g <= 1) { out[0].resize(n); std::iota(out[0].begin(), out[0].end(), (size_t)0); return out; }
    bool equal = true;
    for (size_t i = 1; i < n; i++) equal = equal && len[i] == len[0];
    if (equal) { for (size_t i = 0; i < n; i++) out[i % g].push_back(i); return out; }
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return len[a] > len[b]; });
    std::vector<uint64_t> load(g, 0);
    for (size_t i : order) {
        size_t best = 0;
        for (size_t d = 1; d < g; d++) if (load[d] < load[best]) best = d;
        out[best].push_back(i);
        load[best] += len[i];
    }
    for (auto &v : out) std::sort(v.begin(), v.end());
    return out;
}

// Host-thread budget of G handles working at once (`--gpus G`): every handle's host-pointer entry points fill / drain their pinned
// staging ring with ZARC_GPU_PX_COPY_THREADS threads (default 8) beside two helper threads and the caller's own.  Eight handles at the
// default would be 64 copy threads on one host for a PCIe complex that 16 saturate: the total is capped at 16 (at least 2 per handle).
inline void cap_copy_threads(const std::vector<std::unique_ptr<Engine>> &engines)
{
    const size_t g = engines.size();
    if (g <= 2) return;
    const int per = (int)std::max<size_t>(2, 16 / g);
    for (auto &e : engines) e->check(zarc_gpu_set_parameter(e->get(), ZARC_GPU_PX_COPY_THREADS, per));
}

class Encoder {
  public:
    // Encoder::new: creates the context and writes the 12-byte header (encode.rs:58-78)
    explicit Encoder(std::ostream &writer, int device = 0) : Encoder(writer, std::vector<int>{device}) {}
    // ... on several devices: one engine handle each (`zarc pack --gpus N`)
    Encoder(std::ostream &writer, const std::vector<int> &devices) : writer_(writer)
    {
        if (devices.empty()) throw Error(ZARC_GPU_E_PARAM, "no device");
        for (int d : devices) engines_.emplace_back(new Engine(d));
        cap_copy_threads(engines_);
        writer_.write((const char *)FILE_MAGIC, sizeof FILE_MAGIC);
        offset_ = sizeof FILE_MAGIC;
    }
    size_t devices() const { return engines_.size(); }
    // Encoder::set_zstd_parameter -- sticky for future frames (encode.rs:84-89); id = ZSTD_cParameter value
    void set_zstd_parameter(int id, int value) { for (auto &e : engines_) e->check(zarc_gpu_set_parameter(e->get(), id, value)); }
    // Encoder::enable_compression (encode.rs:95-97)
    void enable_compression(bool compress) { for (auto &e : engines_) zarc_gpu_enable_compression(e->get(), compress ? 1 : 0); }

    // Encoder::add_data_frame for one entry (content_frame.rs:20)
    Digest add_data_frame(const uint8_t *content, size_t len)
    {
        const void *p = content;
        return add_data_frames(&p, &len, 1)[0];
    }

    // Batched add_data_frame: returns the digest of every entry in call order.
    // Hash first, like the reference (content_frame.rs:26-33): the engine digests the batch, asks `claim` below for every entry, and
    // compresses only content that neither an earlier call nor an earlier entry of this batch has brought -- on the reference's own
    // benchmark tree (half of the bytes are duplicates) that halves the work.  With several devices the claims of one digest may come
    // in any order; whichever device compresses it, the bytes are the same (the encoder is deterministic), and they are written where
    // the FIRST entry with that digest stands in call order.
    std::vector<Digest> add_data_frames(const void *const *content, const size_t *len, size_t n)
    {
        std::vector<Digest> digests(n);
        if (n == 0) return digests;
        // 1. one fresh session per frame (reset(SessionOnly), content_frame.rs:37-39) == one independent frame each.  The batch
        //    is dealt to the devices; every device packs its share into its own buffer, concurrently, with no exchange.
        const size_t g = engines_.size();
        const auto share = shard_assign(len, n, g);
        std::vector<std::vector<uint8_t>> buffers(g);
        std::vector<int> status(n, ZARC_GPU_FRAME_OK);
        std::vector<int> rc(g, ZARC_GPU_OK);
        struct Made { const uint8_t *at; size_t len; };
        std::map<Digest, Made> made;   // digest -> the frame some device compressed in this call
        struct Claims { std::mutex mu; std::map<Digest, int> taken; const std::map<Digest, Frame> *written; } claims;
        claims.written = &frames_;
        auto claim = [](void *ctx, const uint8_t *d, size_t) -> int { // "frame already exists, skipping" (content_frame.rs:30-33)
            Claims *c = (Claims *)ctx;
            Digest dg;
            std::memcpy(dg.bytes.data(), d, 32);
            std::lock_guard<std::mutex> lk(c->mu);
            if (c->written->count(dg)) return 1;
            return c->taken.emplace(dg, 1).second ? 0 : 1;
        };
        std::mutex made_mu;
        auto pack_share = [&](size_t d) {
            const std::vector<size_t> &idx = share[d];
            const size_t m = idx.size();
            if (m == 0) return;
            std::vector<const void *> src(m);
            std::vector<size_t> l(m), off(m), out_len(m);
            std::vector<Digest> dig(m);
            std::vector<int> st(m);
            size_t cap = 0;
            for (size_t j = 0; j < m; j++) { src[j] = content[idx[j]]; l[j] = len[idx[j]]; cap += zarc_gpu_bound(l[j]); }
            buffers[d].resize(cap);
            rc[d] = zarc_gpu_pack_batch_dedup(engines_[d]->get(), m, src.data(), l.data(), buffers[d].data(), cap, off.data(), out_len.data(),
                                              (uint8_t(*)[32])dig.data(), st.data(), hash_first_ ? (zarc_gpu_known_fn)claim : nullptr, &claims);
            if (rc[d] != ZARC_GPU_OK) return;
            std::lock_guard<std::mutex> lk(made_mu);
            for (size_t j = 0; j < m; j++) {
                digests[idx[j]] = dig[j];
                status[idx[j]] = st[j];
                if (st[j] == ZARC_GPU_FRAME_OK) made.emplace(dig[j], Made{buffers[d].data() + off[j], out_len[j]}); // (without hash-first: the first copy inserted stays, all copies are equal)
            }
        };
        if (g == 1) pack_share(0);
        else {
            std::vector<std::thread> th;
            for (size_t d = 0; d < g; d++) th.emplace_back(pack_share, d);
            for (auto &t : th) t.join();
        }
        for (size_t d = 0; d < g; d++) engines_[d]->check(rc[d]);
        // 2. append in call order -- whichever device made the frame; first occurrence wins -- against frames already written and
        //    inside this batch, across devices -- later duplicates write nothing; offsets are the running sum of the lengths of
        //    what was written (content_frame.rs:22,45-57)
        for (size_t k = 0; k < n; k++) {
            if (status[k] != ZARC_GPU_FRAME_OK && status[k] != ZARC_GPU_FRAME_DUPLICATE) throw Error(status[k], zarc_gpu_frame_status_name(status[k]));
            if (frames_.count(digests[k])) continue; // "frame already exists, skipping"
            const auto it = made.find(digests[k]);
            if (it == made.end()) throw Error(ZARC_GPU_E_DEVICE, "internal: no frame was made for a new digest");
            Frame f;
            f.edition = edition_;
            f.offset = offset_;
            f.digest = digests[k];
            f.length = it->second.len;
            f.uncompressed = len[k];
            writer_.write((const char *)it->second.at, (std::streamsize)it->second.len);
            if (!writer_) throw Error(ZARC_GPU_E_DEVICE, "write failed");
            offset_ += it->second.len;
            frames_.emplace(f.digest, f);
            order_.push_back(f.digest);
        }
        return digests;
    }
    // hash-first dedup (default on; off = every entry is compressed and duplicates are dropped afterwards: same archive, more work)
    void set_hash_first(bool on) { hash_first_ = on; }

    const std::map<Digest, Frame> &frames() const { return frames_; }
    const std::vector<Digest> &frame_order() const { return order_; } // insertion order (the reference uses a HashMap)
    uint64_t offset() const { return offset_; }

  protected: // ArchiveWriter (zarc_container.hpp) adds add_file_entry / finalise on top
    std::ostream &writer_;
    std::vector<std::unique_ptr<Engine>> engines_; // one per device; engines_[0] also serves the directory frame / digest
    Engine &engine0() { return *engines_[0]; }
    uint16_t edition_ = 1;
    bool hash_first_ = true;
    std::map<Digest, Frame> frames_;
    std::vector<Digest> order_;
    uint64_t offset_ = 0;
};

// Decoder::read_content_frame + FrameIterator for a batch of frames of one archive image held in memory.
// The reference's read side is a serial loop over frames that each get a fresh reader and DCtx (zarc-cli/src/unpack.rs:62-88,
// decode/zstd_iterator.rs:28-29): frames are as independent on the way out as on the way in.  A FrameReader constructed with G
// devices deals the wanted frames with the same `shard_assign` as the Encoder -- by UNCOMPRESSED bytes, what the decoder's work
// is proportional to -- decodes the shares concurrently (one host thread per handle, no collective) and returns the results in the
// caller's order; statuses, digests and bytes are identical to the single-device ones.
class FrameReader {
  public:
    struct Result {
        std::vector<uint8_t> data;       // the concatenation of what FrameIterator::next would yield
        Digest digest;                   // FrameIterator::digest() once the frame is done
        std::optional<bool> verify;      // FrameIterator::verify(): None if the frame did not decode
        int status = ZARC_GPU_FRAME_OK;  // Error::Zstd analogue (decode/error.rs:35-38) via zarc_gpu_frame_status_name
    };
    explicit FrameReader(int device = 0) : FrameReader(std::vector<int>{device}) {}
    explicit FrameReader(const std::vector<int> &devices)
    {
        if (devices.empty()) throw Error(ZARC_GPU_E_PARAM, "no device");
        for (int d : devices) engines_.emplace_back(new Engine(d));
        cap_copy_threads(engines_);
    }
    size_t devices() const { return engines_.size(); }

    // `archive` is the whole file; `wanted` are directory records (offset/length/uncompressed/digest).
    std::vector<Result> read_content_frames(const uint8_t *archive, size_t archive_len, const std::vector<Frame> &wanted)
    {
        const size_t n = wanted.size();
        std::vector<Result> out(n);
        std::vector<size_t> rl(n);
        for (size_t i = 0; i < n; i++) {
            // untrusted directory records: no u64 wrap-around, no frame beyond the file, no allocation the engine would refuse anyway
            if (wanted[i].length > archive_len || wanted[i].offset > archive_len - wanted[i].length) throw Error(ZARC_GPU_E_PARAM, "frame outside the archive");
            if (wanted[i].uncompressed >= 0xFFFFFFF0ull || wanted[i].length >= 0xFFFFFFF0ull) throw Error(ZARC_GPU_E_UNSUPPORTED, "frames of 4 GiB or more are not supported");
            // Zstandard cannot expand a frame by more than a factor of ~(128 KiB block from a 4-byte RLE block): a larger claim is corrupt
            if (wanted[i].uncompressed > (wanted[i].length + 16) * (uint64_t)65536) throw Error(ZARC_GPU_E_PARAM, "frame claims an impossible uncompressed size");
            out[i].data.resize(wanted[i].uncompressed);
            rl[i] = wanted[i].uncompressed;
        }
        if (n == 0) return out;
        const size_t g = engines_.size();
        const auto share = shard_assign(rl.data(), n, g);
        std::vector<int> rc(g, ZARC_GPU_OK);
        auto unpack_share = [&](size_t d) {
            const std::vector<size_t> &idx = share[d];
            const size_t m = idx.size();
            if (m == 0) return;
            std::vector<const void *> fp(m);
            std::vector<void *> dp(m);
            std::vector<size_t> fl(m), ul(m);
            std::vector<Digest> expect(m), got(m);
            std::vector<int> status(m);
            for (size_t j = 0; j < m; j++) {
                const Frame &f = wanted[idx[j]];
                fp[j] = archive + f.offset; fl[j] = f.length; ul[j] = f.uncompressed; dp[j] = out[idx[j]].data.data(); expect[j] = f.digest;
            }
            rc[d] = zarc_gpu_unpack_batch(engines_[d]->get(), m, fp.data(), fl.data(), ul.data(), dp.data(), (const uint8_t(*)[32])expect.data(),
                                          (uint8_t(*)[32])got.data(), status.data());
            if (rc[d] != ZARC_GPU_OK) return;
            for (size_t j = 0; j < m; j++) {
                Result &r = out[idx[j]];
                r.status = status[j];
                r.digest = got[j];
                const bool decoded = status[j] == ZARC_GPU_FRAME_OK || status[j] == ZARC_GPU_FRAME_DIGEST;
                if (decoded) r.verify = got[j] == expect[j]; // a mismatch is reported, not fatal (zarc-cli/src/unpack.rs:118-120)
                else r.data.clear();
            }
        };
        if (g == 1) unpack_share(0);
        else {
            std::vector<std::thread> th;
            for (size_t d = 0; d < g; d++) th.emplace_back(unpack_share, d);
            for (auto &t : th) t.join();
        }
        for (size_t d = 0; d < g; d++) engines_[d]->check(rc[d]);
        return out;
    }

  private:
    std::vector<std::unique_ptr<Engine>> engines_; // one per device
};

} // namespace zarc
