// zarc_amd/host/zarc_container.hpp -- the Zarc container around the content frames.
//
// Host plumbing only (no bulk data): header, directory (CBOR elements), trailer -- what the reference does in
//   crates/zarc/src/header.rs:35-40              FILE_MAGIC
//   crates/zarc/src/directory/elements.rs:10-25  ElementFrame {kind u8, len u16 LE, 1 pad byte, CBOR payload}
//   crates/zarc/src/directory/{edition.rs:11-34, file.rs:16-62, frame.rs:10-32, strings.rs, timestamps.rs, specials.rs}
//   crates/zarc/src/encode/add_file.rs:22-46     add_file_entry
//   crates/zarc/src/encode/directory.rs:40-122   finalise (element order, directory digest, directory frame, trailer)
//   crates/zarc/src/trailer.rs:51-203            Trailer / Epilogue (54-byte payload, XOR check byte)
//   crates/zarc/src/decode/open.rs:21-159        header check + trailer from the file tail
//   crates/zarc/src/decode/directory.rs:55-119   read_directory
// CBOR shapes follow minicbor-derive: `#[cbor(map)]` = definite map keyed by field index with absent options omitted,
// `#[cbor(array)]` = definite array with trailing None dropped, Timestamp = tag 0 + RFC 3339 text.
// The bulk work (directory digest, directory frame, content frames) goes through the engine like every other frame.
// Deliberate differences from the reference: the reader buffers the whole directory before parsing (the reference
// fails when an element straddles a 128 KiB chunk); leftover frames are written in insertion order.
#pragma once
#include "zarc_host.hpp"
#include <algorithm>
#include <ctime>

namespace zarc {

// ------------------------------------------------------------------ minimal CBOR -----------------------------
class CborWriter {
  public:
    std::vector<uint8_t> buf;
    void head(uint8_t major, uint64_t v)
    {
        const uint8_t m = (uint8_t)(major << 5);
        if (v < 24) buf.push_back((uint8_t)(m | v));
        else if (v <= 0xFF) { buf.push_back(m | 24); buf.push_back((uint8_t)v); }
        else if (v <= 0xFFFF) { buf.push_back(m | 25); for (int i = 1; i >= 0; i--) buf.push_back((uint8_t)(v >> (8 * i))); }
        else if (v <= 0xFFFFFFFFull) { buf.push_back(m | 26); for (int i = 3; i >= 0; i--) buf.push_back((uint8_t)(v >> (8 * i))); }
        else { buf.push_back(m | 27); for (int i = 7; i >= 0; i--) buf.push_back((uint8_t)(v >> (8 * i))); }
    }
    void uint(uint64_t v) { head(0, v); }
    void bytes(const uint8_t *p, size_t n) { head(2, n); buf.insert(buf.end(), p, p + n); }
    void text(const std::string &s) { head(3, s.size()); buf.insert(buf.end(), s.begin(), s.end()); }
    void array(uint64_t n) { head(4, n); }
    void map(uint64_t n) { head(5, n); }
    void tag(uint64_t t) { head(6, t); }
    void null() { buf.push_back(0xF6); }
    void boolean(bool b) { buf.push_back(b ? 0xF5 : 0xF4); }
};

class CborReader {
  public:
    CborReader(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}
    bool done() const { return p_ >= end_; }
    int major() const { need(1); return *p_ >> 5; }
    bool is_null() const { need(1); return *p_ == 0xF6; }
    bool is_bool() const { need(1); return *p_ == 0xF4 || *p_ == 0xF5; }
    bool boolean() { need(1); return *p_++ == 0xF5; }
    void skip_null() { need(1); p_++; }
    uint64_t head(int want_major)
    {
        need(1);
        const uint8_t b = *p_++;
        if ((b >> 5) != want_major) throw Error(ZARC_GPU_E_PARAM, "CBOR: unexpected type");
        const uint8_t ai = b & 31;
        if (ai < 24) return ai;
        const int nbytes = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : -1;
        if (nbytes < 0) throw Error(ZARC_GPU_E_PARAM, "CBOR: indefinite lengths are not produced by zarc");
        need((size_t)nbytes);
        uint64_t v = 0;
        for (int i = 0; i < nbytes; i++) v = (v << 8) | *p_++;
        return v;
    }
    uint64_t uint() { return head(0); }
    std::vector<uint8_t> bytes() { const uint64_t n = head(2); need(n); std::vector<uint8_t> v(p_, p_ + n); p_ += n; return v; }
    std::string text() { const uint64_t n = head(3); need(n); std::string s((const char *)p_, n); p_ += n; return s; }
    void skip() // one data item of any kind
    {
        need(1);
        const int m = *p_ >> 5;
        if (m == 7) { const uint8_t ai = *p_++ & 31; const int nb = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : 0; need((size_t)nb); p_ += nb; return; }
        const uint64_t v = head(m);
        if (m == 2 || m == 3) { need(v); p_ += v; }
        else if (m == 4) for (uint64_t i = 0; i < v; i++) skip();
        else if (m == 5) for (uint64_t i = 0; i < 2 * v; i++) skip();
        else if (m == 6) skip();
    }

  private:
    void need(size_t n) const { if ((size_t)(end_ - p_) < n) throw Error(ZARC_GPU_E_PARAM, "CBOR: truncated"); }
    const uint8_t *p_, *end_;
};

// ------------------------------------------------------------------ directory model --------------------------
struct Timestamp { // timestamps.rs:26-78: tag 0 + chrono to_rfc3339()
    int64_t secs = 0;
    uint32_t nanos = 0;
    std::string rfc3339() const
    {
        std::time_t t = (std::time_t)secs;
        std::tm tm{};
        gmtime_r(&t, &tm);
        char b[64];
        size_t n = std::strftime(b, sizeof b, "%Y-%m-%dT%H:%M:%S", &tm);
        std::string s(b, n);
        if (nanos) { // chrono prints 3, 6 or 9 fractional digits, as few as needed
            char f[16];
            if (nanos % 1000000 == 0) std::snprintf(f, sizeof f, ".%03u", nanos / 1000000);
            else if (nanos % 1000 == 0) std::snprintf(f, sizeof f, ".%06u", nanos / 1000);
            else std::snprintf(f, sizeof f, ".%09u", nanos);
            s += f;
        }
        return s + "+00:00";
    }
    static Timestamp parse(const std::string &s)
    {
        Timestamp ts;
        std::tm tm{};
        int y, mo, d, h, mi, se;
        if (std::sscanf(s.c_str(), "%d-%d-%dT%d:%d:%d", &y, &mo, &d, &h, &mi, &se) != 6) throw Error(ZARC_GPU_E_PARAM, "bad timestamp");
        tm.tm_year = y - 1900; tm.tm_mon = mo - 1; tm.tm_mday = d; tm.tm_hour = h; tm.tm_min = mi; tm.tm_sec = se;
        ts.secs = (int64_t)timegm(&tm);
        const size_t dot = s.find('.', 19);
        if (dot != std::string::npos) {
            uint64_t frac = 0; int digits = 0;
            for (size_t i = dot + 1; i < s.size() && s[i] >= '0' && s[i] <= '9' && digits < 9; i++, digits++) frac = frac * 10 + (uint64_t)(s[i] - '0');
            while (digits++ < 9) frac *= 10;
            ts.nanos = (uint32_t)frac;
        }
        return ts; // offsets other than +00:00 / Z are not produced by zarc
    }
    bool operator==(const Timestamp &o) const { return secs == o.secs && nanos == o.nanos; }
};

struct Edition { uint16_t number = 1; Timestamp written_at; uint8_t digest_type = 1; };

inline bool valid_utf8(const std::string &s)
{
    size_t i = 0;
    while (i < s.size()) {
        const uint8_t c = (uint8_t)s[i];
        size_t n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
        if (!n || i + n > s.size()) return false;
        for (size_t k = 1; k < n; k++) if ((((uint8_t)s[i + k]) >> 6) != 2) return false;
        i += n;
    }
    return true;
}

struct File { // directory/file.rs:16-62 (fields this host mirror carries; the rest is optional metadata)
    uint16_t edition = 1;
    std::vector<std::string> name;     // path components (Pathname): text when valid UTF-8, bytes otherwise
    std::optional<Digest> digest;
    std::optional<uint32_t> mode;
    std::optional<Timestamp> created, modified, accessed;
    std::optional<uint8_t> special_kind; // 1 = directory, 10..13 symlinks, 20..22 hardlinks (specials.rs:36-61)
    std::optional<std::string> link_target; // LinkTarget::FullPath (specials.rs:158-186); text when valid UTF-8, bytes otherwise
    struct Owner { std::optional<uint64_t> id; std::optional<std::string> name; }; // PosixOwner (posix_owner.rs:17-21)
    std::optional<Owner> user, group;
    // AttributeValue (strings.rs:154-212): a boolean, or a string that is CBOR text when valid UTF-8 and bytes otherwise
    struct Attr { bool is_bool = false; bool b = false; std::string s; bool operator==(const Attr &o) const { return is_bool == o.is_bool && b == o.b && s == o.s; } };
    typedef std::map<std::string, Attr> AttrMap; // the reference keeps a HashMap (any order on disk); a sorted map is reproducible
    std::optional<AttrMap> user_metadata, attributes, extended_attributes; // file.rs:49-61, keys 10 / 11 / 12
    bool is_normal() const { return digest.has_value() && !special_kind.has_value(); }
    bool is_dir() const { return special_kind.has_value() && *special_kind == 1; }
    bool is_symlink() const { return special_kind.has_value() && *special_kind >= 10 && *special_kind <= 13; }
    bool is_hardlink() const { return special_kind.has_value() && *special_kind >= 20 && *special_kind <= 22; }
};

// BTreeMap<Pathname, _> order: component-wise, Text before Binary, then bytewise (strings.rs:8-13,57-62)
inline bool pathname_less(const std::vector<std::string> &a, const std::vector<std::string> &b)
{
    for (size_t i = 0; i < a.size() && i < b.size(); i++) {
        const bool ba = !valid_utf8(a[i]), bb = !valid_utf8(b[i]);
        if (ba != bb) return !ba;
        if (a[i] != b[i]) return a[i] < b[i];
    }
    return a.size() < b.size();
}

inline void encode_frame(CborWriter &w, const Frame &f) // {0: edition, 1: offset, 2: digest, 3: length, 4: uncompressed}
{
    w.map(5);
    w.uint(0); w.uint(f.edition);
    w.uint(1); w.uint(f.offset);
    w.uint(2); w.bytes(f.digest.bytes.data(), 32);
    w.uint(3); w.uint(f.length);
    w.uint(4); w.uint(f.uncompressed);
}
inline void encode_timestamp(CborWriter &w, const Timestamp &t) { w.tag(0); w.text(t.rfc3339()); }
inline void encode_edition(CborWriter &w, const Edition &e) // {0: number, 1: written_at, 2: digest_type}
{
    w.map(3);
    w.uint(0); w.uint(e.number);
    w.uint(1); encode_timestamp(w, e.written_at);
    w.uint(2); w.uint(e.digest_type);
}
inline void encode_file(CborWriter &w, const File &f)
{
    const bool ts = f.created || f.modified || f.accessed;
    w.map(2 + (f.digest ? 1 : 0) + (f.mode ? 1 : 0) + (f.user ? 1 : 0) + (f.group ? 1 : 0) + (ts ? 1 : 0) + (f.special_kind ? 1 : 0) +
          (f.user_metadata ? 1 : 0) + (f.attributes ? 1 : 0) + (f.extended_attributes ? 1 : 0));
    w.uint(0); w.uint(f.edition);
    w.uint(1); w.array(f.name.size());
    for (const auto &c : f.name) { if (valid_utf8(c)) w.text(c); else w.bytes((const uint8_t *)c.data(), c.size()); }
    if (f.digest) { w.uint(2); w.bytes(f.digest->bytes.data(), 32); }
    if (f.mode) { w.uint(3); w.uint(*f.mode); }
    for (int which = 0; which < 2; which++) { // PosixOwner: array of the fields that are present (posix_owner.rs:181-203)
        const auto &o = which == 0 ? f.user : f.group;
        if (!o) continue;
        w.uint(4 + (uint64_t)which);
        w.array((o->id ? 1 : 0) + (o->name ? 1 : 0));
        if (o->id) w.uint(*o->id);
        if (o->name) w.text(*o->name);
    }
    if (ts) {
        w.uint(6);
        w.map((f.created ? 1 : 0) + (f.modified ? 1 : 0) + (f.accessed ? 1 : 0));
        if (f.created) { w.uint(1); encode_timestamp(w, *f.created); }
        if (f.modified) { w.uint(2); encode_timestamp(w, *f.modified); }
        if (f.accessed) { w.uint(3); encode_timestamp(w, *f.accessed); }
    }
    if (f.special_kind) { // SpecialFile [kind, link_target]; a trailing None is dropped
        w.uint(7);
        w.array(f.link_target ? 2 : 1);
        w.uint(*f.special_kind);
        if (f.link_target) { if (valid_utf8(*f.link_target)) w.text(*f.link_target); else w.bytes((const uint8_t *)f.link_target->data(), f.link_target->size()); }
    }
    for (int which = 0; which < 3; which++) { // attribute maps: text key -> bool | text | bytes
        const auto &m = which == 0 ? f.user_metadata : (which == 1 ? f.attributes : f.extended_attributes);
        if (!m) continue;
        w.uint(10 + (uint64_t)which);
        w.map(m->size());
        for (const auto &kv : *m) {
            w.text(kv.first);
            if (kv.second.is_bool) w.boolean(kv.second.b);
            else if (valid_utf8(kv.second.s)) w.text(kv.second.s);
            else w.bytes((const uint8_t *)kv.second.s.data(), kv.second.s.size());
        }
    }
}

inline void append_element(std::vector<uint8_t> &dir, uint8_t kind, const CborWriter &payload) // elements.rs:10-25
{
    if (payload.buf.size() > 0xFFFF) throw Error(ZARC_GPU_E_PARAM, "directory element larger than 65535 bytes");
    dir.push_back(kind);
    dir.push_back((uint8_t)payload.buf.size());
    dir.push_back((uint8_t)(payload.buf.size() >> 8));
    dir.push_back(0);
    dir.insert(dir.end(), payload.buf.begin(), payload.buf.end());
}

constexpr size_t EPILOGUE_LENGTH = 22, TRAILER_LENGTH = 32 + EPILOGUE_LENGTH, SKIPPABLE_FRAME_OVERHEAD = 8;

struct Trailer { // trailer.rs
    Digest digest;
    uint8_t digest_type = 1;
    int64_t directory_offset = 0;
    uint64_t directory_uncompressed_size = 0;
    uint8_t version = 1;
    std::array<uint8_t, EPILOGUE_LENGTH> epilogue(uint8_t check) const
    {
        std::array<uint8_t, EPILOGUE_LENGTH> e{};
        e[0] = digest_type;
        for (int i = 0; i < 8; i++) e[1 + i] = (uint8_t)((uint64_t)directory_offset >> (8 * i));
        for (int i = 0; i < 8; i++) e[9 + i] = (uint8_t)(directory_uncompressed_size >> (8 * i));
        e[17] = check; e[18] = version; e[19] = 0x65; e[20] = 0xAA; e[21] = 0xDC;
        return e;
    }
    uint8_t compute_check() const // XOR over [0, digest_type] || digest || epilogue with check = 0 (trailer.rs:98-108)
    {
        uint8_t c = (uint8_t)(0 ^ digest_type);
        for (uint8_t b : digest.bytes) c ^= b;
        for (uint8_t b : epilogue(0)) c ^= b;
        return c;
    }
    std::vector<uint8_t> to_bytes() const // digest || epilogue: the 2-byte prologue is NOT written
    {
        std::vector<uint8_t> v(digest.bytes.begin(), digest.bytes.end());
        const auto e = epilogue(compute_check());
        v.insert(v.end(), e.begin(), e.end());
        return v;
    }
};

// ------------------------------------------------------------------ writer ------------------------------------
class ArchiveWriter : public Encoder {
  public:
    explicit ArchiveWriter(std::ostream &w, int device = 0) : Encoder(w, device) {}
    ArchiveWriter(std::ostream &w, const std::vector<int> &devices) : Encoder(w, devices) {} // `zarc pack --gpus N`

    // Encoder::add_file_entry (add_file.rs:22-46)
    void add_file_entry(File f) { f.edition = 1; files_.push_back(std::move(f)); }

    // Encoder::finalise (encode/directory.rs:40-122): returns the directory digest
    Digest finalise(Timestamp written_at)
    {
        std::vector<uint8_t> dir;
        { CborWriter w; Edition e; e.written_at = written_at; encode_edition(w, e); append_element(dir, 1, w); }
        std::vector<size_t> order(files_.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pathname_less(files_[a].name, files_[b].name); });
        std::map<Digest, bool> written;
        for (size_t i : order) {
            const File &f = files_[i];
            if (f.digest && !written.count(*f.digest)) { // the frame element goes right before the first file that uses it
                auto it = frames().find(*f.digest);
                if (it != frames().end()) { CborWriter w; encode_frame(w, it->second); append_element(dir, 3, w); written[*f.digest] = true; }
            }
            CborWriter w; encode_file(w, f); append_element(dir, 2, w);
        }
        for (const Digest &d : frame_order()) // frames not linked to any file
            if (!written.count(d)) { CborWriter w; encode_frame(w, frames().at(d)); append_element(dir, 3, w); }
        // the directory is content like any other: BLAKE3 + one Zstandard frame, both through the engine
        const void *p = dir.data();
        const size_t n = dir.size();
        Digest digest;
        std::vector<uint8_t> frame;
        size_t frame_len = raw_frame(p, n, digest, frame);
        writer_.write((const char *)frame.data(), (std::streamsize)frame_len);
        offset_ += frame_len + SKIPPABLE_FRAME_OVERHEAD + TRAILER_LENGTH;
        Trailer t;
        t.digest = digest;
        t.directory_uncompressed_size = n;
        t.directory_offset = -(int64_t)(frame_len + SKIPPABLE_FRAME_OVERHEAD + TRAILER_LENGTH);
        const std::vector<uint8_t> tb = t.to_bytes();
        const uint8_t skip[8] = {0x5F, 0x2A, 0x4D, 0x18, (uint8_t)tb.size(), 0, 0, 0}; // skippable frame, nibble 0xF
        writer_.write((const char *)skip, 8);
        writer_.write((const char *)tb.data(), (std::streamsize)tb.size());
        writer_.flush();
        if (!writer_) throw Error(ZARC_GPU_E_DEVICE, "write failed");
        return digest;
    }

  private:
    // one frame that is NOT entered into the frame table (write_compressed_frame, lowlevel_frames.rs:19-39)
    size_t raw_frame(const void *p, size_t n, Digest &digest, std::vector<uint8_t> &frame)
    {
        frame.resize(zarc_gpu_bound(n));
        size_t off = 0, len = 0;
        int st = 0;
        engine0().check(zarc_gpu_pack_batch(engine0().get(), 1, &p, &n, frame.data(), frame.size(), &off, &len, (uint8_t(*)[32]) & digest, &st));
        if (st != ZARC_GPU_FRAME_OK) throw Error(st, zarc_gpu_frame_status_name(st));
        if (off) std::memmove(frame.data(), frame.data() + off, len);
        return len;
    }
    std::vector<File> files_;
};

// ------------------------------------------------------------------ reader ------------------------------------
class ArchiveReader {
  public:
    // Decoder::open + read_directory over an archive image in memory
    ArchiveReader(const uint8_t *data, size_t len, int device = 0) : ArchiveReader(data, len, std::vector<int>{device}) {}
    // ... on several devices (`zarc unpack --gpus N`): read_files deals the frames of a batch to one engine handle per device
    ArchiveReader(const uint8_t *data, size_t len, const std::vector<int> &devices) : data_(data), len_(len), reader_(devices)
    {
        if (len < 12 + SKIPPABLE_FRAME_OVERHEAD + TRAILER_LENGTH) throw Error(ZARC_GPU_E_PARAM, "not a zarc archive: too short");
        if (std::memcmp(data, FILE_MAGIC, 11) != 0) throw Error(ZARC_GPU_E_PARAM, "not a zarc archive: bad header");   // open.rs:48-67
        if (data[11] != 1) throw Error(ZARC_GPU_E_UNSUPPORTED, "unsupported zarc version");
        const uint8_t *e = data + len - EPILOGUE_LENGTH;                                                                // open.rs:76-133
        if (!(e[19] == 0x65 && e[20] == 0xAA && e[21] == 0xDC)) throw Error(ZARC_GPU_E_PARAM, "parse error: trailer magic");
        trailer_.digest_type = e[0];
        if (trailer_.digest_type != 1) throw Error(ZARC_GPU_E_UNSUPPORTED, "unknown digest type");
        uint64_t off = 0, usz = 0;
        for (int i = 0; i < 8; i++) { off |= (uint64_t)e[1 + i] << (8 * i); usz |= (uint64_t)e[9 + i] << (8 * i); }
        trailer_.directory_offset = (int64_t)off;
        trailer_.directory_uncompressed_size = usz;
        trailer_.version = e[18];
        std::memcpy(trailer_.digest.bytes.data(), data + len - TRAILER_LENGTH, 32);
        if (trailer_.compute_check() != e[17]) throw Error(ZARC_GPU_E_PARAM, "parse error: trailer check byte doesn't match");
        const int64_t raw_off = trailer_.directory_offset;
        if (trailer_.directory_offset < 0) trailer_.directory_offset += (int64_t)len;                                   // trailer.rs:91-95
        // the directory frame lies between the header and the trailer frame: anything else (a crafted offset inside the last 62 bytes
        // would make the length below wrap) is a parse error, as in the reference (decode/open.rs:118-133)
        if (trailer_.directory_offset < 12 || (uint64_t)trailer_.directory_offset > len - SKIPPABLE_FRAME_OVERHEAD - TRAILER_LENGTH)
            throw Error(ZARC_GPU_E_PARAM, "parse error: directory offset");
        // a zstd frame cannot expand more than ~2^7 per byte at these sizes; refuse absurd claims before allocating for them
        if (trailer_.directory_uncompressed_size > ((uint64_t)1 << 32)) throw Error(ZARC_GPU_E_PARAM, "parse error: directory size");
        const uint64_t dir_frame_len = len - SKIPPABLE_FRAME_OVERHEAD - TRAILER_LENGTH - (uint64_t)trailer_.directory_offset;
        (void)raw_off;
        // read_directory (decode/directory.rs:55-119): decode the directory frame, check its digest, parse elements
        Frame df;
        df.offset = (uint64_t)trailer_.directory_offset;
        df.length = dir_frame_len;
        df.uncompressed = trailer_.directory_uncompressed_size;
        df.digest = trailer_.digest;
        auto res = reader_.read_content_frames(data, len, {df});
        if (res[0].status != ZARC_GPU_FRAME_OK && res[0].status != ZARC_GPU_FRAME_DIGEST) throw Error(res[0].status, zarc_gpu_frame_status_name(res[0].status));
        if (!res[0].verify.value_or(false)) throw Error(ZARC_GPU_FRAME_DIGEST, "directory integrity: digest");
        parse_directory(res[0].data);
    }
    const Trailer &trailer() const { return trailer_; }
    const std::vector<File> &files() const { return files_; }
    const std::map<Digest, Frame> &frames() const { return frames_; }
    const std::vector<Edition> &editions() const { return editions_; }

    // extract_file for a batch (zarc-cli/src/unpack.rs:94-124): entries whose digest has no frame are skipped (:107-110)
    std::vector<FrameReader::Result> read_files(const std::vector<size_t> &indices)
    {
        std::vector<Frame> wanted;
        for (size_t i : indices) {
            const File &f = files_.at(i);
            if (!f.digest) throw Error(ZARC_GPU_E_PARAM, "not a normal file");
            auto it = frames_.find(*f.digest);
            if (it == frames_.end()) throw Error(ZARC_GPU_E_PARAM, "file digest has no frame");
            wanted.push_back(it->second);
        }
        return reader_.read_content_frames(data_, len_, wanted);
    }

  private:
    void parse_directory(const std::vector<uint8_t> &dir)
    {
        size_t pos = 0;
        while (pos < dir.size()) {
            if (pos + 4 > dir.size()) throw Error(ZARC_GPU_E_PARAM, "parse error: truncated element");
            const uint8_t kind = dir[pos];
            const size_t n = dir[pos + 1] | ((size_t)dir[pos + 2] << 8);
            pos += 4;
            if (pos + n > dir.size()) throw Error(ZARC_GPU_E_PARAM, "parse error: truncated element payload");
            CborReader r(dir.data() + pos, n);
            pos += n;
            if (kind == 1) editions_.push_back(parse_edition(r));
            else if (kind == 2) files_.push_back(parse_file(r));
            else if (kind == 3) { Frame f = parse_frame(r); frames_[f.digest] = f; }
            // unknown kinds are skipped (decode/directory.rs:76-79)
        }
    }
    static Timestamp parse_ts(CborReader &r)
    {
        const uint64_t tag = r.head(6);
        if (tag == 0) return Timestamp::parse(r.text());
        if (tag == 1) { Timestamp t; t.secs = (int64_t)r.uint(); return t; } // numeric epoch form (timestamps.rs:89-118)
        throw Error(ZARC_GPU_E_PARAM, "expected Timestamp or DateTime tag");
    }
    static Edition parse_edition(CborReader &r)
    {
        Edition e;
        const uint64_t n = r.head(5);
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t k = r.uint();
            if (k == 0) e.number = (uint16_t)r.uint();
            else if (k == 1) e.written_at = parse_ts(r);
            else if (k == 2) e.digest_type = (uint8_t)r.uint();
            else r.skip();
        }
        return e;
    }
    static Frame parse_frame(CborReader &r)
    {
        Frame f;
        const uint64_t n = r.head(5);
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t k = r.uint();
            if (k == 0) f.edition = (uint16_t)r.uint();
            else if (k == 1) f.offset = r.uint();
            else if (k == 2) { auto b = r.bytes(); if (b.size() != 32) throw Error(ZARC_GPU_E_PARAM, "digest length"); std::memcpy(f.digest.bytes.data(), b.data(), 32); }
            else if (k == 3) f.length = r.uint();
            else if (k == 4) f.uncompressed = r.uint();
            else r.skip();
        }
        return f;
    }
    static File parse_file(CborReader &r)
    {
        File f;
        const uint64_t n = r.head(5);
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t k = r.uint();
            if (k == 0) f.edition = (uint16_t)r.uint();
            else if (k == 1) {
                const uint64_t c = r.head(4);
                for (uint64_t j = 0; j < c; j++) {
                    if (r.major() == 3) f.name.push_back(r.text());
                    else { auto b = r.bytes(); f.name.emplace_back((const char *)b.data(), b.size()); }
                }
            } else if (k == 2) { auto b = r.bytes(); if (b.size() != 32) throw Error(ZARC_GPU_E_PARAM, "digest length"); Digest d; std::memcpy(d.bytes.data(), b.data(), 32); f.digest = d; }
            else if (k == 3) f.mode = (uint32_t)r.uint();
            else if (k == 4 || k == 5) {
                File::Owner o;
                const uint64_t c = r.head(4);
                for (uint64_t j = 0; j < c; j++) {
                    if (r.major() == 0) o.id = r.uint();
                    else if (r.major() == 3) o.name = r.text();
                    else r.skip();
                }
                (k == 4 ? f.user : f.group) = o;
            } else if (k == 6) {
                const uint64_t m = r.head(5);
                for (uint64_t j = 0; j < m; j++) {
                    const uint64_t tk = r.uint();
                    Timestamp t = parse_ts(r);
                    if (tk == 1) f.created = t; else if (tk == 2) f.modified = t; else if (tk == 3) f.accessed = t;
                }
            } else if (k == 7) {
                const uint64_t c = r.head(4);
                for (uint64_t j = 0; j < c; j++) {
                    if (j == 0 && !r.is_null()) f.special_kind = (uint8_t)r.uint();
                    else if (j == 1 && r.major() == 3) f.link_target = r.text();
                    else if (j == 1 && r.major() == 2) { auto b = r.bytes(); f.link_target = std::string((const char *)b.data(), b.size()); }
                    else r.skip(); // component-array targets are what the reference cannot read back either (specials.rs:193-196)
                }
            } else if (k >= 10 && k <= 12) {
                File::AttrMap m;
                const uint64_t c = r.head(5);
                for (uint64_t j = 0; j < c; j++) {
                    const std::string key = r.text();
                    File::Attr a;
                    if (r.is_bool()) { a.is_bool = true; a.b = r.boolean(); }
                    else if (r.major() == 3) a.s = r.text();
                    else { auto b = r.bytes(); a.s.assign((const char *)b.data(), b.size()); }
                    m[key] = a;
                }
                (k == 10 ? f.user_metadata : (k == 11 ? f.attributes : f.extended_attributes)) = m;
            } else r.skip();
        }
        return f;
    }
    const uint8_t *data_;
    size_t len_;
    FrameReader reader_;
    Trailer trailer_;
    std::vector<Edition> editions_;
    std::vector<File> files_;
    std::map<Digest, Frame> frames_;
};

} // namespace zarc
