// tests/host/container_test.cpp -- zarc::ArchiveWriter / zarc::ArchiveReader (zarc_amd/host/zarc_container.hpp):
// a whole archive (header, content frames, directory frame, trailer) written and read back the way
// crates/zarc-cli/src/pack.rs:219-272 and unpack.rs:60-124 drive the reference library.
// usage: container_test <out.zarc> <out.contents> [big_size]; the python test re-parses <out.zarc> independently.
#include "../../zarc_amd/host/zarc_container.hpp"
#include "../../zarc_amd/csrc/corpus.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static std::string hex(const std::vector<uint8_t> &v)
{
    std::string s;
    char b[4];
    for (uint8_t x : v) { std::snprintf(b, sizeof b, "%02x", x); s += b; }
    return s;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const size_t big = argc > 3 ? (size_t)std::atol(argv[3]) : 70000;
    // --- known-answer bytes: CBOR shapes of the directory elements (frame.rs:10-32, edition.rs:11-34, elements.rs:10-25) ---
    {
        zarc::Frame f;
        f.edition = 1; f.offset = 12; f.length = 300; f.uncompressed = 70000;
        for (int i = 0; i < 32; i++) f.digest.bytes[i] = (uint8_t)i;
        zarc::CborWriter w;
        zarc::encode_frame(w, f);
        CHECK(hex(w.buf) == "a5000101" "0c" "025820" "000102030405060708090a0b0c0d0e0f101112131415161718191a1b1c1d1e1f" "0319012c" "041a00011170");
        std::vector<uint8_t> dir;
        zarc::append_element(dir, 3, w);
        CHECK(dir.size() == 4 + w.buf.size() && dir[0] == 3 && dir[1] == w.buf.size() && dir[2] == 0 && dir[3] == 0);
        zarc::Edition e;
        e.written_at.secs = 1703567043; // 2023-12-26T05:04:03Z
        zarc::CborWriter we;
        zarc::encode_edition(we, e);
        const std::string ts = "2023-12-26T05:04:03+00:00";
        std::vector<uint8_t> expect = {0xA3, 0x00, 0x01, 0x01, 0xC0, 0x78, (uint8_t)ts.size()};
        expect.insert(expect.end(), ts.begin(), ts.end());
        expect.push_back(0x02); expect.push_back(0x01);
        CHECK(we.buf == expect);
        zarc::Timestamp t2{1703567043, 120000000};
        CHECK(t2.rfc3339() == "2023-12-26T05:04:03.120+00:00");
        CHECK(zarc::Timestamp::parse(t2.rfc3339()) == t2);
        zarc::Timestamp t3{1, 1};
        CHECK(t3.rfc3339() == "1970-01-01T00:00:01.000000001+00:00" && zarc::Timestamp::parse(t3.rfc3339()) == t3);
        // trailer: 54 bytes, check byte makes the XOR of prologue+digest+epilogue zero (trailer.rs:98-108)
        zarc::Trailer tr;
        tr.digest = f.digest; tr.directory_offset = -1000; tr.directory_uncompressed_size = 4242;
        auto tb = tr.to_bytes();
        CHECK(tb.size() == zarc::TRAILER_LENGTH);
        uint8_t x = 0 ^ 1;
        for (uint8_t b : tb) x ^= b;
        CHECK(x == 0);
        CHECK(tb[32] == 1 && tb[33] == 0x18 && tb[34] == 0xFC && tb[40] == 0xFF && tb[41] == 0x92 && tb[42] == 0x10 && tb[50] == 1 && tb[51] == 0x65 && tb[52] == 0xAA && tb[53] == 0xDC);
    }
    // --- write an archive ---
    std::vector<std::vector<uint8_t>> ents;
    const size_t sizes[] = {big, 0, 513, big + 1000, 4096};
    for (size_t i = 0; i < 5; i++) { ents.emplace_back(sizes[i]); zarc_corpus_entry(ents.back().data(), sizes[i], 500 + i, (int)(i & 3)); }
    struct Spec { std::vector<std::string> name; int content; };
    const std::string bin = std::string("caf") + (char)0xE9;           // not UTF-8 -> CBOR bytes component
    const std::vector<Spec> specs = {
        {{"src", "main.rs"}, 0}, {{"src", "empty"}, 1}, {{"README.md"}, 2}, {{"src", "copy-of-main.rs"}, 0},
        {{"data", "blob.bin"}, 3}, {{"data", bin}, 4}, {{"src"}, -1}, {{"data"}, -1},
    };
    std::vector<uint8_t> orphan(777);
    zarc_corpus_entry(orphan.data(), orphan.size(), 999, 0);
    std::stringstream file;
    zarc::Digest dir_digest;
    std::vector<zarc::Digest> dig;
    uint64_t end_offset = 0;
    {
        zarc::ArchiveWriter enc(file);
        enc.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1); // crates/zarc-cli/src/pack.rs:227
        enc.set_zstd_parameter(ZARC_GPU_P_COMPRESSION_LEVEL, 3);
        std::vector<const void *> ptr;
        std::vector<size_t> len;
        for (auto &e : ents) { ptr.push_back(e.data()); len.push_back(e.size()); }
        dig = enc.add_data_frames(ptr.data(), len.data(), ptr.size());
        // one content frame that no file refers to: it must still be listed in the directory (directory.rs:86-92)
        dig.push_back(enc.add_data_frame(orphan.data(), orphan.size()));
        for (const auto &s : specs) {
            zarc::File f;
            f.name = s.name;
            f.mode = s.content < 0 ? 040755u : 0100644u;
            f.modified = zarc::Timestamp{1700000000 + (int64_t)f.name.size(), 500000};
            if (s.content < 0) f.special_kind = 1; else f.digest = dig[(size_t)s.content];
            enc.add_file_entry(f);
        }
        dir_digest = enc.finalise(zarc::Timestamp{1703567043, 0});
        end_offset = enc.offset();
    }
    const std::string image = file.str();
    CHECK(end_offset == image.size());
    CHECK(std::memcmp(image.data(), zarc::FILE_MAGIC, 12) == 0);
    CHECK((uint8_t)image[image.size() - 62] == 0x5F && (uint8_t)image[image.size() - 61] == 0x2A && (uint8_t)image[image.size() - 58] == 54);
    { std::ofstream o(argv[1], std::ios::binary); o.write(image.data(), (std::streamsize)image.size()); }
    { // contents in frame order, for the python side (whole-stream decode == contents + directory)
        std::ofstream o(argv[2], std::ios::binary);
        for (auto &e : ents) o.write((const char *)e.data(), (std::streamsize)e.size());
        o.write((const char *)orphan.data(), (std::streamsize)orphan.size());
    }
    // --- read it back ---
    zarc::ArchiveReader rd((const uint8_t *)image.data(), image.size());
    CHECK(rd.trailer().digest == dir_digest && rd.trailer().version == 1 && rd.trailer().digest_type == 1);
    CHECK(rd.editions().size() == 1 && rd.editions()[0].number == 1 && rd.editions()[0].written_at.secs == 1703567043 && rd.editions()[0].digest_type == 1);
    CHECK(rd.frames().size() == 6 && rd.files().size() == specs.size());
    // files come back in BTreeMap<Pathname> order: component-wise, text before bytes
    for (size_t i = 1; i < rd.files().size(); i++) CHECK(!zarc::pathname_less(rd.files()[i].name, rd.files()[i - 1].name));
    CHECK(rd.files()[0].name == std::vector<std::string>({"README.md"}) && rd.files()[1].name == std::vector<std::string>({"data"}));
    CHECK(rd.files()[2].name == std::vector<std::string>({"data", "blob.bin"}) && rd.files()[3].name == std::vector<std::string>({"data", bin}));
    std::vector<size_t> normal;
    std::vector<const std::vector<uint8_t> *> want;
    for (size_t i = 0; i < rd.files().size(); i++) {
        const zarc::File &f = rd.files()[i];
        const Spec *s = nullptr;
        for (const auto &c : specs) if (c.name == f.name) s = &c;
        CHECK(s != nullptr);
        CHECK(f.modified && f.modified->secs == 1700000000 + (int64_t)f.name.size() && f.modified->nanos == 500000 && !f.created && !f.accessed);
        if (s->content < 0) { CHECK(f.special_kind && *f.special_kind == 1 && !f.digest && *f.mode == 040755u); continue; }
        CHECK(f.is_normal() && *f.digest == dig[(size_t)s->content] && *f.mode == 0100644u);
        normal.push_back(i);
        want.push_back(&ents[(size_t)s->content]);
    }
    auto res = rd.read_files(normal);
    for (size_t k = 0; k < normal.size(); k++) CHECK(res[k].status == ZARC_GPU_FRAME_OK && res[k].verify.value_or(false) && res[k].data == *want[k]);
    // --- corruption is detected where the reference detects it ---
    auto open_fails = [&](std::string img, const char *what) {
        try { zarc::ArchiveReader r((const uint8_t *)img.data(), img.size()); }
        catch (const zarc::Error &e) { return std::string(e.what()).find(what) != std::string::npos; }
        return false;
    };
    { std::string b = image; b[b.size() - 5] ^= 1; CHECK(open_fails(b, "check byte")); }          // open.rs:113-121
    { std::string b = image; b[b.size() - 1] ^= 1; CHECK(open_fails(b, "trailer magic")); }
    { std::string b = image; b[9] ^= 1; CHECK(open_fails(b, "bad header")); }                      // open.rs:48-67
    { std::string b = image; b[11] = 2; CHECK(open_fails(b, "unsupported zarc version")); }
    { // a trailer whose digest (and check byte) were rewritten: the directory no longer matches it (decode/directory.rs:112-117)
        std::string b = image;
        b[b.size() - 54] ^= 0x10; b[b.size() - 5] ^= 0x10;
        CHECK(open_fails(b, "directory integrity"));
    }
    { // damage inside the directory frame: the frame's own XXH64 / entropy checks fire first
        std::string b = image;
        b[(size_t)rd.trailer().directory_offset + 40] ^= 0x55;
        bool threw = false;
        try { zarc::ArchiveReader r((const uint8_t *)b.data(), b.size()); } catch (const zarc::Error &) { threw = true; }
        CHECK(threw);
    }
    { // crafted trailers: a directory offset inside the last 62 bytes used to make the frame length wrap around; a
      // consistent check byte is recomputed so that only the offset test can reject it
        auto with_offset = [&](int64_t off) {
            std::string b = image;
            uint8_t *e = (uint8_t *)&b[b.size() - 22];
            for (int i = 0; i < 8; i++) e[1 + i] = (uint8_t)((uint64_t)off >> (8 * i));
            e[17] = 0;
            uint8_t x = 0 ^ e[0];
            for (size_t i = b.size() - 54; i < b.size(); i++) x ^= (uint8_t)b[i];
            e[17] = x;
            return b;
        };
        CHECK(open_fails(with_offset(-30), "directory offset"));
        CHECK(open_fails(with_offset(-61), "directory offset"));
        CHECK(open_fails(with_offset((int64_t)image.size() - 10), "directory offset"));
        CHECK(open_fails(with_offset(3), "directory offset"));
    }
    { // directory records that point outside the file or wrap around u64 are refused before anything is staged
        zarc::FrameReader fr(0);
        zarc::Frame f;
        f.offset = ~0ull - 5; f.length = 100; f.uncompressed = 10;
        bool threw = false;
        try { fr.read_content_frames((const uint8_t *)image.data(), image.size(), {f}); } catch (const zarc::Error &) { threw = true; }
        CHECK(threw);
        f.offset = 12; f.length = ~0ull; threw = false;
        try { fr.read_content_frames((const uint8_t *)image.data(), image.size(), {f}); } catch (const zarc::Error &) { threw = true; }
        CHECK(threw);
        f.offset = 12; f.length = 20; f.uncompressed = (uint64_t)1 << 40; threw = false;
        try { fr.read_content_frames((const uint8_t *)image.data(), image.size(), {f}); } catch (const zarc::Error &) { threw = true; }
        CHECK(threw);
    }
    std::printf("container OK: %zu files, %zu frames, %zu bytes, directory %llu bytes\n", rd.files().size(), rd.frames().size(), image.size(),
                (unsigned long long)rd.trailer().directory_uncompressed_size);
    return 0;
}
