// tests/host/host_mirror_test.cpp -- exercises zarc::Encoder / zarc::FrameReader (zarc_amd/host/zarc_host.hpp) the
// way the reference's CLI drives its library (crates/zarc-cli/src/pack.rs:219-272, unpack.rs:94-124).
// Linked against the product library on a GPU box, or against the emulated build of the same sources on CPU.
#include "../../zarc_amd/host/zarc_host.hpp"
#include "../../zarc_amd/csrc/corpus.h"
#include <cstdio>
#include <cstdlib>
#include <sstream>

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t big = argc > 1 ? (size_t)std::atol(argv[1]) : 65536;
    // --- config C1 (BASELINE.json configs[0]): 10 x 64 KiB random entries, checksum on, default level ---
    {
        std::ostringstream file;
        zarc::Encoder enc(file);
        enc.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1); // crates/zarc-cli/src/pack.rs:227
        std::vector<std::vector<uint8_t>> ents(10, std::vector<uint8_t>(65536));
        std::vector<const void *> ptr;
        std::vector<size_t> len;
        for (size_t i = 0; i < ents.size(); i++) { zarc_corpus_entry(ents[i].data(), ents[i].size(), i, 3); ptr.push_back(ents[i].data()); len.push_back(ents[i].size()); }
        auto dig = enc.add_data_frames(ptr.data(), len.data(), ptr.size());
        const std::string bytes = file.str();
        CHECK(bytes.size() == 12 + 10 * 65550);
        CHECK(std::memcmp(bytes.data(), zarc::FILE_MAGIC, 12) == 0);
        CHECK(enc.offset() == bytes.size());
        for (size_t i = 0; i < 10; i++) {
            const zarc::Frame &f = enc.frames().at(dig[i]);
            CHECK(f.offset == 12 + 65550 * i && f.length == 65550 && f.uncompressed == 65536 && f.edition == 1);
            CHECK((uint8_t)bytes[f.offset] == 0x28 && (uint8_t)bytes[f.offset + 3] == 0xFD);
        }
    }
    // --- mixed entries with duplicates: first occurrence wins, later identical content writes nothing ---
    std::ostringstream file;
    zarc::Encoder enc(file);
    enc.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1);
    enc.set_zstd_parameter(ZARC_GPU_P_COMPRESSION_LEVEL, 3);
    std::vector<std::vector<uint8_t>> ents;
    const size_t sizes[] = {0, 1, 300, big, big + 17, 1000};
    for (size_t i = 0; i < 6; i++) { ents.emplace_back(sizes[i]); zarc_corpus_entry(ents.back().data(), sizes[i], 100 + i, (int)(i & 3)); }
    ents.push_back(ents[3]); // duplicate of entry 3 inside the same batch
    std::vector<const void *> ptr;
    std::vector<size_t> len;
    for (auto &e : ents) { ptr.push_back(e.data()); len.push_back(e.size()); }
    auto dig = enc.add_data_frames(ptr.data(), len.data(), ptr.size());
    CHECK(dig.size() == 7 && dig[6] == dig[3]);
    CHECK(enc.frames().size() == 6);
    const uint64_t off_after = enc.offset();
    auto again = enc.add_data_frame(ents[4].data(), ents[4].size()); // duplicate across calls
    CHECK(again == dig[4] && enc.offset() == off_after && enc.frames().size() == 6);
    // offsets are running sums starting at 12, in call order
    uint64_t expect_off = 12;
    for (const auto &d : enc.frame_order()) { const zarc::Frame &f = enc.frames().at(d); CHECK(f.offset == expect_off); expect_off += f.length; }
    CHECK(expect_off == enc.offset());
    // --- read back: read_content_frame + verify ---
    const std::string image = file.str();
    std::vector<zarc::Frame> wanted;
    for (size_t i = 0; i < 6; i++) wanted.push_back(enc.frames().at(dig[i]));
    zarc::Frame bad = wanted[2];
    bad.digest.bytes[0] ^= 0xFF; // wrong expected digest: reported, bytes still delivered (unpack.rs:118-120)
    wanted.push_back(bad);
    zarc::FrameReader rd;
    auto res = rd.read_content_frames((const uint8_t *)image.data(), image.size(), wanted);
    for (size_t i = 0; i < 6; i++) {
        CHECK(res[i].status == ZARC_GPU_FRAME_OK && res[i].verify.has_value() && *res[i].verify);
        CHECK(res[i].data == ents[i] && res[i].digest == dig[i]);
    }
    CHECK(res[6].status == ZARC_GPU_FRAME_DIGEST && res[6].verify.has_value() && !*res[6].verify && res[6].data == ents[2]);
    // --- Encoder::enable_compression(false): stored (raw-block) frame, bookkeeping as for any other frame (content_frame.rs:35-44) ---
    {
        std::vector<uint8_t> e(200000);
        zarc_corpus_entry(e.data(), e.size(), 4242, 0);
        const uint64_t before = enc.offset();
        enc.enable_compression(false);
        const zarc::Digest d = enc.add_data_frame(e.data(), e.size());
        enc.enable_compression(true);
        const zarc::Frame &f = enc.frames().at(d);
        CHECK(f.offset == before && f.uncompressed == e.size() && f.length == 14 + e.size() + 3 * 2 && enc.offset() == before + f.length);
        const std::string img = file.str();
        auto r = rd.read_content_frames((const uint8_t *)img.data(), img.size(), {f});
        CHECK(r[0].status == ZARC_GPU_FRAME_OK && r[0].verify.value_or(false) && r[0].data == e);
    }
    // --- several devices: the same batch dealt to two engine handles gives the same archive, byte for byte ---
    {
        std::vector<std::vector<uint8_t>> es;
        const size_t mixed[] = {70000, 3000, big, 65536, 0, 90001, 20000, big + 5};
        for (size_t i = 0; i < 8; i++) { es.emplace_back(mixed[i]); zarc_corpus_entry(es.back().data(), mixed[i], 700 + i, -1); }
        es.push_back(es[2]); es.push_back(es[1]); es.push_back(es[0]); // duplicates whose first copy is packed by the other handle
        std::vector<const void *> p2;
        std::vector<size_t> l2;
        for (auto &e : es) { p2.push_back(e.data()); l2.push_back(e.size()); }
        auto run = [&](const std::vector<int> &devices, std::string *image, std::vector<zarc::Digest> *dg, uint64_t *end) {
            std::ostringstream f2;
            zarc::Encoder e2(f2, devices);
            e2.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1);
            *dg = e2.add_data_frames(p2.data(), l2.data(), p2.size());
            *end = e2.offset();
            *image = f2.str();
            return e2.frames().size();
        };
        std::string one, two, eq;
        std::vector<zarc::Digest> d1, d2, d3;
        uint64_t e1 = 0, e2 = 0, e3 = 0;
        const size_t f1 = run({0}, &one, &d1, &e1), f2 = run({0, 0}, &two, &d2, &e2);
        CHECK(f1 == 8 && f2 == 8 && one == two && d1 == d2 && e1 == e2 && e1 == one.size());
        { // hash-first dedup (the default) against compress-everything-and-drop-later: the same archive (content_frame.rs:26-33)
            std::ostringstream f3;
            zarc::Encoder e3x(f3, std::vector<int>{0});
            e3x.set_hash_first(false);
            e3x.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1);
            d3 = e3x.add_data_frames(p2.data(), l2.data(), p2.size());
            CHECK(f3.str() == one && d3 == d1 && e3x.frames().size() == 8);
        }
        const auto share = zarc::shard_assign(l2.data(), l2.size(), 2);
        CHECK(share[0].size() + share[1].size() == 11);
        // equal sizes: index mod G
        std::vector<size_t> same(10, 4096);
        const auto rr = zarc::shard_assign(same.data(), same.size(), 3);
        CHECK(rr[0] == (std::vector<size_t>{0, 3, 6, 9}) && rr[1] == (std::vector<size_t>{1, 4, 7}) && rr[2] == (std::vector<size_t>{2, 5, 8}));
        (void)eq; (void)d3; (void)e3;
        // ... and the way back: the frames of that archive dealt to two handles by uncompressed bytes decode to the same results, in
        // the caller's order, a corrupt frame and a wrong expected digest included (unpack.rs:62-88: frames are independent).
        // With two real devices (or HIPEMU_DEVICES=2) the second handle is device 1, otherwise the same device twice.
        const int second = zarc_gpu_device_count() >= 2 ? 1 : 0;
        std::printf("multi-device section: devices {0, %d} (%d visible)\n", second, zarc_gpu_device_count());
        {
            std::ostringstream f2;
            zarc::Encoder e2(f2, {0, second});
            e2.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1);
            auto dg = e2.add_data_frames(p2.data(), l2.data(), p2.size());
            CHECK(f2.str() == one && dg == d1);
            std::string img = f2.str();
            std::vector<zarc::Frame> want;
            for (size_t i = 0; i < es.size(); i++) want.push_back(e2.frames().at(dg[i]));
            zarc::Frame wrong = want[3];
            wrong.digest.bytes[5] ^= 1;                       // reported, bytes delivered
            want.push_back(wrong);
            const zarc::Frame &victim = want[7];              // corrupt one byte in the middle of the largest frame
            img[victim.offset + victim.length / 2] ^= 0x55;
            zarc::FrameReader r1(std::vector<int>{0}), r2(std::vector<int>{0, second});
            CHECK(r2.devices() == 2);
            auto a1 = r1.read_content_frames((const uint8_t *)img.data(), img.size(), want);
            auto a2 = r2.read_content_frames((const uint8_t *)img.data(), img.size(), want);
            CHECK(a1.size() == want.size() && a2.size() == want.size());
            for (size_t i = 0; i < want.size(); i++) {
                CHECK(a1[i].status == a2[i].status && a1[i].data == a2[i].data && a1[i].digest == a2[i].digest && a1[i].verify == a2[i].verify);
                if (i == 7) CHECK(a2[i].status != ZARC_GPU_FRAME_OK);
                else if (i == want.size() - 1) CHECK(a2[i].status == ZARC_GPU_FRAME_DIGEST && a2[i].data == es[3]);
                else CHECK(a2[i].status == ZARC_GPU_FRAME_OK && a2[i].verify.value_or(false) && a2[i].data == es[i]);
            }
            const auto us = zarc::shard_assign(l2.data(), l2.size(), 2);
            CHECK(!us[0].empty() && !us[1].empty());
        }
    }
    // --- parameter errors surface as exceptions carrying the libzstd-style name ---
    bool threw = false;
    try { enc.set_zstd_parameter(ZARC_GPU_P_COMPRESSION_LEVEL, 99); } catch (const zarc::Error &e) { threw = e.code == ZARC_GPU_E_PARAM; }
    CHECK(threw);
    std::printf("host mirror OK: %zu frames, archive body %llu bytes\n", enc.frames().size(), (unsigned long long)enc.offset());
    return 0;
}
