// zarc_amd/host/zarc_cli.cpp -- `zarc pack | unpack | list-files` over the engine (SURVEY.md section 8 rows f2 + f3).
//
// Same verbs, flags and outputs as the reference CLI (crates/zarc-cli/src/args.rs:17-84):
//   pack        --output PATH [--level N] [--zstd PARAM=VALUE]... [--store] [-L|--follow-symlinks] PATH...
//               crates/zarc-cli/src/pack.rs:9-84,219-272: ChecksumFlag(true) always, walk every PATH, file contents
//               -> add_data_frame, entry -> add_file_entry, finalise, prints "digest: <base64>"
//   unpack      INPUT [--filter REGEX]... [--verify DIGEST]        crates/zarc-cli/src/unpack.rs:18-138
//   list-files  INPUT [--only-files] [--decorate] [--filter REGEX]...   crates/zarc-cli/src/list_files.rs:8-63
// What differs on purpose: file contents are gathered into batches of about 1 GiB before they go to the engine (frames
// and directory order do not change: frames in walk order, first occurrence of a content wins); the walk is sorted by
// name (WalkDir yields directory order); metadata carried: mode, owner/group (id + name), modified / accessed times,
// directories, symlinks with their target as a full path (metadata/encode.rs:28-75) -- no chattr flags or xattrs.
#include "zarc_container.hpp"
#include <dirent.h>
#include <fcntl.h>
#include <fstream>
#include <grp.h>
#include <iostream>
#include <pwd.h>
#include <regex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

std::string base64(const uint8_t *p, size_t n) // base64ct::Base64: standard alphabet, padded
{
    static const char *T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string s;
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t v = (uint32_t)p[i] << 16 | (i + 1 < n ? (uint32_t)p[i + 1] << 8 : 0) | (i + 2 < n ? p[i + 2] : 0);
        s += T[v >> 18]; s += T[(v >> 12) & 63];
        s += i + 1 < n ? T[(v >> 6) & 63] : '=';
        s += i + 2 < n ? T[v & 63] : '=';
    }
    return s;
}

std::vector<std::string> normal_components(const std::string &path) // Pathname::from_normal_components (strings.rs:18-31)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < path.size()) {
        size_t j = path.find('/', i);
        if (j == std::string::npos) j = path.size();
        const std::string c = path.substr(i, j - i);
        if (!c.empty() && c != "." && c != "..") out.push_back(c);
        i = j + 1;
    }
    return out;
}
std::string to_path(const std::vector<std::string> &name) // Pathname::to_path (strings.rs:33-55)
{
    std::string p;
    for (const auto &c : name) { if (!p.empty()) p += '/'; p += c; }
    return p;
}

// A pathname from an archive is untrusted: a component that is empty, ".", "..", or contains '/' or NUL would let a crafted
// archive write outside the extraction directory (the reference joins components with PathBuf::push and has the same hole).
bool safe_name(const std::vector<std::string> &name)
{
    if (name.empty()) return false;
    for (const auto &c : name)
        if (c.empty() || c == "." || c == ".." || c.find('/') != std::string::npos || c.find('\0') != std::string::npos) return false;
    return true;
}

int usage()
{
    std::fprintf(stderr, "usage: zarc pack --output PATH [--level N] [--zstd PARAM=VALUE]... [--store] [-L] [--gpus N] PATH...\n"
                         "       zarc unpack INPUT [--filter REGEX]... [--verify DIGEST]\n"
                         "       zarc list-files INPUT [--only-files] [--decorate] [--filter REGEX]...\n");
    return 2;
}

// ---------------------------------------------------------------- pack --------------------------------------
struct ZstdParam { int id; int value; };
bool parse_zstd_param(const std::string &s, ZstdParam *out) // pack.rs:86-217: NAME=VALUE, names of zstd_safe::CParameter
{
    const size_t eq = s.find('=');
    if (eq == std::string::npos) return false;
    const std::string name = s.substr(0, eq), val = s.substr(eq + 1);
    static const struct { const char *n; int id; } ids[] = {
        {"CompressionLevel", 100}, {"WindowLog", 101}, {"HashLog", 102}, {"ChainLog", 103}, {"SearchLog", 104}, {"MinMatch", 105},
        {"TargetLength", 106}, {"Strategy", 107}, {"EnableLongDistanceMatching", 160}, {"LdmHashLog", 161}, {"LdmMinMatch", 162},
        {"LdmBucketSizeLog", 163}, {"LdmHashRateLog", 164}, {"ContentSizeFlag", 200}, {"ChecksumFlag", 201}, {"DictIdFlag", 202},
        {"NbWorkers", 400}, {"JobSize", 401}, {"OverlapSizeLog", 402}};
    static const char *strategies[] = {"", "fast", "dfast", "greedy", "lazy", "lazy2", "btlazy2", "btopt", "btultra", "btultra2"};
    for (const auto &e : ids)
        if (name == e.n) {
            out->id = e.id;
            if (val == "true") out->value = 1;
            else if (val == "false") out->value = 0;
            else if (e.id == 107 && !val.empty() && !std::isdigit((unsigned char)val[0])) {
                out->value = 0;
                for (int k = 1; k < 10; k++) if (val == strategies[k]) out->value = k;
                if (!out->value) return false;
            } else out->value = std::atoi(val.c_str());
            return true;
        }
    return false;
}

struct Walked { std::string path; struct stat st; bool is_link; std::string target; };
void walk(const std::string &path, bool follow, std::vector<Walked> &out) // WalkDir::new(path).follow_links(follow), sorted
{
    Walked w;
    w.path = path;
    struct stat lst;
    if (lstat(path.c_str(), &lst) != 0) { std::fprintf(stderr, "read error: %s: %s\n", path.c_str(), std::strerror(errno)); return; }
    w.is_link = S_ISLNK(lst.st_mode);
    if (w.is_link) {
        char buf[4096];
        const ssize_t n = readlink(path.c_str(), buf, sizeof buf);
        if (n > 0) w.target.assign(buf, (size_t)n);
    }
    w.st = lst;
    if (w.is_link && follow && stat(path.c_str(), &w.st) != 0) w.st = lst; // dangling link: keep what lstat said
    out.push_back(w);
    if (!S_ISDIR(w.st.st_mode) || (w.is_link && !follow)) return;
    std::vector<std::string> names;
    if (DIR *d = opendir(path.c_str())) {
        while (dirent *e = readdir(d)) { const std::string n = e->d_name; if (n != "." && n != "..") names.push_back(n); }
        closedir(d);
    } else std::fprintf(stderr, "read error: %s: %s\n", path.c_str(), std::strerror(errno));
    std::sort(names.begin(), names.end());
    for (const auto &n : names) walk(path + (path.back() == '/' ? "" : "/") + n, follow, out);
}

zarc::File build_file_with_metadata(const Walked &w) // metadata/encode.rs:28-75
{
    zarc::File f;
    f.name = normal_components(w.path);
    f.mode = (uint32_t)w.st.st_mode;
    zarc::File::Owner u, g;
    u.id = (uint64_t)w.st.st_uid;
    if (const passwd *pw = getpwuid(w.st.st_uid)) u.name = pw->pw_name;
    g.id = (uint64_t)w.st.st_gid;
    if (const group *gr = getgrgid(w.st.st_gid)) g.name = gr->gr_name;
    f.user = u; f.group = g;
    f.modified = zarc::Timestamp{(int64_t)w.st.st_mtim.tv_sec, (uint32_t)w.st.st_mtim.tv_nsec};
    f.accessed = zarc::Timestamp{(int64_t)w.st.st_atim.tv_sec, (uint32_t)w.st.st_atim.tv_nsec};
    if (S_ISDIR(w.st.st_mode)) f.special_kind = 1;
    else if (w.is_link && S_ISLNK(w.st.st_mode)) { f.special_kind = 10; f.link_target = w.target; }
    return f;
}

int cmd_pack(const std::vector<std::string> &a)
{
    std::string output;
    std::vector<std::string> paths;
    std::vector<ZstdParam> params;
    bool store = false, follow = false, have_level = false;
    int level = 0, gpus = 1;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i] == "--output" && i + 1 < a.size()) output = a[++i];
        else if (a[i] == "--level" && i + 1 < a.size()) { level = std::atoi(a[++i].c_str()); have_level = true; }
        else if (a[i] == "--zstd" && i + 1 < a.size()) { ZstdParam p; if (!parse_zstd_param(a[++i], &p)) { std::fprintf(stderr, "error: invalid --zstd value\n"); return 2; } params.push_back(p); }
        else if (a[i] == "--store") store = true;
        else if (a[i] == "-L" || a[i] == "--follow-symlinks") follow = true;
        else if (a[i] == "--gpus" && i + 1 < a.size()) gpus = std::atoi(a[++i].c_str()); // engine extension: deal every batch to N devices
        else if (!a[i].empty() && a[i][0] == '-') return usage();
        else paths.push_back(a[i]);
    }
    if (output.empty() || gpus < 1 || gpus > 64) return usage();
    std::ofstream file(output, std::ios::binary | std::ios::trunc);
    if (!file) { std::fprintf(stderr, "Error: %s: %s\n", output.c_str(), std::strerror(errno)); return 1; }
    std::vector<int> devices;
    for (int d = 0; d < gpus; d++) devices.push_back(d);
    zarc::ArchiveWriter enc(file, devices);
    enc.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1); // pack.rs:227
    if (have_level) enc.set_zstd_parameter(ZARC_GPU_P_COMPRESSION_LEVEL, level);
    for (const auto &p : params) enc.set_zstd_parameter(p.id, p.value);
    if (store) enc.enable_compression(false);

    std::vector<Walked> entries;
    for (const auto &p : paths) walk(p, follow, entries);
    // contents go to the engine in batches; entries are added in walk order once their digest is known
    const size_t BATCH = (size_t)1 << 30;
    size_t first = 0;
    while (first < entries.size()) {
        size_t last = first, bytes = 0;
        std::vector<std::vector<uint8_t>> contents;
        std::vector<size_t> owner;
        while (last < entries.size() && (bytes < BATCH || last == first)) {
            const Walked &w = entries[last];
            if (S_ISREG(w.st.st_mode)) {
                std::ifstream in(w.path, std::ios::binary);
                if (!in) { std::fprintf(stderr, "Error: %s: %s\n", w.path.c_str(), std::strerror(errno)); return 1; }
                contents.emplace_back((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
                owner.push_back(last);
                bytes += contents.back().size();
            }
            last++;
        }
        std::vector<const void *> ptr;
        std::vector<size_t> len;
        for (auto &c : contents) { ptr.push_back(c.data()); len.push_back(c.size()); }
        const std::vector<zarc::Digest> dig = enc.add_data_frames(ptr.data(), len.data(), ptr.size());
        size_t k = 0;
        for (size_t i = first; i < last; i++) {
            zarc::File f = build_file_with_metadata(entries[i]);
            if (k < owner.size() && owner[k] == i) f.digest = dig[k++];
            enc.add_file_entry(f);
        }
        first = last;
    }
    timespec now;
    clock_gettime(CLOCK_REALTIME, &now);
    const zarc::Digest digest = enc.finalise(zarc::Timestamp{(int64_t)now.tv_sec, (uint32_t)now.tv_nsec});
    std::printf("digest: %s\n", base64(digest.bytes.data(), 32).c_str());
    return 0;
}

// ---------------------------------------------------------------- unpack / list-files -----------------------
struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;
    explicit Mapped(const std::string &path)
    {
        fd = open(path.c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) throw zarc::Error(ZARC_GPU_E_PARAM, path + ": " + std::strerror(errno));
        n = (size_t)st.st_size;
        if (n) { void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) throw zarc::Error(ZARC_GPU_E_PARAM, path + ": mmap failed"); p = (const uint8_t *)m; }
    }
    ~Mapped() { if (p) munmap((void *)p, n); if (fd >= 0) close(fd); }
};

bool passes(const std::vector<std::regex> &filters, const std::string &name)
{
    if (filters.empty()) return true;
    for (const auto &f : filters) if (std::regex_search(name, f)) return true;
    return false;
}

void mkdirs(const std::string &path, mode_t mode)
{
    for (size_t i = 1; i <= path.size(); i++)
        if (i == path.size() || path[i] == '/') { const std::string sub = path.substr(0, i); if (!sub.empty()) (void)mkdir(sub.c_str(), i == path.size() ? mode : 0777); }
}

void set_metadata(const zarc::File &f, int fd) // unpack.rs:126-138: ownership, permissions, timestamps
{
    uid_t uid = (uid_t)-1; gid_t gid = (gid_t)-1;
    if (f.user) { if (f.user->name) { if (const passwd *pw = getpwnam(f.user->name->c_str())) uid = pw->pw_uid; else if (f.user->id) uid = (uid_t)*f.user->id; } else if (f.user->id) uid = (uid_t)*f.user->id; }
    if (f.group) { if (f.group->name) { if (const group *gr = getgrnam(f.group->name->c_str())) gid = gr->gr_gid; else if (f.group->id) gid = (gid_t)*f.group->id; } else if (f.group->id) gid = (gid_t)*f.group->id; }
    if ((uid != (uid_t)-1 || gid != (gid_t)-1) && fchown(fd, uid, gid) != 0) { /* needs privileges; an unprivileged unpack keeps the caller's ids */ }
    if (f.mode) (void)fchmod(fd, (mode_t)(*f.mode & 07777));
    if (f.modified || f.accessed) {
        timespec ts[2];
        ts[0].tv_sec = f.accessed ? (time_t)f.accessed->secs : 0; ts[0].tv_nsec = f.accessed ? (long)f.accessed->nanos : UTIME_OMIT;
        ts[1].tv_sec = f.modified ? (time_t)f.modified->secs : 0; ts[1].tv_nsec = f.modified ? (long)f.modified->nanos : UTIME_OMIT;
        (void)futimens(fd, ts);
    }
}

int cmd_unpack(const std::vector<std::string> &a)
{
    std::string input, verify;
    std::vector<std::regex> filters;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i] == "--filter" && i + 1 < a.size()) filters.emplace_back(a[++i]);
        else if (a[i] == "--verify" && i + 1 < a.size()) verify = a[++i];
        else if (!a[i].empty() && a[i][0] == '-') return usage();
        else input = a[i];
    }
    if (input.empty()) return usage();
    Mapped m(input);
    zarc::ArchiveReader rd(m.p, m.n);
    const std::string digest = base64(rd.trailer().digest.bytes.data(), 32);
    if (!verify.empty()) {
        if (verify != digest) { std::fprintf(stderr, "Error: integrity failure: zarc file digest is %s\n", digest.c_str()); return 1; }
    } else std::fprintf(stderr, "digest: %s\n", digest.c_str());
    unsigned long long unpacked = 0;
    const size_t BATCH = (size_t)1 << 30;
    std::vector<size_t> batch;
    size_t batch_bytes = 0;
    auto flush = [&]() {
        if (batch.empty()) return;
        auto res = rd.read_files(batch);
        for (size_t k = 0; k < batch.size(); k++) {
            const zarc::File &f = rd.files()[batch[k]];
            const std::string path = to_path(f.name);
            if (res[k].status != ZARC_GPU_FRAME_OK && res[k].status != ZARC_GPU_FRAME_DIGEST)
                throw zarc::Error(res[k].status, path + ": " + zarc_gpu_frame_status_name(res[k].status));
            const size_t slash = path.rfind('/');
            if (slash != std::string::npos) mkdirs(path.substr(0, slash), 0777); // parent, in case its entry was not in the zarc
            const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_NOFOLLOW, 0666);
            if (fd < 0) throw zarc::Error(ZARC_GPU_E_PARAM, path + ": " + std::strerror(errno));
            size_t off = 0;
            while (off < res[k].data.size()) { const ssize_t w = write(fd, res[k].data.data() + off, res[k].data.size() - off); if (w <= 0) { close(fd); throw zarc::Error(ZARC_GPU_E_PARAM, path + ": write failed"); } off += (size_t)w; }
            if (!res[k].verify.value_or(false)) std::fprintf(stderr, "ERROR frame verification failed! path=%s\n", path.c_str()); // unpack.rs:118-120
            set_metadata(f, fd);
            close(fd);
            unpacked++;
        }
        batch.clear();
        batch_bytes = 0;
    };
    for (size_t i = 0; i < rd.files().size(); i++) {
        const zarc::File &f = rd.files()[i];
        const std::string name = to_path(f.name);
        if (!passes(filters, name)) continue;
        if (!safe_name(f.name)) { std::fprintf(stderr, "WARN unsafe pathname skipped: %s\n", name.c_str()); continue; }
        if (f.is_dir()) {
            flush();
            mkdirs(name, f.mode ? (mode_t)(*f.mode & 07777) : 0777);
            const int fd = open(name.c_str(), O_RDONLY | O_DIRECTORY);
            if (fd >= 0) { set_metadata(f, fd); close(fd); }
        } else if (f.is_normal()) {
            auto it = rd.frames().find(*f.digest);
            if (it == rd.frames().end()) { std::fprintf(stderr, "WARN frame not found\n"); continue; } // unpack.rs:107-110
            batch.push_back(i);
            batch_bytes += (size_t)it->second.uncompressed;
            if (batch_bytes >= BATCH) flush();
        }
    }
    flush();
    std::fprintf(stderr, "unpacked %llu files\n", unpacked);
    return 0;
}

int cmd_list_files(const std::vector<std::string> &a)
{
    std::string input;
    std::vector<std::regex> filters;
    bool only_files = false;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i] == "--filter" && i + 1 < a.size()) filters.emplace_back(a[++i]);
        else if (a[i] == "--only-files") only_files = true;
        else if (a[i] == "--decorate") {} // accepted; the reference decorates whether or not it is given (list_files.rs:50-56)
        else if (!a[i].empty() && a[i][0] == '-') return usage();
        else input = a[i];
    }
    if (input.empty()) return usage();
    Mapped m(input);
    zarc::ArchiveReader rd(m.p, m.n);
    for (const zarc::File &f : rd.files()) {
        if (only_files && f.special_kind) continue;
        const std::string name = to_path(f.name);
        if (!passes(filters, name)) continue;
        std::printf("%s%s\n", name.c_str(), f.is_dir() ? "/" : (f.is_symlink() ? "@" : (f.is_hardlink() ? "#" : "")));
    }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 2) return usage();
    const std::string verb = argv[1];
    std::vector<std::string> rest(argv + 2, argv + argc);
    try {
        // `infer_subcommands = true` (args.rs:23): unambiguous prefixes select the subcommand
        if (!verb.empty() && std::string("pack").rfind(verb, 0) == 0) return cmd_pack(rest);
        if (!verb.empty() && std::string("unpack").rfind(verb, 0) == 0) return cmd_unpack(rest);
        if (!verb.empty() && std::string("list-files").rfind(verb, 0) == 0) return cmd_list_files(rest);
        return usage();
    } catch (const zarc::Error &e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    } catch (const std::regex_error &e) {
        std::fprintf(stderr, "error: invalid regex: %s\n", e.what());
        return 2;
    }
}
