// zarc_amd/host/zarc_cli.cpp -- `zarc pack | unpack | list-files` over the engine (SURVEY.md section 8 rows f2 + f3).
//
// Same verbs, flags and outputs as the reference CLI (crates/zarc-cli/src/args.rs:17-84):
//   pack        --output PATH [--level N] [--zstd PARAM=VALUE]... [--store] [-L|--follow-symlinks] PATH...
//               crates/zarc-cli/src/pack.rs:9-84,219-272: ChecksumFlag(true) always, walk every PATH, file contents
//               -> add_data_frame, entry -> add_file_entry, finalise, prints "digest: <base64>"
//   unpack      INPUT [--filter REGEX]... [--verify DIGEST]        crates/zarc-cli/src/unpack.rs:18-138
//   list-files  INPUT [--only-files] [--decorate] [--filter REGEX]...   crates/zarc-cli/src/list_files.rs:8-63
//   global      -v... (warn / info / debug / trace), --log-file [PATH] (JSON lines; a directory gets zarc.<UTC time>.log), $RUST_LOG
//               takes precedence -- crates/zarc-cli/src/args.rs:39-65, logs.rs:12-67
// What differs on purpose: file contents are gathered into batches of about 1 GiB before they go to the engine (frames
// and directory order do not change: frames in walk order, first occurrence of a content wins); the walk is sorted by
// name (WalkDir yields directory order).  Metadata carried (metadata/encode.rs:28-372): mode, owner / group (id + name through a
// uid / gid cache, owner_cache.rs:14-77), modified / accessed times, directories, symlinks with their target as a full path,
// chattr flags as `linux.*` attributes, extended attributes.
// I/O pipeline: pack reads the files of batch k+1 on a reader thread while the engine packs batch k (whose
// frames the engine's own staging moves over PCIe in chunks); unpack writes the files of batch k on a writer thread while
// batch k+1 is decoded.  The reference does all of it on one thread, file by file (pack.rs:244-265, unpack.rs:94-124).
#include "zarc_container.hpp"
#include <dirent.h>
#include <fcntl.h>
#include <fstream>
#include <grp.h>
#include <iostream>
#include <pwd.h>
#include <regex>
#include <condition_variable>
#include <deque>
#include <linux/fs.h>
#include <mutex>
#include <sys/ioctl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/xattr.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>

namespace {

// ---------------------------------------------------------------- diagnostics (logs.rs:12-67) ----------------
// Levels as tracing's: 1 warn, 2 info, 3 debug, 4 trace.  Plain lines on stderr, or JSON lines in --log-file.
struct Log {
    int level = 0;
    FILE *file = nullptr;
    std::mutex mu;
    void init(int verbosity, const std::string &log_file, bool have_log_file)
    {
        if (const char *e = getenv("RUST_LOG")) { // "If $RUST_LOG is set, this flag is ignored" (args.rs:49)
            const std::string v = e;
            level = v.find("trace") != std::string::npos ? 4 : v.find("debug") != std::string::npos ? 3 : v.find("info") != std::string::npos ? 2 :
                    v.find("warn") != std::string::npos ? 1 : (v.find("error") != std::string::npos ? 1 : 2);
            return;
        }
        if (have_log_file && verbosity == 0) verbosity = 3; // "If a log level was not already specified, this will set it to -vvv"
        if (verbosity <= 0) return;
        level = verbosity > 4 ? 4 : verbosity;
        if (have_log_file) {
            std::string path = log_file.empty() ? "." : log_file;
            struct stat st;
            if (stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) {
                char name[64];
                const time_t now = time(nullptr);
                std::tm tm{};
                gmtime_r(&now, &tm);
                std::strftime(name, sizeof name, "zarc.%Y-%m-%dT%H-%M-%SZ.log", &tm);
                path += "/" + std::string(name);
            }
            file = std::fopen(path.c_str(), "w");
            if (!file) std::fprintf(stderr, "Failed to initialise logging, continuing with none\n%s: %s\n", path.c_str(), std::strerror(errno));
        }
        event(2, "logging initialised", "");
    }
    void event(int lvl, const char *msg, const std::string &fields) // fields: pre-formatted `key=value ...`
    {
        if (lvl > level || level == 0) return;
        static const char *names[] = {"", "WARN", "INFO", "DEBUG", "TRACE"};
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        std::tm tm{};
        gmtime_r(&ts.tv_sec, &tm);
        char t[40];
        std::strftime(t, sizeof t, "%Y-%m-%dT%H:%M:%S", &tm);
        std::lock_guard<std::mutex> g(mu);
        if (file) {
            std::string esc;
            for (char c : fields) { if (c == '"' || c == '\\') esc += '\\'; esc += c; }
            std::fprintf(file, "{\"timestamp\":\"%s.%06ldZ\",\"level\":\"%s\",\"fields\":{\"message\":\"%s\",\"detail\":\"%s\"},\"target\":\"zarc\"}\n", t, ts.tv_nsec / 1000,
                         names[lvl], msg, esc.c_str());
            std::fflush(file);
        } else std::fprintf(stderr, "%s.%06ldZ %5s zarc: %s %s\n", t, ts.tv_nsec / 1000, names[lvl], msg, fields.c_str());
    }
    ~Log() { if (file) std::fclose(file); }
};
Log g_log;
#define LOGF(lvl, msg, ...) do { if ((lvl) <= g_log.level) { char b_[600]; std::snprintf(b_, sizeof b_, __VA_ARGS__); g_log.event((lvl), (msg), b_); } } while (0)

// uid / gid <-> name lookups are slow (over 90 % of a pack before the reference cached them, owner_cache.rs:3-6): cache both ways
struct OwnerCache {
    std::unordered_map<uint32_t, std::optional<std::string>> users, groups;
    std::unordered_map<std::string, std::optional<uint32_t>> uid_by_name, gid_by_name;
    std::optional<std::string> user_from_uid(uint32_t uid)
    {
        auto it = users.find(uid);
        if (it != users.end()) return it->second;
        std::optional<std::string> n;
        if (const passwd *pw = getpwuid((uid_t)uid)) { n = pw->pw_name; uid_by_name[*n] = uid; }
        return users[uid] = n;
    }
    std::optional<std::string> group_from_gid(uint32_t gid)
    {
        auto it = groups.find(gid);
        if (it != groups.end()) return it->second;
        std::optional<std::string> n;
        if (const group *gr = getgrgid((gid_t)gid)) { n = gr->gr_name; gid_by_name[*n] = gid; }
        return groups[gid] = n;
    }
    std::optional<uint32_t> uid_from_name(const std::string &name)
    {
        auto it = uid_by_name.find(name);
        if (it != uid_by_name.end()) return it->second;
        std::optional<uint32_t> v;
        if (const passwd *pw = getpwnam(name.c_str())) v = (uint32_t)pw->pw_uid;
        return uid_by_name[name] = v;
    }
    std::optional<uint32_t> gid_from_name(const std::string &name)
    {
        auto it = gid_by_name.find(name);
        if (it != gid_by_name.end()) return it->second;
        std::optional<uint32_t> v;
        if (const group *gr = getgrnam(name.c_str())) v = (uint32_t)gr->gr_gid;
        return gid_by_name[name] = v;
    }
};
OwnerCache g_owners;

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
    std::fprintf(stderr, "usage: zarc [-v...] [--log-file [PATH]] <pack|unpack|list-files> ...\n"
                         "       zarc pack --output PATH [--level N] [--zstd PARAM=VALUE]... [--store] [-L] [--gpus N] PATH...\n"
                         "       zarc unpack INPUT [--filter REGEX]... [--verify DIGEST] [--gpus N]\n"
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

// chattr flags as `linux.*` booleans plus the portable aliases (metadata/encode.rs:212-329)
std::optional<zarc::File::AttrMap> file_attributes(const Walked &w)
{
    if (!S_ISREG(w.st.st_mode) && !S_ISDIR(w.st.st_mode)) return std::nullopt; // FS_IOC_GETFLAGS needs an open regular file or directory
    const int fd = open(w.path.c_str(), O_RDONLY | O_NONBLOCK | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return std::nullopt;
    int flags = 0;
    const int rc = ioctl(fd, FS_IOC_GETFLAGS, &flags);
    close(fd);
    if (rc != 0) return std::nullopt; // file systems without flags (tmpfs, overlay ...)
    static const struct { const char *name; int bit; } names[] = {
        {"append-only", FS_APPEND_FL}, {"casefold", 0x40000000 /* FS_CASEFOLD_FL */}, {"compressed", FS_COMPR_FL}, {"delete-undo", FS_UNRM_FL},
        {"delete-zero", FS_SECRM_FL}, {"dir-sync", FS_DIRSYNC_FL}, {"encrypted", 0x00000800 /* FS_ENCRYPT_FL */}, {"file-sync", FS_SYNC_FL},
        {"immutable", FS_IMMUTABLE_FL}, {"no-atime", FS_NOATIME_FL}, {"no-backup", FS_NODUMP_FL}, {"no-cow", 0x00800000 /* FS_NOCOW_FL */},
        {"not-compressed", FS_NOCOMP_FL}};
    zarc::File::AttrMap m;
    zarc::File::Attr yes;
    yes.is_bool = true; yes.b = true;
    for (const auto &n : names) if (flags & n.bit) m[std::string("linux.") + n.name] = yes;
    if (m.empty()) return std::nullopt;
    if (m.count("linux.append-only")) m["append-only"] = yes;
    if (m.count("linux.immutable")) m["immutable"] = yes;
    if (m.count("linux.compressed")) m["compressed"] = yes;
    if (!(w.st.st_mode & 0222)) m["read-only"] = yes;
    return m;
}

// extended attributes, names that are valid UTF-8 only (metadata/encode.rs:343-372)
std::optional<zarc::File::AttrMap> file_extended_attributes(const Walked &w)
{
    std::vector<char> names(1024);
    ssize_t n;
    while ((n = llistxattr(w.path.c_str(), names.data(), names.size())) < 0 && errno == ERANGE) names.resize(names.size() * 4);
    if (n < 0) return std::nullopt; // unsupported here
    zarc::File::AttrMap m;
    for (ssize_t i = 0; i < n;) {
        const std::string name(names.data() + i);
        i += (ssize_t)name.size() + 1;
        if (!zarc::valid_utf8(name)) { LOGF(1, "not storing non-Unicode xattr", "path=%s", w.path.c_str()); continue; }
        std::vector<char> val(256);
        ssize_t v;
        while ((v = lgetxattr(w.path.c_str(), name.c_str(), val.data(), val.size())) < 0 && errno == ERANGE) val.resize(val.size() * 4);
        if (v < 0) continue;
        zarc::File::Attr a;
        a.s.assign(val.data(), (size_t)v);
        m[name] = a;
    }
    return m;
}

zarc::File build_file_with_metadata(const Walked &w) // metadata/encode.rs:28-85
{
    zarc::File f;
    f.name = normal_components(w.path);
    f.mode = (uint32_t)w.st.st_mode;
    zarc::File::Owner u, g;
    u.id = (uint64_t)w.st.st_uid;
    u.name = g_owners.user_from_uid((uint32_t)w.st.st_uid);
    g.id = (uint64_t)w.st.st_gid;
    g.name = g_owners.group_from_gid((uint32_t)w.st.st_gid);
    f.user = u; f.group = g;
    f.modified = zarc::Timestamp{(int64_t)w.st.st_mtim.tv_sec, (uint32_t)w.st.st_mtim.tv_nsec};
    f.accessed = zarc::Timestamp{(int64_t)w.st.st_atim.tv_sec, (uint32_t)w.st.st_atim.tv_nsec};
    if (S_ISDIR(w.st.st_mode)) f.special_kind = 1;
    else if (w.is_link && S_ISLNK(w.st.st_mode)) { f.special_kind = 10; f.link_target = w.target; }
    f.attributes = file_attributes(w);
    f.extended_attributes = file_extended_attributes(w);
    LOGF(4, "build_file_with_metadata", "path=%s mode=%o", w.path.c_str(), (unsigned)w.st.st_mode);
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
        else if (a[i].rfind("--level=", 0) == 0) { level = std::atoi(a[i].c_str() + 8); have_level = true; }
        else if (a[i] == "--zstd" && i + 1 < a.size()) { ZstdParam p; if (!parse_zstd_param(a[++i], &p)) { std::fprintf(stderr, "error: invalid --zstd value\n"); return 2; } params.push_back(p); }
        else if (a[i] == "--store") store = true;
        else if (a[i] == "-L" || a[i] == "--follow-symlinks") follow = true;
        else if (a[i] == "--gpus" && i + 1 < a.size()) gpus = std::atoi(a[++i].c_str()); // engine extension: deal every batch to N devices
        else if (!a[i].empty() && a[i][0] == '-') return usage();
        else paths.push_back(a[i]);
    }
    if (output.empty() || gpus < 1 || gpus > 64) return usage();
    if (gpus > zarc_gpu_device_count()) { std::fprintf(stderr, "Error: --gpus %d but %d device(s) are usable\n", gpus, zarc_gpu_device_count()); return 1; }
    std::ofstream file(output, std::ios::binary | std::ios::trunc);
    if (!file) { std::fprintf(stderr, "Error: %s: %s\n", output.c_str(), std::strerror(errno)); return 1; }
    std::vector<int> devices;
    for (int d = 0; d < gpus; d++) devices.push_back(d);
    zarc::ArchiveWriter enc(file, devices);
    enc.set_zstd_parameter(ZARC_GPU_P_CHECKSUM_FLAG, 1); // pack.rs:227
    if (have_level) enc.set_zstd_parameter(ZARC_GPU_P_COMPRESSION_LEVEL, level);
    for (const auto &p : params) enc.set_zstd_parameter(p.id, p.value);
    // What the engine does differently from libzstd with these is said once, on stderr and in the log: a level runs its tier's finder, the
    // search-effort and long-distance-matching hints are accepted (the reference forwards them all, pack.rs:86-217) but change nothing.
    if (have_level && level != 0 && zarc_gpu_level_finder(level) != level) {
        std::fprintf(stderr, "warning: --level %d packs with the engine's level-%d finder (tiers: <= 1, 2..8, 9..14, 15..22)\n", level, zarc_gpu_level_finder(level));
        LOGF(1, "level mapped to a finder tier", "level=%d finder=%d", level, zarc_gpu_level_finder(level));
    }
    for (const auto &p : params)
        if (zarc_gpu_parameter_advisory(p.id)) {
            std::fprintf(stderr, "warning: --zstd parameter %d=%d is accepted but advisory on this engine: the frames are the level's frames\n", p.id, p.value);
            LOGF(1, "advisory zstd parameter", "id=%d value=%d", p.id, p.value);
        }
    if (store) enc.enable_compression(false);

    std::vector<Walked> entries;
    for (const auto &p : paths) walk(p, follow, entries);
    LOGF(2, "walked", "entries=%zu devices=%d", entries.size(), gpus);
    // Contents go to the engine in batches of about 1 GiB.  A reader thread fills batch k+1 (file reads) while this thread has the
    // engine pack batch k and appends its frames to the archive; entries are added in walk order once their digest is known.
    struct Batch { size_t first = 0, last = 0; std::vector<std::vector<uint8_t>> contents; std::vector<size_t> owner; std::string error; };
    const size_t BATCH = (size_t)1 << 30;
    std::deque<Batch> ready;
    std::mutex mu;
    std::condition_variable cv;
    bool reader_done = false, abandon = false;
    std::thread reader([&] {
        size_t first = 0;
        while (first < entries.size()) {
            { std::lock_guard<std::mutex> lk(mu); if (abandon) break; }
            Batch b;
            b.first = first;
            size_t last = first, bytes = 0;
            while (last < entries.size() && (bytes < BATCH || last == first) && b.error.empty()) {
                const Walked &w = entries[last];
                if (S_ISREG(w.st.st_mode)) {
                    std::ifstream in(w.path, std::ios::binary);
                    if (!in) { b.error = w.path + ": " + std::strerror(errno); break; }
                    b.contents.emplace_back((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
                    b.owner.push_back(last);
                    bytes += b.contents.back().size();
                }
                last++;
            }
            b.last = last;
            const bool failed = !b.error.empty();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready.size() < 2 || abandon; }); // at most two batches of file contents in memory
                if (abandon) break;
                ready.push_back(std::move(b));
            }
            cv.notify_all();
            if (failed) break;
            first = last;
        }
        { std::lock_guard<std::mutex> lk(mu); reader_done = true; }
        cv.notify_all();
    });
    // Whatever ends the loop below -- a read error, or an exception out of the engine / the archive writer (zarc::Error: engine
    // failure, an entry of 4 GiB or more, a failed write) -- the reader thread is told to stop and joined before this frame unwinds:
    // a joinable std::thread must never be destroyed (std::terminate), the error goes to main()'s handler ("Error: ...", exit 1).
    struct Joiner {
        std::thread &t; std::mutex &mu; std::condition_variable &cv; bool &stop;
        ~Joiner() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); }
    } joiner{reader, mu, cv, abandon};
    std::string failure;
    for (;;) {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || reader_done; });
            if (ready.empty()) break;
            b = std::move(ready.front());
            ready.pop_front();
        }
        cv.notify_all();
        if (!b.error.empty()) { failure = b.error; break; }
        std::vector<const void *> ptr;
        std::vector<size_t> len;
        for (auto &c : b.contents) { ptr.push_back(c.data()); len.push_back(c.size()); }
        LOGF(3, "add_data_frames", "entries=%zu files=%zu", b.last - b.first, ptr.size());
        const std::vector<zarc::Digest> dig = enc.add_data_frames(ptr.data(), len.data(), ptr.size());
        size_t k = 0;
        for (size_t i = b.first; i < b.last; i++) {
            zarc::File f = build_file_with_metadata(entries[i]);
            if (k < b.owner.size() && b.owner[k] == i) f.digest = dig[k++];
            enc.add_file_entry(f);
        }
    }
    if (!failure.empty()) { std::fprintf(stderr, "Error: %s\n", failure.c_str()); return 1; }
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
    // by name first, then by id (directory/posix_owner.rs:25-110), through the cache
    if (f.user) { std::optional<uint32_t> v; if (f.user->name) v = g_owners.uid_from_name(*f.user->name); if (!v && f.user->id) v = (uint32_t)*f.user->id; if (v) uid = (uid_t)*v; }
    if (f.group) { std::optional<uint32_t> v; if (f.group->name) v = g_owners.gid_from_name(*f.group->name); if (!v && f.group->id) v = (uint32_t)*f.group->id; if (v) gid = (gid_t)*v; }
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
    int gpus = 1;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i] == "--filter" && i + 1 < a.size()) filters.emplace_back(a[++i]);
        else if (a[i] == "--verify" && i + 1 < a.size()) verify = a[++i];
        else if (a[i] == "--gpus" && i + 1 < a.size()) gpus = std::atoi(a[++i].c_str());
        else if (!a[i].empty() && a[i][0] == '-') return usage();
        else input = a[i];
    }
    if (input.empty() || gpus < 1 || gpus > 64) return usage();
    if (gpus > zarc_gpu_device_count()) { std::fprintf(stderr, "Error: --gpus %d but %d device(s) are usable\n", gpus, zarc_gpu_device_count()); return 1; }
    Mapped m(input);
    std::vector<int> devices;
    for (int d = 0; d < gpus; d++) devices.push_back(d);
    zarc::ArchiveReader rd(m.p, m.n, devices); // frames of a batch are dealt to the devices by uncompressed bytes (zarc_host.hpp)
    const std::string digest = base64(rd.trailer().digest.bytes.data(), 32);
    if (!verify.empty()) {
        if (verify != digest) { std::fprintf(stderr, "Error: integrity failure: zarc file digest is %s\n", digest.c_str()); return 1; }
    } else std::fprintf(stderr, "digest: %s\n", digest.c_str());
    unsigned long long unpacked = 0;
    const size_t BATCH = (size_t)1 << 30;
    std::vector<size_t> batch;
    size_t batch_bytes = 0;
    // A writer thread creates and fills the files of batch k while this thread has the engine decode batch k+1.
    struct Done { std::vector<size_t> idx; std::vector<zarc::FrameReader::Result> res; };
    std::deque<Done> todo;
    std::mutex mu;
    std::condition_variable cv;
    bool no_more = false;
    std::string failure;
    std::thread writer([&] {
        for (;;) {
            Done d;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !todo.empty() || no_more; });
                if (todo.empty()) return;
                d = std::move(todo.front());
                todo.pop_front();
            }
            cv.notify_all();
            for (size_t k = 0; k < d.idx.size(); k++) {
                const zarc::File &f = rd.files()[d.idx[k]];
                const std::string path = to_path(f.name);
                std::string err;
                if (d.res[k].status != ZARC_GPU_FRAME_OK && d.res[k].status != ZARC_GPU_FRAME_DIGEST) err = path + ": " + zarc_gpu_frame_status_name(d.res[k].status);
                int fd = -1;
                if (err.empty()) {
                    const size_t slash = path.rfind('/');
                    if (slash != std::string::npos) mkdirs(path.substr(0, slash), 0777); // parent, in case its entry was not in the zarc
                    fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_NOFOLLOW, 0666);
                    if (fd < 0) err = path + ": " + std::strerror(errno);
                }
                size_t off = 0;
                while (err.empty() && off < d.res[k].data.size()) {
                    const ssize_t w = write(fd, d.res[k].data.data() + off, d.res[k].data.size() - off);
                    if (w <= 0) err = path + ": write failed"; else off += (size_t)w;
                }
                if (!err.empty()) { if (fd >= 0) close(fd); std::lock_guard<std::mutex> lk(mu); if (failure.empty()) failure = err; continue; }
                if (!d.res[k].verify.value_or(false)) std::fprintf(stderr, "ERROR frame verification failed! path=%s\n", path.c_str()); // unpack.rs:118-120
                set_metadata(f, fd);
                close(fd);
                LOGF(3, "extract_file", "path=%s bytes=%zu", path.c_str(), d.res[k].data.size());
                std::lock_guard<std::mutex> lk(mu);
                unpacked++;
            }
        }
    });
    auto flush = [&]() {
        if (batch.empty()) return;
        Done d;
        d.idx = batch;
        d.res = rd.read_files(batch);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return todo.size() < 2; }); // at most two decoded batches in memory
            todo.push_back(std::move(d));
        }
        cv.notify_all();
        batch.clear();
        batch_bytes = 0;
    };
    // As in cmd_pack: the writer thread is stopped and joined on every way out of this function, exceptions included (a corrupt
    // frame record makes read_files throw in the middle of the loop).
    struct Joiner {
        std::thread &t; std::mutex &mu; std::condition_variable &cv; bool &stop;
        ~Joiner() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); }
    } joiner{writer, mu, cv, no_more};
    // Directories are MADE when their entry comes up (the files below them follow in the archive), but their metadata is applied
    // last, deepest first, once the writer thread has put every file on disk: creating a file inside a directory moves the
    // directory's mtime, and a read-only mode would refuse the files.  (The reference applies it at once, unpack.rs:62-88, and so
    // restores directory mtimes only for empty directories.)
    std::vector<size_t> dirs;
    for (size_t i = 0; i < rd.files().size(); i++) {
        const zarc::File &f = rd.files()[i];
        const std::string name = to_path(f.name);
        if (!passes(filters, name)) continue;
        if (!safe_name(f.name)) { std::fprintf(stderr, "WARN unsafe pathname skipped: %s\n", name.c_str()); continue; }
        if (f.is_dir()) {
            mkdirs(name, 0777);
            dirs.push_back(i);
        } else if (f.is_normal()) {
            auto it = rd.frames().find(*f.digest);
            if (it == rd.frames().end()) { std::fprintf(stderr, "WARN frame not found\n"); continue; } // unpack.rs:107-110
            batch.push_back(i);
            batch_bytes += (size_t)it->second.uncompressed;
            if (batch_bytes >= BATCH) flush();
        }
    }
    flush();
    { std::lock_guard<std::mutex> lk(mu); no_more = true; }
    cv.notify_all();
    writer.join();
    if (!failure.empty()) throw zarc::Error(ZARC_GPU_E_PARAM, failure);
    std::stable_sort(dirs.begin(), dirs.end(), [&](size_t x, size_t y) { return rd.files()[x].name.size() > rd.files()[y].name.size(); }); // more components first
    for (size_t i : dirs) {
        const zarc::File &f = rd.files()[i];
        const int fd = open(to_path(f.name).c_str(), O_RDONLY | O_DIRECTORY);
        if (fd >= 0) { set_metadata(f, fd); close(fd); }
    }
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
    // global flags come before the subcommand (args.rs:39-65): -v / -vv / --verbose (counted), --log-file [PATH]
    int verbosity = 0, i = 1;
    bool have_log_file = false;
    std::string log_file;
    for (; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--verbose") verbosity++;
        else if (a.size() >= 2 && a[0] == '-' && a[1] == 'v' && a.find_first_not_of('v', 1) == std::string::npos) verbosity += (int)a.size() - 1;
        else if (a.rfind("--log-file=", 0) == 0) { have_log_file = true; log_file = a.substr(11); }
        else if (a == "--log-file") {
            have_log_file = true;
            // num_args = 0..=1: a following word that is not a subcommand is the path
            if (i + 1 < argc) { const std::string n = argv[i + 1]; const bool verb = !n.empty() && (std::string("pack").rfind(n, 0) == 0 || std::string("unpack").rfind(n, 0) == 0 || std::string("list-files").rfind(n, 0) == 0); if (!verb && n[0] != '-') log_file = argv[++i]; }
        } else break;
    }
    if (i >= argc) return usage();
    g_log.init(verbosity, log_file, have_log_file);
    const std::string verb = argv[i];
    std::vector<std::string> rest(argv + i + 1, argv + argc);
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
