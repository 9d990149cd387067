// hostpass.cpp -- the host-side half of the hashes.yaml pass: tree walk, record
// model, yaml.v2-compatible emitter, tolerant parser, shard assignment.
// No device code and no hashing here (digests come from the HIP kernels).
//
// Mirrors, from the reference tree:
//   snappy/build.go:228-259    the filepath.Walk callback of writeHashes
//   snappy/hashes.go:33-88     yamlFileMode.MarshalYAML / UnmarshalYAML
//   snappy/hashes.go:93-110    fileHash / hashesYaml field order and omitempty
//   snappy/hashes_test.go:89-103  the byte layout yaml.v2 gives that schema
#include "hostpass.h"

#include "hostfill.h"
#include "walk.h"

#include <dirent.h>
#include <errno.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <string_view>
#include <functional>
#include <queue>
#include <thread>

namespace snaphash {

// ---- yamlFileMode -------------------------------------------------------------

int mode_string(uint32_t st_mode, char out[11])
{
    memcpy(out, "----------", 11);
    // Go: ModeDir -> 'd', ModeSymlink -> 'l', (mode & ModeType)==0 -> 'f',
    // anything else (device, fifo, socket) -> "Unknown file mode" (hashes.go:36-47)
    if (S_ISDIR(st_mode)) out[0] = 'd';
    else if (S_ISLNK(st_mode)) out[0] = 'l';
    else if (S_ISREG(st_mode)) out[0] = 'f';
    else return SNAPHASH_EMODE;
    static const char rwx[] = "rwxrwxrwx";
    for (int i = 0; i < 9; ++i)
        if (st_mode & (1u << (8 - i))) out[i + 1] = rwx[i]; // setuid/setgid/sticky are not in Go's low 9 bits
    return SNAPHASH_OK;
}

int mode_parse(const char* s, uint32_t* st_mode)
{
    if (!s || !s[0]) return SNAPHASH_EPARSE; // the reference indexes modeAsStr[0] unguarded (hashes.go:67)
    uint32_t m;
    switch (s[0]) {
    case 'd': m = S_IFDIR; break;
    case 'f': m = S_IFREG; break;
    case 'l': m = S_IFLNK; break;
    default: return SNAPHASH_EPARSE;
    }
    static const char rwx[] = "rwxrwxrwx";
    for (int i = 0; i < 9 && s[i + 1]; ++i)
        if (s[i + 1] == rwx[i]) m |= 1u << (8 - i); // a perm char counts iff it is the expected letter (hashes.go:81-85)
    *st_mode = m;
    return SNAPHASH_OK;
}

// ---- filepath.Walk as writeHashes drives it ------------------------------------

// The records of the entries writeHashes' callback keeps (walk.h gives the entries in Walk's order with their Lstat).
// SNAPHASH_EMODE for the first entry whose type yamlFileMode cannot express (hashes.go:33-57).
// steal: the entries' path strings move into the records (the caller is done with the entries)
static int records_from_entries_impl(std::vector<WalkEntry>& ents, std::vector<Record>& out, bool steal)
{
    if (ents.empty()) return SNAPHASH_OK;
    const size_t rootlen = ents[0].path.size();
    const size_t n = ents.size();
    // Two passes, both in ranges on a few threads when the tree is large (10 100 entries were 0.4 ms of string work on the
    // caller's thread, a seventh of the walk): which entries the callback keeps and whether their type has a mode string
    // (the FIRST that has none fails the pass, as the serial loop would: what lies behind it is never looked at), then the
    // records, each at the place the count of kept entries in front of it gives.
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(8u, usable_cpus()), n / 2048));
    std::vector<uint8_t> keep(n, 0);
    std::vector<size_t> kept(T + 1, 0), bad(T, n);
    auto range = [&](unsigned t, size_t& lo, size_t& hi) { lo = n * t / T; hi = n * (t + 1) / T; };
    run_on_threads(T, [&](unsigned t) {
        size_t lo, hi, c = 0;
        range(t, lo, hi);
        for (size_t i = lo; i < hi; ++i) {
            const char* rel = ents[i].path.c_str() + rootlen;
            // build.go:229: string prefix, not path component -- "/DEBIAN-extra" is skipped too;
            // build.go:232: the root itself.  The callback returns nil (not SkipDir), so
            // Walk still descends into DEBIAN and skips its children one by one.
            if (rel[0] == 0 || strncmp(rel, "/DEBIAN", 7) == 0) continue;
            char m[11];
            if (mode_string(ents[i].st.st_mode, m) != SNAPHASH_OK) { bad[t] = i; break; }
            keep[i] = 1;
            ++c;
        }
        kept[t + 1] = c;
    });
    size_t first_bad = n;
    for (unsigned t = 0; t < T; ++t) first_bad = std::min(first_bad, bad[t]);
    if (first_bad < n) return SNAPHASH_EMODE; // (entries kept in front of it are of no use to anybody: the pass fails)
    for (unsigned t = 0; t < T; ++t) kept[t + 1] += kept[t];
    const size_t base = out.size();
    out.resize(base + kept[T]);
    run_on_threads(T, [&](unsigned t) {
        size_t lo, hi;
        range(t, lo, hi);
        size_t at = base + kept[t];
        for (size_t i = lo; i < hi; ++i) {
            if (!keep[i]) continue;
            WalkEntry& e = ents[i];
            Record& r = out[at++];
            r.name.assign(e.path.c_str() + rootlen + 1, e.path.size() - rootlen - 1); // build.go:250
            r.st_mode = e.st.st_mode;
            r.is_regular = S_ISREG(e.st.st_mode); // build.go:240
            r.size = r.is_regular ? (int64_t)e.st.st_size : 0;
            if (steal) r.path = std::move(e.path);
            else r.path = e.path;
        }
    });
    return SNAPHASH_OK;
}

// The first record whose name the YAML emitter does not restate (yamlscalar.cpp), or recs.size(): in ranges on a few
// threads for a large tree.
size_t first_unemittable_name(const std::vector<Record>& recs)
{
    const size_t n = recs.size();
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(8u, usable_cpus()), n / 2048));
    std::vector<size_t> bad(T, n);
    run_on_threads(T, [&](unsigned t) {
        for (size_t i = n * t / T; i < n * (t + 1) / T; ++i)
            if (!name_emittable(recs[i].name)) { bad[t] = i; break; }
    });
    size_t first = n;
    for (size_t b : bad) first = std::min(first, b);
    return first;
}

int records_from_entries(const std::vector<WalkEntry>& ents, std::vector<Record>& out)
{
    return records_from_entries_impl(const_cast<std::vector<WalkEntry>&>(ents), out, false);
}

int walk_tree(const char* build_dir, std::vector<Record>& out, int* err_no)
{
    std::vector<WalkEntry> ents;
    int e = 0;
    const int wrc = walk_entries(build_dir, ents, &e, nullptr);
    if (err_no) *err_no = e;
    const int rc = records_from_entries_impl(ents, out, true); // what was visited before a failure may already hold the first error
    if (rc) return rc;
    return wrc ? SNAPHASH_EIO : SNAPHASH_OK;
}

// ---- yaml.v2 emitter for hashesYaml ---------------------------------------------

// The only scalar style the reference pins is the plain one (hashes_test.go:89-103): names inside this
// conservative set are written as they are.  Every other name goes through yamlscalar.cpp, the restatement of
// yaml.v2's style selection and scalar writers (unpinned by the reference's tests; see that file).
bool plain_safe_name(const std::string& s)
{
    if (s.empty() || s.size() > 72) return false; // 8 + 72 = column 80: no folding can be due either
    bool alpha = false;
    for (unsigned char c : s) {
        const bool ok = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' ||
                        c == '.' || c == '/' || c == '-' || c == '+';
        if (!ok) return false;
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) alpha = true;
    }
    const unsigned char c0 = s[0];
    if (c0 == '-' || c0 == '.' || c0 == '+' || (c0 >= '0' && c0 <= '9')) return false;
    if (!alpha) return false;
    static const char* const resolved[] = {"y", "Y", "yes", "Yes", "YES", "on", "On", "ON", "n", "N", "no", "No", "NO",
                                           "off", "Off", "OFF", "true", "True", "TRUE", "false", "False", "FALSE",
                                           "null", "Null", "NULL", nullptr};
    if (s.size() <= 5) // (the longest of them is "false")
        for (int i = 0; resolved[i]; ++i)
            if (s == resolved[i]) return false;
    return true;
}

void hex_lower(const uint8_t d[64], char out[128])
{
    static const char x[] = "0123456789abcdef"; // encoding/hex: lowercase (helpers.go:200)
    for (int i = 0; i < 64; ++i) { out[2 * i] = x[d[i] >> 4]; out[2 * i + 1] = x[d[i] & 15]; }
}

// the records [lo, hi) of recs; file_digests already points at the digest of the first regular record among them.
// hex_at (skeleton): no digests yet -- 128 zeros stand in, and where each run starts (relative to out) is noted.
static int emit_records(const std::vector<Record>& recs, size_t lo, size_t hi, const uint8_t* file_digests, std::string& out,
                        std::vector<size_t>* hex_at = nullptr)
{
    char hex[128], num[32];
    size_t fi = 0;
    if (hex_at) memset(hex, '0', sizeof hex);
    for (size_t k = lo; k < hi; ++k) {
        const Record& r = recs[k];
        char mode[11];
        int rc = mode_string(r.st_mode, mode);
        if (rc) return rc;
        out += "- name:";
        if (plain_safe_name(r.name)) { out += ' '; out += r.name; } // the pinned case (hashes_test.go:89-103), short cut
        else if ((rc = yaml_append_name_scalar(r.name, 7, 4, out)) != SNAPHASH_OK) return rc;
        out += '\n';
        if (r.is_regular) { // size (*int64, omitempty on nil only: "size: 0" IS emitted), then sha512
            snprintf(num, sizeof num, "%lld", (long long)r.size);
            out += "  size: "; out += num; out += '\n';
            if (!hex_at) hex_lower(file_digests + 64 * fi++, hex);
            out += "  sha512: ";
            if (hex_at) hex_at->push_back(out.size());
            out.append(hex, 128); out += '\n';
        }
        out += "  mode: "; out.append(mode, 10); out += '\n';
    }
    return SNAPHASH_OK;
}

static int emit_yaml_impl(const std::vector<Record>& recs, const uint8_t* archive_digest, const uint8_t* file_digests,
                          std::string& out, std::vector<size_t>* hex_at, unsigned max_threads = 8)
{
    char hex[128];
    out.clear();
    out.reserve(64 + recs.size() * 220);
    if (hex_at) { hex_at->clear(); memset(hex, '0', sizeof hex); }
    else hex_lower(archive_digest, hex);
    out += "archive-sha512: "; // hashes.go:106
    if (hex_at) hex_at->push_back(out.size());
    out.append(hex, 128);
    out += '\n';
    if (recs.empty()) { // yaml.v2 renders an empty slice in flow style (unpinned by the reference)
        out += "files: []\n";
        return SNAPHASH_OK;
    }
    out += "files:\n"; // untagged field Files -> lower-cased key (hashes.go:109)
    // Records are independent: a large tree is written in ranges on a few threads and the pieces are joined in order
    // (10 100 records: 2.1 ms on one thread -- on rank 0 of an 8-GPU pass that is serial time behind the gather).
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(std::max(1u, max_threads), usable_cpus()), recs.size() / 1024));
    if (T <= 1) return emit_records(recs, 0, recs.size(), file_digests, out, hex_at);
    std::vector<size_t> first_digest(T + 1, 0); // regular records in front of each range
    {
        size_t fi = 0;
        for (unsigned t = 0; t < T; ++t) {
            first_digest[t] = fi;
            for (size_t k = recs.size() * t / T; k < recs.size() * (t + 1) / T; ++k) fi += recs[k].is_regular ? 1 : 0;
        }
    }
    std::vector<std::string> piece(T);
    std::vector<std::vector<size_t>> piece_at(hex_at ? T : 0);
    std::vector<int> prc(T, 0);
    auto work = [&](unsigned t) {
        const size_t lo = recs.size() * t / T, hi = recs.size() * (t + 1) / T;
        piece[t].reserve((hi - lo) * 220);
        prc[t] = emit_records(recs, lo, hi, file_digests ? file_digests + 64 * first_digest[t] : nullptr, piece[t], hex_at ? &piece_at[t] : nullptr);
    };
    run_on_threads(T, work);
    for (unsigned t = 0; t < T; ++t) { // the first error in record order, as the serial loop would have met it
        if (prc[t]) return prc[t];
        if (hex_at)
            for (size_t at : piece_at[t]) hex_at->push_back(out.size() + at);
        out += piece[t];
    }
    return SNAPHASH_OK;
}

int emit_yaml(const std::vector<Record>& recs, const uint8_t archive_digest[64], const uint8_t* file_digests,
              std::string& out)
{
    return emit_yaml_impl(recs, archive_digest, file_digests, out, nullptr);
}

int emit_yaml_skeleton(const std::vector<Record>& recs, YamlSkeleton& sk, unsigned max_threads)
{
    return emit_yaml_impl(recs, nullptr, nullptr, sk.text, &sk.hex_at, max_threads);
}

void yaml_fill_digests(YamlSkeleton& sk, const uint8_t archive_digest[64], const uint8_t* file_digests)
{
    if (sk.hex_at.empty()) return;
    hex_lower(archive_digest, &sk.text[sk.hex_at[0]]);
    const size_t n = sk.hex_at.size() - 1;
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(8u, usable_cpus()), n / 2048));
    auto work = [&](unsigned t) {
        for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) hex_lower(file_digests + 64 * i, &sk.text[sk.hex_at[i + 1]]);
    };
    run_on_threads(T, work);
}

// ---- the shared walk of a one-process-per-GPU job (hostpass.h) ---------------------------------------------------

namespace {
constexpr uint32_t kListMagic = 0x4c504e53u; // "SNPL"
struct BlobOut {
    std::string b;
    void u32(uint32_t v) { b.append((const char*)&v, 4); }
    void u64(uint64_t v) { b.append((const char*)&v, 8); }
};
struct BlobIn {
    const uint8_t* p;
    size_t n, at = 0;
    bool ok = true;
    uint32_t u32() { uint32_t v = 0; if (at + 4 > n) { ok = false; return 0; } memcpy(&v, p + at, 4); at += 4; return v; }
    uint64_t u64() { uint64_t v = 0; if (at + 8 > n) { ok = false; return 0; } memcpy(&v, p + at, 8); at += 8; return v; }
    const char* bytes(size_t k) { if (k > n - at || at > n) { ok = false; return nullptr; } const char* r = (const char*)p + at; at += k; return r; }
};
// the root's entries, byte-wise sorted; false: the root cannot be listed (Walk then visits nothing below it)
bool list_root(const std::string& root, std::vector<std::string>& names)
{
    names.clear();
    DIR* d = opendir(root.c_str());
    if (!d) return false;
    while (struct dirent* de = readdir(d)) {
        if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
        names.emplace_back(de->d_name);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    return true;
}
uint64_t hash_names(const std::vector<std::string>& names)
{
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (const std::string& s : names)
        for (size_t i = 0; i <= s.size(); ++i) { h = (h ^ (uint8_t)(i < s.size() ? s[i] : 0)) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
    return h;
}
} // namespace


int shard_listing(const char* build_dir, uint32_t rank, uint32_t world, std::string& blob, int* err_no)
{
    if (err_no) *err_no = 0;
    std::string root(build_dir);
    while (root.size() > 1 && root.back() == '/') root.pop_back();
    struct stat rst;
    if (lstat(root.c_str(), &rst) != 0) { if (err_no) *err_no = errno; return SNAPHASH_EIO; }
    std::vector<std::string> top;
    if (S_ISDIR(rst.st_mode)) (void)list_root(root, top); // (a root that cannot be listed has nothing below it: no records, as in the full walk)
    std::vector<uint32_t> mine;
    for (size_t i = 0; i < top.size(); ++i)
        if (i % world == rank) mine.push_back((uint32_t)i);
    std::vector<std::vector<WalkEntry>> sub(mine.size());
    std::vector<int> sub_rc(mine.size(), 0), sub_errno(mine.size(), 0);
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, usable_cpus()), mine.size()));
    std::atomic<size_t> next{0};
    run_on_threads(T, [&](unsigned) {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= mine.size()) return;
            const std::string& name = top[mine[k]];
            if (name.compare(0, 6, "DEBIAN") == 0) continue; // never a record, nor anything below it
            sub_rc[k] = walk_entries((root + "/" + name).c_str(), sub[k], &sub_errno[k], nullptr);
        }
    });
    for (size_t k = 0; k < mine.size(); ++k)
        if (sub_rc[k]) { if (err_no) *err_no = sub_errno[k]; return SNAPHASH_EIO; } // the serial walk stops at its first Lstat that fails: so does the plan
    BlobOut o;
    o.u32(kListMagic); o.u32(1); o.u32(world); o.u32(rank); o.u32((uint32_t)top.size()); o.u32((uint32_t)mine.size());
    o.u64(hash_names(top));
    const size_t cut = root.size() + 1;
    for (size_t k = 0; k < mine.size(); ++k) {
        o.u32(mine[k]);
        o.u32((uint32_t)sub[k].size());
        for (const WalkEntry& e : sub[k]) {
            o.u32((uint32_t)e.st.st_mode);
            o.u32((uint32_t)(e.path.size() - cut));
            o.u64(S_ISREG(e.st.st_mode) ? (uint64_t)e.st.st_size : 0);
            o.b.append(e.path, cut, std::string::npos);
        }
    }
    blob.swap(o.b);
    return SNAPHASH_OK;
}

int records_from_listings(const char* build_dir, uint32_t world, const void* const* blobs, const size_t* blob_lens, std::vector<Record>& recs)
{
    std::string root(build_dir);
    while (root.size() > 1 && root.back() == '/') root.pop_back();
    // pass 1: every blob's header, and where each root entry's records lie
    struct Piece { uint32_t r = 0; size_t at = 0; uint32_t n = 0; bool have = false; };
    std::vector<Piece> piece;
    uint32_t n_top = 0;
    uint64_t names_hash = 0;
    for (uint32_t r = 0; r < world; ++r) {
        if (!blobs[r] && blob_lens[r]) return SNAPHASH_EINVAL;
        BlobIn in{(const uint8_t*)blobs[r], blob_lens[r]};
        const uint32_t magic = in.u32(), version = in.u32(), w = in.u32(), rk = in.u32(), nt = in.u32(), nm = in.u32();
        const uint64_t nh = in.u64();
        if (!in.ok || magic != kListMagic || version != 1 || w != world || rk != r) return SNAPHASH_EPARSE;
        if (r == 0) {
            if (nt > (1u << 26)) return SNAPHASH_EPARSE; // (a directory of 64 M entries is not a listing anybody sent)
            n_top = nt; names_hash = nh; piece.assign(n_top, Piece());
        } else if (nt != n_top || nh != names_hash) return SNAPHASH_EMISMATCH; // the ranks listed different roots: the tree changed under them
        for (uint32_t k = 0; k < nm; ++k) {
            const uint32_t i = in.u32(), ne = in.u32();
            if (!in.ok || i >= n_top || i % world != r || piece[i].have) return SNAPHASH_EPARSE;
            piece[i].r = r; piece[i].at = in.at; piece[i].n = ne; piece[i].have = true;
            for (uint32_t e = 0; e < ne; ++e) { // over the entries (they are read in order below)
                (void)in.u32();
                const uint32_t len = in.u32();
                (void)in.u64();
                (void)in.bytes(len);
                if (!in.ok) return SNAPHASH_EPARSE;
            }
        }
        if (in.at != in.n) return SNAPHASH_EPARSE;
    }
    std::vector<size_t> first(n_top + 1, 0); // the record a root entry's listing begins at
    for (uint32_t i = 0; i < n_top; ++i) { if (!piece[i].have) return SNAPHASH_EPARSE; first[i + 1] = first[i] + piece[i].n; }
    const size_t total = first[n_top];
    recs.clear();
    recs.resize(total);
    // pass 2: the records, root entry after root entry in the root's sorted order = filepath.Walk's order; the entries of
    // the root are independent, so a few threads take ranges of them (10 100 records were 1.8 ms of string building on one
    // thread: as much as the walk this replaces)
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::min(8u, usable_cpus()), total / 1024, (size_t)n_top}));
    std::vector<int> trc(T, SNAPHASH_OK);
    std::vector<size_t> tbad(T, (size_t)-1); // the first record (in Walk's order) a range failed at
    run_on_threads(T, [&](unsigned t) {
        for (uint32_t i = (uint32_t)((uint64_t)n_top * t / T); i < (uint32_t)((uint64_t)n_top * (t + 1) / T); ++i) {
            BlobIn in{(const uint8_t*)blobs[piece[i].r], blob_lens[piece[i].r]};
            in.at = piece[i].at;
            for (uint32_t e = 0; e < piece[i].n; ++e) {
                const uint32_t mode = in.u32(), len = in.u32();
                const uint64_t size = in.u64();
                const char* name = in.bytes(len);
                int rc = SNAPHASH_OK;
                char m[11];
                if (!in.ok || len == 0 || memchr(name, 0, len)) rc = SNAPHASH_EPARSE;
                else if (mode_string(mode, m) != SNAPHASH_OK) rc = SNAPHASH_EMODE; // the first in Walk's order, as the serial loop would meet it
                if (rc) { trc[t] = rc; tbad[t] = first[i] + e; return; }
                Record& r = recs[first[i] + e];
                r.name.assign(name, len);
                r.path.reserve(root.size() + 1 + len);
                r.path = root;
                r.path += '/';
                r.path.append(name, len);
                r.st_mode = mode;
                r.is_regular = S_ISREG(mode);
                r.size = r.is_regular ? (int64_t)size : 0;
            }
        }
    });
    size_t bad = (size_t)-1;
    int rc = SNAPHASH_OK;
    for (unsigned t = 0; t < T; ++t)
        if (trc[t] && tbad[t] < bad) { bad = tbad[t]; rc = trc[t]; }
    if (rc) recs.clear();
    return rc;
}

// ---- tolerant parser for yaml.v2's rendering of hashesYaml ------------------------

namespace {

int hexval(int c)
{
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

// Scalar after "key: ".  Plain, 'single' or "double" quoted; no multi-line forms.
bool parse_scalar(const std::string& v, std::string& out)
{
    out.clear();
    if (v.empty()) return true;
    if (v[0] == '\'') {
        size_t i = 1;
        for (; i < v.size(); ++i) {
            if (v[i] == '\'') {
                if (i + 1 < v.size() && v[i + 1] == '\'') { out += '\''; ++i; continue; }
                break;
            }
            out += v[i];
        }
        return i < v.size();
    }
    if (v[0] == '"') {
        size_t i = 1;
        for (; i < v.size() && v[i] != '"'; ++i) {
            if (v[i] != '\\') { out += v[i]; continue; }
            if (++i >= v.size()) return false;
            switch (v[i]) {
            case 'n': out += '\n'; break;
            case 't': out += '\t'; break;
            case 'r': out += '\r'; break;
            case '0': out += '\0'; break;
            case 'a': out += '\a'; break;
            case 'b': out += '\b'; break;
            case 'e': out += '\x1b'; break;
            case 'f': out += '\f'; break;
            case 'v': out += '\v'; break;
            case ' ': out += ' '; break;
            case '"': out += '"'; break;
            case '/': out += '/'; break;
            case '\\': out += '\\'; break;
            case 'N': out += "\xC2\x85"; break;     // NEL
            case '_': out += "\xC2\xA0"; break;     // NBSP
            case 'L': out += "\xE2\x80\xA8"; break; // LS
            case 'P': out += "\xE2\x80\xA9"; break; // PS
            case 'x': case 'u': case 'U': {
                const int nd = v[i] == 'x' ? 2 : (v[i] == 'u' ? 4 : 8);
                if (i + nd >= v.size()) return false;
                uint32_t cp = 0;
                for (int k = 0; k < nd; ++k) {
                    const int h = hexval(v[i + 1 + k]);
                    if (h < 0) return false;
                    cp = cp * 16 + h;
                }
                i += nd;
                if (cp < 0x80) out += (char)cp; // UTF-8 encode
                else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 63)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                break;
            }
            default: return false;
            }
        }
        return i < v.size();
    }
    // plain: strip a trailing " #comment" and trailing spaces
    std::string p = v;
    size_t h = p.find(" #");
    if (h != std::string::npos) p.resize(h);
    while (!p.empty() && (p.back() == ' ' || p.back() == '\t' || p.back() == '\r')) p.pop_back();
    out = p;
    return true;
}

bool split_key(const std::string& line, size_t from, std::string& key, std::string& val)
{
    size_t c = line.find(':', from);
    if (c == std::string::npos) return false;
    key = line.substr(from, c - from);
    size_t v = c + 1;
    while (v < line.size() && line[v] == ' ') ++v;
    val = line.substr(v);
    while (!val.empty() && (val.back() == '\r' || val.back() == ' ')) val.pop_back();
    return true;
}

bool unhex64(const std::string& s, uint8_t out[64])
{
    if (s.size() != 128) return false;
    for (int i = 0; i < 64; ++i) {
        const int a = hexval(s[2 * i]), b = hexval(s[2 * i + 1]);
        if (a < 0 || b < 0) return false;
        out[i] = (uint8_t)(a * 16 + b);
    }
    return true;
}

} // namespace

namespace {

// lines [lo, hi) of a hashes.yaml; in_files_at_start: the range begins inside the `files:` list (at an item's first
// line).  *top_level (may be NULL) is set when a top-level key is met -- a range parsed on its own must not meet one.
int parse_lines(const std::vector<std::string_view>& lines, size_t lo, size_t hi, bool in_files_at_start, ParsedHashes& out, bool* top_level)
{
    bool in_files = in_files_at_start;
    int skip_indent = -1; // inside a nested block we ignore (xattr)
    ParsedRecord* cur = nullptr;
    auto indent_of = [](std::string_view l) { size_t k = 0; while (k < l.size() && l[k] == ' ') ++k; return k; };
    std::string line;
    std::string key, val, sv; // (outside the loop: their capacity is kept from line to line)
    for (size_t li = lo; li < hi; ++li) {
        line.assign(lines[li].data(), lines[li].size());
        while (!line.empty() && (line.back() == '\r')) line.pop_back();
        size_t ind = 0;
        while (ind < line.size() && line[ind] == ' ') ++ind;
        // A scalar the emitter folded (yaml.v2 breaks lines past column 80 at a space, yamlscalar.cpp): the lines that
        // follow a "key: value" of a list item and are indented deeper than the item's keys continue the value; the
        // break and the indentation stand for one space.
        if (in_files && (skip_indent < 0 || (int)ind <= skip_indent) && ind < line.size() && line[ind] != '#') { // (a line that ends a skipped block is worked on)
            const size_t key_col = line[ind] == '-' ? ind + 2 : ind;
            const size_t colon = line.find(": ", ind);
            if (colon != std::string::npos && colon + 2 < line.size()) {
                while (li + 1 < hi) {
                    std::string_view nx = lines[li + 1];
                    while (!nx.empty() && (nx.back() == '\r' || nx.back() == ' ')) nx.remove_suffix(1);
                    const size_t ni = indent_of(nx);
                    if (nx.empty() || ni <= key_col) break;
                    while (!line.empty() && line.back() == ' ') line.pop_back();
                    line += ' ';
                    line.append(nx.data() + ni, nx.size() - ni);
                    ++li;
                }
            }
        }
        if (ind == line.size() || line[ind] == '#') continue;
        if (line == "---" || line == "...") continue;
        if (skip_indent >= 0) {
            if ((int)ind > skip_indent) continue;
            skip_indent = -1;
        }
        if (ind == 0 && line[0] != '-') { // top-level key
            if (top_level) *top_level = true;
            if (line == "{}") continue;   // common_test.go:77-80: an empty document is acceptable
            if (!split_key(line, 0, key, val)) return SNAPHASH_EPARSE;
            in_files = false;
            cur = nullptr;
            if (key == "archive-sha512") {
                if (!parse_scalar(val, sv)) return SNAPHASH_EPARSE;
                out.archive_hex = sv;
                out.has_archive = true;
            } else if (key == "files") {
                if (val == "[]" || val == "") in_files = (val == "");
                else return SNAPHASH_EPARSE;
            } else if (val.empty()) {
                skip_indent = 0; // unknown nested block
            }
            continue;
        }
        if (!in_files) return SNAPHASH_EPARSE;
        size_t kstart = ind;
        if (line[ind] == '-') { // new sequence item: "- key: value"
            out.files.emplace_back();
            cur = &out.files.back();
            kstart = ind + 1;
            while (kstart < line.size() && line[kstart] == ' ') ++kstart;
            if (kstart >= line.size()) continue;
        }
        if (!cur) return SNAPHASH_EPARSE;
        if (!split_key(line, kstart, key, val)) return SNAPHASH_EPARSE;
        if (key == "xattr") { // map[string]string, unused upstream (hashes.go:98-100)
            if (val.empty()) skip_indent = (int)kstart;
            continue;
        }
        if (!parse_scalar(val, sv)) return SNAPHASH_EPARSE;
        if (key == "name") { cur->name = sv; cur->has_name = true; }
        else if (key == "size") {
            if (sv.empty()) return SNAPHASH_EPARSE;
            char* end = nullptr;
            errno = 0;
            long long v = strtoll(sv.c_str(), &end, 10);
            if (errno || !end || *end) return SNAPHASH_EPARSE;
            cur->size = v; cur->has_size = true;
        } else if (key == "sha512") { cur->sha512_hex = sv; }
        else if (key == "mode") {
            if (mode_parse(sv.c_str(), &cur->st_mode) != SNAPHASH_OK) return SNAPHASH_EPARSE;
            cur->has_mode = true;
        }
        // unknown keys are ignored, as yaml.v2 does for non-strict Unmarshal
    }
    return SNAPHASH_OK;
}

} // namespace

int parse_yaml(const char* text, size_t len, ParsedHashes& out)
{
    out = ParsedHashes();
    std::vector<std::string_view> lines; // views into the caller's text: a line is copied only when it is worked on
    {
        size_t i = 0;
        while (i < len) {
            const char* nl = (const char*)memchr(text + i, '\n', len - i);
            const size_t j = nl ? (size_t)(nl - text) : len;
            lines.emplace_back(text + i, j - i);
            i = j + 1;
        }
    }
    // A big document (100 000 files: 20 MB, 400 000 lines) is parsed in ranges that begin at an item of the `files:` list,
    // each by a thread of its own: an item's lines depend on nothing before them.  The head up to `files:` first; a
    // range that meets anything but list items (a further top-level key) sends the whole text down the serial way.
    const unsigned T = (unsigned)std::min<size_t>(std::min(8u, usable_cpus()), lines.size() / 20000);
    size_t f_line = lines.size();
    if (T > 1)
        for (size_t i = 0; i < lines.size(); ++i) {
            std::string_view l = lines[i];
            while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.remove_suffix(1);
            if (l == "files:") { f_line = i; break; }
            if (!l.empty() && l[0] == '-') break; // an item before any `files:`: not the shape this shortcut is for
        }
    bool done = false;
    if (T > 1 && f_line + 1 < lines.size()) {
        ParsedHashes head;
        int rc = parse_lines(lines, 0, f_line + 1, false, head, nullptr);
        if (rc == SNAPHASH_OK) {
            std::vector<size_t> cut(T + 1, lines.size());
            cut[0] = f_line + 1;
            for (unsigned t = 1; t < T; ++t) {
                size_t at = f_line + 1 + (lines.size() - f_line - 1) * t / T;
                while (at < lines.size() && !(lines[at].size() >= 2 && lines[at][0] == '-' && lines[at][1] == ' ')) ++at;
                cut[t] = std::max(at, cut[t - 1]);
            }
            std::vector<ParsedHashes> part(T);
            std::vector<int> prc(T, SNAPHASH_OK);
            std::vector<char> top(T, 0);
            // (run_on_threads raises what a worker threw -- an allocation that failed: a lost worker must not read as a clean,
            // short file list)
            run_on_threads(T, [&](unsigned t) {
                bool tl = false;
                prc[t] = cut[t] < cut[t + 1] ? parse_lines(lines, cut[t], cut[t + 1], true, part[t], &tl) : SNAPHASH_OK;
                top[t] = tl;
            });
            bool clean = true;
            for (unsigned t = 0; t < T; ++t) clean = clean && prc[t] == SNAPHASH_OK && !top[t];
            if (clean) {
                out = std::move(head);
                size_t total = out.files.size();
                for (unsigned t = 0; t < T; ++t) total += part[t].files.size();
                out.files.reserve(total);
                for (unsigned t = 0; t < T; ++t)
                    for (ParsedRecord& r : part[t].files) out.files.push_back(std::move(r));
                done = true;
            }
        }
    }
    if (!done) {
        out = ParsedHashes();
        const int rc = parse_lines(lines, 0, lines.size(), false, out, nullptr);
        if (rc != SNAPHASH_OK) return rc;
    }
    for (const ParsedRecord& r : out.files)
        if (!r.has_name || !r.has_mode) return SNAPHASH_EPARSE;
    if (out.has_archive && !out.archive_hex.empty()) {
        // stays a string: the reader side only carries it (snapp.go:466-478)
    }
    return SNAPHASH_OK;
}

bool digest_matches_hex(const uint8_t d[64], const std::string& hex)
{
    uint8_t e[64];
    if (!unhex64(hex, e)) return false;
    return memcmp(d, e, 64) == 0;
}

// ---- shard assignment --------------------------------------------------------------

// LPT (longest processing time first): sort by SHA-512 block count descending,
// give each file to the currently lightest shard.  Ties broken by index so the
// result is deterministic on every rank.
int lpt_assign(const uint64_t* lens, size_t n, int nshards, int32_t* shard_of)
{
    if (nshards <= 0 || (!lens && n) || (!shard_of && n)) return SNAPHASH_EINVAL;
    // longest first, ties in list order (what a stable sort by block count gives): one sort of packed keys when the
    // counts fit 32 bits (a stream under 512 GiB), the comparator form otherwise
    auto blocks = [&](size_t i) { return (lens[i] + 17 + 127) / 128; };
    std::vector<uint32_t> order(n);
    bool small = n <= 0xffffffffull;
    for (size_t i = 0; i < n && small; ++i) small = blocks(i) <= 0xffffffffull;
    if (small) {
        std::vector<uint64_t> key(n);
        for (size_t i = 0; i < n; ++i) key[i] = (blocks(i) << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
        std::sort(key.begin(), key.end(), std::greater<uint64_t>());
        for (size_t i = 0; i < n; ++i) order[i] = 0xffffffffu - (uint32_t)key[i];
    } else {
        for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return blocks(a) > blocks(b); });
    }
    typedef std::pair<uint64_t, int> Load; // (blocks so far, shard)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int s = 0; s < nshards; ++s) heap.push(Load(0, s));
    for (uint32_t i : order) {
        Load l = heap.top();
        heap.pop();
        shard_of[i] = l.second;
        l.first += blocks(i);
        heap.push(l);
    }
    return SNAPHASH_OK;
}

} // namespace snaphash
