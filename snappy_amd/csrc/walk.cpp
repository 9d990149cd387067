// walk.cpp -- see walk.h.
#include "walk.h"

#include "hostfill.h"

#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

namespace snaphash {

namespace {

// One directory, listed: its children byte-wise sorted (sort.Strings), each with the type the filesystem gave for it
// (or, where it gave none, the Lstat taken on the spot).
struct Child {
    std::string name;
    unsigned char type = DT_UNKNOWN;
    struct stat st;
    bool have_st = false;
    int lstat_errno = 0;
    int64_t dir = -1; // index of this child's own listing when it is a directory
};
struct DirList {
    std::string path;
    bool open_failed = false;
    std::vector<Child> kids;
    DIR* kept = nullptr; // the directory, still open: its entries are Lstat'ed relative to it (below)
    DirList() = default;
    DirList(DirList&& o) noexcept : path(std::move(o.path)), open_failed(o.open_failed), kids(std::move(o.kids)), kept(o.kept) { o.kept = nullptr; }
    DirList& operator=(DirList&& o) noexcept
    {
        if (this != &o) { release(); path = std::move(o.path); open_failed = o.open_failed; kids = std::move(o.kids); kept = o.kept; o.kept = nullptr; }
        return *this;
    }
    DirList(const DirList&) = delete;
    DirList& operator=(const DirList&) = delete;
    ~DirList() { release(); }
    void release() { if (kept) { closedir(kept); kept = nullptr; } }
};

// Directories kept open between their listing and the Lstat of their entries: fstatat(dirfd, name) resolves ONE path
// component where lstat(path) resolves all of them (six on the bench's tree: a quarter of an Lstat).  Only so many at a
// time -- a descriptor each -- and a tree of more directories Lstats the rest by path as before.
constexpr int kKeptDirs = 192;
std::atomic<int> g_kept_dirs{0};

void list_dir(DirList& dl)
{
    DIR* d = opendir(dl.path.c_str());
    if (!d) { dl.open_failed = true; return; }
    while (struct dirent* de = readdir(d)) {
        if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
        Child c;
        c.name = de->d_name;
        c.type = de->d_type;
        dl.kids.push_back(std::move(c));
    }
    if (dl.kids.size() >= 16 && g_kept_dirs.fetch_add(1) < kKeptDirs) dl.kept = d; // (released by the walk when the entries are Lstat'ed, or by ~DirList)
    else { if (dl.kids.size() >= 16) g_kept_dirs.fetch_sub(1); closedir(d); }
    std::sort(dl.kids.begin(), dl.kids.end(), [](const Child& a, const Child& b) { return a.name < b.name; });
    for (Child& c : dl.kids)
        if (c.type == DT_UNKNOWN) { // this filesystem does not say: look now
            if (lstat((dl.path + "/" + c.name).c_str(), &c.st) != 0) c.lstat_errno = errno;
            else { c.have_st = true; c.type = S_ISDIR(c.st.st_mode) ? DT_DIR : DT_REG; }
        }
}

unsigned walk_threads(size_t items, size_t per_thread)
{
    return (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, usable_cpus()), items / per_thread));
}

// Lists the tree under root level by level: the directories of one level are independent, so they are listed on a few
// threads (a 10 000-file tree in 100 directories: ~6 ms of readdir + sort on one thread).
void list_tree(const std::string& root, std::vector<DirList>& dirs)
{
    dirs.clear();
    dirs.emplace_back();
    dirs[0].path = root;
    size_t lo = 0;
    while (lo < dirs.size()) {
        const size_t hi = dirs.size();
        const unsigned T = walk_threads(hi - lo, 4);
        std::atomic<size_t> next{lo};
        run_on_threads(T, [&](unsigned) {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= hi) return;
                list_dir(dirs[i]);
            }
        });
        for (size_t i = lo; i < hi; ++i) // the next level (dirs may reallocate: by index)
            for (size_t k = 0; k < dirs[i].kids.size(); ++k)
                if (dirs[i].kids[k].type == DT_DIR && !dirs[i].kids[k].lstat_errno) {
                    dirs[i].kids[k].dir = (int64_t)dirs.size();
                    DirList sub;
                    sub.path = dirs[i].path + "/" + dirs[i].kids[k].name;
                    dirs.push_back(std::move(sub));
                }
        lo = hi;
    }
}

} // namespace

int walk_entries(const char* root_c, std::vector<WalkEntry>& ents, int* err_no, std::string* err_path)
{
    std::string root(root_c);
    while (root.size() > 1 && root.back() == '/') root.pop_back();
    if (err_no) *err_no = 0;
    ents.clear();
    ents.resize(1);
    ents[0].path = root;
    if (lstat(root.c_str(), &ents[0].st) != 0) {
        if (err_no) *err_no = errno;
        if (err_path) *err_path = root;
        ents.clear();
        return -1;
    }
    ents[0].have_st = true;
    if (!S_ISDIR(ents[0].st.st_mode)) return 0;
    const bool trace = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<DirList> dirs;
    list_tree(root, dirs);
    const auto t1 = std::chrono::steady_clock::now();

    // Where every entry goes in Walk's pre-order: a directory's entries follow it at once, each child followed by its own
    // subtree.  A directory that cannot be listed is followed by ONE entry: itself again --
    // filepath.Walk: `names, err := readDirNames(path); if err != nil { return walkFn(path, info, err) }`: the callback
    // runs a SECOND time for the directory, and neither of the reference's callbacks looks at the err it is handed
    // (snappy/build.go:228 ignores it, clickdeb/deb.go:285-286 shadows it with its own Lstat): the entry is emitted
    // again and the walk goes on behind it.
    const size_t nd = dirs.size();
    std::vector<size_t> count(nd, 0), start(nd, 0); // entries below a directory; index of the first of them
    for (size_t d = nd; d-- > 0;) { // children were listed after their parents: their counts are final here
        if (dirs[d].open_failed) { count[d] = 1; continue; }
        size_t c = 0;
        for (const Child& k : dirs[d].kids) c += 1 + (k.dir >= 0 ? count[(size_t)k.dir] : 0);
        count[d] = c;
    }
    start[0] = 1;
    struct Task { uint32_t dir, k0, k1; size_t first; };
    std::vector<Task> tasks;
    for (size_t d = 0; d < nd; ++d) { // parents before children: start[d] is known when d is reached
        if (dirs[d].open_failed) { tasks.push_back(Task{(uint32_t)d, 0, 0, start[d]}); continue; }
        size_t at = start[d];
        const std::vector<Child>& kids = dirs[d].kids;
        for (size_t k = 0; k < kids.size(); ++k) {
            if (k % 256 == 0) tasks.push_back(Task{(uint32_t)d, (uint32_t)k, (uint32_t)std::min(kids.size(), k + 256), at});
            if (kids[k].dir >= 0) { start[(size_t)kids[k].dir] = at + 1; at += 1 + count[(size_t)kids[k].dir]; }
            else at += 1;
        }
    }
    const size_t n = 1 + count[0];
    ents.resize(n);

    // Every entry's path and Lstat, a few hundred entries a task, on a few threads.
    const unsigned T = walk_threads(n, 512);
    std::vector<int64_t> bad(T, -1);
    std::vector<int> bad_errno(T, 0);
    std::atomic<size_t> next{0};
    auto work = [&](unsigned t) {
        auto fail = [&](size_t idx, int e) { if (bad[t] < 0 || (int64_t)idx < bad[t]) { bad[t] = (int64_t)idx; bad_errno[t] = e; } };
        for (;;) {
            const size_t ti = next.fetch_add(1);
            if (ti >= tasks.size()) return;
            const Task& tk = tasks[ti];
            const DirList& dl = dirs[tk.dir];
            if (dl.open_failed) { // the directory again
                WalkEntry& e = ents[tk.first];
                e.path = dl.path;
                if (lstat(e.path.c_str(), &e.st) != 0) fail(tk.first, errno);
                else e.have_st = true;
                continue;
            }
            size_t at = tk.first;
            for (uint32_t k = tk.k0; k < tk.k1; ++k) {
                const Child& c = dl.kids[k];
                WalkEntry& e = ents[at];
                e.path.reserve(dl.path.size() + 1 + c.name.size());
                e.path = dl.path;
                e.path += '/';
                e.path += c.name;
                if (c.lstat_errno) fail(at, c.lstat_errno); // (the look-ahead of a file system that gives no types)
                else if (c.have_st) { e.st = c.st; e.have_st = true; }
                else if ((dl.kept ? fstatat(dirfd(dl.kept), c.name.c_str(), &e.st, AT_SYMLINK_NOFOLLOW) : lstat(e.path.c_str(), &e.st)) != 0) fail(at, errno);
                else e.have_st = true;
                at += 1 + (c.dir >= 0 ? count[(size_t)c.dir] : 0);
            }
        }
    };
    struct ReleaseDirs { // the kept directories go back whatever way the walk ends
        std::vector<DirList>& dirs;
        ~ReleaseDirs()
        {
            int n = 0;
            for (DirList& d : dirs) if (d.kept) { d.release(); ++n; }
            g_kept_dirs.fetch_sub(n);
        }
    } release_dirs{dirs};
    run_on_threads(T, work); // (every thread is waited for whatever happens; a worker's exception leaves through the C entry point's catch)
    if (trace) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "snaphash walk: listing %zu directories (readdir + sort) %.2f ms, paths + Lstat of %zu entries on %u threads %.2f ms\n", nd,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), n, T, std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    int64_t first_bad = -1;
    int first_errno = 0;
    for (unsigned t = 0; t < T; ++t)
        if (bad[t] >= 0 && (first_bad < 0 || bad[t] < first_bad)) { first_bad = bad[t]; first_errno = bad_errno[t]; }
    if (first_bad < 0) return 0;
    // the serial walk stops at the first entry it cannot Lstat: everything in front of it was visited
    if (err_no) *err_no = first_errno;
    if (err_path) *err_path = ents[(size_t)first_bad].path;
    ents.resize((size_t)first_bad);
    return -1;
}

} // namespace snaphash
