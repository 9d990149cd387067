// walk.cpp -- see walk.h.
#include "walk.h"

#include "hostfill.h"

#include <dirent.h>
#include <errno.h>
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
};

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
    closedir(d);
    std::sort(dl.kids.begin(), dl.kids.end(), [](const Child& a, const Child& b) { return a.name < b.name; });
    for (Child& c : dl.kids)
        if (c.type == DT_UNKNOWN) { // this filesystem does not say: look now
            if (lstat((dl.path + "/" + c.name).c_str(), &c.st) != 0) c.lstat_errno = errno;
            else { c.have_st = true; c.type = S_ISDIR(c.st.st_mode) ? DT_DIR : DT_REG; }
        }
}

// Pre-order assembly of the listings: appends dirs[di]'s children (recursively); dirs[di] itself is ents[self].  Returns
// the index of the entry at which the serial walk would have failed (errno in *err_no), or -1.
int64_t assemble(const std::vector<DirList>& dirs, size_t di, std::vector<WalkEntry>& ents, size_t self, int* err_no)
{
    const DirList& dl = dirs[di];
    if (dl.open_failed) {
        // filepath.Walk: `names, err := readDirNames(path); if err != nil { return walkFn(path, info, err) }` -- the
        // callback runs a SECOND time for the directory, and neither of the reference's callbacks looks at the err
        // it is handed (snappy/build.go:228 ignores it, clickdeb/deb.go:285-286 shadows it with its own Lstat): the
        // entry is emitted again and the walk goes on behind it.
        WalkEntry again;
        again.path = dl.path;
        again.st = ents[self].st;
        again.have_st = ents[self].have_st;
        ents.push_back(std::move(again));
        return -1;
    }
    for (const Child& c : dl.kids) {
        WalkEntry e;
        e.path = dl.path + "/" + c.name;
        if (c.lstat_errno) { *err_no = c.lstat_errno; ents.push_back(std::move(e)); return (int64_t)ents.size() - 1; }
        if (c.have_st) { e.st = c.st; e.have_st = true; }
        const size_t me = ents.size();
        ents.push_back(std::move(e));
        if (c.dir >= 0) {
            const int64_t bad = assemble(dirs, (size_t)c.dir, ents, me, err_no);
            if (bad >= 0) return bad;
        }
    }
    return -1;
}

unsigned walk_threads(size_t items, size_t per_thread)
{
    return (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, usable_cpus()), items / per_thread));
}

// Lists the tree under root level by level: the directories of one level are independent, so they are listed on a few
// threads (a 10 000-file tree in 100 directories: ~6 ms of readdir + sort on one thread); then the listings are put
// together in Walk's pre-order.
int64_t walk_names(const std::string& root, std::vector<WalkEntry>& ents, int* err_no)
{
    std::vector<DirList> dirs(1);
    dirs[0].path = root;
    size_t lo = 0;
    while (lo < dirs.size()) {
        const size_t hi = dirs.size();
        const unsigned T = walk_threads(hi - lo, 4);
        std::atomic<size_t> next{lo};
        auto work = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= hi) return;
                list_dir(dirs[i]);
            }
        };
        {
            ThreadJoiner th;
            for (unsigned t = 1; t < T; ++t) th.th.emplace_back(work);
            work();
            th.join_all();
        }
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
    return assemble(dirs, 0, ents, 0, err_no);
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
    int dir_errno = 0;
    int64_t dir_bad = -1; // entry whose children could not be listed, or whose look-ahead Lstat failed
    const bool trace = getenv("SNAPHASH_TRACE_TREE") != nullptr;
    const auto t_list0 = std::chrono::steady_clock::now();
    if (S_ISDIR(ents[0].st.st_mode)) dir_bad = walk_names(root, ents, &dir_errno);
    const auto t_list1 = std::chrono::steady_clock::now();

    const size_t n = ents.size();
    const unsigned T = walk_threads(n, 512);
    std::vector<int64_t> bad(T, -1);
    std::vector<int> bad_errno(T, 0);
    auto work = [&](unsigned t) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; ++i) {
            if (ents[i].have_st) continue;
            if (lstat(ents[i].path.c_str(), &ents[i].st) != 0) { bad[t] = (int64_t)i; bad_errno[t] = errno; return; }
            ents[i].have_st = true;
        }
    };
    {
        ThreadJoiner th; // joined even when a thread cannot be started (the exception then leaves through the C entry point's catch)
        for (unsigned t = 1; t < T; ++t) th.th.emplace_back(work, t);
        work(0);
        th.join_all();
    }
    if (trace) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "snaphash walk: names (list + sort + pre-order) %.2f ms, Lstat of %zu entries on %u threads %.2f ms\n",
                std::chrono::duration<double, std::milli>(t_list1 - t_list0).count(), n, T, std::chrono::duration<double, std::milli>(t2 - t_list1).count());
    }
    int64_t first_bad = -1;
    int first_errno = 0;
    for (unsigned t = 0; t < T; ++t)
        if (bad[t] >= 0 && (first_bad < 0 || bad[t] < first_bad)) { first_bad = bad[t]; first_errno = bad_errno[t]; }
    // a directory that could not be listed fails AFTER its own Lstat and visit, before anything behind it
    bool keep_bad = false;
    if (dir_bad >= 0 && (first_bad < 0 || dir_bad < first_bad)) {
        first_bad = dir_bad;
        first_errno = dir_errno;
        keep_bad = ents[(size_t)dir_bad].have_st;
    }
    if (first_bad < 0) return 0;
    if (err_no) *err_no = first_errno;
    if (err_path) *err_path = ents[(size_t)first_bad].path;
    ents.resize((size_t)first_bad + (keep_bad ? 1 : 0));
    return -1;
}

} // namespace snaphash
