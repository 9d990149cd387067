// walk.cpp -- see walk.h.
#include "walk.h"

#include "hostfill.h"

#include <dirent.h>
#include <errno.h>
#include <string.h>

#include <algorithm>
#include <thread>

namespace snaphash {

namespace {

// appends path's children (recursively); `path` itself is ents[self].  Returns the index of the entry at which the
// serial walk would have failed (errno in *err_no), or -1.
int64_t walk_names(const std::string& path, std::vector<WalkEntry>& ents, size_t self, int* err_no)
{
    DIR* d = opendir(path.c_str());
    if (!d) {
        // filepath.Walk: `names, err := readDirNames(path); if err != nil { return walkFn(path, info, err) }` -- the
        // callback runs a SECOND time for the directory, and neither of the reference's callbacks looks at the err
        // it is handed (snappy/build.go:228 ignores it, clickdeb/deb.go:285-286 shadows it with its own Lstat): the
        // entry is emitted again and the walk goes on behind it.
        WalkEntry again;
        again.path = path;
        again.st = ents[self].st;
        again.have_st = ents[self].have_st;
        ents.push_back(std::move(again));
        (void)err_no;
        return -1;
    }
    std::vector<std::pair<std::string, unsigned char>> names;
    while (struct dirent* de = readdir(d)) {
        if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
        names.emplace_back(de->d_name, de->d_type);
    }
    closedir(d);
    std::sort(names.begin(), names.end(), [](const auto& a, const auto& b) { return a.first < b.first; }); // sort.Strings: byte-wise
    for (const auto& nt : names) {
        WalkEntry e;
        e.path = path + "/" + nt.first;
        bool is_dir = nt.second == DT_DIR;
        if (nt.second == DT_UNKNOWN) { // this filesystem does not say: look now
            if (lstat(e.path.c_str(), &e.st) != 0) { *err_no = errno; ents.push_back(std::move(e)); return (int64_t)ents.size() - 1; }
            e.have_st = true;
            is_dir = S_ISDIR(e.st.st_mode);
        }
        const size_t me = ents.size();
        ents.push_back(std::move(e));
        if (is_dir) {
            const std::string sub = ents[me].path; // ents may reallocate below
            const int64_t bad = walk_names(sub, ents, me, err_no);
            if (bad >= 0) return bad;
        }
    }
    return -1;
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
    if (S_ISDIR(ents[0].st.st_mode)) dir_bad = walk_names(root, ents, 0, &dir_errno);

    const size_t n = ents.size();
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(12u, std::max(1u, std::thread::hardware_concurrency())), n / 2048));
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
