// walk.h -- Go's filepath.Walk as the reference drives it (snappy/build.go:228, clickdeb/deb.go:285), shared by the
// hash pass and the tar planner.  Internal.
#pragma once
#include <sys/stat.h>

#include <string>
#include <vector>

namespace snaphash {

struct WalkEntry {
    std::string path; // root-joined
    struct stat st;   // Lstat
    bool have_st = false;
};

// Every entry under root (root itself first), in Walk's order: per directory the names byte-wise sorted, pre-order;
// Lstat semantics (a symlink to a directory is an entry, not descended).  Two phases: the names are listed serially,
// descending by the directory entry's type where the filesystem gives one (one opendir per directory instead of one
// lstat per entry), then every entry is Lstat'ed on a few threads.
// A directory that cannot be listed is NOT an error, as in the reference: Walk hands the ReadDir error to the callback
// in a second call for that directory, both callbacks ignore it (snappy/build.go:228, clickdeb/deb.go:285-286), so the
// directory appears twice and the walk continues.
// Returns 0, or -1 when an Lstat failed: then *err_no / *err_path say what, and `ents` holds exactly the entries the
// serial loop would have visited before the failure.
int walk_entries(const char* root, std::vector<WalkEntry>& ents, int* err_no, std::string* err_path);

} // namespace snaphash
