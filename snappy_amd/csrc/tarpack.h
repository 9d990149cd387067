// tarpack.h -- host side of the data.tar.gz producer: tar layout,
// ustar headers, gzip framing, CRC-32.  Internal; the public entry points are in include/snaphash.h.
//
// Mirrors tarCreate (reference clickdeb/deb.go:261-344): filepath.Walk order, Lstat, only regular
// files / symlinks / directories, the caller's exclude rule, member names "./<relative path>",
// every member owned by root (uid/gid 0, uname/gname "root"), file content via io.Copy, and
// gzip.NewWriterLevel(w, 9) around the tar stream.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace snaphash {

struct TarMember {
    std::string path;     // on disk
    std::string name;     // "./..." inside the archive
    std::string linkname; // symlink target
    uint32_t st_mode = 0;
    int64_t size = 0;     // regular files only
    int64_t mtime = 0;
    char typeflag = '0';  // '0' regular, '2' symlink, '5' directory
    uint64_t hdr_off = 0; // offset of the 512-byte header in the tar stream
    uint64_t data_off = 0; // offset of the content (hdr_off + 512)
    // A name or link target the ustar fields cannot hold travels in a PAX extended header in front of the member
    // (POSIX.1-2001 typeflag 'x', records "<len> path=<name>\n" / "<len> linkpath=<target>\n"), as Go's archive/tar
    // falls back to (its writePAXHeader; pseudo-file name <dir>/PaxHeaders.<pid>/<file>, here with pid 0 so that the
    // archive is a function of the tree).  pax: the records (empty = none); pax_off: offset of the 'x' header record,
    // its data follows at pax_off + 512, padded to the record size, then hdr_off.
    std::string pax;
    uint64_t pax_off = 0;
    uint64_t first_off() const { return pax.empty() ? hdr_off : pax_off; }
};

struct TarPlan {
    std::vector<TarMember> members;
    uint64_t total = 0; // bytes of the whole stream, including the two trailing zero blocks
};

// Walks source_dir as tarCreate does and lays the stream out.  exclude_prefix: full-path string
// prefix to skip (Build passes filepath.Join(sourceDir, "DEBIAN"), deb.go:361-363); empty = none.
// Returns 0 or a SNAPHASH_E* code; *err_no carries errno for EIO.
int tar_plan(const char* source_dir, const std::string& exclude_prefix, TarPlan& out, int* err_no, std::string* err_what);
// the same from a walk already made (walk.h): the fused pass walks once for the archive and for hashes.yaml
struct WalkEntry;
// keep (may be NULL): tarCreate's exclude function (deb.go:261, 295-299), called with the full path of every supported
// entry in walk order on the calling thread; zero leaves the entry out
typedef int (*TarKeepFn)(const char* path, void* user);
int tar_plan_entries(const std::vector<WalkEntry>& ents, const std::string& exclude_prefix, TarPlan& out, TarKeepFn keep = nullptr,
                     void* user = nullptr);

// The 512-byte ustar header of a member (POSIX.1-1988 ustar as Go's archive/tar writes it: octal
// fields of width-1 digits + NUL, checksum as six digits + NUL + space, magic "ustar\0" "00").
// A field that does not fit is truncated when m.pax carries it (the reader takes the PAX record), else SNAPHASH_ENAME.
int tar_header(const TarMember& m, uint8_t out[512]);
// The header record of m's PAX extended header (typeflag 'x', size = m.pax.size()).
void tar_pax_header(const TarMember& m, uint8_t out[512]);

// CRC-32 (IEEE 802.3, the gzip trailer's), slice-by-8, and the combination of the CRCs of two
// adjacent ranges (the second of len2 bytes).
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n);
uint32_t crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

// gzip member framing as compress/gzip writes it at level 9 with a zero Header: 10 bytes.
extern const uint8_t kGzipHeader[10];

} // namespace snaphash
