// tarpack.cpp -- see tarpack.h.  Host-only; no hashing, no compression here.
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "tarpack.h"
#include "walk.h"

#include <dirent.h>
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

#include "../../include/snaphash.h"

namespace snaphash {

const uint8_t kGzipHeader[10] = {0x1f, 0x8b, 0x08, 0x00, 0x00, 0x00, 0x00, 0x00, 0x02, 0xff}; // deflate, no name/mtime, XFL=best, OS unknown

// ---- filepath.Walk as tarCreate drives it (deb.go:283-341) ------------------------------------

// name: up to 100 bytes, or prefix (<= 155) + "/" + name (<= 100) split at a slash
static bool ustar_name_fits(const std::string& nm, size_t* cut_out)
{
    if (cut_out) *cut_out = std::string::npos;
    if (nm.size() <= 100) return true;
    for (size_t i = std::min<size_t>(nm.size() - 1, 155); i > 0; --i)
        if (nm[i] == '/' && nm.size() - i - 1 <= 100 && nm.size() - i - 1 > 0) { if (cut_out) *cut_out = i; return true; }
    return false;
}

// "<len> <key>=<value>\n", len counting its own digits
static std::string pax_record(const char* key, const std::string& value)
{
    const size_t body = 1 + strlen(key) + 1 + value.size() + 1; // ' ' key '=' value '\n'
    size_t len = body + 1;
    while (std::to_string(len).size() + body != len) len = std::to_string(len).size() + body;
    return std::to_string(len) + " " + key + "=" + value + "\n";
}

int tar_plan_entries(const std::vector<WalkEntry>& ents, const std::string& exclude_prefix, TarPlan& out, TarKeepFn keep, void* user)
{
    out.members.clear();
    const size_t rootlen = ents.empty() ? 0 : ents[0].path.size();
    for (const WalkEntry& e : ents) {
        const std::string& path = e.path;
        const struct stat& st = e.st;
        const bool supported = S_ISREG(st.st_mode) || S_ISLNK(st.st_mode) || S_ISDIR(st.st_mode); // deb.go:290-292
        const bool excluded = !exclude_prefix.empty() && path.compare(0, exclude_prefix.size(), exclude_prefix) == 0; // deb.go:295-299
        const bool is_root = path.size() == rootlen; // relativePath == "." (deb.go:309-312)
        if (!supported || excluded) continue;
        if (keep && !keep(path.c_str(), user)) continue; // deb.go:295-299: fn is asked before the root is dropped
        if (is_root) continue;
        TarMember m;
        m.path = path;
        m.name = "." + path.substr(rootlen); // deb.go:309
        m.st_mode = st.st_mode;
        m.mtime = (int64_t)st.st_mtime;
        if (S_ISREG(st.st_mode)) { m.typeflag = '0'; m.size = (int64_t)st.st_size; }
        else if (S_ISDIR(st.st_mode)) m.typeflag = '5';
        else {
            m.typeflag = '2';
            char buf[4096];
            const ssize_t n = readlink(path.c_str(), buf, sizeof buf); // os.Readlink, error ignored (deb.go:302)
            if (n > 0) m.linkname.assign(buf, (size_t)n);
        }
        if (!ustar_name_fits(m.name, nullptr)) m.pax += pax_record("path", m.name);
        if (m.linkname.size() > 100) m.pax += pax_record("linkpath", m.linkname);
        out.members.push_back(std::move(m));
    }
    uint64_t off = 0;
    for (TarMember& m : out.members) {
        if (!m.pax.empty()) {
            m.pax_off = off;
            off += 512 + (((uint64_t)m.pax.size() + 511) & ~(uint64_t)511);
        }
        m.hdr_off = off;
        m.data_off = off + 512;
        off += 512 + (((uint64_t)m.size + 511) & ~(uint64_t)511); // content is padded to the 512-byte record
    }
    out.total = off + 1024; // tar.Writer.Close: two zero blocks
    return SNAPHASH_OK;
}

int tar_plan(const char* source_dir, const std::string& exclude_prefix, TarPlan& out, int* err_no, std::string* err_what)
{
    std::vector<WalkEntry> ents;
    int e = 0;
    const int wrc = walk_entries(source_dir, ents, &e, err_what);
    if (err_no) *err_no = e;
    if (wrc) { out.members.clear(); return SNAPHASH_EIO; } // deb.go:286-289: the first Lstat error ends the walk
    return tar_plan_entries(ents, exclude_prefix, out);
}

// ---- ustar header ---------------------------------------------------------------------------

static void octal(uint8_t* f, int width, uint64_t v) // width-1 digits, zero padded, NUL terminated
{
    for (int i = width - 2; i >= 0; --i) { f[i] = (uint8_t)('0' + (v & 7)); v >>= 3; }
    f[width - 1] = 0;
}

int tar_header(const TarMember& m, uint8_t h[512])
{
    memset(h, 0, 512);
    const std::string& nm = m.name;
    size_t cut;
    if (nm.size() <= 100) {
        memcpy(h, nm.data(), nm.size());
    } else if (ustar_name_fits(nm, &cut)) {
        memcpy(h + 345, nm.data(), cut);
        memcpy(h, nm.data() + cut + 1, nm.size() - cut - 1);
    } else if (m.pax.find(" path=") != std::string::npos) {
        memcpy(h, nm.data(), 100); // the PAX record in front carries the whole name
    } else {
        return SNAPHASH_ENAME;
    }
    if (m.linkname.size() > 100 && m.pax.find(" linkpath=") == std::string::npos) return SNAPHASH_ENAME;
    // mode: permission bits + setuid/setgid/sticky + the file type bits, as tar.FileInfoHeader fills it
    uint64_t mode = m.st_mode & 07777;
    if (m.typeflag == '0') mode |= 0100000;
    else if (m.typeflag == '5') mode |= 040000;
    else mode |= 0120000;
    octal(h + 100, 8, mode);
    octal(h + 108, 8, 0); // uid: all files belong to root (deb.go:316-319)
    octal(h + 116, 8, 0); // gid
    if ((uint64_t)m.size >= (1ull << 33)) {
        // 11 octal digits hold 8 GiB - 1: beyond that the binary form every reader of the last decades knows (GNU
        // base-256: a set top bit, then the value big-endian), as archive/tar writes it for such a member
        h[124] = 0x80;
        for (int i = 0; i < 8; ++i) h[124 + 4 + i] = (uint8_t)((uint64_t)m.size >> (56 - 8 * i));
    } else {
        octal(h + 124, 12, m.typeflag == '0' ? (uint64_t)m.size : 0);
    }
    octal(h + 136, 12, m.mtime < 0 ? 0 : (uint64_t)m.mtime);
    memset(h + 148, ' ', 8); // checksum field counts as spaces
    h[156] = (uint8_t)m.typeflag;
    memcpy(h + 157, m.linkname.data(), std::min<size_t>(m.linkname.size(), 100));
    memcpy(h + 257, "ustar", 6); // magic "ustar\0"
    h[263] = '0'; h[264] = '0';  // version "00"
    memcpy(h + 265, "root", 4);  // uname
    memcpy(h + 297, "root", 4);  // gname
    octal(h + 329, 8, 0);        // devmajor
    octal(h + 337, 8, 0);        // devminor
    unsigned sum = 0;
    for (int i = 0; i < 512; ++i) sum += h[i];
    octal(h + 148, 7, sum); // six digits + NUL, then the space that is already there
    h[155] = ' ';
    return SNAPHASH_OK;
}

void tar_pax_header(const TarMember& m, uint8_t h[512])
{
    memset(h, 0, 512);
    // <dir>/PaxHeaders.0/<file>, cut to the name field
    const size_t slash = m.name.rfind('/');
    std::string nm = (slash == std::string::npos ? std::string() : m.name.substr(0, slash + 1)) + "PaxHeaders.0/" +
                     (slash == std::string::npos ? m.name : m.name.substr(slash + 1));
    if (nm.size() > 100) nm.resize(100);
    memcpy(h, nm.data(), nm.size());
    octal(h + 100, 8, 0);
    octal(h + 108, 8, 0);
    octal(h + 116, 8, 0);
    octal(h + 124, 12, (uint64_t)m.pax.size());
    octal(h + 136, 12, m.mtime < 0 ? 0 : (uint64_t)m.mtime); // a reader wants a valid ModTime here
    memset(h + 148, ' ', 8);
    h[156] = 'x';
    memcpy(h + 257, "ustar", 6);
    h[263] = '0'; h[264] = '0';
    octal(h + 329, 8, 0);
    octal(h + 337, 8, 0);
    unsigned sum = 0;
    for (int i = 0; i < 512; ++i) sum += h[i];
    octal(h + 148, 7, sum);
    h[155] = ' ';
}

// ---- CRC-32 -----------------------------------------------------------------------------------

static uint32_t g_crc[8][256];
static bool g_crc_ready = false;
static void crc_init()
{
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        g_crc[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc[t][i] = g_crc[0][g_crc[t - 1][i] & 0xff] ^ (g_crc[t - 1][i] >> 8);
    g_crc_ready = true;
}
namespace { struct CrcInit { CrcInit() { crc_init(); } } g_crc_init; }

static uint32_t crc32_tables(uint32_t crc, const uint8_t* p, size_t n);

#if defined(__x86_64__)
// Carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Intel 2009):
// four 128-bit accumulators are folded 64 bytes forward per step, then into one, then reduced to 32 bits (Barrett).
// The constants are x^n mod P of the reflected gzip polynomial for the fold distances used (4 x 128 + 32 / - 32 bits,
// 128 +- 32, 64, and P with its Barrett inverse).  State in and out is the raw register (no final inversion).
// ~8 bytes a cycle where the tables do 0.4: the CRC of a 256 MiB staging slot no longer keeps the compressor's first
// piece from the consumers.  n >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t c, const uint8_t* p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    const __m128i* v = reinterpret_cast<const __m128i*>(p);
    __m128i x1 = _mm_xor_si128(_mm_loadu_si128(v), _mm_cvtsi32_si128((int)c));
    __m128i x2 = _mm_loadu_si128(v + 1), x3 = _mm_loadu_si128(v + 2), x4 = _mm_loadu_si128(v + 3);
    v += 4;
    n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00),
                      a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), a1), _mm_loadu_si128(v));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), a2), _mm_loadu_si128(v + 1));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), a3), _mm_loadu_si128(v + 2));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), a4), _mm_loadu_si128(v + 3));
        v += 4;
        n -= 64;
    }
    // four accumulators into one, 16 bytes forward each time
    for (const __m128i* nx : {&x2, &x3, &x4}) {
        const __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), *nx);
    }
    while (n >= 16) {
        const __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), _mm_loadu_si128(v));
        ++v;
        n -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_set_epi32(0, ~0, 0, ~0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    // 64 -> 32 bits
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10), mask32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(t, poly, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static const bool g_have_clmul = [] {
    __builtin_cpu_init();
    return __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
}();
#endif

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n)
{
#if defined(__x86_64__)
    if (g_have_clmul && n >= 256) {
        const size_t body = n & ~(size_t)15;
        const uint32_t c = ~crc32_clmul(~crc, p, body);
        return body == n ? c : crc32_tables(c, p + body, n - body);
    }
#endif
    return crc32_tables(crc, p, n);
}

// (test hook: the table form alone)
uint32_t crc32_update_tables(uint32_t crc, const uint8_t* p, size_t n) { return crc32_tables(crc, p, n); }

static uint32_t crc32_tables(uint32_t crc, const uint8_t* p, size_t n)
{
    if (!g_crc_ready) crc_init();
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = g_crc[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        const uint32_t lo = (uint32_t)v ^ c, hi = (uint32_t)(v >> 32);
        c = g_crc[7][lo & 0xff] ^ g_crc[6][(lo >> 8) & 0xff] ^ g_crc[5][(lo >> 16) & 0xff] ^ g_crc[4][lo >> 24] ^
            g_crc[3][hi & 0xff] ^ g_crc[2][(hi >> 8) & 0xff] ^ g_crc[1][(hi >> 16) & 0xff] ^ g_crc[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}

// CRC of a concatenation from the CRCs of its parts: multiply crc1 by x^(8*len2) modulo the polynomial
// (square-and-multiply over GF(2) 32x32 matrices), then xor crc2.
static uint32_t gf2_times(const uint32_t* mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (int i = 0; vec; vec >>= 1, ++i)
        if (vec & 1) sum ^= mat[i];
    return sum;
}
static void gf2_square(uint32_t* sq, const uint32_t* mat)
{
    for (int n = 0; n < 32; ++n) sq[n] = gf2_times(mat, mat[n]);
}
uint32_t crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    if (len2 == 0) return crc1;
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u; // the operator for one zero bit
    uint32_t row = 1;
    for (int n = 1; n < 32; ++n) { odd[n] = row; row <<= 1; }
    gf2_square(even, odd); // two zero bits
    gf2_square(odd, even); // four
    do {
        gf2_square(even, odd); // first pass: one zero byte
        if (len2 & 1) crc1 = gf2_times(even, crc1);
        len2 >>= 1;
        if (!len2) break;
        gf2_square(odd, even);
        if (len2 & 1) crc1 = gf2_times(odd, crc1);
        len2 >>= 1;
    } while (len2);
    return crc1 ^ crc2;
}

} // namespace snaphash
