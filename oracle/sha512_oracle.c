/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C CPU restatement of the reference's hashes.yaml pass, used only as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under snappy_amd/ may include, link or call this file.
 *
 * What it restates (paths relative to /root/reference):
 *   helpers/helpers.go:187-201   Sha512sum: open, sha512.New(), io.Copy
 *                                (32 KiB reads), lowercase hex
 *   snappy/build.go:216-270      writeHashes: archive digest, filepath.Walk,
 *                                skip "/DEBIAN" prefix, skip root, regular
 *                                files get size+sha512, yaml.Marshal
 *   snappy/hashes.go:33-57       yamlFileMode.MarshalYAML ("frw-r--r--")
 *   snappy/hashes.go:93-110      fileHash / hashesYaml field order
 *   helpers/cmp.go:28-114        FilesAreEqual / streamsEqual (16 KiB ReadAtLeast
 *                                loop) / DirUpdated -- SURVEY row f4, pinned by the
 *                                cases of helpers/cmp_test.go:30-140
 *
 * The arithmetic itself lives outside /root/reference: Go's standard library
 * crypto/sha512 (Go version unpinned, debian/control:11 says golang-go).  Its
 * published algorithm is FIPS 180-4 sec. 5.1.2 / 6.4, restated here.  YAML bytes
 * come from gopkg.in/yaml.v2 @ 49c95bdc (dependencies.tsv:7): the layout is
 * pinned by snappy/hashes_test.go:89-103 for plain names; quoted and folded
 * names follow the library's published algorithm (name_scalar below) and are
 * PARITY UNPINNED -- no reference fixture holds one.
 *
 * Parity pinning: checked in tests/test_oracle.py against the reference's own
 * known answers (helpers/helpers_test.go:175, snappy/hashes_test.go:30-33 and
 * :89-103, snappy/systemimage_test.go:104/117) and against hashlib.sha512 on
 * FIPS boundary lengths.  The reference itself (Go) cannot be built here.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <locale.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>
#include <unistd.h>

/* ---- FIPS 180-4 sec. 4.2.3 constants ---------------------------------- */
static const uint64_t K512[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL,
    0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL,
    0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
    0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL,
    0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
    0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL,
    0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL,
    0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
    0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL,
    0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
    0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL,
    0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL,
    0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL,
    0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL,
    0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
    0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL,
    0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL,
};

typedef struct {
    uint64_t h[8];
    uint64_t nbytes; /* files here are < 2^61 bytes, one word suffices */
    uint8_t buf[128];
    size_t fill;
} oracle_sha512_ctx;

#define ROTR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))

static void compress(uint64_t h[8], const uint8_t *p)
{
    uint64_t w[80];
    for (int t = 0; t < 16; t++) {
        w[t] = 0;
        for (int b = 0; b < 8; b++)
            w[t] = (w[t] << 8) | p[8 * t + b];
    }
    for (int t = 16; t < 80; t++) {
        uint64_t s0 = ROTR(w[t - 15], 1) ^ ROTR(w[t - 15], 8) ^ (w[t - 15] >> 7);
        uint64_t s1 = ROTR(w[t - 2], 19) ^ ROTR(w[t - 2], 61) ^ (w[t - 2] >> 6);
        w[t] = s1 + w[t - 7] + s0 + w[t - 16];
    }
    uint64_t a = h[0], b = h[1], c = h[2], d = h[3];
    uint64_t e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int t = 0; t < 80; t++) {
        uint64_t S1 = ROTR(e, 14) ^ ROTR(e, 18) ^ ROTR(e, 41);
        uint64_t ch = (e & f) ^ (~e & g);
        uint64_t t1 = hh + S1 + ch + K512[t] + w[t];
        uint64_t S0 = ROTR(a, 28) ^ ROTR(a, 34) ^ ROTR(a, 39);
        uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint64_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1;
        d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void oracle_sha512_init(oracle_sha512_ctx *c)
{
    static const uint64_t iv[8] = {
        0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
        0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
        0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    memcpy(c->h, iv, sizeof iv);
    c->nbytes = 0;
    c->fill = 0;
}

void oracle_sha512_update(oracle_sha512_ctx *c, const uint8_t *p, size_t n)
{
    c->nbytes += n;
    if (c->fill) {
        size_t take = 128 - c->fill;
        if (take > n) take = n;
        memcpy(c->buf + c->fill, p, take);
        c->fill += take; p += take; n -= take;
        if (c->fill == 128) { compress(c->h, c->buf); c->fill = 0; }
    }
    while (n >= 128) { compress(c->h, p); p += 128; n -= 128; }
    if (n) { memcpy(c->buf, p, n); c->fill = n; }
}

void oracle_sha512_final(oracle_sha512_ctx *c, uint8_t out[64])
{
    uint64_t bits_lo = c->nbytes << 3, bits_hi = c->nbytes >> 61;
    uint8_t pad[256];
    size_t r = c->fill, padlen = (r < 112) ? 128 - r : 256 - r;
    memset(pad, 0, sizeof pad);
    pad[0] = 0x80;
    for (int i = 0; i < 8; i++) {
        pad[padlen - 16 + i] = (uint8_t)(bits_hi >> (56 - 8 * i));
        pad[padlen - 8 + i] = (uint8_t)(bits_lo >> (56 - 8 * i));
    }
    uint64_t keep = c->nbytes;
    oracle_sha512_update(c, pad, padlen);
    c->nbytes = keep;
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 8; b++)
            out[8 * i + b] = (uint8_t)(c->h[i] >> (56 - 8 * b));
}

/* One-shot digest of a memory buffer (raw 64 bytes). */
void oracle_sha512(const uint8_t *data, size_t len, uint8_t out[64])
{
    oracle_sha512_ctx c;
    oracle_sha512_init(&c);
    oracle_sha512_update(&c, data, len);
    oracle_sha512_final(&c, out);
}

/* n buffers packed in one base array; serial, one core (cpu_baseline leg). */
void oracle_sha512_batch(const uint8_t *base, const uint64_t *offsets,
                         const uint64_t *lens, size_t n, uint8_t *digests)
{
    for (size_t i = 0; i < n; i++)
        oracle_sha512(base + offsets[i], lens[i], digests + 64 * i);
}

static void to_hex(const uint8_t d[64], char hex[129])
{
    static const char x[] = "0123456789abcdef"; /* hex.EncodeToString: lowercase */
    for (int i = 0; i < 64; i++) { hex[2 * i] = x[d[i] >> 4]; hex[2 * i + 1] = x[d[i] & 15]; }
    hex[128] = 0;
}

/* helpers/helpers.go:187-201 -- returns 0 or errno; hex gets 128 chars + NUL. */
int oracle_sha512sum(const char *infile, char hex[129])
{
    int fd = open(infile, O_RDONLY);
    if (fd < 0) return errno;
    oracle_sha512_ctx c;
    oracle_sha512_init(&c);
    static __thread uint8_t buf[32 * 1024]; /* io.Copy's default buffer */
    for (;;) {
        ssize_t r = read(fd, buf, sizeof buf);
        if (r < 0) { if (errno == EINTR) continue; int e = errno; close(fd); return e; }
        if (r == 0) break;
        oracle_sha512_update(&c, buf, (size_t)r);
    }
    close(fd);
    uint8_t d[64];
    oracle_sha512_final(&c, d);
    to_hex(d, hex);
    return 0;
}

/* ---- growable text buffer --------------------------------------------- */
typedef struct { char *p; size_t n, cap; } sbuf;
static int sb_put(sbuf *s, const char *t, size_t n)
{
    if (s->n + n + 1 > s->cap) {
        size_t c = s->cap ? s->cap * 2 : 4096;
        while (c < s->n + n + 1) c *= 2;
        char *q = realloc(s->p, c);
        if (!q) return -1;
        s->p = q; s->cap = c;
    }
    memcpy(s->p + s->n, t, n);
    s->n += n; s->p[s->n] = 0;
    return 0;
}
static int sb_puts(sbuf *s, const char *t) { return sb_put(s, t, strlen(t)); }

/* snappy/hashes.go:33-57 -- returns -1 for "Unknown file mode". */
int oracle_mode_string(unsigned st_mode, char out[11])
{
    memcpy(out, "----------", 11);
    if (S_ISDIR(st_mode)) out[0] = 'd';
    else if (S_ISLNK(st_mode)) out[0] = 'l';
    else if (S_ISREG(st_mode)) out[0] = 'f';
    else return -1;
    static const char rwx[] = "rwxrwxrwx";
    for (int i = 0; i < 9; i++)
        if (st_mode & (1u << (8 - i))) out[i + 1] = rwx[i];
    return 0;
}

/* The value of `name:` as gopkg.in/yaml.v2 @ 49c95bdc writes a Go string (snappy/build.go:249-264 marshals every
 * file name).  The library is not in the reference tree: its published algorithm is restated -- resolve.go (a text
 * that would read back as bool/null/int/float is double-quoted; so are base-60 floats, encode.go isBase60Float),
 * emitterc.go yaml_emitter_analyze_scalar / select_scalar_style / the three scalar writers (the libyaml port: plain
 * unless indicators, leading or trailing space forbid it in block context, then single-quoted, double-quoted when a
 * character is not printable; lines fold at a space past column 80, continuation indented by 4).
 * PARITY UNPINNED: the reference's fixtures hold plain names only (snappy/hashes_test.go:89-103).
 * Returns 0, or -1 for what is not restated (invalid UTF-8 -> !!binary, line breaks, spellings whose type depends on
 * the Go release: 0o17, hex floats, "<<"). */
static int nm_digit(int c) { return c >= '0' && c <= '9'; }
static int nm_hex(int c) { return nm_digit(c) || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }
static int nm_ieq(const char *s, const char *t) { return strcasecmp(s, t) == 0; }

static int nm_float(const char *s) /* strconv.ParseFloat, decimal grammar + inf/infinity/nan */
{
    const char *p = s;
    if (*p == '+' || *p == '-') { if (nm_ieq(p + 1, "inf") || nm_ieq(p + 1, "infinity")) return 1; p++; }
    else if (nm_ieq(p, "inf") || nm_ieq(p, "infinity") || nm_ieq(p, "nan")) return 1;
    int nd = 0;
    while (nm_digit(*p)) { p++; nd++; }
    if (*p == '.') { p++; while (nm_digit(*p)) { p++; nd++; } }
    if (!nd) return 0;
    if (*p == 'e' || *p == 'E') {
        p++;
        if (*p == '+' || *p == '-') p++;
        if (!nm_digit(*p)) return 0;
        while (nm_digit(*p)) p++;
    }
    return *p == 0;
}
/* resolve.go asks strconv with `err == nil`, and a value out of range is an error there: what follows answers "does
 * the call succeed", by the C library's own conversions (strtoull / strtod: ERANGE is Go's ErrRange; both round a
 * decimal correctly, so the overflow threshold is the same one). */
static int nm_uint_digits(const char *s, int base, unsigned long long *mag) /* all of s digits of `base` and the value fits 64 bits */
{
    if (!*s) return 0;
    for (const char *q = s; *q; q++) {
        int d = nm_digit(*q) ? *q - '0' : (*q >= 'a' && *q <= 'f') ? *q - 'a' + 10 : (*q >= 'A' && *q <= 'F') ? *q - 'A' + 10 : 99;
        if (d >= base) return 0;
    }
    char *end = 0;
    errno = 0;
    *mag = strtoull(s, &end, base);
    return errno == 0 && end && *end == 0;
}
static int nm_int_base(const char *s, int base, int try_unsigned) /* ParseInt(s, base, 64), then ParseUint if asked */
{
    int neg = *s == '-', sign = *s == '+' || *s == '-';
    const char *q = s + sign;
    if (base == 0) { /* Go before 1.13: 0x hex, a leading 0 octal, else decimal */
        if (q[0] == '0' && (q[1] == 'x' || q[1] == 'X')) { base = 16; q += 2; }
        else base = q[0] == '0' ? 8 : 10;
    }
    unsigned long long mag;
    if (!nm_uint_digits(q, base, &mag)) return 0;
    if (neg ? mag <= 9223372036854775808ull : mag <= 9223372036854775807ull) return 1;
    return try_unsigned && !sign;
}
static int nm_int(const char *s) { return nm_int_base(s, 0, 1); }
static int nm_float_ok(const char *s) /* ParseFloat(s, 64) without ErrRange */
{
    if (!nm_float(s)) return 0;
    const char *q = s + (*s == '+' || *s == '-');
    if (!nm_digit(*q) && *q != '.') return 1; /* inf, infinity, nan */
    static locale_t cloc;
    if (!cloc) cloc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    errno = 0;
    double v = cloc ? strtod_l(s, 0, cloc) : strtod(s, 0);
    return !(v == HUGE_VAL || v == -HUGE_VAL); /* underflow to zero is no error in Go */
}
static int nm_bin_spelling(const char *s) /* 0b / -0b followed by binary digits only */
{
    if (*s == '-') s++;
    if (s[0] != '0' || s[1] != 'b' || !s[2]) return 0;
    for (s += 2; *s; s++) if (*s != '0' && *s != '1') return 0;
    return 1;
}
static int nm_bin(const char *s) /* resolve.go: ParseInt(plain[2:], 2, 64) then ParseUint; behind -0b ParseInt alone */
{
    if (!strncmp(s, "0b", 2)) return nm_int_base(s + 2, 2, 1);
    if (!strncmp(s, "-0b", 3)) return nm_int_base(s + 3, 2, 0);
    return 0;
}
static int nm_version_dependent(const char *s)
{
    const char *q = s;
    if (*q == '+' || *q == '-') q++;
    if (q[0] != '0' || !q[1] || !q[2]) return 0;
    if (q[1] == 'o' || q[1] == 'O') { for (q += 2; *q; q++) if (*q < '0' || *q > '7') return 0; return 1; }
    if (q[1] == 'x' || q[1] == 'X') {
        int has_p = 0;
        for (q += 2; *q; q++) {
            if (*q == 'p' || *q == 'P') has_p = 1;
            else if (!nm_hex(*q) && *q != '.' && *q != '+' && *q != '-') return 0;
        }
        return has_p;
    }
    if (q[1] == 'b' || q[1] == 'B') {
        if (nm_bin_spelling(s)) return 0;
        for (q += 2; *q; q++) if (*q != '0' && *q != '1') return 0;
        return 1;
    }
    return 0;
}
static int nm_base60(const char *s)
{
    if (!(*s == '+' || *s == '-' || nm_digit(*s)) || !strchr(s, ':')) return 0;
    if (*s == '+' || *s == '-') s++;
    if (!nm_digit(*s)) return 0;
    s++;
    while (nm_digit(*s) || *s == '_') s++;
    int groups = 0;
    while (*s == ':') {
        s++;
        if (*s >= '0' && *s <= '5' && nm_digit(s[1])) s += 2;
        else if (nm_digit(*s)) s += 1;
        else return 0;
        groups++;
    }
    if (!groups) return 0;
    if (*s == '.') { s++; while (nm_digit(*s) || *s == '_') s++; }
    return *s == 0;
}
/* 1: resolves to another type than !!str; 0: a string; -1: not restated */
static int nm_resolves(const char *s)
{
    static const char *const mapped[] = {"y", "Y", "yes", "Yes", "YES", "on", "On", "ON", "n", "N", "no", "No", "NO", "off",
        "Off", "OFF", "true", "True", "TRUE", "false", "False", "FALSE", "~", "null", "Null", "NULL", ".nan", ".NaN", ".NAN",
        ".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF", "-.inf", "-.Inf", "-.INF", 0};
    int c0 = (unsigned char)s[0];
    int hm = c0 && strchr("yYnNtTfFoO~", c0) != 0, hn = c0 == '+' || c0 == '-' || nm_digit(c0), hd = c0 == '.';
    if (!hm && !hn && !hd) return strcmp(s, "<<") == 0 ? -1 : 0;
    for (int i = 0; mapped[i]; i++) if (!strcmp(s, mapped[i])) return 1;
    if (hm) return 0;
    if (hd) return nm_float_ok(s);
    char plain[4200];
    size_t k = 0;
    for (const char *p = s; *p && k + 1 < sizeof plain; p++) if (*p != '_') plain[k++] = *p;
    plain[k] = 0;
    if (nm_version_dependent(plain)) return -1;
    return nm_int(plain) || nm_float_ok(plain) || nm_bin(plain);
}
static int nm_decode(const unsigned char *p, uint32_t *cp) /* bytes of the well-formed UTF-8 sequence at p, 0 if none */
{
    if (p[0] < 0x80) { *cp = p[0]; return 1; }
    if (p[0] >= 0xC2 && p[0] <= 0xDF && (p[1] & 0xC0) == 0x80) { *cp = ((p[0] & 0x1Fu) << 6) | (p[1] & 0x3Fu); return 2; }
    if (p[0] >= 0xE0 && p[0] <= 0xEF && (p[1] & 0xC0) == 0x80 && (p[2] & 0xC0) == 0x80) {
        uint32_t v = ((p[0] & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu);
        if (v < 0x800 || (v >= 0xD800 && v <= 0xDFFF)) return 0;
        *cp = v; return 3;
    }
    if (p[0] >= 0xF0 && p[0] <= 0xF4 && (p[1] & 0xC0) == 0x80 && (p[2] & 0xC0) == 0x80 && (p[3] & 0xC0) == 0x80) {
        uint32_t v = ((p[0] & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu);
        if (v < 0x10000 || v > 0x10FFFF) return 0;
        *cp = v; return 4;
    }
    return 0;
}
static int nm_printable(uint32_t c)
{
    return c == 0x0A || (c >= 0x20 && c <= 0x7E) || (c >= 0xA0 && c <= 0xD7FF) || (c >= 0xE000 && c <= 0xFFFD && c != 0xFEFF);
}
typedef struct { sbuf *out; int col; } nm_emit;
static void nm_put(nm_emit *e, char c) { sb_put(e->out, &c, 1); e->col++; }
static void nm_fold(nm_emit *e) { sb_puts(e->out, "\n    "); e->col = 4; }

static int name_scalar(sbuf *out, const char *s)
{
    size_t len = strlen(s);
    if (len == 0 || len > 4096) return -1;
    uint32_t cp[4100]; unsigned char w[4100]; size_t n = 0;
    for (size_t i = 0; i < len;) {
        uint32_t c; int k = nm_decode((const unsigned char *)s + i, &c);
        if (!k || c == '\n' || c == '\r' || c == 0x85 || c == 0x2028 || c == 0x2029) return -1;
        cp[n] = c; w[n] = (unsigned char)k; n++; i += (size_t)k;
    }
    int res = nm_resolves(s);
    if (res < 0) return -1;
    int style = (res == 1 || nm_base60(s)) ? 2 : 0; /* 0 plain, 1 single, 2 double */
    int block_ind = len >= 3 && (!strncmp(s, "---", 3) || !strncmp(s, "...", 3));
    int special = 0, lead = 0, trail = 0, prev_ws = 1;
    for (size_t k = 0; k < n; k++) {
        uint32_t c = cp[k];
        int next_ws = k + 1 >= n || cp[k + 1] == ' ' || cp[k + 1] == '\t';
        if (k == 0) {
            if (c && c < 0x80 && strchr("#,[]{}&*!|>'\"%@`", (int)c)) block_ind = 1;
            else if ((c == '?' || c == ':' || c == '-') && next_ws) block_ind = 1;
        } else if ((c == ':' && next_ws) || (c == '#' && prev_ws)) block_ind = 1;
        if (!nm_printable(c)) special = 1;
        if (c == ' ') { if (k == 0) lead = 1; if (k + 1 == n) trail = 1; }
        prev_ws = c == ' ' || c == '\t';
    }
    if (style == 0 && (lead || trail || special || block_ind)) style = 1;
    if (style == 1 && special) style = 2;
    nm_emit e = {out, 7}; /* behind "- name:" */
    nm_put(&e, ' ');
    if (style == 1) nm_put(&e, '\'');
    if (style == 2) nm_put(&e, '"');
    int spaces = 0;
    size_t bi = 0;
    for (size_t k = 0; k < n; bi += w[k], k++) {
        uint32_t c = cp[k];
        if (style == 2 && (!nm_printable(c) || c == 0xFEFF || c == '"' || c == '\\')) {
            nm_put(&e, '\\');
            const char *esc = c == 0 ? "0" : c == 7 ? "a" : c == 8 ? "b" : c == 9 ? "t" : c == 0xB ? "v" : c == 0xC ? "f" :
                              c == 0xD ? "r" : c == 0x1B ? "e" : c == '"' ? "\"" : c == '\\' ? "\\" : 0;
            if (esc) nm_put(&e, esc[0]);
            else {
                int digits = c <= 0xFF ? 2 : c <= 0xFFFF ? 4 : 8;
                nm_put(&e, digits == 2 ? 'x' : digits == 4 ? 'u' : 'U');
                for (int sh = (digits - 1) * 4; sh >= 0; sh -= 4) nm_put(&e, "0123456789ABCDEF"[(c >> sh) & 15]);
            }
            spaces = 0;
        } else if (c == ' ') {
            int fold = !spaces && e.col > 80;
            if (style == 0) fold = fold && !(k + 1 < n && cp[k + 1] == ' ');
            else if (style == 1) fold = fold && k > 0 && k + 1 < n && cp[k + 1] != ' ';
            else fold = fold && k > 0 && k + 1 < n;
            if (fold) { nm_fold(&e); if (style == 2 && cp[k + 1] == ' ') nm_put(&e, '\\'); }
            else nm_put(&e, ' ');
            spaces = 1;
        } else {
            if (style == 1 && c == '\'') nm_put(&e, '\'');
            sb_put(out, s + bi, w[k]); e.col++;
            spaces = 0;
        }
    }
    if (style == 1) nm_put(&e, '\'');
    if (style == 2) nm_put(&e, '"');
    return 0;
}

enum { ORACLE_OK = 0, ORACLE_EIO = -1, ORACLE_EMODE = -2, ORACLE_EUNSAFE = -3, ORACLE_ENOMEM = -4 };

typedef struct { sbuf *out; const char *root; size_t rootlen; int err; int saved_errno; size_t nfiles; } walkst;

static int cmp_names(const void *a, const void *b)
{
    return strcmp(*(char *const *)a, *(char *const *)b); /* sort.Strings: bytewise */
}

/* path/filepath.Walk as writeHashes uses it (snappy/build.go:228-259). */
static void visit(walkst *w, const char *path, const struct stat *stp);

static void walk(walkst *w, const char *path)
{
    struct stat st;
    if (w->err) return;
    if (lstat(path, &st) != 0) { w->err = ORACLE_EIO; w->saved_errno = errno; return; }
    visit(w, path, &st);
    if (w->err) return;
    /* returning nil (not SkipDir) for /DEBIAN means Walk still descends; the
     * children carry the same prefix and are skipped one by one. */
    if (!S_ISDIR(st.st_mode)) return;
    DIR *d = opendir(path);
    if (!d) {
        /* filepath.Walk: `names, err := readDirNames(path); if err != nil { return walkFn(path, info, err) }` -- the
         * callback is called a SECOND time for the directory, and writeHashes' callback never looks at its err
         * argument (build.go:228): the record is appended again, nil is returned, the walk goes on. */
        visit(w, path, &st);
        return;
    }
    char **names = 0; size_t n = 0, cap = 0;
    struct dirent *de;
    while ((de = readdir(d))) {
        if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
        if (n == cap) { cap = cap ? cap * 2 : 64; names = realloc(names, cap * sizeof *names); }
        names[n++] = strdup(de->d_name);
    }
    closedir(d);
    qsort(names, n, sizeof *names, cmp_names);
    for (size_t i = 0; i < n; i++) {
        size_t L = strlen(path) + 1 + strlen(names[i]) + 1;
        char *child = malloc(L);
        snprintf(child, L, "%s/%s", path, names[i]);
        walk(w, child);
        free(child); free(names[i]);
    }
    free(names);
}

/* the callback of writeHashes (snappy/build.go:228-259) */
static void visit(walkst *w, const char *path, const struct stat *stp)
{
    const struct stat st = *stp;
    const char *rel = path + w->rootlen;
    int skip = (strncmp(rel, "/DEBIAN", 7) == 0) || rel[0] == 0; /* build.go:229, :232 */
    if (!skip) {
        char mode[11], hex[129], line[64];
        const char *name = rel + 1; /* build.go:250 */
        if (oracle_mode_string(st.st_mode, mode) != 0) { w->err = ORACLE_EMODE; return; }
        size_t mark = w->out->n;
        sb_puts(w->out, "- name:");
        if (name_scalar(w->out, name) != 0) { w->out->n = mark; w->err = ORACLE_EUNSAFE; return; }
        sb_puts(w->out, "\n");
        if (S_ISREG(st.st_mode)) { /* build.go:240-247 */
            int e = oracle_sha512sum(path, hex);
            if (e) { w->err = ORACLE_EIO; w->saved_errno = e; return; }
            snprintf(line, sizeof line, "  size: %lld\n", (long long)st.st_size);
            sb_puts(w->out, line);
            sb_puts(w->out, "  sha512: "); sb_puts(w->out, hex); sb_puts(w->out, "\n");
        }
        sb_puts(w->out, "  mode: "); sb_puts(w->out, mode); sb_puts(w->out, "\n");
        w->nfiles++;
    }
}

/* writeHashes minus the file write: YAML text in *yaml_out (free with
 * oracle_free).  Returns ORACLE_*; *err_no gets errno for ORACLE_EIO. */
int oracle_hashes_yaml(const char *build_dir, const char *data_tar, char **yaml_out, size_t *yaml_len, int *err_no)
{
    sbuf out = {0, 0, 0};
    char hex[129];
    int e = oracle_sha512sum(data_tar, hex); /* build.go:222 */
    if (e) { if (err_no) *err_no = e; return ORACLE_EIO; }
    sb_puts(&out, "archive-sha512: "); sb_puts(&out, hex); sb_puts(&out, "\n");
    size_t mark = out.n;
    sb_puts(&out, "files:\n");
    size_t rl = strlen(build_dir);
    while (rl > 1 && build_dir[rl - 1] == '/') rl--;
    char *root = strndup(build_dir, rl);
    walkst w = {&out, root, rl, 0, 0, 0};
    walk(&w, root);
    free(root);
    if (w.err) { free(out.p); if (err_no) *err_no = w.saved_errno; return w.err; }
    if (w.nfiles == 0) { out.n = mark; sb_puts(&out, "files: []\n"); } /* yaml.v2 empty slice; unpinned */
    *yaml_out = out.p;
    if (yaml_len) *yaml_len = out.n;
    return ORACLE_OK;
}

/* writeHashes proper: also creates DEBIAN/ (0755) and writes hashes.yaml (0644). */
int oracle_write_hashes(const char *build_dir, const char *data_tar, int *err_no)
{
    char *y = 0; size_t n = 0;
    size_t L = strlen(build_dir) + 32;
    char *p = malloc(L);
    snprintf(p, L, "%s/DEBIAN", build_dir);
    mkdir(p, 0755); /* os.MkdirAll(debianDir, 0755), error ignored (build.go:219) */
    int rc = oracle_hashes_yaml(build_dir, data_tar, &y, &n, err_no);
    if (rc) { free(p); return rc; }
    snprintf(p, L, "%s/DEBIAN/hashes.yaml", build_dir);
    FILE *f = fopen(p, "wb");
    if (!f) { if (err_no) *err_no = errno; free(p); free(y); return ORACLE_EIO; }
    fwrite(y, 1, n, f);
    fclose(f);
    chmod(p, 0644);
    free(p); free(y);
    return ORACLE_OK;
}

void oracle_free(void *p) { free(p); }

/* SURVEY sec. 8(d) synthetic content: little-endian SplitMix64 stream seeded
 * with 0x5eed000000000000 ^ file_index, truncated to len. */
void oracle_fill_synthetic(uint8_t *dst, uint64_t len, uint64_t file_index)
{
    uint64_t s = 0x5eed000000000000ULL ^ file_index;
    uint64_t i = 0;
    while (i < len) {
        s += 0x9e3779b97f4a7c15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z ^= z >> 31;
        for (int b = 0; b < 8 && i < len; b++, i++) dst[i] = (uint8_t)(z >> (8 * b));
    }
}

/* ---- helpers.FilesAreEqual / streamsEqual / DirUpdated (helpers/cmp.go:28-114), row f4 ------ */

/* io.ReadAtLeast(f, buf, bufsz): fills buf unless EOF intervenes; returns bytes read, *eof says
 * whether the stream ended (0 bytes: io.EOF; fewer than bufsz: io.ErrUnexpectedEOF), *err a read error. */
static size_t read_at_least(int fd, uint8_t *buf, size_t want, int *eof, int *err)
{
    size_t got = 0;
    *eof = 0; *err = 0;
    while (got < want) {
        ssize_t r = read(fd, buf + got, want - got);
        if (r < 0) { if (errno == EINTR) continue; *err = 1; break; }
        if (r == 0) { *eof = 1; break; }
        got += (size_t)r;
    }
    return got;
}

int oracle_files_equal(const char *a, const char *b) /* cmp.go:31-60 */
{
    enum { BUFSZ = 16 * 1024 }; /* cmp.go:27 */
    int fa = open(a, O_RDONLY);
    if (fa < 0) return 0;
    int fb = open(b, O_RDONLY);
    if (fb < 0) { close(fa); return 0; }
    struct stat sa, sb;
    int eq = 0;
    if (fstat(fa, &sa) == 0 && fstat(fb, &sb) == 0 && sa.st_size == sb.st_size) {
        static __thread uint8_t bufa[BUFSZ], bufb[BUFSZ];
        for (;;) { /* streamsEqual, cmp.go:62-86 */
            int eofa, eofb, erra, errb;
            size_t ra = read_at_least(fa, bufa, BUFSZ, &eofa, &erra);
            size_t rb = read_at_least(fb, bufb, BUFSZ, &eofb, &errb);
            if (erra || errb) { eq = 0; break; }
            if (eofa && ra == 0 && eofb && rb == 0) { eq = 1; break; }     /* both io.EOF */
            if ((eofa || eofb) && !(eofa && eofb && ra && rb)) { eq = 0; break; } /* only "both ErrUnexpectedEOF" may still be equal */
            if (ra != rb || memcmp(bufa, bufb, ra) != 0) { eq = 0; break; }
        }
    }
    close(fa); close(fb);
    return eq;
}

/* DirUpdated: names (pfx prepended) one per line into *out (oracle_free). cmp.go:88-114 */
int oracle_dir_updated(const char *dir_a, const char *dir_b, const char *pfx, char **out, size_t *count)
{
    sbuf s = {0, 0, 0};
    size_t k = 0;
    char **names = 0; size_t n = 0, cap = 0;
    DIR *d = opendir(dir_a);
    if (d) {
        struct dirent *de;
        while ((de = readdir(d))) {
            if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
            if (n == cap) { cap = cap ? cap * 2 : 64; names = realloc(names, cap * sizeof *names); }
            names[n++] = strdup(de->d_name);
        }
        closedir(d);
    }
    qsort(names, n, sizeof *names, cmp_names); /* filepath.Glob sorts */
    for (size_t i = 0; i < n; i++) {
        size_t la = strlen(dir_a) + strlen(names[i]) + 2, lb = strlen(dir_b) + strlen(names[i]) + 2;
        char *fa = malloc(la), *fb = malloc(lb);
        snprintf(fa, la, "%s/%s", dir_a, names[i]);
        snprintf(fb, lb, "%s/%s", dir_b, names[i]);
        struct stat st;
        int isdir = (stat(fa, &st) == 0 && S_ISDIR(st.st_mode)); /* IsDirectory (helpers.go:227) */
        if (!isdir && stat(fb, &st) == 0 /* FileExists */ && !oracle_files_equal(fa, fb)) {
            sb_puts(&s, pfx ? pfx : ""); sb_puts(&s, names[i]); sb_puts(&s, "\n");
            k++;
        }
        free(fa); free(fb); free(names[i]);
    }
    free(names);
    if (!s.p) sb_puts(&s, "");
    *out = s.p;
    *count = k;
    return 0;
}
