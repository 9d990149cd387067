// hostsha.cpp -- the library's own SHA-512 on host cores, used ONLY by the opt-in
// hybrid scheduler (snaphash_config.host_threads > 0): streams whose single-stream
// time on the GPU would exceed the batch makespan (a lone stream advances at
// ~40 MB/s on MI355X, a host core at ~0.5 GB/s) are hashed here, concurrently
// with the GPU batch.  Same compression function as the kernels (sha512_core.h,
// FIPS 180-4), continuing from any chaining value, so a stream may also start on
// the GPU and finish here.  This is not a fallback: without a gfx950 device
// snaphash_init still fails, and with host_threads == 0 (the default) nothing in
// this file runs.
//
// What it computes is helpers.Sha512sum (reference helpers/helpers.go:187-201):
// io.Copy in chunks into crypto/sha512, i.e. streaming SHA-512 to EOF.
#include "hostsha.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "sha512_core.h"

namespace snaphash {

void host_sha512_init(HostSha& s)
{
    for (int k = 0; k < 8; ++k) s.H[k] = IV512[k];
    s.total = 0;
    s.ntail = 0;
}

void host_sha512_resume(HostSha& s, const uint64_t H[8], uint64_t total_prev)
{
    for (int k = 0; k < 8; ++k) s.H[k] = H[k];
    s.total = total_prev; // a multiple of 128: only whole blocks are ever handed over
    s.ntail = 0;
}

// One 128-byte block: FIPS 180-4 sec. 6.4.2, rounds fully unrolled over a 16-word rolling schedule.
// (Same function as compress_block in sha512_core.h; spelled for a host core: native 64-bit rotates,
// and a BMI2 clone where the CPU has rorx.)
#define HS_ROTR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
#define HS_S0(x) (HS_ROTR(x, 28) ^ HS_ROTR(x, 34) ^ HS_ROTR(x, 39))
#define HS_S1(x) (HS_ROTR(x, 14) ^ HS_ROTR(x, 18) ^ HS_ROTR(x, 41))
#define HS_s0(x) (HS_ROTR(x, 1) ^ HS_ROTR(x, 8) ^ ((x) >> 7))
#define HS_s1(x) (HS_ROTR(x, 19) ^ HS_ROTR(x, 61) ^ ((x) >> 6))
#define HS_RND(a, b, c, d, e, f, g, h, kw)                                   \
    do {                                                                    \
        const uint64_t t1_ = h + HS_S1(e) + (g ^ (e & (f ^ g))) + (kw);      \
        const uint64_t t2_ = HS_S0(a) + ((a & b) | (c & (a | b)));          \
        d += t1_;                                                           \
        h = t1_ + t2_;                                                      \
    } while (0)

__attribute__((target_clones("default", "bmi2"))) static void blocks(uint64_t H[8], const uint8_t* p, size_t nblocks)
{
    uint64_t a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
    for (; nblocks; --nblocks, p += 128) {
        uint64_t w[16];
        for (int k = 0; k < 16; ++k) {
            uint64_t v;
            memcpy(&v, p + 8 * k, 8);
            w[k] = __builtin_bswap64(v);
        }
        const uint64_t sa = a, sb = b, sc = c, sd = d, se = e, sf = f, sg = g, sh = h;
        for (int t = 0; t < 80; t += 16) {
            if (t)
                for (int k = 0; k < 16; ++k) w[k] += HS_s1(w[(k + 14) & 15]) + w[(k + 9) & 15] + HS_s0(w[(k + 1) & 15]);
            const uint64_t* K = K512 + t;
            HS_RND(a, b, c, d, e, f, g, h, K[0] + w[0]);
            HS_RND(h, a, b, c, d, e, f, g, K[1] + w[1]);
            HS_RND(g, h, a, b, c, d, e, f, K[2] + w[2]);
            HS_RND(f, g, h, a, b, c, d, e, K[3] + w[3]);
            HS_RND(e, f, g, h, a, b, c, d, K[4] + w[4]);
            HS_RND(d, e, f, g, h, a, b, c, K[5] + w[5]);
            HS_RND(c, d, e, f, g, h, a, b, K[6] + w[6]);
            HS_RND(b, c, d, e, f, g, h, a, K[7] + w[7]);
            HS_RND(a, b, c, d, e, f, g, h, K[8] + w[8]);
            HS_RND(h, a, b, c, d, e, f, g, K[9] + w[9]);
            HS_RND(g, h, a, b, c, d, e, f, K[10] + w[10]);
            HS_RND(f, g, h, a, b, c, d, e, K[11] + w[11]);
            HS_RND(e, f, g, h, a, b, c, d, K[12] + w[12]);
            HS_RND(d, e, f, g, h, a, b, c, K[13] + w[13]);
            HS_RND(c, d, e, f, g, h, a, b, K[14] + w[14]);
            HS_RND(b, c, d, e, f, g, h, a, K[15] + w[15]);
        }
        a += sa; b += sb; c += sc; d += sd; e += se; f += sf; g += sg; h += sh;
    }
    H[0] = a; H[1] = b; H[2] = c; H[3] = d; H[4] = e; H[5] = f; H[6] = g; H[7] = h;
}
static inline void block(uint64_t H[8], const uint8_t* p) { blocks(H, p, 1); }

void host_sha512_update(HostSha& s, const uint8_t* p, size_t n)
{
    s.total += n;
    if (s.ntail) {
        const size_t take = (n < 128 - s.ntail) ? n : 128 - s.ntail;
        memcpy(s.tail + s.ntail, p, take);
        s.ntail += (uint32_t)take;
        p += take;
        n -= take;
        if (s.ntail < 128) return;
        block(s.H, s.tail);
        s.ntail = 0;
    }
    if (n >= 128) {
        const size_t nb = n >> 7;
        blocks(s.H, p, nb);
        p += nb << 7;
        n &= 127;
    }
    if (n) {
        memcpy(s.tail, p, n);
        s.ntail = (uint32_t)n;
    }
}

void host_sha512_final(HostSha& s, uint8_t out[64])
{
    uint8_t pad[256];
    memset(pad, 0, sizeof pad);
    memcpy(pad, s.tail, s.ntail);
    pad[s.ntail] = 0x80;
    const size_t plen = (s.ntail < 112) ? 128 : 256;
    const uint64_t bits_hi = s.total >> 61, bits_lo = s.total << 3;
    for (int k = 0; k < 8; ++k) {
        pad[plen - 16 + k] = (uint8_t)(bits_hi >> (56 - 8 * k));
        pad[plen - 8 + k] = (uint8_t)(bits_lo >> (56 - 8 * k));
    }
    block(s.H, pad);
    if (plen == 256) block(s.H, pad + 128);
    for (int k = 0; k < 8; ++k)
        for (int b = 0; b < 8; ++b) out[8 * k + b] = (uint8_t)(s.H[k] >> (56 - 8 * b));
}

int host_sha512_file_from(HostSha& s, const char* path, uint64_t offset, uint64_t expect_len, uint8_t out[64])
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return errno;
    std::vector<uint8_t> buf(1u << 20);
    uint64_t off = offset;
    int err = 0;
    for (;;) {
        const ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)off);
        if (r < 0) {
            if (errno == EINTR) continue;
            err = errno;
            break;
        }
        if (r == 0) break; // EOF, like io.Copy
        host_sha512_update(s, buf.data(), (size_t)r);
        off += (uint64_t)r;
    }
    close(fd);
    if (err) return err;
    if (off != expect_len) return EIO; // the file changed size under the pass: the record's size would disagree
    host_sha512_final(s, out);
    return 0;
}

} // namespace snaphash
