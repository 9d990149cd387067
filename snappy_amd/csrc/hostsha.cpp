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

static inline void block(uint64_t H[8], const uint8_t* p)
{
    uint64_t w[16];
    for (int k = 0; k < 16; ++k) {
        uint32_t d0, d1;
        memcpy(&d0, p + 8 * k, 4);
        memcpy(&d1, p + 8 * k + 4, 4);
        w[k] = be64(d0, d1);
    }
    compress_block(H, w, true, K512);
}

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
    for (; n >= 128; p += 128, n -= 128) block(s.H, p);
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
