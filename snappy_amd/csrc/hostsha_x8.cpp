// hostsha_x8.cpp -- eight SHA-512 streams at once on ONE host core (AVX-512: a stream per 64-bit lane).
//
// The planner's host side (planner.h) hashes whole streams on host threads: BASELINE config 3 (100 x 1 GiB) goes there
// entirely -- a lone stream advances at 44 MB/s on the GPU -- and a thread of the pool then holds six or seven streams.
// One stream keeps a core's scalar pipes busy at 3.4 cycles a byte and cannot go faster (the round's dependency chain);
// EIGHT independent streams in the lanes of a zmm register take ~28 vector instructions a round for all of them.  The
// message words of eight blocks are brought lane-wise with two 8 x 8 transposes, the rounds are FIPS 180-4 as in
// sha512_core.h with vprorq / vpternlogq for the rotates and the three-input functions.
//
// No reference counterpart (helpers.Sha512sum, helpers/helpers.go:187-201, is one goroutine over one file); the digests
// are the same 64 bytes.  Without AVX-512F/BW the callers keep to hostsha.cpp's one-stream code.
#include "hostsha.h"

#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <memory>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "sha512_core.h"

namespace snaphash {

#if defined(__x86_64__)

bool host_sha512_x8_available()
{
    static const bool ok = [] {
        __builtin_cpu_init();
        if (getenv("SNAPHASH_NO_X8")) return false; // (A/B runs)
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    }();
    return ok;
}

namespace {

#define X8_TARGET __attribute__((target("avx512f,avx512bw")))

// rows r[l] = eight consecutive words of lane l  ->  cols c[j] = word j of lanes 0..7
X8_TARGET inline void transpose8(const __m512i r[8], __m512i c[8])
{
    const __m512i t0 = _mm512_unpacklo_epi64(r[0], r[1]), t1 = _mm512_unpackhi_epi64(r[0], r[1]);
    const __m512i t2 = _mm512_unpacklo_epi64(r[2], r[3]), t3 = _mm512_unpackhi_epi64(r[2], r[3]);
    const __m512i t4 = _mm512_unpacklo_epi64(r[4], r[5]), t5 = _mm512_unpackhi_epi64(r[4], r[5]);
    const __m512i t6 = _mm512_unpacklo_epi64(r[6], r[7]), t7 = _mm512_unpackhi_epi64(r[6], r[7]);
    // t0 = {r0[0] r1[0] | r0[2] r1[2] | r0[4] r1[4] | r0[6] r1[6]} and so on: gather the 128-bit pairs
    const __m512i u0 = _mm512_shuffle_i64x2(t0, t2, 0x88), u1 = _mm512_shuffle_i64x2(t0, t2, 0xDD);
    const __m512i u2 = _mm512_shuffle_i64x2(t4, t6, 0x88), u3 = _mm512_shuffle_i64x2(t4, t6, 0xDD);
    const __m512i v0 = _mm512_shuffle_i64x2(t1, t3, 0x88), v1 = _mm512_shuffle_i64x2(t1, t3, 0xDD);
    const __m512i v2 = _mm512_shuffle_i64x2(t5, t7, 0x88), v3 = _mm512_shuffle_i64x2(t5, t7, 0xDD);
    c[0] = _mm512_shuffle_i64x2(u0, u2, 0x88);
    c[4] = _mm512_shuffle_i64x2(u0, u2, 0xDD);
    c[2] = _mm512_shuffle_i64x2(u1, u3, 0x88);
    c[6] = _mm512_shuffle_i64x2(u1, u3, 0xDD);
    c[1] = _mm512_shuffle_i64x2(v0, v2, 0x88);
    c[5] = _mm512_shuffle_i64x2(v0, v2, 0xDD);
    c[3] = _mm512_shuffle_i64x2(v1, v3, 0x88);
    c[7] = _mm512_shuffle_i64x2(v1, v3, 0xDD);
}

#define X8_XOR3(a, b, c) _mm512_ternarylogic_epi64(a, b, c, 0x96)
#define X8_S0(x) X8_XOR3(_mm512_ror_epi64(x, 28), _mm512_ror_epi64(x, 34), _mm512_ror_epi64(x, 39))
#define X8_S1(x) X8_XOR3(_mm512_ror_epi64(x, 14), _mm512_ror_epi64(x, 18), _mm512_ror_epi64(x, 41))
#define X8_s0(x) X8_XOR3(_mm512_ror_epi64(x, 1), _mm512_ror_epi64(x, 8), _mm512_srli_epi64(x, 7))
#define X8_s1(x) X8_XOR3(_mm512_ror_epi64(x, 19), _mm512_ror_epi64(x, 61), _mm512_srli_epi64(x, 6))
#define X8_CH(e, f, g) _mm512_ternarylogic_epi64(e, f, g, 0xCA)  /* e ? f : g */
#define X8_MAJ(a, b, c) _mm512_ternarylogic_epi64(a, b, c, 0xE8)
#define X8_ADD(a, b) _mm512_add_epi64(a, b)
#define X8_ROUND(a, b, c, d, e, f, g, h, t)                                                                     \
    do {                                                                                                        \
        const __m512i t1_ = X8_ADD(X8_ADD(X8_ADD(h, X8_S1(e)), X8_ADD(X8_CH(e, f, g), _mm512_set1_epi64((long long)K512[t]))), W[(t) & 15]); \
        const __m512i t2_ = X8_ADD(X8_S0(a), X8_MAJ(a, b, c));                                                  \
        d = X8_ADD(d, t1_);                                                                                     \
        h = X8_ADD(t1_, t2_);                                                                                   \
    } while (0)
#define X8_SCHED(t) W[(t) & 15] = X8_ADD(X8_ADD(W[(t) & 15], X8_s0(W[((t) + 1) & 15])), X8_ADD(W[((t) + 9) & 15], X8_s1(W[((t) + 14) & 15])))

X8_TARGET void x8_blocks(uint64_t* S, const uint8_t* const* data, size_t nblocks)
{
    __m512i a = _mm512_loadu_si512(S + 0), b = _mm512_loadu_si512(S + 8), c = _mm512_loadu_si512(S + 16), d = _mm512_loadu_si512(S + 24);
    __m512i e = _mm512_loadu_si512(S + 32), f = _mm512_loadu_si512(S + 40), g = _mm512_loadu_si512(S + 48), h = _mm512_loadu_si512(S + 56);
    const __m512i bswap = _mm512_set_epi8(56, 57, 58, 59, 60, 61, 62, 63, 48, 49, 50, 51, 52, 53, 54, 55, 40, 41, 42, 43, 44, 45, 46, 47, 32, 33, 34, 35, 36,
                                          37, 38, 39, 24, 25, 26, 27, 28, 29, 30, 31, 16, 17, 18, 19, 20, 21, 22, 23, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3,
                                          4, 5, 6, 7);
    for (size_t blk = 0; blk < nblocks; ++blk) {
        __m512i W[16];
        for (int half = 0; half < 2; ++half) {
            __m512i r[8];
            for (int l = 0; l < 8; ++l) r[l] = _mm512_shuffle_epi8(_mm512_loadu_si512(data[l] + blk * 128 + half * 64), bswap);
            transpose8(r, W + 8 * half);
        }
        const __m512i sa = a, sb = b, sc = c, sd = d, se = e, sf = f, sg = g, sh = h;
#define X8_EIGHT(t)                                 \
        X8_ROUND(a, b, c, d, e, f, g, h, (t) + 0);  \
        X8_ROUND(h, a, b, c, d, e, f, g, (t) + 1);  \
        X8_ROUND(g, h, a, b, c, d, e, f, (t) + 2);  \
        X8_ROUND(f, g, h, a, b, c, d, e, (t) + 3);  \
        X8_ROUND(e, f, g, h, a, b, c, d, (t) + 4);  \
        X8_ROUND(d, e, f, g, h, a, b, c, (t) + 5);  \
        X8_ROUND(c, d, e, f, g, h, a, b, (t) + 6);  \
        X8_ROUND(b, c, d, e, f, g, h, a, (t) + 7)
        X8_EIGHT(0);
        X8_EIGHT(8);
        for (int t = 16; t < 80; t += 16) {
            X8_SCHED(0); X8_SCHED(1); X8_SCHED(2); X8_SCHED(3); X8_SCHED(4); X8_SCHED(5); X8_SCHED(6); X8_SCHED(7);
            X8_SCHED(8); X8_SCHED(9); X8_SCHED(10); X8_SCHED(11); X8_SCHED(12); X8_SCHED(13); X8_SCHED(14); X8_SCHED(15);
            X8_EIGHT(t);
            X8_EIGHT(t + 8);
        }
#undef X8_EIGHT
        a = X8_ADD(a, sa); b = X8_ADD(b, sb); c = X8_ADD(c, sc); d = X8_ADD(d, sd);
        e = X8_ADD(e, se); f = X8_ADD(f, sf); g = X8_ADD(g, sg); h = X8_ADD(h, sh);
    }
    _mm512_storeu_si512(S + 0, a); _mm512_storeu_si512(S + 8, b); _mm512_storeu_si512(S + 16, c); _mm512_storeu_si512(S + 24, d);
    _mm512_storeu_si512(S + 32, e); _mm512_storeu_si512(S + 40, f); _mm512_storeu_si512(S + 48, g); _mm512_storeu_si512(S + 56, h);
}

} // namespace

void host_sha512_x8_blocks(uint64_t S[64], const uint8_t* const data[8], size_t nblocks)
{
    if (nblocks) x8_blocks(S, data, nblocks);
}

#else // not x86-64

bool host_sha512_x8_available() { return false; }
void host_sha512_x8_blocks(uint64_t*, const uint8_t* const*, size_t) {}

#endif

// ---- a thread's share of the host part, up to eight streams in flight ---------------------------------------------

namespace {

struct Lane {
    int64_t id = -1;       // the stream (caller's index), -1 = free
    HostStream src{};
    int fd = -1;
    uint64_t done = 0;     // bytes of the stream absorbed into the lane's state
    const uint8_t* p = nullptr; // unread bytes in hand ...
    size_t avail = 0;           // ... and how many
    uint64_t file_off = 0;      // files: next byte to read
    std::unique_ptr<uint8_t[]> buf; // files: the chunk in hand (kLaneChunk + 128 bytes, not zeroed: a small call should not pay for that)
};

constexpr size_t kLaneChunk = 256u << 10;

// a file lane: keep what is in hand (less than a block) and read on, up to a chunk; EOF before the stream's length is
// the caller's EIO ("the file shrank"), as in host_sha512_file_from
int lane_refill(Lane& L)
{
    if (!L.buf) L.buf.reset(new uint8_t[kLaneChunk + 128]);
    if (L.avail && L.p != L.buf.get()) memmove(L.buf.get(), L.p, L.avail);
    L.p = L.buf.get();
    const uint64_t left = L.src.len - L.file_off;
    size_t want = (size_t)std::min<uint64_t>(left, kLaneChunk);
    while (want) {
        ssize_t r;
        do r = pread(L.fd, L.buf.get() + L.avail, want, (off_t)L.file_off); while (r < 0 && errno == EINTR);
        if (r < 0) return errno;
        if (r == 0) return EIO; // shorter than its size said
        L.avail += (size_t)r;
        L.file_off += (uint64_t)r;
        want -= (size_t)r;
    }
    return 0;
}

} // namespace

int host_sha512_many(unsigned lanes, const std::function<int64_t()>& next, const std::function<HostStream(int64_t)>& get, int64_t* err_id)
{
    if (err_id) *err_id = -1;
    if (lanes > 8) lanes = 8;
    const bool x8 = lanes >= 3 && host_sha512_x8_available();
    // one at a time: hostsha.cpp's own code (with its read-ahead for a long file where the caller has a core to spare)
    auto one = [&](int64_t id, const HostStream& s) -> int {
        HostSha hs;
        host_sha512_init(hs);
        if (s.mem || !s.path) {
            host_sha512_update(hs, s.mem, s.len);
            host_sha512_final(hs, s.digest);
            return 0;
        }
        // The process is out of descriptors (EMFILE) or the system is (ENFILE): the reference's loop, one file open at a time
        // (helpers.go:189-194), would have gone through -- other threads' lanes and the staging fill's kept descriptors
        // (FdCache) are what holds them, and they let go as their streams end.  Wait for that, a bounded while.
        int err = 0;
        for (unsigned tries = 0;; ++tries) {
            host_sha512_init(hs);
            err = host_sha512_file_from(hs, s.path, 0, s.len, s.digest, s.read_ahead);
            if ((err != EMFILE && err != ENFILE) || tries >= 4000) break;
            usleep(tries < 100 ? 200 : 2000);
        }
        if (err && err_id) *err_id = id;
        return err;
    };
    if (!x8) {
        for (int64_t id; (id = next()) >= 0;) {
            const int err = one(id, get(id));
            if (err) return err;
        }
        return 0;
    }
    std::vector<int64_t> deferred; // streams whose open met EMFILE / ENFILE while lanes were busy: one at a time, when the lanes have drained

    alignas(64) uint64_t S[64];
    static const std::vector<uint8_t> zeros(kLaneChunk, 0);
    Lane L[8];
    unsigned active = 0;
    bool drained = false;
    int rc = 0;
    auto close_lane = [&](Lane& l) {
        if (l.fd >= 0) close(l.fd);
        l.fd = -1;
        l.id = -1;
        --active;
    };
    // the lane's stream ends within what is in hand (fewer than 128 bytes left, or a memory stream's tail): pad + emit
    auto finish_lane = [&](unsigned k) -> int {
        Lane& l = L[k];
        HostSha hs;
        uint64_t H[8];
        for (int w = 0; w < 8; ++w) H[w] = S[w * 8 + k];
        host_sha512_resume(hs, H, l.done);
        host_sha512_update(hs, l.p, l.avail);
        int err = 0;
        if (l.fd >= 0) { // io.Copy reads to EOF: a file that grew since its size was taken is an error too
            uint8_t probe;
            ssize_t r;
            do r = pread(l.fd, &probe, 1, (off_t)l.file_off); while (r < 0 && errno == EINTR);
            if (r < 0) err = errno;
            else if (r != 0) err = EIO;
        }
        if (!err) host_sha512_final(hs, l.src.digest);
        else if (err_id) *err_id = l.id;
        close_lane(l);
        return err;
    };
    for (;;) {
        // take streams into free lanes
        for (unsigned k = 0; k < lanes && !drained && !rc; ++k) {
            if (L[k].id >= 0) continue;
            const int64_t id = next();
            if (id < 0) { drained = true; break; }
            Lane& l = L[k];
            l.src = get(id);
            if (l.src.alone) { // (the callers hand these out first: no lane is in use yet)
                rc = one(id, l.src);
                --k;
                continue;
            }
            l.id = id;
            l.done = 0;
            l.avail = 0;
            l.file_off = 0;
            l.fd = -1;
            ++active;
            for (int w = 0; w < 8; ++w) S[w * 8 + k] = IV512[w];
            if (l.src.mem || !l.src.path) {
                l.p = l.src.mem;
                l.avail = (size_t)l.src.len;
            } else {
                l.fd = open(l.src.path, O_RDONLY | O_CLOEXEC);
                if (l.fd < 0) {
                    const int e = errno;
                    close_lane(l);
                    if (e == EMFILE || e == ENFILE) { deferred.push_back(id); --k; continue; } // (the lane stays free; the stream waits)
                    rc = e;
                    if (err_id) *err_id = id;
                    break;
                }
                l.p = l.buf.get();
            }
        }
        if (rc || active == 0) break;
        // bytes in hand for every lane; lanes at their tail leave
        size_t nblk = SIZE_MAX;
        bool left_one = false;
        for (unsigned k = 0; k < lanes && !rc; ++k) {
            Lane& l = L[k];
            if (l.id < 0) continue;
            if (l.fd >= 0 && l.avail < 128 && l.file_off < l.src.len) {
                rc = lane_refill(l);
                if (rc) { if (err_id) *err_id = l.id; break; }
            }
            const uint64_t rest = l.src.len - l.done; // of the stream, in hand or not
            if (rest < 128 || l.avail < 128) {        // (avail < 128 with rest >= 128 cannot be: the refill read on)
                rc = finish_lane(k);
                left_one = true;
                continue;
            }
            nblk = std::min(nblk, l.avail / 128);
        }
        if (rc) break;
        if (left_one) continue; // fill the freed lanes first
        if (active < 3 && drained) { // too few to pay for eight lanes: the rest one by one from where they stand
            for (unsigned k = 0; k < lanes && !rc; ++k) {
                Lane& l = L[k];
                if (l.id < 0) continue;
                HostSha hs;
                uint64_t H[8];
                for (int w = 0; w < 8; ++w) H[w] = S[w * 8 + k];
                host_sha512_resume(hs, H, l.done);
                host_sha512_update(hs, l.p, l.avail);
                if (l.fd >= 0) {
                    // (the file is opened again by its path: the lane's own descriptor goes first, so that a thread never
                    // needs two for one stream; HostSha is a plain value -- a try that found no descriptor starts from a copy)
                    close(l.fd);
                    l.fd = -1;
                    for (unsigned tries = 0;; ++tries) {
                        HostSha h2 = hs;
                        rc = host_sha512_file_from(h2, l.src.path, l.file_off, l.src.len, l.src.digest, false);
                        if ((rc != EMFILE && rc != ENFILE) || tries >= 4000) break;
                        usleep(tries < 100 ? 200 : 2000);
                    }
                    if (rc && err_id) *err_id = l.id;
                } else {
                    host_sha512_final(hs, l.src.digest);
                }
                close_lane(l);
            }
            break;
        }
        nblk = std::min(nblk, kLaneChunk / 128); // (a free lane reads zeros: that many of them are at hand)
        const uint8_t* ptr[8];
        for (unsigned k = 0; k < 8; ++k) ptr[k] = (k < lanes && L[k].id >= 0) ? L[k].p : zeros.data();
        host_sha512_x8_blocks(S, ptr, nblk); // (a free lane's column of S is rewritten when a stream moves in)
        for (unsigned k = 0; k < lanes; ++k) {
            Lane& l = L[k];
            if (l.id < 0) continue;
            l.p += nblk * 128;
            l.avail -= nblk * 128;
            l.done += (uint64_t)nblk * 128;
        }
    }
    for (unsigned k = 0; k < 8; ++k)
        if (L[k].fd >= 0) close(L[k].fd);
    for (size_t q = 0; q < deferred.size() && !rc; ++q) rc = one(deferred[q], get(deferred[q]));
    return rc;
}

} // namespace snaphash
