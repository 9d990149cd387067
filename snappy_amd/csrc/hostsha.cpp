// hostsha.cpp -- the library's own SHA-512 on host cores.  It runs in the DEFAULT configuration (ABI 3+): the
// planner of snaphash_api.cpp (plan_host_streams) hands it every stream that would be slower on the GPU than
// on a host core -- a lone stream advances at ~45 MB/s on MI355X, a host core does ~1.4 GB/s -- concurrently with
// the GPU batch: the package's data.tar.gz, a 1 GiB member, and whole batches too small to repay a kernel launch
// (the literal one-file helpers.Sha512sum call).  The data.tar.gz producer uses it for the archive digest, the one
// stream of that pass that cannot be parallel.  Same compression function as the kernels (sha512_core.h,
// FIPS 180-4), continuing from any chaining value.  This is not a fallback: without a gfx950 device snaphash_init
// still fails; only SNAPHASH_FLAG_GPU_ONLY keeps this file idle.  (A pool thread with several streams to hash runs them
// eight at a time: hostsha_x8.cpp; this file is the one-stream code and the pieces both share.)
//
// What it computes is helpers.Sha512sum (reference helpers/helpers.go:187-201):
// io.Copy in chunks into crypto/sha512, i.e. streaming SHA-512 to EOF.
#include "hostsha.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
// (Same function as compress_block in sha512_core.h; spelled for a host core: native 64-bit rotates.  Baseline x86-64
// only: a CPU with rorx also has AVX2 and takes blocks_avx2 below, so a BMI2 clone of this one bought nothing -- and
// target_clones makes an IFUNC, whose resolver runs before any runtime is up: the ThreadSanitizer build of
// tests/tsan_host.cpp died in it.)
#define HS_ROTR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
#define HS_S0(x) (HS_ROTR(x, 28) ^ HS_ROTR(x, 34) ^ HS_ROTR(x, 39))
#define HS_S1(x) (HS_ROTR(x, 14) ^ HS_ROTR(x, 18) ^ HS_ROTR(x, 41))
#define HS_s0(x) (HS_ROTR(x, 1) ^ HS_ROTR(x, 8) ^ ((x) >> 7))
#define HS_s1(x) (HS_ROTR(x, 19) ^ HS_ROTR(x, 61) ^ ((x) >> 6))
// HS_ASSOC picks how a round is associated (tools/hostsha_bench.cpp measures them; EPYC 9575F, AVX-512VL schedule:
// 0/1 1.34 GB/s, 2 1.42, 3 1.43 -- hashlib/OpenSSL on the same core: 1.45)
#ifndef HS_ASSOC
#define HS_ASSOC 3
#endif
#define HS_KEEP(x) asm("" : "+r"(x)) /* the compiler keeps this association */
#if HS_ASSOC == 2 || HS_ASSOC == 3
// Both recurrences at Sigma (3 deep) + one add: e' = ((d + h + kw) + Ch) + Sigma1(e), a' = ((T1 + Maj) + Sigma0(a));
// the sums that do not hang on e or a are formed first.
#if HS_ASSOC == 3
#define HS_KEEP3(x) HS_KEEP(x)
#else
#define HS_KEEP3(x) (void)0
#endif
#define HS_RND(a, b, c, d, e, f, g, h, kw)                                   \
    do {                                                                    \
        uint64_t hk_ = h + (kw);                                            \
        uint64_t dhk_ = d + hk_;                                            \
        HS_KEEP3(hk_);                                                       \
        HS_KEEP3(dhk_);                                                      \
        const uint64_t ch_ = ((f ^ g) & e) ^ g;                             \
        const uint64_t s1_ = HS_S1(e);                                      \
        uint64_t x_ = dhk_ + ch_;                                           \
        uint64_t y_ = hk_ + ch_;                                            \
        HS_KEEP3(x_);                                                        \
        HS_KEEP3(y_);                                                        \
        d = x_ + s1_;                                                       \
        const uint64_t t1_ = y_ + s1_;                                      \
        uint64_t z_ = t1_ + ((a & (b | c)) | (b & c));                      \
        HS_KEEP3(z_);                                                        \
        h = z_ + HS_S0(a);                                                  \
    } while (0)
#elif HS_ASSOC
#define HS_RND(a, b, c, d, e, f, g, h, kw)                                   \
    do {                                                                    \
        /* everything that does not hang on e first: the chain through e is Sigma1, one add, one add */ \
        uint64_t t1_ = (h + (kw)) + (g ^ (e & (f ^ g)));                    \
        t1_ += HS_S1(e);                                                    \
        const uint64_t t2_ = HS_S0(a) + ((a & b) | (c & (a | b)));          \
        d += t1_;                                                           \
        h = t1_ + t2_;                                                      \
    } while (0)
#else
#define HS_RND(a, b, c, d, e, f, g, h, kw)                                   \
    do {                                                                    \
        const uint64_t t1_ = h + HS_S1(e) + (g ^ (e & (f ^ g))) + (kw);      \
        const uint64_t t2_ = HS_S0(a) + ((a & b) | (c & (a | b)));          \
        d += t1_;                                                           \
        h = t1_ + t2_;                                                      \
    } while (0)
#endif

static void blocks(uint64_t H[8], const uint8_t* p, size_t nblocks)
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

// The same block function with the message schedule on the vector unit (two schedule words per 128-bit operation,
// W[t..t+1] = W[t-16..] + s0(W[t-15..]) + W[t-7..] + s1(W[t-2..])), interleaved with the scalar rounds two by two:
// the rounds are a serial chain through e (~6 cycles per round), the schedule is independent work that fits beside
// it on the vector pipes.  Two spellings: AVX-512VL (64-bit vector rotates, three-input xor) and AVX2 (shifts).
#if defined(__x86_64__)
#define HS_VEC_BODY(S0V, S1V)                                                                                        \
    const __m128i bswap = _mm_set_epi64x(0x08090a0b0c0d0e0fLL, 0x0001020304050607LL);                                \
    uint64_t a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];                         \
    alignas(16) uint64_t kw[16];                                                                                     \
    for (; nblocks; --nblocks, p += 128) {                                                                           \
        __m128i X0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 0)), bswap);                              \
        __m128i X1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), bswap);                             \
        __m128i X2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), bswap);                             \
        __m128i X3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), bswap);                             \
        __m128i X4 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 64)), bswap);                             \
        __m128i X5 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 80)), bswap);                             \
        __m128i X6 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 96)), bswap);                             \
        __m128i X7 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 112)), bswap);                            \
        const uint64_t sa = a, sb = b, sc = c, sd = d, se = e, sf = f, sg = g, sh = h;                               \
        const uint64_t* K = K512;                                                                                    \
        /* step i: K+W of rounds 2i, 2i+1 to the stack, the two rounds, then the words sixteen rounds ahead */       \
        _Pragma("GCC unroll 1") for (int t = 0; t < 64; t += 16, K += 16)                                           \
        {                                                                                                            \
            HS_VSTEP(X0, X1, X4, X5, X7, 0, a, b, c, d, e, f, g, h, S0V, S1V);                                        \
            HS_VSTEP(X1, X2, X5, X6, X0, 2, g, h, a, b, c, d, e, f, S0V, S1V);                                        \
            HS_VSTEP(X2, X3, X6, X7, X1, 4, e, f, g, h, a, b, c, d, S0V, S1V);                                        \
            HS_VSTEP(X3, X4, X7, X0, X2, 6, c, d, e, f, g, h, a, b, S0V, S1V);                                        \
            HS_VSTEP(X4, X5, X0, X1, X3, 8, a, b, c, d, e, f, g, h, S0V, S1V);                                        \
            HS_VSTEP(X5, X6, X1, X2, X4, 10, g, h, a, b, c, d, e, f, S0V, S1V);                                       \
            HS_VSTEP(X6, X7, X2, X3, X5, 12, e, f, g, h, a, b, c, d, S0V, S1V);                                       \
            HS_VSTEP(X7, X0, X3, X4, X6, 14, c, d, e, f, g, h, a, b, S0V, S1V);                                       \
        }                                                                                                            \
        HS_VLAST(X0, 0, a, b, c, d, e, f, g, h);                                                                     \
        HS_VLAST(X1, 2, g, h, a, b, c, d, e, f);                                                                     \
        HS_VLAST(X2, 4, e, f, g, h, a, b, c, d);                                                                     \
        HS_VLAST(X3, 6, c, d, e, f, g, h, a, b);                                                                     \
        HS_VLAST(X4, 8, a, b, c, d, e, f, g, h);                                                                     \
        HS_VLAST(X5, 10, g, h, a, b, c, d, e, f);                                                                    \
        HS_VLAST(X6, 12, e, f, g, h, a, b, c, d);                                                                    \
        HS_VLAST(X7, 14, c, d, e, f, g, h, a, b);                                                                    \
        a += sa; b += sb; c += sc; d += sd; e += se; f += sf; g += sg; h += sh;                                      \
    }                                                                                                                \
    H[0] = a; H[1] = b; H[2] = c; H[3] = d; H[4] = e; H[5] = f; H[6] = g; H[7] = h;

// XA = W[t..t+1] (becomes W[t+16..t+17]), XB = W[t+2..], XE/XF = W[t+8..]/W[t+10..], XH = W[t+14..t+15]
#define HS_VSTEP(XA, XB, XE, XF, XH, i, a, b, c, d, e, f, g, h, S0V, S1V)                                             \
    _mm_store_si128((__m128i*)(kw + (i)), _mm_add_epi64(XA, _mm_loadu_si128((const __m128i*)(K + (i)))));             \
    HS_RND(a, b, c, d, e, f, g, h, kw[(i)]);                                                                         \
    HS_RND(h, a, b, c, d, e, f, g, kw[(i) + 1]);                                                                     \
    XA = _mm_add_epi64(_mm_add_epi64(XA, S0V(_mm_alignr_epi8(XB, XA, 8))),                                           \
                       _mm_add_epi64(_mm_alignr_epi8(XF, XE, 8), S1V(XH)))
#define HS_VLAST(XA, i, a, b, c, d, e, f, g, h)                                                                      \
    _mm_store_si128((__m128i*)(kw + (i)), _mm_add_epi64(XA, _mm_loadu_si128((const __m128i*)(K + (i)))));             \
    HS_RND(a, b, c, d, e, f, g, h, kw[(i)]);                                                                         \
    HS_RND(h, a, b, c, d, e, f, g, kw[(i) + 1])

#define HS_S0V_512(x) _mm_ternarylogic_epi64(_mm_ror_epi64(x, 1), _mm_ror_epi64(x, 8), _mm_srli_epi64(x, 7), 0x96)
#define HS_S1V_512(x) _mm_ternarylogic_epi64(_mm_ror_epi64(x, 19), _mm_ror_epi64(x, 61), _mm_srli_epi64(x, 6), 0x96)
#define HS_VROR(x, n) _mm_or_si128(_mm_srli_epi64(x, n), _mm_slli_epi64(x, 64 - (n)))
#define HS_S0V_AVX2(x) _mm_xor_si128(_mm_xor_si128(HS_VROR(x, 1), HS_VROR(x, 8)), _mm_srli_epi64(x, 7))
#define HS_S1V_AVX2(x) _mm_xor_si128(_mm_xor_si128(HS_VROR(x, 19), HS_VROR(x, 61)), _mm_srli_epi64(x, 6))

__attribute__((target("avx512f,avx512vl,bmi2"))) static void blocks_avx512(uint64_t H[8], const uint8_t* p, size_t nblocks)
{
    HS_VEC_BODY(HS_S0V_512, HS_S1V_512)
}
// AVX-512VL, second spelling: the a-recurrence lives on the vector unit as well (a, b, c in xmm registers: Sigma0 is
// three vprorq + one vpternlogq, Maj one vpternlogq), the e-recurrence stays scalar.  T1 crosses to the vector side
// every round, c crosses back as the next round's d (computed three rounds earlier, so its latency is hidden).
// That leaves the scalar pipes 15 operations a round instead of 26, and both recurrences are 4 cycles deep -- on
// paper.  Measured on EPYC 9575F: 0.87 GB/s against 1.43 for the spelling above (the two register-file crossings a
// round cost more than the operations they save): kept as a tested variant, never picked.
#define HS_KW(XA, i) _mm_store_si128((__m128i*)(kw + (i)), _mm_add_epi64(XA, _mm_loadu_si128((const __m128i*)(K + (i)))))
#define HS_SCHED(XA, XB, XE, XF, XH)                                                                     \
    XA = _mm_add_epi64(_mm_add_epi64(XA, HS_S0V_512(_mm_alignr_epi8(XB, XA, 8))),                       \
                       _mm_add_epi64(_mm_alignr_epi8(XF, XE, 8), HS_S1V_512(XH)))
// state (a, b, c | d, e, f, g, h) = (A, B, C | d, e, f, g, h) -> (C', A, B | h', d', e, f, g): C' = a', d' = e', h' = old c
#define HS_RV(A, B, C, d, e, f, g, h, kwv)                                                               \
    do {                                                                                                \
        uint64_t hk_ = h + (kwv);                                                                       \
        uint64_t dhk_ = d + hk_;                                                                        \
        HS_KEEP(hk_);                                                                                   \
        HS_KEEP(dhk_);                                                                                  \
        const uint64_t ch_ = ((f ^ g) & e) ^ g;                                                         \
        const uint64_t s1_ = HS_S1(e);                                                                  \
        uint64_t x_ = dhk_ + ch_;                                                                       \
        uint64_t y_ = hk_ + ch_;                                                                        \
        HS_KEEP(x_);                                                                                    \
        HS_KEEP(y_);                                                                                    \
        d = x_ + s1_;                                                                                   \
        const __m128i t1_ = _mm_cvtsi64_si128((long long)(y_ + s1_));                                   \
        const __m128i z_ = _mm_add_epi64(t1_, _mm_ternarylogic_epi64(A, B, C, 0xE8));                   \
        h = (uint64_t)_mm_cvtsi128_si64(C);                                                             \
        C = _mm_add_epi64(z_, _mm_ternarylogic_epi64(_mm_ror_epi64(A, 28), _mm_ror_epi64(A, 34), _mm_ror_epi64(A, 39), 0x96)); \
    } while (0)

__attribute__((target("avx512f,avx512vl,bmi2"))) static void blocks_avx512_va(uint64_t H[8], const uint8_t* p, size_t nblocks)
{
    const __m128i bswap = _mm_set_epi64x(0x08090a0b0c0d0e0fLL, 0x0001020304050607LL);
    __m128i A = _mm_cvtsi64_si128((long long)H[0]), B = _mm_cvtsi64_si128((long long)H[1]), C = _mm_cvtsi64_si128((long long)H[2]);
    uint64_t d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
    alignas(16) uint64_t kw[16];
    for (; nblocks; --nblocks, p += 128) {
        __m128i X0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 0)), bswap);
        __m128i X1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), bswap);
        __m128i X2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), bswap);
        __m128i X3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), bswap);
        __m128i X4 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 64)), bswap);
        __m128i X5 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 80)), bswap);
        __m128i X6 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 96)), bswap);
        __m128i X7 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 112)), bswap);
        const __m128i sA = A, sB = B, sC = C;
        const uint64_t sd = d, se = e, sf = f, sg = g, sh = h;
        const uint64_t* K = K512;
        // sixteen rounds a trip; the names rotate with period 3 (vector) and 5 (scalar), so a trip ends one name off
        // in each: put them back (register moves)
#define HS_UNROTATE()                                                                                   \
    do {                                                                                                \
        const __m128i tv_ = C; C = B; B = A; A = tv_;                                                   \
        const uint64_t ts_ = h; h = g; g = f; f = e; e = d; d = ts_;                                    \
    } while (0)
        _Pragma("GCC unroll 1") for (int t = 0; t < 64; t += 16, K += 16)
        {
            HS_KW(X0, 0);
            HS_RV(A, B, C, d, e, f, g, h, kw[0]);
            HS_RV(C, A, B, h, d, e, f, g, kw[1]);
            HS_SCHED(X0, X1, X4, X5, X7);
            HS_KW(X1, 2);
            HS_RV(B, C, A, g, h, d, e, f, kw[2]);
            HS_RV(A, B, C, f, g, h, d, e, kw[3]);
            HS_SCHED(X1, X2, X5, X6, X0);
            HS_KW(X2, 4);
            HS_RV(C, A, B, e, f, g, h, d, kw[4]);
            HS_RV(B, C, A, d, e, f, g, h, kw[5]);
            HS_SCHED(X2, X3, X6, X7, X1);
            HS_KW(X3, 6);
            HS_RV(A, B, C, h, d, e, f, g, kw[6]);
            HS_RV(C, A, B, g, h, d, e, f, kw[7]);
            HS_SCHED(X3, X4, X7, X0, X2);
            HS_KW(X4, 8);
            HS_RV(B, C, A, f, g, h, d, e, kw[8]);
            HS_RV(A, B, C, e, f, g, h, d, kw[9]);
            HS_SCHED(X4, X5, X0, X1, X3);
            HS_KW(X5, 10);
            HS_RV(C, A, B, d, e, f, g, h, kw[10]);
            HS_RV(B, C, A, h, d, e, f, g, kw[11]);
            HS_SCHED(X5, X6, X1, X2, X4);
            HS_KW(X6, 12);
            HS_RV(A, B, C, g, h, d, e, f, kw[12]);
            HS_RV(C, A, B, f, g, h, d, e, kw[13]);
            HS_SCHED(X6, X7, X2, X3, X5);
            HS_KW(X7, 14);
            HS_RV(B, C, A, e, f, g, h, d, kw[14]);
            HS_RV(A, B, C, d, e, f, g, h, kw[15]);
            HS_SCHED(X7, X0, X3, X4, X6);
            HS_UNROTATE();
        }
        HS_KW(X0, 0);
        HS_RV(A, B, C, d, e, f, g, h, kw[0]);
        HS_RV(C, A, B, h, d, e, f, g, kw[1]);
        HS_KW(X1, 2);
        HS_RV(B, C, A, g, h, d, e, f, kw[2]);
        HS_RV(A, B, C, f, g, h, d, e, kw[3]);
        HS_KW(X2, 4);
        HS_RV(C, A, B, e, f, g, h, d, kw[4]);
        HS_RV(B, C, A, d, e, f, g, h, kw[5]);
        HS_KW(X3, 6);
        HS_RV(A, B, C, h, d, e, f, g, kw[6]);
        HS_RV(C, A, B, g, h, d, e, f, kw[7]);
        HS_KW(X4, 8);
        HS_RV(B, C, A, f, g, h, d, e, kw[8]);
        HS_RV(A, B, C, e, f, g, h, d, kw[9]);
        HS_KW(X5, 10);
        HS_RV(C, A, B, d, e, f, g, h, kw[10]);
        HS_RV(B, C, A, h, d, e, f, g, kw[11]);
        HS_KW(X6, 12);
        HS_RV(A, B, C, g, h, d, e, f, kw[12]);
        HS_RV(C, A, B, f, g, h, d, e, kw[13]);
        HS_KW(X7, 14);
        HS_RV(B, C, A, e, f, g, h, d, kw[14]);
        HS_RV(A, B, C, d, e, f, g, h, kw[15]);
        HS_UNROTATE();
        A = _mm_add_epi64(A, sA); B = _mm_add_epi64(B, sB); C = _mm_add_epi64(C, sC);
        d += sd; e += se; f += sf; g += sg; h += sh;
    }
    H[0] = (uint64_t)_mm_cvtsi128_si64(A); H[1] = (uint64_t)_mm_cvtsi128_si64(B); H[2] = (uint64_t)_mm_cvtsi128_si64(C);
    H[3] = d; H[4] = e; H[5] = f; H[6] = g; H[7] = h;
}

__attribute__((target("avx2,bmi2"))) static void blocks_avx2(uint64_t H[8], const uint8_t* p, size_t nblocks)
{
    HS_VEC_BODY(HS_S0V_AVX2, HS_S1V_AVX2)
}

#ifndef HS_AVX512_BEST
#define HS_AVX512_BEST blocks_avx512
#endif
typedef void (*BlocksFn)(uint64_t*, const uint8_t*, size_t);
static BlocksFn pick_blocks()
{
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("bmi2")) return HS_AVX512_BEST;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) return blocks_avx2;
    return blocks;
}
static const BlocksFn blocks_best = pick_blocks();
#else
static void (*const blocks_best)(uint64_t*, const uint8_t*, size_t) = blocks;
#endif

// which block function a build runs (tests: every spelling must agree)
int host_sha512_variants() { return 4; }
void host_sha512_blocks_variant(int v, uint64_t H[8], const uint8_t* p, size_t nblocks)
{
#if defined(__x86_64__)
    if (v == 1 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) return blocks_avx2(H, p, nblocks);
    if (v == 2 && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("bmi2")) return blocks_avx512(H, p, nblocks);
    if (v == 3 && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("bmi2")) return blocks_avx512_va(H, p, nblocks);
#endif
    blocks(H, p, nblocks);
}

static inline void block(uint64_t H[8], const uint8_t* p) { blocks_best(H, p, 1); }


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
        blocks_best(s.H, p, nb);
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

namespace {

// A stream of tens of megabytes or more (the package's data.tar.gz, a 1 GiB member): ONE thread hashing it spends an
// eighth of its time copying out of the page cache.  A reader thread keeps two buffers ahead of the hasher, which then
// runs at the block function's own rate (1.27 -> 1.4 GB/s on the archive of bench.py's package leg).
int hash_file_read_ahead(HostSha& s, int fd, uint64_t offset, uint64_t* end_off)
{
    constexpr size_t kBuf = 1u << 20;
    constexpr int kRing = 4;
    std::vector<uint8_t> ring(kBuf * kRing);
    ssize_t got[kRing];
    std::mutex mu;
    std::condition_variable cv;
    int produced = 0, consumed = 0, err = 0;
    bool eof = false, stop = false;
    std::thread reader;
    try {
        reader = std::thread([&] {
            uint64_t off = offset;
            for (;;) {
                int slot;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || produced - consumed < kRing; });
                    if (stop) return;
                    slot = produced % kRing;
                }
                ssize_t r;
                do r = pread(fd, ring.data() + (size_t)slot * kBuf, kBuf, (off_t)off); while (r < 0 && errno == EINTR);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (r < 0) err = errno;
                    else if (r == 0) eof = true;
                    else { got[slot] = r; ++produced; off += (uint64_t)r; }
                }
                cv.notify_all();
                if (r <= 0) return;
            }
        });
    } catch (...) {
        return -1; // no thread to be had: the caller reads by itself
    }
    uint64_t off = offset;
    for (;;) {
        int slot;
        ssize_t n;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return produced > consumed || eof || err; });
            if (produced == consumed) break; // EOF or error, and nothing left to hash
            slot = consumed % kRing;
            n = got[slot];
        }
        host_sha512_update(s, ring.data() + (size_t)slot * kBuf, (size_t)n);
        off += (uint64_t)n;
        {
            std::lock_guard<std::mutex> lk(mu);
            ++consumed;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    reader.join();
    *end_off = off;
    return err;
}

} // namespace

int host_sha512_file_from(HostSha& s, const char* path, uint64_t offset, uint64_t expect_len, uint8_t out[64], bool read_ahead)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return errno;
    uint64_t off = offset;
    int err = 0;
    bool done = false;
    if (read_ahead && expect_len >= offset && expect_len - offset >= (32u << 20)) { // long enough to repay a thread, and a core to spare
        const int rc = hash_file_read_ahead(s, fd, offset, &off);
        if (rc >= 0) { err = rc; done = true; }
    }
    if (!done) {
        // one read buffer per thread, kept: a fresh megabyte per file was a memset and 256 page faults in front of every small
        // file; 256 KiB stays in the core's L2 between the copy out of the page cache and the hashing
        constexpr size_t kBuf = 256u << 10;
        static thread_local std::vector<uint8_t> buf;
        if (buf.size() != kBuf) buf.resize(kBuf);
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
    }
    close(fd);
    if (err) return err;
    if (off != expect_len) return EIO; // the file changed size under the pass: the record's size would disagree
    host_sha512_final(s, out);
    return 0;
}

} // namespace snaphash
