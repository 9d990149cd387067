// hostsha.h -- streaming SHA-512 on a host core for the opt-in hybrid scheduler
// (see hostsha.cpp).  Internal; not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>

namespace snaphash {

struct HostSha {
    uint64_t H[8];
    uint64_t total; // bytes absorbed so far (including any GPU-hashed prefix)
    uint8_t tail[128];
    uint32_t ntail;
};

void host_sha512_init(HostSha& s);
// Continue a stream whose first total_prev bytes (a multiple of 128) were hashed elsewhere.
void host_sha512_resume(HostSha& s, const uint64_t H[8], uint64_t total_prev);
void host_sha512_update(HostSha& s, const uint8_t* p, size_t n);
void host_sha512_final(HostSha& s, uint8_t out[64]);
// Reads path from `offset` to EOF into s, checks that EOF is at expect_len, finalises.
// Returns 0 or an errno.
// read_ahead: a second thread reads two buffers ahead of the hasher (streams of 32 MiB or more; for a caller that has a
// core to spare: the lone archive beside its tree, not a pool that already keeps every core busy)
int host_sha512_file_from(HostSha& s, const char* path, uint64_t offset, uint64_t expect_len, uint8_t out[64], bool read_ahead = false);

// The block function exists in several spellings (portable, AVX2 schedule, AVX-512VL schedule), picked once by
// CPU features; tests run every one the CPU supports: variant v in [0, host_sha512_variants()), unsupported -> portable.
int host_sha512_variants();
void host_sha512_blocks_variant(int v, uint64_t H[8], const uint8_t* p, size_t nblocks);

// ---- eight streams at once on one core (hostsha_x8.cpp; AVX-512F + BW) --------------------------------------------------
bool host_sha512_x8_available();
// S[w * 8 + lane]: word w of lane's chaining value; data[lane]: nblocks x 128 bytes of that lane's stream
void host_sha512_x8_blocks(uint64_t S[64], const uint8_t* const data[8], size_t nblocks);
struct HostStream {
    const uint8_t* mem = nullptr; // caller memory, or
    const char* path = nullptr;   // a file (read to EOF, which must be at len)
    uint64_t len = 0;
    uint8_t* digest = nullptr;    // 64 bytes out
    bool read_ahead = false;      // a long file hashed alone may take a reader thread (host_sha512_file_from)
    bool alone = false;           // a stream that sets the host part's makespan by itself keeps a core to itself (lanes share one)
};
// One thread's share of a host part: streams come from next() (-1: none left) and get(id); up to `lanes` (<= 8) of them
// are in flight at once, a stream per 64-bit lane; with fewer than three to run side by side -- or without AVX-512 --
// one at a time as before.  Returns 0 or the errno of the first failure (*err_id: its stream).
int host_sha512_many(unsigned lanes, const std::function<int64_t()>& next, const std::function<HostStream(int64_t)>& get, int64_t* err_id);

} // namespace snaphash
