// hostsha.h -- streaming SHA-512 on a host core for the opt-in hybrid scheduler
// (see hostsha.cpp).  Internal; not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

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

} // namespace snaphash
