// Test-only: compiles snappy_amd/csrc/sha512_core.h as host C++ and walks one
// stream through it exactly as a kernel lane does (full blocks, then the padded
// tail via apply_padding with garbage past the message end, optional segment
// split with chaining-state carry).  Lets the CPU suite check the block/padding
// logic the HIP kernels share; it is not part of the product.
#include <string.h>

#include "../snappy_amd/csrc/sha512_core.h"

using namespace snaphash;

static void load_block(uint64_t w[16], const uint8_t* p, uint32_t valid)
{
    uint8_t tmp[128];
    memset(tmp, 0xAA, sizeof tmp); // what a lane may see past the end of its file
    memcpy(tmp, p, valid);
    for (int k = 0; k < 16; ++k) {
        uint32_t d0, d1;
        memcpy(&d0, tmp + 8 * k, 4);
        memcpy(&d1, tmp + 8 * k + 4, 4);
        w[k] = be64(d0, d1);
    }
}

static void run_segment(uint64_t H[8], const uint8_t* data, uint64_t nbytes, uint64_t total_prev, bool fin)
{
    const uint32_t nfull = (uint32_t)(nbytes >> 7), rem = (uint32_t)(nbytes & 127);
    const uint32_t nblk = padded_blocks(nbytes, fin);
    for (uint32_t b = 0; b < nblk + 1; ++b) { // one extra dead iteration: live == false must not change H
        uint64_t w[16];
        if (b < nfull) load_block(w, data + 128ull * b, 128);
        else if (b == nfull) load_block(w, data + 128ull * b, rem);
        else load_block(w, data, 0);
        apply_padding(w, b >= nfull, b - nfull, rem, total_prev + nbytes);
        compress_block(H, w, b < nblk, K512);
    }
}

extern "C" void core_sha512(const uint8_t* data, uint64_t len, uint64_t split, uint8_t out[64])
{
    uint64_t H[8];
    for (int k = 0; k < 8; ++k) H[k] = IV512[k];
    if (split && split < len && (split & 127) == 0) {
        run_segment(H, data, split, 0, false);
        run_segment(H, data + split, len - split, split, true);
    } else {
        run_segment(H, data, len, 0, true);
    }
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(H[i] >> (56 - 8 * b));
}

// ---- the library's host SHA-512 (hostsha.cpp: hybrid scheduling only) -----------------------
#include "../snappy_amd/csrc/hostsha.cpp"

// one-shot, optionally resumed at `split` (a multiple of 128) from the state after the prefix
extern "C" void hostsha_buffer(const uint8_t* data, uint64_t n, uint64_t split, uint8_t* out)
{
    HostSha a;
    host_sha512_init(a);
    if (split && split <= n && (split & 127) == 0) {
        host_sha512_update(a, data, split);
        HostSha b;
        host_sha512_resume(b, a.H, split); // what a hand-over from the GPU looks like
        for (uint64_t off = split; off < n;) { // ragged update sizes
            const uint64_t step = (off * 7 + 13) % 300 + 1;
            const uint64_t k = step < n - off ? step : n - off;
            host_sha512_update(b, data + off, k);
            off += k;
        }
        host_sha512_final(b, out);
        return;
    }
    host_sha512_update(a, data, n);
    host_sha512_final(a, out);
}
// chaining value after nblocks blocks through spelling v of the block function (portable, AVX2, AVX-512VL)
extern "C" int hostsha_variants() { return host_sha512_variants(); }
extern "C" void hostsha_blocks_variant(int v, const uint8_t* data, uint64_t nblocks, uint64_t* H_out)
{
    for (int k = 0; k < 8; ++k) H_out[k] = IV512[k];
    host_sha512_blocks_variant(v, H_out, data, (size_t)nblocks);
}
extern "C" int hostsha_file(const char* path, uint64_t expect, uint8_t* out)
{
    HostSha a;
    host_sha512_init(a);
    return host_sha512_file_from(a, path, 0, expect, out, true);
}

// ---- eight streams side by side on one core (hostsha_x8.cpp), as a thread of the host pool runs them --------------
#include "../snappy_amd/csrc/hostsha_x8.cpp"

extern "C" int hostsha_x8_available() { return host_sha512_x8_available() ? 1 : 0; }
// streams i with ptrs[i] != NULL are memory, the others files (paths[i]); alone_from: streams at least that long keep the
// core to themselves (0 = none); -> 0 or an errno, *err_id = the failing stream
extern "C" int hostsha_many(const uint8_t* const* ptrs, const char* const* paths, const uint64_t* lens, uint64_t n, unsigned lanes,
                            uint64_t alone_from, uint8_t* digests, int64_t* err_id)
{
    uint64_t nexti = 0;
    return host_sha512_many(
        lanes, [&]() -> int64_t { return nexti < n ? (int64_t)nexti++ : -1; },
        [&](int64_t id) {
            HostStream h;
            h.mem = ptrs[id];
            h.path = ptrs[id] ? nullptr : paths[id];
            h.len = lens[id];
            h.digest = digests + 64 * id;
            h.alone = alone_from != 0 && lens[id] >= alone_from;
            return h;
        },
        err_id);
}
