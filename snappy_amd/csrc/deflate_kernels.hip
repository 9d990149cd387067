// deflate_kernels.hip -- block-parallel DEFLATE for gfx950 (MI355X): the GPU side of the
// data.tar.gz producer (reference clickdeb/deb.go:261-344, which pipes
// the tar stream through compress/gzip at level 9).
//
// Format-compatible, not byte-identical: archive-sha512 is taken over whatever bytes are
// produced (snappy/build.go:222), so the contract is RFC 1951/1952 validity -- any inflater
// must return exactly the tar stream -- not the bytes Go's compressor would emit.
//
// Parallel axis: the stream is cut into 16 KiB chunks; one wave64 compresses one chunk on its
// own (hash table in LDS, seeded with the previous chunk so that matches reach 32 KiB back) into a fixed-Huffman block, ends
// it with an empty stored block so that the chunk's output is byte aligned (what zlib's
// Z_SYNC_FLUSH does), and the chunk outputs are concatenated by a second kernel.  A chunk
// that does not shrink is emitted as a stored block.  Per tile of 64 input positions
// (lane = position): hash 4 bytes, look the candidate up, extend the match, a scalar greedy
// parse with one-byte lazy evaluation over the wave's match mask, then every token-start lane encodes its own token and a
// prefix sum of the bit lengths places it in the LDS bit buffer.
//
// This kernel is integer/LDS work with data-dependent control flow: no MFMA.  Bound: each input
// byte is read ~2x (position + candidate, L2-resident within a chunk) and <= 1.13 B written per
// byte; the measured rate (profiles/) is far below HBM -- the parse loop, not memory, bounds it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deflate_core.h"
#include "deflate_kernels.h"

namespace snaphash {

namespace {

constexpr uint32_t kHashBits = 12;
constexpr uint32_t kTab = 1u << kHashBits;

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

__device__ __forceinline__ uint32_t load32(const uint8_t* p) { return *reinterpret_cast<const u32_unaligned*>(p); }

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

} // namespace

// One wave per chunk, four waves (chunks) per workgroup.  in: the stream (readable up to n_in + 8);
// slots: nchunks * kDeflateSlot bytes; sizes[c]: bytes chunk c produced.
__global__ __launch_bounds__(256) void deflate_chunks_kernel(const uint8_t* __restrict__ in, uint64_t n_in,
                                                             uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                             uint32_t nchunks)
{
    __shared__ uint16_t s_tab[4][kTab];
    __shared__ uint32_t s_out[4][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * 4u + wave;
    if (c >= nchunks) return; // whole wave
    uint16_t* tab = s_tab[wave];
    uint32_t* ob = s_out[wave];
    const uint64_t base = (uint64_t)c * kDeflateChunk;
    const uint32_t len = (uint32_t)((n_in - base < kDeflateChunk) ? (n_in - base) : kDeflateChunk);
    const uint8_t* src = in + base;
    uint8_t* dst = slots + (uint64_t)c * kDeflateSlot;
    uint32_t* dstw = reinterpret_cast<uint32_t*>(dst);

    for (uint32_t i = lane; i < kTab / 2u; i += 64u) reinterpret_cast<uint32_t*>(tab)[i] = 0u;
    ob[lane] = (lane == 0u) ? 2u : 0u; // BFINAL=0, BTYPE=01 (fixed Huffman): bits 0,1,0 LSB first
    __builtin_amdgcn_wave_barrier();
    // Window: the table is seeded with the previous chunk's positions (when this launch holds it), so a match may
    // reach up to 32 KiB back across the chunk boundary -- the inflater does not care about block boundaries.
    // Table entries are position + kDeflateChunk + 1 (0 = empty): previous-chunk positions are 1 .. kDeflateChunk.
    if (c > 0u) {
        const uint8_t* prev = src - kDeflateChunk;
        for (uint32_t p = lane; p < kDeflateChunk; p += 64u) {
            const uint32_t w = load32(prev + p);
            tab[(w * 0x9E3779B1u) >> (32u - kHashBits)] = (uint16_t)(p + 1u);
        }
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t bitpos = 3u;      // bits of this chunk's stream so far (wave-uniform)
    uint32_t flushed = 0u;     // 32-bit words already stored to dst
    uint32_t skip_until = 0u;  // first position not covered by an earlier match

    for (uint32_t p0 = 0; p0 < len; p0 += 64u) {
        const uint32_t pos = p0 + lane;
        const bool valid = pos < len;
        const bool canmatch = pos + 4u <= len;
        uint32_t w = 0;
        if (valid) w = load32(src + pos); // reads at most 3 bytes past the chunk: inside the padded input
        const uint32_t h = (w * 0x9E3779B1u) >> (32u - kHashBits);
        uint32_t cand = 0;
        if (canmatch) cand = tab[h];
        __builtin_amdgcn_wave_barrier();
        if (canmatch) tab[h] = (uint16_t)(pos + kDeflateChunk + 1u); // lanes are served in order: the highest position stays
        __builtin_amdgcn_wave_barrier();
        uint32_t mlen = 0, dist = 0;
        if (canmatch && cand != 0u) {
            const int32_t cp = (int32_t)cand - 1 - (int32_t)kDeflateChunk; // an earlier tile or the previous chunk (< 0)
            const uint32_t maxl = (len - pos < 258u) ? len - pos : 258u;
            uint32_t l = 0;
            while (l < maxl) {
                const uint32_t x = load32(src + pos + l) ^ load32(src + cp + (int32_t)l);
                if (x) { l += (uint32_t)__builtin_ctz(x) >> 3; break; }
                l += 4u;
            }
            if (l > maxl) l = maxl;
            if (l >= 4u) { mlen = l; dist = (uint32_t)((int32_t)pos - cp); }
        }
        // greedy parse, sequential semantics, on the scalar unit: literals up to the next position that
        // has a match, take the match, jump past it
        const uint64_t mm_all = __ballot(mlen >= 4u);
        const uint32_t tile_n = (len - p0 < 64u) ? len - p0 : 64u;
        uint64_t start_mask = 0, match_mask = 0;
        uint32_t rel = (skip_until > p0) ? skip_until - p0 : 0u;
        while (rel < tile_n) {
            const uint64_t mm = mm_all & (~0ull << rel);
            if (mm == 0ull) {
                start_mask |= (~0ull << rel) & ((tile_n == 64u) ? ~0ull : ((1ull << tile_n) - 1ull));
                rel = tile_n;
                break;
            }
            const uint32_t f = (uint32_t)__builtin_ctzll(mm);
            if (f > rel) start_mask |= (~0ull << rel) & ((1ull << f) - 1ull);
            start_mask |= 1ull << f;
            const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)f);
            // lazy evaluation (as zlib from level 4 up): a longer match one byte later wins, this byte goes out as a literal
            if (f + 1u < tile_n && (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)(f + 1u)) > L) {
                rel = f + 1u;
                continue;
            }
            match_mask |= 1ull << f;
            rel = f + L;
        }
        skip_until = p0 + rel;
        const bool my_start = (start_mask >> lane) & 1ull;
        const bool my_match = (match_mask >> lane) & 1ull;
        uint32_t bits = 0, nb = 0;
        if (my_start) {
            if (my_match) enc_match(mlen, dist, bits, nb);
            else enc_literal(w & 0xffu, bits, nb);
        }
        const uint32_t incl = wave_scan_incl(nb, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (nb) {
            const uint32_t at = bitpos + incl - nb;
            const uint32_t widx = (at >> 5) - flushed, sh = at & 31u;
            const uint64_t v = (uint64_t)bits << sh;
            atomicOr(&ob[widx], (uint32_t)v);
            if (v >> 32) atomicOr(&ob[widx + 1u], (uint32_t)(v >> 32));
        }
        bitpos += total;
        __builtin_amdgcn_wave_barrier();
        const uint32_t done = (bitpos >> 5) - flushed; // complete words in the buffer (< 32)
        const uint32_t mine = ob[lane];
        const uint32_t carry = ob[done];               // the partial word (uniform address: broadcast)
        __builtin_amdgcn_wave_barrier();
        if (lane < done) dstw[flushed + lane] = mine;
        ob[lane] = (lane == 0u) ? carry : 0u;
        __builtin_amdgcn_wave_barrier();
        flushed += done;
    }

    // end of block (code 256: seven 0 bits), then an empty stored block: 3 header bits, pad to a byte,
    // LEN=0, NLEN=0xFFFF -- the chunk's output ends on a byte boundary
    bitpos += 7u + 3u;
    bitpos = (bitpos + 7u) & ~7u;
    if (lane == 0u) {
        const uint32_t widx = (bitpos >> 5) - flushed, sh = bitpos & 31u;
        const uint64_t v = 0xFFFF0000ull << sh;
        atomicOr(&ob[widx], (uint32_t)v);
        if (v >> 32) atomicOr(&ob[widx + 1u], (uint32_t)(v >> 32));
    }
    bitpos += 32u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t nbytes = bitpos >> 3;
    const uint32_t words = ((bitpos + 31u) >> 5) - flushed;
    if (lane < words) dstw[flushed + lane] = ob[lane];

    if (nbytes >= len + 5u) { // did not shrink: one stored block (BFINAL=0, BTYPE=00 in a whole byte; LEN; ~LEN; the bytes)
        __builtin_amdgcn_wave_barrier();
        if (lane == 0u) {
            dst[0] = 0u;
            dst[1] = (uint8_t)len; dst[2] = (uint8_t)(len >> 8);
            dst[3] = (uint8_t)~len; dst[4] = (uint8_t)(~len >> 8);
        }
        const uint32_t nw = len >> 2;
        for (uint32_t i = lane; i < nw; i += 64u)
            *reinterpret_cast<u32_unaligned*>(dst + 5u + 4u * i) = *reinterpret_cast<const uint32_t*>(src + 4u * i);
        for (uint32_t i = (nw << 2) + lane; i < len; i += 64u) dst[5u + i] = src[i];
        if (lane == 0u) sizes[c] = len + 5u;
    } else if (lane == 0u) {
        sizes[c] = nbytes;
    }
}

// Concatenates the chunk outputs: chunk c's sizes[c] bytes go to out + prefix[c].
__global__ __launch_bounds__(256) void deflate_compact_kernel(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ sizes,
                                                              const uint64_t* __restrict__ prefix, uint8_t* __restrict__ out,
                                                              uint32_t nchunks)
{
    const uint32_t c = blockIdx.x;
    if (c >= nchunks) return;
    const uint8_t* src = slots + (uint64_t)c * kDeflateSlot;
    uint8_t* dst = out + prefix[c];
    const uint32_t n = sizes[c];
    const uint32_t nw = n >> 2;
    for (uint32_t i = threadIdx.x; i < nw; i += 256u)
        *reinterpret_cast<u32_unaligned*>(dst + 4u * i) = *reinterpret_cast<const uint32_t*>(src + 4u * i);
    for (uint32_t i = (nw << 2) + threadIdx.x; i < n; i += 256u) dst[i] = src[i];
}

hipError_t launch_deflate_chunks(const uint8_t* d_in, uint64_t n_in, uint8_t* d_slots, uint32_t* d_sizes, uint32_t nchunks,
                                 hipStream_t s)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(deflate_chunks_kernel, dim3((nchunks + 3u) / 4u), dim3(256), 0, s, d_in, n_in, d_slots, d_sizes, nchunks);
    return hipGetLastError();
}

hipError_t launch_deflate_compact(const uint8_t* d_slots, const uint32_t* d_sizes, const uint64_t* d_prefix, uint8_t* d_out,
                                  uint32_t nchunks, hipStream_t s)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(deflate_compact_kernel, dim3(nchunks), dim3(256), 0, s, d_slots, d_sizes, d_prefix, d_out, nchunks);
    return hipGetLastError();
}

} // namespace snaphash
