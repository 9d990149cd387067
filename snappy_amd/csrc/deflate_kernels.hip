// deflate_kernels.hip -- block-parallel DEFLATE for gfx950 (MI355X): the GPU side of the
// data.tar.gz producer (reference clickdeb/deb.go:261-344, which pipes
// the tar stream through compress/gzip at level 9).
//
// Format-compatible, not byte-identical: archive-sha512 is taken over whatever bytes are
// produced (snappy/build.go:222), so the contract is RFC 1951/1952 validity -- any inflater
// must return exactly the tar stream -- not the bytes Go's compressor would emit.
//
// Parallel axis: the stream is cut into 16 KiB chunks; one wave64 compresses one chunk on its
// own (four-way hash buckets in LDS, seeded with the previous chunk so that matches reach 32 KiB back) into a
// dynamic- or fixed-Huffman block (parse and count, build the codes, emit the remembered tokens), ends
// it with an empty stored block so that the chunk's output is byte aligned (what zlib's
// Z_SYNC_FLUSH does), and the chunk outputs are concatenated by a second kernel.  A chunk
// that does not shrink is emitted as a stored block.  Per tile of 64 input positions
// (lane = position): hash 4 bytes, look up to four candidates up, extend the matches, a scalar greedy
// parse with one-byte lazy evaluation over the wave's match mask; in the second pass every token-start lane encodes
// its own token and a prefix sum of the bit lengths places it in the LDS bit buffer.
//
// This kernel is integer/LDS work with data-dependent control flow: no MFMA.  Algorithmic bytes: 1 read and
// <= 1.13 written per input byte; the token scratch adds 4 B out and back per byte.  The measured rate (profiles/)
// is far below HBM -- instruction issue and LDS latency bound it, and the pass it serves is bound elsewhere (the
// serial digest of the archive), see DESIGN.md sec. 9.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deflate_core.h"
#include "deflate_kernels.h"

namespace snaphash {

namespace {

constexpr uint32_t kHashBits = 11;
constexpr uint32_t kTab = 1u << kHashBits; // buckets of four candidates, newest first, in one 64-bit word
constexpr uint32_t kGroup = 16;            // positions that look the table up together, before any of them enters it

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
__device__ __forceinline__ uint32_t load32(const uint8_t* p) { return *reinterpret_cast<const u32_unaligned*>(p); }
__device__ __forceinline__ uint64_t load64(const uint8_t* p) { return *reinterpret_cast<const u64_unaligned*>(p); }

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

constexpr uint32_t kTokStart = 1u << 31, kTokMatch = 1u << 30;

// Per-wave LDS: hash table, bit buffer, symbol counts / codes, scratch of the code construction.
// The scratch of the code construction is live after the parse, when the table is spent: it lives inside the
// table's bytes.
struct HuffScratch {
    uint32_t w[2 * kNumLL];
    uint32_t cnt[260];
    uint16_t parent[2 * kNumLL];
    uint16_t order[kNumLL + 2];
    // the block header: run-length tokens of the code lengths, the code length code
    uint16_t rle[kNumLL + kNumD + 4];
    uint32_t clfreq[kNumCL + 1];
    uint32_t clcode[kNumCL + 1];
    uint8_t cllen[kNumCL + 1];
    DynHeader hdr;
    uint32_t hdr_tail; // the header's last, partial word
};
struct WaveLds {
    union {
        unsigned long long tab[kTab];
        HuffScratch hs;
    };
    uint32_t ob[64];
    uint32_t freq[320];  // literal/length symbols 0..285, distance symbols at 288..317; after the first pass the
                         // codes take the counts' place: code << 8 | length, same indexing
    uint8_t len[320];
};
static_assert(sizeof(HuffScratch) <= kTab * sizeof(unsigned long long), "scratch must fit in the table");

__device__ __forceinline__ void put_bits(uint32_t* ob, uint32_t at, uint32_t flushed, uint32_t bits)
{
    const uint32_t widx = (at >> 5) - flushed, sh = at & 31u;
    const uint64_t v = (uint64_t)bits << sh;
    atomicOr(&ob[widx], (uint32_t)v);
    if (v >> 32) atomicOr(&ob[widx + 1u], (uint32_t)(v >> 32));
}

// One wave per chunk, four waves (chunks) per workgroup.  in: the stream (readable up to n_in + 8);
// slots: nchunks * kDeflateSlot bytes; sizes[c]: bytes chunk c produced; toks: one scratch word per input byte.
// Pass 1 parses the chunk, counts symbols (and prices a fixed-Huffman block) and writes its tokens to `toks`; one
// lane builds the dynamic codes and the block header (deflate_core.h: the host model runs the same routines);
// pass 2 re-reads the tokens (each lane its own) and emits a dynamic or a fixed block, whichever is smaller; a
// chunk that does not shrink is stored.
__global__ __launch_bounds__(256) void deflate_chunks_kernel(const uint8_t* __restrict__ in, uint64_t n_in,
                                                             uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                             uint32_t* __restrict__ toks, uint32_t nchunks)
{
    __shared__ WaveLds s_lds[4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * 4u + wave;
    if (c >= nchunks) return; // whole wave
    WaveLds& L = s_lds[wave];
    unsigned long long* tab = L.tab;
    uint32_t* ob = L.ob;
    const uint64_t base = (uint64_t)c * kDeflateChunk;
    const uint32_t len = (uint32_t)((n_in - base < kDeflateChunk) ? (n_in - base) : kDeflateChunk);
    const uint8_t* src = in + base;
    uint8_t* dst = slots + (uint64_t)c * kDeflateSlot;
    uint32_t* dstw = reinterpret_cast<uint32_t*>(dst);
    uint32_t* tok = toks + base; // one word per position: 0 = inside a match, kTokStart | byte, or kTokMatch | length << 16 | distance

    // ---------------- pass 1: parse, count, remember the tokens ----------------
    for (uint32_t i = lane; i < 320u; i += 64u) L.freq[i] = 0u;
    for (uint32_t i = lane; i < kTab; i += 64u) tab[i] = 0ull;
    __builtin_amdgcn_wave_barrier();
    // Window: the table is seeded with the previous chunk's positions (when this launch holds it), so a match may
    // reach up to 32 KiB back across the chunk boundary -- the inflater does not care about block boundaries.
    // Entries are position + kDeflateChunk + 1 (0 = empty): previous-chunk positions are 1 .. kDeflateChunk.
    // A bucket is updated tile-wise: every lane reads its bucket, then offers (its entry, the three newest it read);
    // of the lanes that share a bucket the highest position wins (atomic max: the entry is the most significant field).
    if (c > 0u) {
        const uint8_t* prev = src - kDeflateChunk;
        for (uint32_t p = lane; p < kDeflateChunk; p += 64u) {
            const uint32_t h = (load32(prev + p) * 0x9E3779B1u) >> (32u - kHashBits);
            for (uint32_t g = 0; g < 64u / kGroup; ++g) {
                if (lane / kGroup == g) {
                    const unsigned long long old = tab[h];
                    atomicMax(&tab[h], ((unsigned long long)(p + 1u) << 48) | (old >> 16));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    uint32_t fixed_bits = 3u + 7u, extra_bits = 0; // cost of a fixed block / extra bits of the matches (wave-uniform)
    uint32_t skip_until = 0u;                      // first position not covered by an earlier match
    for (uint32_t p0 = 0; p0 < len; p0 += 64u) {
        const uint32_t pos = p0 + lane;
        const bool valid = pos < len;
        const bool canmatch = pos + 4u <= len;
        uint32_t w = 0;
        if (valid) w = load32(src + pos); // reads at most 3 bytes past the chunk: inside the padded input
        const uint32_t h = (w * 0x9E3779B1u) >> (32u - kHashBits);
        unsigned long long cand = 0;
        for (uint32_t g = 0; g < 64u / kGroup; ++g) { // a group sees what the groups before it in the tile have entered
            if (canmatch && lane / kGroup == g) {
                cand = tab[h];
                atomicMax(&tab[h], ((unsigned long long)(pos + kDeflateChunk + 1u) << 48) | (cand >> 16));
            }
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t mlen = 0, dist = 0;
        if (canmatch && pos >= skip_until) { // a position inside the match that reaches into this tile can start no token
            const uint32_t maxl = (len - pos < 258u) ? len - pos : 258u;
            for (uint32_t k = 0; k < 4u; ++k) { // newest first; the longest wins, ties stay with the nearer one
                const uint32_t e = (uint32_t)(cand >> (48u - 16u * k)) & 0xffffu;
                if (e == 0u || mlen >= maxl) break;
                const uint8_t* cs = src + ((int32_t)e - 1 - (int32_t)kDeflateChunk); // an earlier tile or the previous chunk
                // to beat mlen the candidate must agree in the bytes mlen-3 .. mlen (all inside the chunk: mlen < maxl)
                if (mlen >= 4u && load32(src + pos + mlen - 3u) != load32(cs + mlen - 3u)) continue;
                uint32_t l = 0;
                while (l < maxl) { // eight bytes a step: reads at most 7 bytes past the chunk's last byte
                    const uint64_t x = load64(src + pos + l) ^ load64(cs + l);
                    if (x) { l += (uint32_t)__builtin_ctzll(x) >> 3; break; }
                    l += 8u;
                }
                if (l > maxl) l = maxl;
                if (l >= 4u && l > mlen) { mlen = l; dist = (uint32_t)(src + pos - cs); }
            }
        }
        // greedy parse with one-byte lazy evaluation, sequential semantics, on the scalar unit
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
            const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)f);
            // lazy evaluation (as zlib from level 4 up): a longer match one byte later wins, this byte goes out as a literal
            if (f + 1u < tile_n && (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)(f + 1u)) > ml) {
                rel = f + 1u;
                continue;
            }
            match_mask |= 1ull << f;
            rel = f + ml;
        }
        skip_until = p0 + rel;
        const bool my_start = (start_mask >> lane) & 1ull;
        const bool my_match = (match_mask >> lane) & 1ull;
        uint32_t ls = w & 0xffu, le = 0, lv = 0, ds = 0, de = 0, dv = 0;
        if (my_match) { len_symbol(mlen, ls, le, lv); dist_symbol(dist, ds, de, dv); }
        uint32_t fb = 0, eb = 0, t = 0;
        if (my_start) {
            atomicAdd(&L.freq[ls], 1u);
            fb = fixed_ll_bits(ls);
            t = kTokStart | ls;
            if (my_match) {
                atomicAdd(&L.freq[288u + ds], 1u);
                fb += 5u;
                eb = le + de;
                t = kTokStart | kTokMatch | (mlen << 16) | dist;
            }
        }
        if (valid) tok[pos] = t;
        const uint32_t tf = wave_scan_incl(fb, lane), te = wave_scan_incl(eb, lane);
        fixed_bits += (uint32_t)__builtin_amdgcn_readlane((int)tf, 63);
        extra_bits += (uint32_t)__builtin_amdgcn_readlane((int)te, 63);
    }

    // ---------------- the codes and the header: one lane, sequential and deterministic (the host model runs the same routines) ----------------
    __builtin_amdgcn_wave_barrier();
    HuffScratch& S = L.hs;
    if (lane == 0u) {
        L.freq[256] += 1u; // end of block
        if (L.freq[288] == 0u) L.freq[288] = 1u; // at least two distance codes, as zlib sends
        if (L.freq[289] == 0u) L.freq[289] = 1u;
        huff_lengths(L.freq, kNumLL, (uint32_t)kMaxBits, L.len, S.w, S.parent, S.order, S.cnt);
        huff_lengths(L.freq + 288, kNumD, (uint32_t)kMaxBits, L.len + 288, S.w, S.parent, S.order, S.cnt);
        build_dyn_header(L.len, L.len + 288, S.rle, S.clfreq, S.cllen, S.clcode, S.w, S.parent, S.order, S.cnt, S.hdr);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t db = 0; // bits a dynamic block spends on the symbols
    for (uint32_t i = lane; i < 318u; i += 64u)
        if (i < (uint32_t)kNumLL || i >= 288u) db += L.freq[i] * (uint32_t)L.len[i];
    const uint32_t hdr_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.hdr.bits);
    const uint32_t dyn_bits = hdr_bits + extra_bits + (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(db, lane), 63);
    fixed_bits += extra_bits;
    const bool dynamic = dyn_bits < fixed_bits;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0u) {
        if (dynamic) {
            // the counts are spent: the codes go where they were
            huff_codes(L.len, kNumLL, L.freq, S.cnt);
            huff_codes(L.len + 288, kNumD, L.freq + 288, S.cnt);
            // the header goes straight to the slot, whole words; its partial last word opens the bit buffer
            uint64_t acc = 0;
            uint32_t nacc = 0, widx = 0;
            write_dyn_header(S.hdr, S.rle, S.cllen, S.clcode, [&](uint32_t bits, uint32_t nb) {
                acc |= (uint64_t)bits << nacc;
                nacc += nb;
                if (nacc >= 32u) { dstw[widx++] = (uint32_t)acc; acc >>= 32; nacc -= 32u; }
            });
            S.hdr_tail = (uint32_t)acc;
        } else {
            S.hdr_tail = 2u; // BFINAL=0, BTYPE=01: bits 0,1,0 LSB first
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t bitpos = dynamic ? hdr_bits : 3u;
    uint32_t flushed = bitpos >> 5;
    ob[lane] = (lane == 0u) ? S.hdr_tail : 0u;
    __builtin_amdgcn_wave_barrier();

    // ---------------- pass 2: emit the remembered tokens ----------------
    bool overflow = false;
    for (uint32_t p0 = 0; p0 < len; p0 += 64u) {
        const uint32_t pos = p0 + lane;
        const uint32_t t = (pos < len) ? tok[pos] : 0u;
        // part A = literal/length code + its extra bits, part B = distance code + its extra bits (each <= 32 bits)
        uint32_t ba = 0, na = 0, bb = 0, nbb = 0;
        if (t & kTokStart) {
            if (dynamic) {
                uint32_t ls = t & 0x1ffu, le = 0, lv = 0, ds = 0, de = 0, dv = 0;
                if (t & kTokMatch) { len_symbol((t >> 16) & 0x1ffu, ls, le, lv); dist_symbol(t & 0xffffu, ds, de, dv); }
                const uint32_t ca = L.freq[ls];
                ba = (ca >> 8) | (lv << (ca & 0xffu));
                na = (ca & 0xffu) + le;
                if (t & kTokMatch) {
                    const uint32_t cb = L.freq[288u + ds];
                    bb = (cb >> 8) | (dv << (cb & 0xffu));
                    nbb = (cb & 0xffu) + de;
                }
            } else if (t & kTokMatch) {
                enc_match((t >> 16) & 0x1ffu, t & 0xffffu, ba, na);
            } else {
                enc_literal(t & 0xffu, ba, na);
            }
        }
        const uint32_t nb = na + nbb;
        const uint32_t incl = wave_scan_incl(nb, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (nb) {
            const uint32_t at = bitpos + incl - nb;
            put_bits(ob, at, flushed, ba);
            if (nbb) put_bits(ob, at + na, flushed, bb);
        }
        bitpos += total;
        __builtin_amdgcn_wave_barrier();
        const uint32_t done = (bitpos >> 5) - flushed; // complete words in the buffer (< 34)
        const uint32_t mine = ob[lane];
        const uint32_t carry = ob[done];               // the partial word (uniform address: broadcast)
        __builtin_amdgcn_wave_barrier();
        if ((flushed + done) * 4u + 64u > kDeflateSlot) overflow = true; // only a chunk that ends up stored comes here
        if (lane < done && !overflow) dstw[flushed + lane] = mine;
        ob[lane] = (lane == 0u) ? carry : 0u;
        __builtin_amdgcn_wave_barrier();
        flushed += done;
    }

    // end of block (fixed: seven 0 bits; dynamic: the code of symbol 256), then an empty stored block: 3 header
    // bits, pad to a byte, LEN=0, NLEN=0xFFFF -- the chunk's output ends on a byte boundary
    if (dynamic) {
        const uint32_t ce = L.freq[256];
        if (lane == 0u) put_bits(ob, bitpos, flushed, ce >> 8);
        bitpos += ce & 0xffu;
    } else {
        bitpos += 7u;
    }
    bitpos += 3u;
    bitpos = (bitpos + 7u) & ~7u;
    if (lane == 0u) {
        put_bits(ob, bitpos, flushed, 0xFFFF0000u);
    }
    bitpos += 32u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t nbytes = bitpos >> 3;
    const uint32_t words = ((bitpos + 31u) >> 5) - flushed;
    if (lane < words && !overflow) dstw[flushed + lane] = ob[lane];

    if (overflow || nbytes >= len + 5u) { // did not shrink: one stored block (BFINAL=0, BTYPE=00 in a whole byte; LEN; ~LEN; the bytes)
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

hipError_t launch_deflate_chunks(const uint8_t* d_in, uint64_t n_in, uint8_t* d_slots, uint32_t* d_sizes, uint32_t* d_toks,
                                 uint32_t nchunks, hipStream_t s)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(deflate_chunks_kernel, dim3((nchunks + 3u) / 4u), dim3(256), 0, s, d_in, n_in, d_slots, d_sizes, d_toks,
                       nchunks);
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
