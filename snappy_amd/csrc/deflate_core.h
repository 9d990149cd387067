// deflate_core.h -- the token encoder of the block-parallel DEFLATE kernel (fixed Huffman codes,
// RFC 1951 sec. 3.2.5-3.2.6), shared with a host harness in tests/ so that the code tables and the
// chunk framing are checked against zlib on the CPU before the kernel runs on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DF_HD __host__ __device__ __forceinline__
#define DF_HD_CALL __host__ __device__ __noinline__ // the sequential code construction: a call, so that its registers
                                                    // do not widen the kernel's parse loop
#else
#define DF_HD inline
#define DF_HD_CALL inline
#endif

namespace snaphash {

// Huffman codes are packed starting from their most significant bit, everything else LSB first.
DF_HD uint32_t rev_bits(uint32_t code, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(code) >> (32u - n);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) r |= ((code >> i) & 1u) << (n - 1u - i);
    return r;
#endif
}
DF_HD uint32_t floor_log2(uint32_t v) // v > 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 31u - (uint32_t)__clz((int)v);
#else
    return 31u - (uint32_t)__builtin_clz(v);
#endif
}

// LSB-first bit string of one literal
DF_HD void enc_literal(uint32_t lit, uint32_t& bits, uint32_t& n)
{
    if (lit < 144u) { bits = rev_bits(0x30u + lit, 8); n = 8; }
    else { bits = rev_bits(0x190u + (lit - 144u), 9); n = 9; }
}

// length 3..258, distance 1..32768 -> at most 31 bits
DF_HD void enc_match(uint32_t len, uint32_t dist, uint32_t& bits, uint32_t& n)
{
    uint32_t l = len - 3u, le = 0, lcode;
    if (len == 258u) lcode = 285u;
    else if (l < 8u) lcode = 257u + l;
    else {
        le = floor_log2(l) - 2u;
        lcode = 257u + 4u * (le + 1u) + ((l >> le) & 3u);
    }
    uint32_t b, nl;
    if (lcode < 280u) { b = rev_bits(lcode - 256u, 7); nl = 7; }
    else { b = rev_bits(0xC0u + (lcode - 280u), 8); nl = 8; }
    b |= (l & ((1u << le) - 1u)) << nl;
    nl += le;
    const uint32_t d = dist - 1u;
    uint32_t de = 0, dcode;
    if (d < 4u) dcode = d;
    else {
        de = floor_log2(d) - 1u;
        dcode = 2u * de + 2u + ((d >> de) & 1u);
    }
    b |= rev_bits(dcode, 5) << nl;
    nl += 5u;
    b |= (d & ((1u << de) - 1u)) << nl;
    nl += de;
    bits = b;
    n = nl;
}

} // namespace snaphash

// ---- dynamic Huffman blocks (RFC 1951 sec. 3.2.7) -----------------------------------------------------------
// Per chunk the kernel parses once, counting symbols and remembering its tokens, then ONE lane builds the codes and
// the block header with the sequential, deterministic routines below (the host model in tests/ runs the very same
// code), and a second pass emits the tokens with them.  Code lengths come from a plain two-queue Huffman
// construction over counting-sorted frequencies (symbols that do not occur get no code); if the tree is deeper than
// the limit the frequencies are halved and it is rebuilt.  The two sequences of code lengths are run-length coded
// and Huffman coded themselves, as the format provides (symbols 16, 17, 18).

namespace snaphash {

// ---- the parse (round 3: hash chains, one workgroup per chunk) -- shared by the kernel and its CPU model ---------
// A chunk is one DEFLATE block; inside it every position's candidates come from hash chains over the chunk and the
// kDfMaxDist bytes in front of it (3-byte hash, 16-bit distance links in a ring), the chain is walked at most
// kDfDepth links starting at the position's own link.  The parse then picks, segment by segment (kDfSeg positions),
// the cheapest way through the segment that those matches allow (a shortest path over the positions: a literal, or
// the first L bytes of the position's match for every L from 3 to 62), at prices taken from the symbol counts of the
// chunk so far -- see "the price parse" below.
constexpr uint32_t kDfChunk = 65536;    // input bytes per chunk (one workgroup, one DEFLATE block)
constexpr uint32_t kDfSeg = 1920;       // positions indexed / searched / parsed per pipeline step: 30 tiles, two for each of 15 searching waves
constexpr uint32_t kDfRing = 32768;     // entries of the link ring (>= kDfMaxDist + kDfSeg)
constexpr uint32_t kDfHashBits = 12;
constexpr uint32_t kDfMaxDist = 28800;  // farthest match; also the bytes in front of the chunk that are indexed (15 segments)
constexpr uint32_t kDfMinMatch = 3;
constexpr uint32_t kDfTooFar = 4096;    // a 3-byte match farther than this costs more than its literals (zlib's TOO_FAR)
#if !defined(SNAPHASH_DF_DEPTH) // (tuning builds override the two search parameters; the CPU model follows)
#define SNAPHASH_DF_DEPTH 96 // (round 5: 32 before; the bytes of the reference's gzip level 9 -- text 0.2392-0.2397 against zlib -9's 0.2394-0.2397 -- for 1.1 x what depth 32 cost before the search was counted out instruction by instruction, inside a fused pass that got faster: DESIGN.md sec. 9)
#define SNAPHASH_DF_GOOD 32
#endif
constexpr uint32_t kDfDepth = SNAPHASH_DF_DEPTH; // links walked per position
constexpr uint32_t kDfGood = SNAPHASH_DF_GOOD;   // a match this long cuts what is left of the walk to a quarter (zlib's good_length idea)
constexpr uint32_t kDfNice = 128;       // a match this long ends the walk
static_assert(kDfRing >= kDfMaxDist + kDfSeg, "a segment is indexed whole before it is searched");
static_assert(kDfMaxDist % kDfSeg == 0 && kDfSeg % 64 == 0 && kDfSeg % 16 == 0, "whole window segments, whole tiles (a chunk's last segment may be short)");

// ---- the price parse ---------------------------------------------------------------------------------------------
// Prices are in quarter bits.  Before a chunk has kDfPriceWarm tokens they are fixed (literal 6 bits, length symbol 7,
// distance symbol 5); afterwards a symbol seen f times among n costs log2(n + 1) - log2(f + 1), log2 linear between
// the powers of two, at least one bit and at most 14 (12 for a distance symbol).  Extra bits cost what they are.
// A match of kDfLongMatch bytes or more is not weighed: the path goes through its position and takes all of it.
// Among equally cheap ways to a position the longest last token wins.  A segment is parsed as kDfParseWaves windows,
// one wave each, all at the segment's prices: no token leaves its window.  A window ends at the last position of the
// kDfCutSpan up to its nominal end that no match from in front of it reaches across (every way passes such a position
// anyhow: cutting there costs nothing), at the nominal end itself when there is none.
constexpr uint32_t kDfParseWaves = 4;
constexpr uint32_t kDfCutSpan = 64;
// where window w of a segment nominally begins (8 + 8 + 7 + 7 tiles of 64 positions; a short last segment ends them early)
DF_HD uint32_t df_window_begin(uint32_t w) { return w < 2u ? w * 512u : (w == 2u ? 1024u : (w == 3u ? 1472u : kDfSeg)); }
static_assert(kDfSeg == 1920, "the windows above are cut for this segment");
constexpr uint32_t kDfPriceUnit = 4;
constexpr uint32_t kDfPriceWarm = 64;
constexpr uint32_t kDfLongMatch = 63;
constexpr uint32_t kDfLitPrice0 = 6 * kDfPriceUnit, kDfLenPrice0 = 7 * kDfPriceUnit, kDfDistPrice0 = 5 * kDfPriceUnit;
constexpr uint32_t kDfLLCap = 14 * kDfPriceUnit, kDfDistCap = 12 * kDfPriceUnit;
DF_HD uint32_t df_ilog(uint32_t x) // x >= 1
{
    const uint32_t e = floor_log2(x);
    const uint32_t frac = e >= 2u ? (x >> (e - 2u)) & 3u : (x << (2u - e)) & 3u;
    return e * 4u + frac;
}
DF_HD uint32_t df_price(uint32_t f, uint32_t log_total, uint32_t cap)
{
    const uint32_t p = log_total - df_ilog(f + 1u); // f <= total: never negative
    return p < kDfPriceUnit ? kDfPriceUnit : (p > cap ? cap : p);
}
// the key of a way to a position: its price, and in the low byte 255 - the length of its last token (1 = a literal)
DF_HD uint32_t df_key(uint32_t price, uint32_t last) { return (price << 8) | (255u - last); }

// bytes a chunk of len input bytes takes as stored blocks: LEN is a 16-bit field, so a full 64 KiB chunk is two blocks
DF_HD uint32_t deflate_stored_size(uint32_t len) { return len + (len > 65535u ? 10u : 5u); }

DF_HD uint32_t df_hash(uint32_t w) { return ((w & 0xffffffu) * 0x9E3779B1u) >> (32u - kDfHashBits); }

constexpr int kNumLL = 286; // literal/length symbols
constexpr int kNumD = 30;   // distance symbols
constexpr int kNumCL = 19;  // symbols of the code length code
constexpr int kMaxBits = 15;
constexpr int kMaxCLBits = 7;

DF_HD void len_symbol(uint32_t len, uint32_t& sym, uint32_t& ebits, uint32_t& eval)
{
    const uint32_t l = len - 3u;
    ebits = 0;
    if (len == 258u) sym = 285u;
    else if (l < 8u) sym = 257u + l;
    else {
        ebits = floor_log2(l) - 2u;
        sym = 257u + 4u * (ebits + 1u) + ((l >> ebits) & 3u);
    }
    eval = l & ((1u << ebits) - 1u);
}

DF_HD void dist_symbol(uint32_t dist, uint32_t& sym, uint32_t& ebits, uint32_t& eval)
{
    const uint32_t d = dist - 1u;
    ebits = 0;
    if (d < 4u) sym = d;
    else {
        ebits = floor_log2(d) - 1u;
        sym = 2u * ebits + 2u + ((d >> ebits) & 1u);
    }
    eval = d & ((1u << ebits) - 1u);
}

// bits a fixed-Huffman block spends on one literal/length symbol (without extra bits)
DF_HD uint32_t fixed_ll_bits(uint32_t sym) { return sym < 144u ? 8u : (sym < 256u ? 9u : (sym < 280u ? 7u : 8u)); }

// Code lengths (1..max_bits; 0 for a symbol with freq 0) for n symbols.  Scratch: w[2n] u32, parent[2n] u16,
// order[n] u16, cnt[257] u32.  Deterministic: stable counting sort (ties by symbol index), ties between a leaf and
// an internal node go to the leaf.  A single used symbol gets length 1.
template <typename U32P, typename U16P>
DF_HD_CALL void huff_lengths(const U32P freq, int n, uint32_t max_bits, uint8_t* len, U32P w, U16P parent, U16P order, U32P cnt)
{
    for (uint32_t shift = 0;; ++shift) {
        // stable radix sort of the symbols by key = 0 (unused) or freq >> shift clamped to 1..0xffff, two 8-bit passes
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k <= 256; ++k) cnt[k] = 0;
            for (int i = 0; i < n; ++i) {
                uint32_t f = freq[i] >> shift;
                if (f == 0 && freq[i] != 0) f = 1;
                if (f > 0xffffu) f = 0xffffu;
                cnt[((f >> (8 * pass)) & 0xffu) + 1u]++;
            }
            for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
            // pass 0 reads symbols in index order into parent[] (used as a temporary), pass 1 reads that into order[]
            for (int i = 0; i < n; ++i) {
                const int sym = pass ? (int)parent[i] : i;
                uint32_t f = freq[sym] >> shift;
                if (f == 0 && freq[sym] != 0) f = 1;
                if (f > 0xffffu) f = 0xffffu;
                const uint32_t at = cnt[(f >> (8 * pass)) & 0xffu]++;
                if (pass) order[at] = (uint16_t)sym;
                else parent[at] = (uint16_t)sym;
            }
        }
        int z = 0; // unused symbols sort first
        while (z < n && freq[order[z]] == 0) { len[order[z]] = 0; ++z; }
        const int m = n - z;
        if (m == 0) return;
        if (m == 1) { len[order[z]] = 1; return; }
        // two-queue merge: leaves in sorted order (node id = position in `order`), internal nodes n .. n+m-2
        for (int i = z; i < n; ++i) {
            uint32_t f = freq[order[i]] >> shift;
            if (f > 0xffffu) f = 0xffffu;
            w[i] = f ? f : 1u;
        }
        int li = z, ii = n, nn = n; // heads of the leaf and internal queues, next internal node
        for (int k = 0; k < m - 1; ++k) {
            int a, b;
            if (li < n && (ii >= nn || w[li] <= w[ii])) a = li++; else a = ii++;
            if (li < n && (ii >= nn || w[li] <= w[ii])) b = li++; else b = ii++;
            w[nn] = w[a] + w[b];
            parent[a] = (uint16_t)nn;
            parent[b] = (uint16_t)nn;
            ++nn;
        }
        // depths, root first (internal nodes were created in increasing order, so a parent has the larger id)
        w[nn - 1] = 0;
        uint32_t deepest = 0;
        for (int node = nn - 2; node >= z; --node) {
            w[node] = w[parent[node]] + 1u;
            if (node < n && w[node] > deepest) deepest = w[node];
        }
        if (deepest <= max_bits) {
            for (int i = z; i < n; ++i) len[order[i]] = (uint8_t)w[i];
            return;
        }
    }
}

// Canonical codes from lengths, already bit-reversed for the LSB-first stream: out[i] = code << 8 | length.
// scratch: 2 * (kMaxBits + 2) u32.
template <typename U32P>
DF_HD_CALL void huff_codes(const uint8_t* len, int n, U32P out, U32P scratch)
{
    U32P count = scratch;
    U32P next = scratch + (kMaxBits + 2);
    for (int b = 0; b <= kMaxBits + 1; ++b) count[b] = 0;
    for (int i = 0; i < n; ++i) count[len[i]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int b = 1; b <= kMaxBits; ++b) {
        code = (code + count[b - 1]) << 1;
        next[b] = code;
    }
    for (int i = 0; i < n; ++i) {
        const uint32_t l = len[i];
        out[i] = l ? ((rev_bits(next[l]++, l) << 8) | l) : 0u;
    }
}

// The header of a dynamic block, as data: what write_dyn_header() sends.
struct DynHeader {
    uint32_t nll;   // literal/length code lengths sent (HLIT + 257)
    uint32_t nd;    // distance code lengths sent (HDIST + 1)
    uint32_t ncl;   // code length code lengths sent (HCLEN + 4)
    uint32_t ntok;  // run-length tokens
    uint32_t bits;  // whole header, the three block-type bits included
};

DF_HD uint32_t cl_order(uint32_t i) // RFC 1951 sec. 3.2.7: the order in which the code length code lengths are sent
{
    // 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
    if (i < 3u) return 16u + i;
    if (i == 3u) return 0u;
    const uint32_t k = i - 4u; // 8,7,9,6,10,5,...: 8 + (k+1)/2 alternating down/up
    return (k & 1u) ? 8u - ((k + 1u) >> 1) : 8u + (k >> 1);
}

// Run-length codes the code lengths of both alphabets (as ONE sequence, runs may cross from one into the other),
// builds the code length code.  tok[i] = symbol | extra value << 8.  clfreq/cl_len/cl_code: kNumCL entries.
// Scratch as huff_lengths (for n = kNumCL).
template <typename U32P, typename U16P>
DF_HD_CALL void dyn_header_tokens(const uint8_t* ll_len, const uint8_t* d_len, U16P tok, U32P clfreq, DynHeader& h)
{
    uint32_t nll = kNumLL, nd = kNumD;
    while (nll > 257u && ll_len[nll - 1u] == 0) --nll;
    while (nd > 1u && d_len[nd - 1u] == 0) --nd;
    for (int k = 0; k < kNumCL; ++k) clfreq[k] = 0;
    const uint32_t total = nll + nd;
    uint32_t nt = 0, i = 0;
    while (i < total) {
        const uint32_t v = i < nll ? ll_len[i] : d_len[i - nll];
        uint32_t run = 1;
        while (i + run < total && (i + run < nll ? ll_len[i + run] : d_len[i + run - nll]) == v) ++run;
        i += run;
        if (v == 0) {
            while (run >= 11u) { const uint32_t r = run < 138u ? run : 138u; tok[nt++] = (uint16_t)(18u | ((r - 11u) << 8)); clfreq[18]++; run -= r; }
            if (run >= 3u) { tok[nt++] = (uint16_t)(17u | ((run - 3u) << 8)); clfreq[17]++; run = 0; }
        } else {
            tok[nt++] = (uint16_t)v; clfreq[v]++; --run;
            while (run >= 3u) { const uint32_t r = run < 6u ? run : 6u; tok[nt++] = (uint16_t)(16u | ((r - 3u) << 8)); clfreq[16]++; run -= r; }
        }
        for (; run; --run) { tok[nt++] = (uint16_t)v; clfreq[v]++; }
    }
    // an inflater wants a complete code length code: at least two symbols
    uint32_t used = 0;
    for (int k = 0; k < kNumCL; ++k) used += clfreq[k] != 0;
    if (used < 2u) { if (clfreq[0] == 0) clfreq[0] = 1; else clfreq[1] = 1; }
    h.nll = nll; h.nd = nd; h.ntok = nt;
}
// bits one run-length token takes behind its code
DF_HD uint32_t cl_extra_bits(uint32_t sym) { return sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u; }

template <typename U32P, typename U16P>
DF_HD_CALL void build_dyn_header(const uint8_t* ll_len, const uint8_t* d_len, U16P tok, U32P clfreq, uint8_t* cl_len, U32P cl_code,
                                 U32P w, U16P parent, U16P order, U32P cnt, DynHeader& h)
{
    dyn_header_tokens(ll_len, d_len, tok, clfreq, h);
    huff_lengths(clfreq, kNumCL, (uint32_t)kMaxCLBits, cl_len, w, parent, order, cnt);
    huff_codes(cl_len, kNumCL, cl_code, cnt);
    uint32_t ncl = kNumCL;
    while (ncl > 4u && cl_len[cl_order(ncl - 1u)] == 0) --ncl;
    uint32_t bits = 3u + 5u + 5u + 4u + 3u * ncl;
    for (uint32_t t = 0; t < h.ntok; ++t) {
        const uint32_t sym = tok[t] & 0xffu;
        bits += cl_len[sym] + cl_extra_bits(sym);
    }
    h.ncl = ncl; h.bits = bits;
}

// Sends the header through sink(bits, nbits) (LSB first, nbits <= 16 per call).
template <typename U32P, typename U16P, typename Sink>
DF_HD void write_dyn_header(const DynHeader& h, const U16P tok, const uint8_t* cl_len, const U32P cl_code, Sink&& sink)
{
    sink(4u, 3u); // BFINAL=0, BTYPE=10
    sink(h.nll - 257u, 5u);
    sink(h.nd - 1u, 5u);
    sink(h.ncl - 4u, 4u);
    for (uint32_t k = 0; k < h.ncl; ++k) sink((uint32_t)cl_len[cl_order(k)], 3u);
    for (uint32_t t = 0; t < h.ntok; ++t) {
        const uint32_t sym = tok[t] & 0xffu, ev = (uint32_t)tok[t] >> 8;
        const uint32_t c = cl_code[sym];
        sink(c >> 8, c & 0xffu);
        if (sym >= 16u) sink(ev, sym == 16u ? 2u : sym == 17u ? 3u : 7u);
    }
}

} // namespace snaphash
