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
// Per chunk the kernel runs its parse twice: the first pass only counts symbols, then ONE lane builds the two
// codes with the sequential, deterministic routines below (the host model in tests/ runs the very same code), the
// second pass emits with them.  Code lengths come from a plain two-queue Huffman construction over counting-
// sorted frequencies; if the tree is deeper than 15 the frequencies are halved and it is rebuilt.

namespace snaphash {

constexpr int kNumLL = 286; // literal/length symbols sent (HLIT = 29)
constexpr int kNumD = 30;   // distance symbols sent (HDIST = 29)
constexpr int kMaxBits = 15;
// 3 block header bits + HLIT/HDIST/HCLEN + 19 x 3 bits + one 4-bit code per code length (no run-length codes)
constexpr uint32_t kDynHeaderBits = 3 + 5 + 5 + 4 + 19 * 3 + (kNumLL + kNumD) * 4;

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

// Code lengths (1..kMaxBits) for n symbols, freq[i] >= 1.  Scratch: w[2n] u32, parent[2n] u16, order[n] u16,
// cnt[257] u32.  Deterministic: stable counting sort (ties by symbol index), ties between a leaf and an internal
// node go to the leaf.
template <typename U32P, typename U16P>
DF_HD_CALL void huff_lengths(const U32P freq, int n, uint8_t* len, U32P w, U16P parent, U16P order, U32P cnt)
{
    for (uint32_t shift = 0;; ++shift) {
        // stable radix sort of the symbols by (freq >> shift, at least 1), two 8-bit passes
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k <= 256; ++k) cnt[k] = 0;
            for (int i = 0; i < n; ++i) {
                uint32_t f = freq[i] >> shift;
                if (f == 0) f = 1;
                if (f > 0xffffu) f = 0xffffu;
                cnt[((f >> (8 * pass)) & 0xffu) + 1u]++;
            }
            for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
            // pass 0 reads symbols in index order into parent[] (used as a temporary), pass 1 reads that into order[]
            for (int i = 0; i < n; ++i) {
                const int sym = pass ? (int)parent[i] : i;
                uint32_t f = freq[sym] >> shift;
                if (f == 0) f = 1;
                if (f > 0xffffu) f = 0xffffu;
                const uint32_t at = cnt[(f >> (8 * pass)) & 0xffu]++;
                if (pass) order[at] = (uint16_t)sym;
                else parent[at] = (uint16_t)sym;
            }
        }
        // two-queue merge: leaves in sorted order (node id = position in `order`), internal nodes n .. 2n-2
        for (int i = 0; i < n; ++i) {
            uint32_t f = freq[order[i]] >> shift;
            w[i] = f ? f : 1u;
        }
        int li = 0, ii = n, nn = n; // heads of the leaf and internal queues, next internal node
        for (int k = 0; k < n - 1; ++k) {
            int a, b;
            if (li < n && (ii >= nn || w[li] <= w[ii])) a = li++; else a = ii++;
            if (li < n && (ii >= nn || w[li] <= w[ii])) b = li++; else b = ii++;
            w[nn] = w[a] + w[b];
            parent[a] = (uint16_t)nn;
            parent[b] = (uint16_t)nn;
            ++nn;
        }
        // depths, root first (internal nodes were created in increasing order, so a parent has the larger id)
        w[2 * n - 2] = 0;
        uint32_t deepest = 0;
        for (int node = 2 * n - 3; node >= 0; --node) {
            w[node] = w[parent[node]] + 1u;
            if (node < n && w[node] > deepest) deepest = w[node];
        }
        if (deepest <= (uint32_t)kMaxBits) {
            for (int i = 0; i < n; ++i) len[order[i]] = (uint8_t)w[i];
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

} // namespace snaphash
