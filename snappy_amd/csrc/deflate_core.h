// deflate_core.h -- the token encoder of the block-parallel DEFLATE kernel (fixed Huffman codes,
// RFC 1951 sec. 3.2.5-3.2.6), shared with a host harness in tests/ so that the code tables and the
// chunk framing are checked against zlib on the CPU before the kernel runs on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DF_HD __host__ __device__ __forceinline__
#else
#define DF_HD inline
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
