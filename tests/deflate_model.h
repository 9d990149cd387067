// deflate_model.h -- CPU model of the parse of deflate_chunks_kernel (round 3: hash chains in a ring, workgroup per
// chunk).  Test infrastructure: tests/f3_host_harness.cpp builds the expected bytes with it, tools/deflate_lab.cpp
// explores its parameters.  Serial restatement of exactly what the kernel computes:
//   links    link[p] = p - q for the nearest q < p (q >= first indexed position) whose first three bytes hash alike,
//            0 when there is none or it is more than 65 534 back; positions are indexed while p + 3 <= n_in
//   search   from p, follow links: at most depth candidates, none farther than max_dist; a candidate is extended only
//            if it agrees with p in the four bytes ending at the current best length (it cannot be longer otherwise);
//            longer wins, the first found keeps ties; nice ends the walk, good cuts the links left to a quarter;
//            a 3-byte match farther than too_far is dropped
//   parse    greedy with one-byte lazy evaluation, tile (64 positions) by tile: no look-ahead across a tile's end
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../snappy_amd/csrc/deflate_core.h"

namespace dfmodel {

struct Params {
    uint32_t chunk = snaphash::kDfChunk, max_dist = snaphash::kDfMaxDist, depth = snaphash::kDfDepth, good = snaphash::kDfGood,
             nice = snaphash::kDfNice, too_far = snaphash::kDfTooFar, hash_bits = snaphash::kDfHashBits;
};
struct Tok { uint32_t lit, len, dist; }; // len == 0: literal
struct Stats { uint64_t steps = 0, tile_max_steps = 0, tiles = 0, positions = 0; };

inline uint32_t ld32(const uint8_t* p, const uint8_t* end) // bytes past the buffer read as zero (the kernel's buffers are padded)
{
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) if (p + k < end) v |= (uint32_t)p[k] << (8 * k);
    return v;
}

// in[0, n_in): the staged piece; [c0, c1) the chunk.  Appends the chunk's tokens.
inline void parse_chunk(const uint8_t* in, size_t n_in, size_t c0, size_t c1, const Params& P, std::vector<Tok>& toks, Stats* st = nullptr)
{
    const uint8_t* end = in + n_in;
    const size_t s0 = c0 >= P.max_dist ? c0 - P.max_dist : 0;
    std::vector<uint32_t> head(1u << P.hash_bits, 0);
    std::vector<uint16_t> link(c1 - s0, 0);
    auto hash = [&](uint32_t w) { return ((w & 0xffffffu) * 0x9E3779B1u) >> (32u - P.hash_bits); };
    for (size_t p = s0; p < c1; ++p) {
        if (p + 3 > n_in) break;
        const uint32_t h = hash(ld32(in + p, end));
        const uint32_t q1 = head[h]; // q + 1 - s0
        uint32_t d = q1 ? (uint32_t)(p - s0) + 1u - q1 : 0u;
        if (d >= 65535u) d = 0; // 65 535 is the kernel's "first of its hash in the tile" marker
        link[p - s0] = (uint16_t)d;
        head[h] = (uint32_t)(p - s0) + 1u;
    }
    const size_t len = c1 - c0;
    std::vector<uint32_t> mlen(len, 0), mdist(len, 0);
    for (size_t t0 = 0; t0 < len; t0 += 64) {
        uint32_t tile_max = 0;
        for (size_t i = t0; i < std::min(len, t0 + 64); ++i) {
            const size_t p = c0 + i;
            const uint32_t maxl = (uint32_t)std::min<size_t>(258, c1 - p);
            if (maxl < snaphash::kDfMinMatch || p + 3 > n_in) continue;
            uint32_t best = snaphash::kDfMinMatch - 1, bdist = 0, left = P.depth, steps = 0;
            size_t cur = p;
            while (left) {
                const uint32_t d = link[cur - s0];
                if (!d) break;
                cur -= d;
                if (p - cur > P.max_dist) break;
                --left;
                ++steps;
                if (best >= 3u && ld32(in + cur + best - 3, end) != ld32(in + p + best - 3, end)) continue;
                uint32_t l = 0;
                while (l < maxl && in[cur + l] == in[p + l]) ++l;
                if (l > best) {
                    best = l;
                    bdist = (uint32_t)(p - cur);
                    if (l >= P.nice || l >= maxl) break;
                    if (l >= P.good && left > P.depth / 4) left = P.depth / 4;
                }
            }
            if (best == 3 && bdist > P.too_far) best = 0;
            if (best >= snaphash::kDfMinMatch) { mlen[i] = best; mdist[i] = bdist; }
            if (st) { st->steps += steps; st->positions++; }
            tile_max = std::max(tile_max, steps);
        }
        if (st) { st->tile_max_steps += tile_max; st->tiles++; }
    }
    size_t rel = 0;
    while (rel < len) {
        const size_t tile_end = std::min(len, (rel / 64 + 1) * 64);
        if (mlen[rel] >= snaphash::kDfMinMatch && rel + 1 < tile_end && mlen[rel + 1] > mlen[rel]) { // lazy: the next byte matches longer
            toks.push_back(Tok{in[c0 + rel], 0, 0});
            rel += 1;
        } else if (mlen[rel] >= snaphash::kDfMinMatch) {
            toks.push_back(Tok{0, mlen[rel], mdist[rel]});
            rel += mlen[rel];
        } else {
            toks.push_back(Tok{in[c0 + rel], 0, 0});
            rel += 1;
        }
    }
}

} // namespace dfmodel
