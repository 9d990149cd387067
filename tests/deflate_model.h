// deflate_model.h -- CPU model of the parse of deflate_chunks_kernel (round 3: hash chains in a ring, workgroup per
// chunk).  Test infrastructure: tests/f3_host_harness.cpp builds the expected bytes with it, tools/deflate_lab.cpp
// explores its parameters.  Serial restatement of exactly what the kernel computes:
//   links    link[p] = p - q for the nearest q < p (q >= first indexed position) whose first three bytes hash alike,
//            0 when there is none or it is more than 65 534 back; positions are indexed while p + 3 <= n_in
//   search   from p, follow links: at most depth candidates, none farther than max_dist; a candidate is extended only
//            if it agrees with p in the four bytes ending at the current best length (it cannot be longer otherwise);
//            longer wins, the first found keeps ties; nice ends the walk, good cuts the links left to a quarter;
//            a 3-byte match farther than too_far is dropped
//   parse    the price parse of deflate_core.h, segment (1 920 positions) by segment, window (about 512 or 448: deflate_core.h) by window: forward,
//            every position offering a literal and the first L bytes of its match (L from 3, or 4 for a distance
//            beyond too_far, to 62) to the positions behind it; a match of 63 or more ends the run: the way to its
//            position is traced back, the match is taken whole, and a new run starts behind it
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
    uint32_t hashn = 3;   // LAB: bytes the chain's hash covers (3 = shipped); 4 / 5: longer contexts, fewer and better candidates
    uint32_t first3 = 0;  // LAB (with hashn > 3): also try the nearest earlier position with the same THREE bytes' hash (one candidate)
    uint32_t via = 0;     // LAB: N > 0 = a candidate from an earlier segment within N bytes also offers ITS best match's source (what it found is older than anything left of this walk)
    uint32_t inherit = 0; // LAB: 1 = a position takes over what is left of its predecessor's match when that is longer than its own (within a tile of 64), 2 = across the whole chunk
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
    auto hash3 = [&](uint32_t w) { return ((w & 0xffffffu) * 0x9E3779B1u) >> (32u - P.hash_bits); };
    auto hashw = [&](const uint8_t* q) -> uint32_t {
        if (P.hashn <= 3) return hash3(ld32(q, end));
        uint64_t w = ld32(q, end);
        if (P.hashn >= 5) w |= (uint64_t)(q + 4 < end ? q[4] : 0) << 32;
        if (P.hashn >= 6) w |= (uint64_t)(q + 5 < end ? q[5] : 0) << 40;
        return (uint32_t)((w * 0x9E3779B97F4A7C15ull) >> (64u - P.hash_bits));
    };
    std::vector<uint32_t> head3(P.first3 ? (1u << P.hash_bits) : 0, 0);
    std::vector<uint16_t> link3(P.first3 ? c1 - s0 : 0, 0);
    for (size_t p = s0; p < c1; ++p) {
        if (p + 3 > n_in) break;
        if (P.first3) {
            const uint32_t h3 = hash3(ld32(in + p, end));
            const uint32_t q1 = head3[h3];
            uint32_t d3 = q1 ? (uint32_t)(p - s0) + 1u - q1 : 0u;
            if (d3 >= 65535u) d3 = 0;
            link3[p - s0] = (uint16_t)d3;
            head3[h3] = (uint32_t)(p - s0) + 1u;
        }
        const uint32_t h = hashw(in + p);
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
            if (P.first3 && link3[p - s0] && link3[p - s0] <= P.max_dist) {
                const size_t c = p - link3[p - s0];
                uint32_t l = 0;
                while (l < maxl && in[c + l] == in[p + l]) ++l;
                ++steps;
                if (l > best) { best = l; bdist = (uint32_t)(p - c); }
            }
            while (left && best < P.nice && best < maxl) {
                const uint32_t d = link[cur - s0];
                if (!d) break;
                cur -= d;
                if (p - cur > P.max_dist) break;
                --left;
                ++steps;
                bool stop = false;
                for (int hop = 0; hop < (P.via ? 2 : 1) && !stop; ++hop) {
                    size_t c = cur;
                    if (hop == 1) { // the candidate's own best match, if its result is still at hand: an earlier segment of this chunk, not too far back
                        const size_t seg_p = c0 + ((p - c0) / snaphash::kDfSeg) * snaphash::kDfSeg;
                        if (cur < c0 || cur >= seg_p || p - cur > P.via || mlen[cur - c0] < 3) break;
                        c = cur - mdist[cur - c0];
                        if (p - c > P.max_dist) break;
                        ++steps;
                    }
                    if (best >= 3u && ld32(in + c + best - 3, end) != ld32(in + p + best - 3, end)) continue;
                    uint32_t l = 0;
                    while (l < maxl && in[c + l] == in[p + l]) ++l;
                    if (l > best) {
                        best = l;
                        bdist = (uint32_t)(p - c);
                        if (l >= P.nice || l >= maxl) { stop = true; break; }
                        if (l >= P.good && left > P.depth / 4) left = P.depth / 4;
                    }
                }
                if (stop) break;
            }
            if (best == 3 && bdist > P.too_far) best = 0;
            if (best >= snaphash::kDfMinMatch) { mlen[i] = best; mdist[i] = bdist; }
            if (st) { st->steps += steps; st->positions++; }
            tile_max = std::max(tile_max, steps);
        }
        if (st) { st->tile_max_steps += tile_max; st->tiles++; }
    }
    if (P.inherit) {
        for (size_t i = 1; i < len; ++i) {
            if (P.inherit == 1 && (i & 63) == 0) continue;
            if (mlen[i - 1] >= 4 && mlen[i - 1] - 1 > mlen[i]) { mlen[i] = mlen[i - 1] - 1; mdist[i] = mdist[i - 1]; }
        }
    }
    // the price parse
    uint32_t f_ll[snaphash::kNumLL] = {0}, f_d[snaphash::kNumD] = {0}, ntok = 0, nmatch = 0;
    uint32_t lit_price[256], len_price[snaphash::kDfLongMatch], dsym_price[snaphash::kNumD];
    std::vector<uint32_t> key(snaphash::kDfSeg + 1);
    std::vector<uint32_t> tok_len(snaphash::kDfSeg);
    for (size_t g0 = 0; g0 < len; g0 += snaphash::kDfSeg) {
        const uint32_t m = (uint32_t)std::min<size_t>(snaphash::kDfSeg, len - g0);
        using namespace snaphash;
        if (ntok < kDfPriceWarm) {
            for (uint32_t b = 0; b < 256u; ++b) lit_price[b] = kDfLitPrice0;
            for (uint32_t L = 3; L < kDfLongMatch; ++L) { uint32_t sy, eb, ev; len_symbol(L, sy, eb, ev); len_price[L] = kDfLenPrice0 + eb * kDfPriceUnit; }
            for (uint32_t d = 0; d < (uint32_t)kNumD; ++d) dsym_price[d] = kDfDistPrice0;
        } else {
            const uint32_t lt = df_ilog(ntok + 1u);
            for (uint32_t b = 0; b < 256u; ++b) lit_price[b] = df_price(f_ll[b], lt, kDfLLCap);
            for (uint32_t L = 3; L < kDfLongMatch; ++L) { uint32_t sy, eb, ev; len_symbol(L, sy, eb, ev); len_price[L] = df_price(f_ll[sy], lt, kDfLLCap) + eb * kDfPriceUnit; }
            const uint32_t dt = df_ilog(nmatch + 1u);
            for (uint32_t d = 0; d < (uint32_t)kNumD; ++d) dsym_price[d] = nmatch ? df_price(f_d[d], dt, kDfDistCap) : kDfDistPrice0;
        }
        std::fill(tok_len.begin(), tok_len.end(), 0u); // tok_len[i] != 0: a token of that length starts at i
        // where the windows end: far[p] = how far the matches of the positions in front of p reach
        uint32_t bounds[kDfParseWaves + 1u];
        bounds[0] = 0;
        {
            std::vector<uint32_t> far(m + 1u, 0);
            uint32_t reach = 0;
            for (uint32_t i = 0; i < m; ++i) {
                far[i] = reach;
                reach = std::max(reach, i + std::max(1u, std::min(mlen[g0 + i], m - i)));
            }
            far[m] = reach;
            for (uint32_t wi = 1; wi <= kDfParseWaves; ++wi) {
                uint32_t b = std::min(m, df_window_begin(wi));
                if (b < m)
                    for (uint32_t p = b; p + kDfCutSpan > b; --p)
                        if (far[p] <= p) { b = p; break; }
                bounds[wi] = b;
            }
        }
        for (uint32_t wi = 0; wi < kDfParseWaves && bounds[wi] < m; ++wi) { // the windows of the segment: no token leaves its own
        const uint32_t wa = bounds[wi], wb = bounds[wi + 1u];
        uint32_t pos = wa;
        while (pos < wb) {
            const uint32_t run0 = pos;
            std::fill(key.begin(), key.end(), 0xffffffffu);
            key[run0] = 0;
            uint32_t i = run0;
            for (; i < wb; ++i) {
                const uint32_t mlc = std::min(mlen[g0 + i], wb - i);
                if (mlc >= kDfLongMatch) break;
                const uint32_t c = key[i] >> 8;
                key[i + 1] = std::min(key[i + 1], df_key(c + lit_price[in[c0 + g0 + i]], 1u));
                if (mlc < kDfMinMatch) continue;
                uint32_t ds, de, dv;
                dist_symbol(mdist[g0 + i], ds, de, dv);
                const uint32_t dp = dsym_price[ds] + de * kDfPriceUnit;
                for (uint32_t L = mdist[g0 + i] > P.too_far ? 4u : 3u; L <= std::min(mlc, kDfLongMatch - 1u); ++L)
                    key[i + L] = std::min(key[i + L], df_key(c + len_price[L] + dp, L));
            }
            for (uint32_t p = i; p > run0;) { // the way back
                const uint32_t L = 255u - (key[p] & 255u);
                p -= L;
                tok_len[p] = L;
            }
            if (i < wb) { tok_len[i] = std::min(mlen[g0 + i], wb - i); pos = i + tok_len[i]; }
            else pos = wb;
        }
        }
        for (uint32_t i = 0; i < m;) {
            const uint32_t L = tok_len[i];
            if (L == 1u) { toks.push_back(Tok{in[c0 + g0 + i], 0, 0}); f_ll[in[c0 + g0 + i]]++; }
            else {
                toks.push_back(Tok{0, L, mdist[g0 + i]});
                uint32_t sy, eb, ev, ds, de, dv;
                len_symbol(L, sy, eb, ev); dist_symbol(mdist[g0 + i], ds, de, dv);
                f_ll[sy]++; f_d[ds]++; nmatch++;
            }
            ntok++;
            i += L;
        }
    }
}

} // namespace dfmodel
