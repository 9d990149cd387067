// deflate_lab.cpp -- CPU laboratory for the parse of the block-parallel DEFLATE kernel (row f3): what a change of
// chunk size, table shape, search depth or window buys in output size, BEFORE it is written for the GPU.  Sizes are
// exact for the format the kernel emits (dynamic / fixed / stored per chunk, the routines of deflate_core.h, the
// 5-byte byte-aligning empty stored block behind every chunk).  Not part of the product.
//   g++ -O2 -std=c++17 -o /tmp/deflate_lab tools/deflate_lab.cpp && /tmp/deflate_lab FILE [key=value ...]
//   keys: chunk=16384 seed=16384 hb=11 ways=4 chain=0 (0 = bucket finder, N = hash chains of depth N) hb2=0 (bits of an
//         8-byte "long" hash table, one entry per bucket) lazy=1 min=4 nice=258 maxdist=32768
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../snappy_amd/csrc/deflate_core.h"
#include "../tests/deflate_model.h"

using namespace snaphash;

struct Opt { uint32_t chunk = 16384, seed = 16384, hb = 11, ways = 4, chain = 0, hb2 = 0, lazy = 1, minm = 4, nice = 258, maxdist = 32768, lazy2 = 0, toofar = 4096; };

static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct Finder {
    const Opt& o;
    const uint8_t* base; // whole buffer
    size_t n;
    std::vector<uint32_t> tab;   // bucket finder: ways entries per bucket (position + 1, 0 = empty), newest first
    std::vector<uint32_t> head, prev; // chain finder
    std::vector<uint32_t> tab2;  // long hash
    size_t prev_base = 0;
    Finder(const Opt& o_, const uint8_t* b, size_t n_) : o(o_), base(b), n(n_) {}
    uint32_t h4(size_t p) const { const uint32_t w = o.minm == 3 ? (ld32(base + p) & 0xffffffu) : ld32(base + p); return (w * 0x9E3779B1u) >> (32 - o.hb); }
    uint32_t h8(size_t p) const { return (uint32_t)((ld64(base + p) * 0x9E3779B97F4A7C15ull) >> (64 - o.hb2)); }
    void reset(size_t from)
    {
        if (o.chain) { head.assign(1u << o.hb, 0); prev.assign(o.chunk + o.seed + 8, 0); prev_base = from; }
        else tab.assign((size_t)o.ways << o.hb, 0);
        if (o.hb2) tab2.assign(1u << o.hb2, 0);
    }
    void insert(size_t p)
    {
        if (p + 8 > n) return;
        if (o.chain) {
            const uint32_t h = h4(p);
            prev[p - prev_base] = head[h];
            head[h] = (uint32_t)(p + 1);
        } else {
            uint32_t* b = &tab[(size_t)h4(p) * o.ways];
            for (uint32_t k = o.ways - 1; k > 0; --k) b[k] = b[k - 1];
            b[0] = (uint32_t)(p + 1);
        }
        if (o.hb2) tab2[h8(p)] = (uint32_t)(p + 1);
    }
    uint32_t extend(size_t a, size_t c, uint32_t maxl) const
    {
        uint32_t l = 0;
        while (l < maxl && base[a + l] == base[c + l]) ++l;
        return l;
    }
    // best match at p (not yet inserted); chunk_end bounds the length
    void find(size_t p, size_t chunk_end, uint32_t& mlen, uint32_t& dist) const
    {
        mlen = 0; dist = 0;
        if (p + 8 > n) return;
        const uint32_t maxl = (uint32_t)std::min<size_t>(258, chunk_end - p);
        if (maxl < o.minm) return;
        auto consider = [&](uint32_t e) {
            if (!e) return;
            const size_t c = e - 1;
            if (p - c > o.maxdist) return;
            const uint32_t l = extend(p, c, maxl);
            if (l >= o.minm && l > mlen) { mlen = l; dist = (uint32_t)(p - c); }
        };
        if (o.hb2) consider(tab2[h8(p)]);
        if (o.chain) {
            uint32_t e = head[h4(p)];
            for (uint32_t d = 0; d < o.chain && e; ++d) {
                if (p - (e - 1) > o.maxdist) break;
                consider(e);
                if (mlen >= o.nice) break;
                e = prev[(e - 1) - prev_base];
            }
        } else {
            const uint32_t* b = &tab[(size_t)h4(p) * o.ways];
            for (uint32_t k = 0; k < o.ways; ++k) { consider(b[k]); if (mlen >= o.nice) break; }
        }
        if (mlen == 3 && dist > o.toofar) mlen = 0; // zlib's TOO_FAR
    }
};

typedef dfmodel::Tok Tok;

static uint64_t block_bits(const std::vector<Tok>& toks, uint32_t rawlen, int* kind)
{
    uint32_t llf[kNumLL] = {0}, df[kNumD] = {0};
    uint64_t extra = 0, fixed = 3 + 7;
    for (const Tok& t : toks) {
        if (!t.len) { llf[t.lit]++; fixed += fixed_ll_bits(t.lit); continue; }
        uint32_t ls, le, lv, ds, de, dv;
        len_symbol(t.len, ls, le, lv);
        dist_symbol(t.dist, ds, de, dv);
        llf[ls]++; df[ds]++;
        extra += le + de;
        fixed += fixed_ll_bits(ls) + 5;
    }
    llf[256]++;
    if (!df[0]) df[0] = 1;
    if (!df[1]) df[1] = 1;
    fixed += extra;
    uint8_t lll[kNumLL], dl[kNumD], cll[kNumCL];
    std::vector<uint32_t> w(2 * kNumLL), cnt(257), clf(kNumCL), clc(kNumCL);
    std::vector<uint16_t> parent(2 * kNumLL), order(kNumLL), rle(kNumLL + kNumD);
    huff_lengths(llf, kNumLL, (uint32_t)kMaxBits, lll, w.data(), parent.data(), order.data(), cnt.data());
    huff_lengths(df, kNumD, (uint32_t)kMaxBits, dl, w.data(), parent.data(), order.data(), cnt.data());
    DynHeader hdr;
    build_dyn_header(lll, dl, rle.data(), clf.data(), cll, clc.data(), w.data(), parent.data(), order.data(), cnt.data(), hdr);
    uint64_t dyn = hdr.bits + extra;
    for (int i = 0; i < kNumLL; ++i) dyn += (uint64_t)llf[i] * lll[i];
    for (int i = 0; i < kNumD; ++i) dyn += (uint64_t)df[i] * dl[i];
    uint64_t best = std::min(dyn, fixed);
    *kind = dyn < fixed ? 2 : 1;
    best += 3;                       // empty stored block header
    best = (best + 7) / 8 * 8 + 32;  // pad + LEN/NLEN
    const uint64_t stored = (uint64_t)(rawlen + 5) * 8;
    if (best >= stored) { *kind = 0; return stored; }
    return best;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    Opt o;
    unsigned v2 = 0, good = snaphash::kDfGood, inherit = 0, hashn = 3, first3 = 0, via = 0;
    for (int a = 2; a < argc; ++a) {
        char k[32]; unsigned v;
        if (sscanf(argv[a], "%31[^=]=%u", k, &v) != 2) return 2;
        std::string key = k;
        if (key == "v2") { v2 = v; o.chunk = snaphash::kDfChunk; o.maxdist = snaphash::kDfMaxDist; o.chain = snaphash::kDfDepth; o.nice = snaphash::kDfNice; o.toofar = snaphash::kDfTooFar; o.hb = snaphash::kDfHashBits; continue; }
        if (key == "good") { good = v; continue; }
        if (key == "inherit") { inherit = v; continue; }
        if (key == "hashn") { hashn = v; continue; }
        if (key == "via") { via = v; continue; }
        if (key == "first3") { first3 = v; continue; }
        if (key == "chunk") o.chunk = v; else if (key == "seed") o.seed = v; else if (key == "hb") o.hb = v; else if (key == "ways") o.ways = v;
        else if (key == "chain") o.chain = v; else if (key == "hb2") o.hb2 = v; else if (key == "lazy") o.lazy = v; else if (key == "min") o.minm = v;
        else if (key == "nice") o.nice = v; else if (key == "maxdist") o.maxdist = v; else if (key == "lazy2") o.lazy2 = v; else if (key == "toofar") o.toofar = v; else return 2;
    }
    const Opt& v2p = o;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t r;
    while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
    fclose(f);
    const size_t n = buf.size();
    buf.resize(n + 16, 0);
    std::string label; for (int a = 2; a < argc; ++a) { label += argv[a]; label += " "; }
    if (v2) { // the shipped parse (tests/deflate_model.h) with the given parameters
        dfmodel::Params P;
        P.chunk = v2p.chunk; P.max_dist = v2p.maxdist; P.depth = v2p.chain ? v2p.chain : P.depth; P.nice = v2p.nice; P.too_far = v2p.toofar;
        P.hash_bits = v2p.hb; P.good = good; P.inherit = inherit; P.hashn = hashn; P.first3 = first3; P.via = via;
        dfmodel::Stats st;
        uint64_t bits = 80;
        int kinds[3] = {0, 0, 0};
        for (size_t c0 = 0; c0 < n; c0 += P.chunk) {
            const size_t c1 = std::min(n, c0 + P.chunk);
            std::vector<Tok> toks;
            dfmodel::parse_chunk(buf.data(), n, c0, c1, P, toks, &st);
            int kind;
            bits += block_bits(toks, (uint32_t)(c1 - c0), &kind);
            kinds[kind]++;
        }
        bits += 16 + 64;
        printf("v2 %-52s ratio %.4f  steps/pos %.1f  max steps/tile %.1f  (stored/fixed/dyn %d/%d/%d)\n", label.c_str(), (double)(bits / 8) / (double)n,
               (double)st.steps / st.positions, (double)st.tile_max_steps / st.tiles, kinds[0], kinds[1], kinds[2]);
        return 0;
    }
    Finder fd(o, buf.data(), n);
    uint64_t total_bits = 80; // gzip header
    uint64_t nmatch = 0, nlit = 0, mbytes = 0;
    int kinds[3] = {0, 0, 0};
    for (size_t c0 = 0; c0 < n; c0 += o.chunk) {
        const size_t c1 = std::min(n, c0 + o.chunk);
        const size_t s0 = c0 >= o.seed ? c0 - o.seed : 0;
        fd.reset(s0);
        for (size_t p = s0; p < c0; ++p) fd.insert(p);
        std::vector<Tok> toks;
        size_t p = c0;
        uint32_t ml = 0, md = 0;
        bool have = false; // (ml, md) already computed for p
        while (p < c1) {
            if (!have) fd.find(p, c1, ml, md);
            have = false;
            if (ml >= o.minm) {
                if (o.lazy && p + 1 < c1) { // one-byte lazy evaluation
                    fd.insert(p);
                    uint32_t ml2, md2;
                    fd.find(p + 1, c1, ml2, md2);
                    if (ml2 > ml) {
                        toks.push_back(Tok{buf[p], 0, 0}); ++nlit;
                        ++p; ml = ml2; md = md2; have = true;
                        continue;
                    }
                    toks.push_back(Tok{0, ml, md}); ++nmatch; mbytes += ml;
                    for (size_t q = p + 1; q < p + ml; ++q) fd.insert(q);
                    p += ml;
                    continue;
                }
                toks.push_back(Tok{0, ml, md}); ++nmatch; mbytes += ml;
                for (size_t q = p; q < p + ml; ++q) fd.insert(q);
                p += ml;
            } else {
                toks.push_back(Tok{buf[p], 0, 0}); ++nlit;
                fd.insert(p);
                ++p;
            }
        }
        int kind;
        total_bits += block_bits(toks, (uint32_t)(c1 - c0), &kind);
        kinds[kind]++;
    }
    total_bits += 16 + 64; // final empty fixed block (2 bytes) + CRC + ISIZE
    printf("%-60s ratio %.4f  (%zu -> %llu bytes; %llu matches avg %.1f, %llu literals; blocks stored/fixed/dyn %d/%d/%d)\n",
           label.c_str(), (double)(total_bits / 8) / (double)n, n, (unsigned long long)(total_bits / 8),
           (unsigned long long)nmatch, nmatch ? (double)mbytes / nmatch : 0.0, (unsigned long long)nlit, kinds[0], kinds[1], kinds[2]);
    return 0;
}
