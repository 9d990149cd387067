// Test-only (CPU): (1) a serial model of deflate_chunks_kernel -- same chunking, hash, greedy parse,
// token encoder (snappy_amd/csrc/deflate_core.h) and framing -- so that the code tables and the
// byte-aligned chunk framing are checked against zlib before the kernel runs on a GPU; (2) the host
// half of the tar producer (tarpack.cpp): plan + ustar headers + CRC-32.  Not part of the product.
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../snappy_amd/csrc/deflate_core.h"
#include "../snappy_amd/csrc/tarpack.cpp"
#include "../snappy_amd/csrc/walk.cpp"

using namespace snaphash;

namespace {
constexpr uint32_t kChunk = 16384, kHashBits = 11, kGroup = 16;

struct BitW {
    std::vector<uint8_t>& o;
    uint64_t acc = 0;
    uint32_t n = 0;
    explicit BitW(std::vector<uint8_t>& out) : o(out) {}
    void put(uint32_t bits, uint32_t nb) { acc |= (uint64_t)bits << n; n += nb; while (n >= 8) { o.push_back((uint8_t)acc); acc >>= 8; n -= 8; } }
    void align() { if (n) { o.push_back((uint8_t)acc); acc = 0; n = 0; } }
};

uint32_t ld32(const uint8_t* p, const uint8_t* end) // like the kernel: bytes past the buffer read as padding
{
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) if (p + k < end) v |= (uint32_t)p[k] << (8 * k);
    return v;
}

struct Tok { uint32_t lit, len, dist; }; // len == 0: literal

// The kernel's parse of one chunk: a four-way bucket per hash value (newest first, one 64-bit word), updated
// in groups of kGroup positions -- every lane of a group reads its bucket before any lane of the group writes, and of
// the lanes that share a bucket the highest position wins (the kernel's LDS atomic max) -- seeded the same way from
// the previous chunk; the longest of the up to four candidates (ties: the nearest), greedy with one-byte lazy
// evaluation, tile (64 positions) by tile.
void parse_chunk(const uint8_t* src, uint32_t len, const uint8_t* bufend, bool has_prev, std::vector<Tok>& toks)
{
    std::vector<uint64_t> tab(1u << kHashBits, 0);
    auto hash = [](uint32_t w) { return (w * 0x9E3779B1u) >> (32 - kHashBits); };
    // entry = position + kChunk + 1 (0 = empty): the previous chunk's positions are 1 .. kChunk
    auto tile_update = [&](const uint32_t* word, const uint64_t* old, uint32_t first_entry, uint32_t n) {
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t v = ((uint64_t)(first_entry + i) << 48) | (old[i] >> 16);
            uint64_t& t = tab[hash(word[i])];
            if (v > t) t = v;
        }
    };
    if (has_prev)
        for (uint32_t p0 = 0; p0 < kChunk; p0 += 64) {
            uint32_t word[64];
            uint64_t old[64];
            for (uint32_t i = 0; i < 64; ++i) word[i] = ld32(src - kChunk + p0 + i, bufend);
            for (uint32_t g0 = 0; g0 < 64; g0 += kGroup) {
                for (uint32_t i = g0; i < g0 + kGroup; ++i) old[i] = tab[hash(word[i])];
                tile_update(word + g0, old + g0, p0 + g0 + 1, kGroup);
            }
        }
    uint32_t skip_until = 0;
    for (uint32_t p0 = 0; p0 < len; p0 += 64) {
        uint32_t mlen[64] = {0}, dist[64] = {0}, word[64] = {0};
        uint64_t cand[64] = {0};
        const uint32_t tile_n = len - p0 < 64 ? len - p0 : 64;
        uint32_t n_can = 0; // positions with four bytes left: a prefix of the tile
        for (uint32_t i = 0; i < tile_n; ++i) {
            word[i] = ld32(src + p0 + i, bufend);
            if (p0 + i + 4 <= len) n_can = i + 1;
        }
        for (uint32_t g0 = 0; g0 < n_can; g0 += kGroup) {
            const uint32_t gn = std::min<uint32_t>(kGroup, n_can - g0);
            for (uint32_t i = g0; i < g0 + gn; ++i) cand[i] = tab[hash(word[i])];
            tile_update(word + g0, cand + g0, p0 + g0 + kChunk + 1, gn);
        }
        for (uint32_t i = 0; i < n_can; ++i) {
            const uint32_t pos = p0 + i;
            const uint32_t maxl = len - pos < 258 ? len - pos : 258;
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = (uint32_t)(cand[i] >> (48 - 16 * k)) & 0xffffu;
                if (!e) break;
                const int64_t cp = (int64_t)e - 1 - kChunk;
                uint32_t l = 0;
                while (l < maxl && src[pos + l] == src[cp + (int64_t)l]) ++l;
                if (l >= 4 && l > mlen[i]) { mlen[i] = l; dist[i] = (uint32_t)((int64_t)pos - cp); }
            }
        }
        uint32_t rel = skip_until > p0 ? skip_until - p0 : 0;
        while (rel < tile_n) {
            if (mlen[rel] >= 4 && rel + 1 < tile_n && mlen[rel + 1] > mlen[rel]) { // lazy: the next byte matches longer
                toks.push_back(Tok{word[rel] & 0xff, 0, 0}); rel += 1;
            } else if (mlen[rel] >= 4) { toks.push_back(Tok{0, mlen[rel], dist[rel]}); rel += mlen[rel]; }
            else { toks.push_back(Tok{word[rel] & 0xff, 0, 0}); rel += 1; }
        }
        skip_until = p0 + rel;
    }
}

// has_prev: the previous chunk lies in the same staging piece (the kernel seeds its table from it)
void deflate_chunk(const uint8_t* src, uint32_t len, const uint8_t* bufend, bool has_prev, std::vector<uint8_t>& out)
{
    std::vector<Tok> toks;
    parse_chunk(src, len, bufend, has_prev, toks);
    // symbol counts and the cost of both block kinds
    uint32_t llf[kNumLL] = {0}, df[kNumD] = {0};
    uint64_t extra_bits = 0, fixed_bits = 3 + 7;
    for (const Tok& t : toks) {
        if (t.len == 0) { llf[t.lit]++; fixed_bits += fixed_ll_bits(t.lit); continue; }
        uint32_t ls, le, lv, ds, de, dv;
        len_symbol(t.len, ls, le, lv);
        dist_symbol(t.dist, ds, de, dv);
        llf[ls]++; df[ds]++;
        extra_bits += le + de;
        fixed_bits += fixed_ll_bits(ls) + 5;
    }
    llf[256]++;
    if (df[0] == 0) df[0] = 1; // at least two distance codes, as zlib sends (old inflaters want a complete code)
    if (df[1] == 0) df[1] = 1;
    fixed_bits += extra_bits;
    uint8_t lll[kNumLL], dl[kNumD], cll[kNumCL];
    std::vector<uint32_t> w(2 * kNumLL), cnt(257), llc(kNumLL), dc(kNumD), clf(kNumCL), clc(kNumCL);
    std::vector<uint16_t> parent(2 * kNumLL), order(kNumLL), rle(kNumLL + kNumD);
    huff_lengths(llf, kNumLL, (uint32_t)kMaxBits, lll, w.data(), parent.data(), order.data(), cnt.data());
    huff_lengths(df, kNumD, (uint32_t)kMaxBits, dl, w.data(), parent.data(), order.data(), cnt.data());
    DynHeader hdr;
    build_dyn_header(lll, dl, rle.data(), clf.data(), cll, clc.data(), w.data(), parent.data(), order.data(), cnt.data(), hdr);
    huff_codes(lll, kNumLL, llc.data(), cnt.data());
    huff_codes(dl, kNumD, dc.data(), cnt.data());
    uint64_t dyn_bits = hdr.bits + extra_bits;
    for (int i = 0; i < kNumLL; ++i) dyn_bits += (uint64_t)llf[i] * lll[i];
    for (int i = 0; i < kNumD; ++i) dyn_bits += (uint64_t)df[i] * dl[i];
    const bool dynamic = dyn_bits < fixed_bits;

    std::vector<uint8_t> z;
    BitW bw(z);
    if (dynamic) write_dyn_header(hdr, rle.data(), cll, clc.data(), [&](uint32_t bits, uint32_t nb) { bw.put(bits, nb); });
    else bw.put(2, 3);
    for (const Tok& t : toks) {
        uint32_t bits, nb;
        if (t.len == 0) {
            if (dynamic) bw.put(llc[t.lit] >> 8, llc[t.lit] & 0xff);
            else { enc_literal(t.lit, bits, nb); bw.put(bits, nb); }
            continue;
        }
        if (!dynamic) { enc_match(t.len, t.dist, bits, nb); bw.put(bits, nb); continue; }
        uint32_t ls, le, lv, ds, de, dv;
        len_symbol(t.len, ls, le, lv);
        dist_symbol(t.dist, ds, de, dv);
        bw.put(llc[ls] >> 8, llc[ls] & 0xff);
        bw.put(lv, le);
        bw.put(dc[ds] >> 8, dc[ds] & 0xff);
        bw.put(dv, de);
    }
    if (dynamic) bw.put(llc[256] >> 8, llc[256] & 0xff); else bw.put(0, 7);  // end of block
    bw.put(0, 3);  // empty stored block
    bw.align();
    z.push_back(0); z.push_back(0); z.push_back(0xff); z.push_back(0xff);
    if (z.size() >= len + 5u) {
        out.push_back(0);
        out.push_back((uint8_t)len); out.push_back((uint8_t)(len >> 8));
        out.push_back((uint8_t)~len); out.push_back((uint8_t)(~len >> 8));
        out.insert(out.end(), src, src + len);
    } else {
        out.insert(out.end(), z.begin(), z.end());
    }
}
} // namespace

extern "C" {

// -> malloc'd gzip member; model of snaphash_gzip_buffer's output.  piece = bytes per staging piece (a multiple of
// the chunk size; 0 = everything in one): a chunk's table is seeded from the previous chunk only within a piece.
uint8_t* f3_model_gzip2(const uint8_t* in, size_t n, size_t piece, size_t* out_len)
{
    std::vector<uint8_t> out(kGzipHeader, kGzipHeader + 10);
    for (size_t off = 0; off < n; off += kChunk) {
        const size_t pend = piece ? std::min(n, (off / piece + 1) * piece) : n; // the kernel never reads past its piece (+3)
        const bool has_prev = piece ? (off % piece) != 0 : off != 0;
        deflate_chunk(in + off, (uint32_t)(n - off < kChunk ? n - off : kChunk), in + pend, has_prev, out);
    }
    out.push_back(0x03); out.push_back(0x00);
    const uint32_t crc = crc32_update(0, in, n);
    for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(crc >> (8 * k)));
    for (int k = 0; k < 4; ++k) out.push_back((uint8_t)((uint32_t)n >> (8 * k)));
    uint8_t* p = (uint8_t*)malloc(out.size());
    memcpy(p, out.data(), out.size());
    *out_len = out.size();
    return p;
}
uint8_t* f3_model_gzip(const uint8_t* in, size_t n, size_t* out_len) { return f3_model_gzip2(in, n, 0, out_len); }

uint32_t f3_crc32(uint32_t crc, const uint8_t* p, size_t n) { return crc32_update(crc, p, n); }
uint32_t f3_crc32_combine(uint32_t a, uint32_t b, uint64_t len2) { return crc32_combine(a, b, len2); }

// the whole tar stream of a directory, as the producer lays it out (host reads; tests only)
int f3_tar_stream(const char* dir, const char* exclude_prefix, uint8_t** out, size_t* out_len)
{
    TarPlan plan;
    int en = 0;
    std::string what;
    int rc = tar_plan(dir, exclude_prefix ? exclude_prefix : "", plan, &en, &what);
    if (rc) return rc;
    uint8_t* buf = (uint8_t*)calloc(plan.total ? plan.total : 1, 1);
    for (const TarMember& m : plan.members) {
        if (!m.pax.empty()) {
            tar_pax_header(m, buf + m.pax_off);
            memcpy(buf + m.pax_off + 512, m.pax.data(), m.pax.size());
        }
        rc = tar_header(m, buf + m.hdr_off);
        if (rc) { free(buf); return rc; }
        if (m.typeflag == '0' && m.size) {
            const int fd = open(m.path.c_str(), O_RDONLY);
            if (fd < 0 || pread(fd, buf + m.data_off, (size_t)m.size, 0) != (ssize_t)m.size) { if (fd >= 0) close(fd); free(buf); return -3; }
            close(fd);
        }
    }
    *out = buf;
    *out_len = plan.total;
    return 0;
}
// the header record of a regular member of the given size (tests: the fields beyond what a real tree can afford)
int f3_tar_header_of(const char* name, int64_t size, uint8_t* out512)
{
    TarMember m;
    m.name = name;
    m.size = size;
    m.st_mode = 0100644;
    m.mtime = 1;
    return tar_header(m, out512);
}
void f3_free(void* p) { free(p); }
}
