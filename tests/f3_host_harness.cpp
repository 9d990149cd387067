// Test-only (CPU): (1) a serial model of deflate_chunks_kernel -- same chunking, hash chains, lazy parse
// (tests/deflate_model.h), token encoder (snappy_amd/csrc/deflate_core.h) and framing -- so that the code tables and the
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
#include "deflate_model.h"
#include "../snappy_amd/csrc/tarpack.cpp"
#include "../snappy_amd/csrc/walk.cpp"
#include "../snappy_amd/csrc/hostfill.cpp" // walk.cpp sizes its thread pools with usable_cpus()

using namespace snaphash;

namespace {
constexpr uint32_t kChunk = kDfChunk;

struct BitW {
    std::vector<uint8_t>& o;
    uint64_t acc = 0;
    uint32_t n = 0;
    explicit BitW(std::vector<uint8_t>& out) : o(out) {}
    void put(uint32_t bits, uint32_t nb) { acc |= (uint64_t)bits << n; n += nb; while (n >= 8) { o.push_back((uint8_t)acc); acc >>= 8; n -= 8; } }
    void align() { if (n) { o.push_back((uint8_t)acc); acc = 0; n = 0; } }
};

typedef dfmodel::Tok Tok; // len == 0: literal

// piece[0, n_piece): the staging piece the kernel is launched on; [c0, c1) the chunk inside it.  The parse is
// tests/deflate_model.h (the serial restatement of the kernel's pipeline); the code construction and the block
// framing below run the kernel's own routines (deflate_core.h).
static uint32_t g_model_depth = 0; // 0 = kDfDepth; f3_model_gzip3 sets it (the product: snaphash_config.deflate_depth)

void deflate_chunk(const uint8_t* piece, size_t n_piece, size_t c0, size_t c1, std::vector<uint8_t>& out)
{
    const uint8_t* src = piece + c0;
    const uint32_t len = (uint32_t)(c1 - c0);
    std::vector<Tok> toks;
    dfmodel::Params P;
    if (g_model_depth) P.depth = g_model_depth;
    dfmodel::parse_chunk(piece, n_piece, c0, c1, P, toks);
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
    if (z.size() >= deflate_stored_size(len)) { // stored: LEN is 16 bits, a full 64 KiB chunk goes out as two blocks of 32 KiB
        const uint32_t first = len > 65535u ? 32768u : len;
        for (uint32_t at = 0, n = first; at < len || at == 0; at += n, n = len - at) {
            out.push_back(0);
            out.push_back((uint8_t)n); out.push_back((uint8_t)(n >> 8));
            out.push_back((uint8_t)~n); out.push_back((uint8_t)(~n >> 8));
            out.insert(out.end(), src + at, src + at + n);
            if (len == 0) break;
        }
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
    if (piece == 0) piece = n ? n : 1;
    for (size_t p0 = 0; p0 < n; p0 += piece) { // the window in front of a chunk never reaches into the previous piece
        const size_t pn = std::min(piece, n - p0);
        for (size_t off = 0; off < pn; off += kChunk) deflate_chunk(in + p0, pn, off, std::min(pn, off + kChunk), out);
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
// the same at another search depth (links walked per position; not thread-safe: a test harness)
uint8_t* f3_model_gzip3(const uint8_t* in, size_t n, size_t piece, uint32_t depth, size_t* out_len)
{
    g_model_depth = depth;
    uint8_t* p = f3_model_gzip2(in, n, piece, out_len);
    g_model_depth = 0;
    return p;
}

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
// the price parse's arithmetic (deflate_core.h)
uint32_t f3_ilog(uint32_t x) { return df_ilog(x); }
uint32_t f3_price(uint32_t f, uint32_t log_total, uint32_t cap) { return df_price(f, log_total, cap); }
void f3_free(void* p) { free(p); }
}
