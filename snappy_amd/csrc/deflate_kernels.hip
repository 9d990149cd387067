// deflate_kernels.hip -- block-parallel DEFLATE for gfx950 (MI355X): the GPU side of the
// data.tar.gz producer (reference clickdeb/deb.go:261-344, which pipes
// the tar stream through compress/gzip at level 9).
//
// Format-compatible, not byte-identical: archive-sha512 is taken over whatever bytes are
// produced (snappy/build.go:222), so the contract is RFC 1951/1952 validity -- any inflater
// must return exactly the tar stream -- not the bytes Go's compressor would emit.
//
// Round 3: hash chains.  The stream is cut into 64 KiB chunks, ONE WORKGROUP (16 waves) per chunk, one DEFLATE
// block per chunk.  What zlib does one byte after the other is spread over the workgroup as a pipeline of
// 1 920-byte segments (30 tiles of 64 positions), everything it shares living in LDS (157 KB, one workgroup per CU):
//   staging   the bytes of the segment (272 ahead) go from HBM into a 32 KiB data ring in LDS: everything below
//             reads the stream from there -- a search is thousands of scattered 4- and 8-byte reads, which the L1
//             serves one cache line per cycle and LDS 64 banks per cycle;
//   index     3-byte hash -> head[4096] and a ring of 32 768 16-bit links (distance to the previous position with
//             the same hash): inside a tile (64 positions) by ballots, all waves at once; across tiles through
//             head[], one wave, tile after tile;
//   search    every position of the segment walks its chain (its own link first, so it only ever sees older
//             positions) up to kDfDepth candidates within the last 28 800 bytes, four links at a time, and leaves
//             (length, distance, its byte) in an LDS result word;
//   parse     waves 1-4, one segment behind, a window (a quarter of the segment) each: the price parse of
//             deflate_core.h -- the cheapest way through the window at prices from the chunk's symbol counts so
//             far.  A shortest path, forward: the ways to the next 63 positions live in ONE register, a lane each
//             (lane = token length); a step offers every length of the position's match to them in one masked
//             v_min and shifts the register by a lane (DPP), the literal's way stays scalar; the way back hops from
//             token to token through a byte per position -> bitmaps, and the chosen lengths into the results;
//   finish    one more segment behind: symbol counts (LDS atomics), match tokens to an HBM scratch (4 B per
//             MATCH, not per byte), the two block prices.
// Only two barriers a step wait for everybody (staging | indexing inside tiles | everything else): the links across
// tiles announce their progress tile by tile and a searcher waits for its own tile; the finishers count their tiles
// and the parse waits for the last (its prices are the counts); search tiles and finish tiles come from queues.
// Then two waves build the dynamic Huffman codes (wave 0 the literal/length tree, wave 1 the distance tree: the sort is
// a rank count over the wave, the two-queue merge one lane's job, the depths walked leaf by leaf in parallel -- the
// lengths are those of the serial routines in deflate_core.h, which the CPU model runs), wave 0 the block header;
// every wave prices its tiles, a scan places them, and all waves encode their tokens into an LDS image of the block
// (the ring's bytes, spent by then) that is copied out in one coalesced sweep.  A chunk ends with an empty stored
// block (byte aligned, zlib's Z_SYNC_FLUSH); a second kernel concatenates the chunks.
//
// Integer/LDS work with data-dependent control flow: no MFMA.  Algorithmic bytes: 1 read and <= 1.0002 written
// per input byte (a chunk that does not shrink is stored).  HBM traffic beyond that: match tokens out and back
// (~0.3 B per input byte on text); the 28 KiB window in front of a chunk, staged by two workgroups, the candidates'
// check words and the emission's byte loads come out of the XCD's L2, because an XCD takes a run of consecutive
// chunks (measured: reads 1.34x the input at the counter's upper reading, 3.2x with every eighth chunk per XCD).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deflate_core.h"
#include "deflate_kernels.h"

namespace snaphash {

namespace {

constexpr uint32_t kWaves = 16;
constexpr uint32_t kThreads = kWaves * 64u;
constexpr uint32_t kSegTiles = kDfSeg / 64u;                 // 30
constexpr uint32_t kChunkTiles = kDfChunk / 64u;             // 1024
constexpr uint32_t kWinSegs = kDfMaxDist / kDfSeg;           // 15
constexpr uint32_t kRingMask = kDfRing - 1u;
constexpr uint32_t kHeadN = 1u << kDfHashBits;
constexpr uint32_t kDataRing = 32768;                        // bytes of the stream kept in LDS
constexpr uint32_t kDataMask = kDataRing - 1u;
constexpr uint32_t kLook = 272;                              // bytes in front of the indexed positions that are staged: 258 + 8, in whole 16s
static_assert(kDfMaxDist + kDfSeg + kLook <= kDataRing, "window + segment + look-ahead live in the data ring");

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

// Inclusive sum over the wave in six DPP additions (no LDS traffic): within a row of 16 lanes by row_shr 1, 2, 4, 8
// (the bank masks keep a lane from adding what lies outside its row), then a row's last lane into the rows behind it.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t lane)
{
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}
// The same with max (values >= 0: the lanes a step leaves out contribute 0).
__device__ __forceinline__ uint32_t wave_scan_max_incl(uint32_t v)
{
    const auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v, uint32_t lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v, lane), 63);
}

// result word of one position: length (9 bits, 0 = no match) | distance << 9 (15 bits) | the byte << 24
__device__ __forceinline__ uint32_t res_pack(uint32_t len, uint32_t dist, uint32_t byte) { return len | (dist << 9) | (byte << 24); }

struct HuffScratch {
    uint32_t w[2 * kNumLL];
    uint32_t cnt[260];
    uint16_t parent[2 * kNumLL];
    uint16_t order[kNumLL + 2];
    uint16_t rle[kNumLL + kNumD + 4];
    uint32_t clfreq[kNumCL + 1];
    uint32_t clcode[kNumCL + 1];
    uint8_t cllen[kNumCL + 1];
    DynHeader hdr;
};

constexpr uint32_t kMirror = 48; // the ring's first bytes again behind its end: a run of reads that starts inside the ring never wraps (below)

struct ChunkLds {
    // FIRST, at LDS address 0 (round 5): a read of the stream is then `pos & mask` and nothing else -- behind the ring and the
    // heads it was a mask AND an add of 0x14000 per read, which no ds_read's 16-bit offset field holds; the links' ring now
    // sits at 32 816, which the field does hold
    uint8_t data[kDataRing + kMirror]; // 32 KB: the stream around the segment in work, a ring by position; its first 48
                                       // bytes are mirrored behind its end so that an unaligned read -- or the four 8-byte
                                       // reads of one extension step, 32 bytes from one masked address -- never wraps
    // the block image (emission) overlays the ring and the heads: both are spent when the codes are built
    union {
        struct {
            uint16_t ring[kDfRing];   // 64 KB: distance to the previous position with the same hash, kNoLink = none
            uint32_t head[kHeadN];    // 16 KB: newest indexed position + 1 - s0, 0 = none
        } ix;
        uint32_t image[(kDfChunk + 16384u) / 4u]; // 80 KB: a block that shrinks is < 64 KiB + framing
    };
    // search results of three segments in flight (searched | being parsed | being finished); afterwards the scratch of
    // the code construction and the tile offsets
    union {
        uint32_t res[3][kDfSeg];      // 24 KB
        struct {
            HuffScratch hs;                  // literal/length tree, then the code length code
            HuffScratch hs_d;                // distance tree (another wave, at the same time)
            uint32_t tile_bits[kChunkTiles]; // bits a tile's tokens take, then their exclusive scan
        } em;
    };
    unsigned long long startbits[kChunkTiles]; // 8 KB: positions that begin a token
    unsigned long long matchbits[kChunkTiles]; // 8 KB: of those, matches
    uint32_t match_base[kChunkTiles];          // 4 KB: matches in the tiles before this one
    uint32_t freq[320];   // literal/length symbols 0..285, distance symbols at 288..317; after the codes are built: code << 8 | length
    uint8_t len[320];
    unsigned long long lastmask[kSegTiles]; // indexing: the lanes of each tile that are the last of their hash in it
    uint8_t from8[kDfSeg + 64];  // the parse: length of the last token of the cheapest way to each position of the segment
    uint8_t price_ll[288];       // the parse: prices (quarter bits) of the literal/length symbols, of the distance symbols
    uint8_t price_d[32];
    uint32_t bounds[kDfParseWaves + 1u]; // the parse: where the windows of the segment in work begin and end
    uint32_t across_done;  // tiles of the step's segment whose links are complete (the searchers of a tile wait for it)
    uint32_t finish_done;  // tiles of segment j - 2 whose tokens are counted (the parse waits for all: its prices)
    uint32_t ntok, nmatch; // the parse: tokens / matches of the segments whose tokens are counted (token_counts)
    uint32_t queue[2];    // work items handed out in the current step (the other counter is reset for the next one)
    uint32_t fixed_bits, extra_bits, total_bits, dyn_bits, use_dynamic, ghosts;
};
static_assert(sizeof(ChunkLds) <= 160 * 1024, "one workgroup per CU");

__device__ __forceinline__ uint32_t d32(const ChunkLds& L, uint32_t pos) { return *reinterpret_cast<const u32_unaligned*>(L.data + (pos & kDataMask)); }
__device__ __forceinline__ uint64_t d64(const ChunkLds& L, uint32_t pos) { return *reinterpret_cast<const u64_unaligned*>(L.data + (pos & kDataMask)); }

__device__ __forceinline__ void put_bits(uint32_t* img, uint32_t at, uint32_t bits, uint32_t nbits)
{
    if (nbits == 0u) return;
    const uint32_t widx = at >> 5, sh = at & 31u;
    const uint64_t v = (uint64_t)bits << sh;
    atomicOr(&img[widx], (uint32_t)v);
    if (v >> 32) atomicOr(&img[widx + 1u], (uint32_t)(v >> 32));
}

// ---- staging: bytes [from, from + kDfSeg) of the stream into the data ring (waves 0 and 1, 16 bytes per lane) --------
// Positions are relative to the staged piece (32 bits: a piece is at most 4 GiB - the library's staging buffers are
// far smaller); bytes behind n_in read as zero.
__device__ __forceinline__ void stage_bytes(ChunkLds& L, const uint8_t* __restrict__ in, uint64_t n_in, uint64_t from, uint32_t tid)
{
    if (tid >= kDfSeg / 16u) return;
    const uint64_t p = from + (uint64_t)tid * 16u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p + 16u <= n_in) v = *reinterpret_cast<const uint4*>(in + p);
    else if (p < n_in) {
        uint8_t b[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) b[k] = (p + k < n_in) ? in[p + k] : (uint8_t)0;
        v.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
        v.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
        v.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
        v.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((uint32_t)b[15] << 24);
    }
    const uint32_t at = (uint32_t)p & kDataMask;
    *reinterpret_cast<uint4*>(L.data + at) = v;
    if (at < kMirror) *reinterpret_cast<uint4*>(L.data + kDataRing + at) = v; // the mirror behind the ring's end
}

// ---- indexing one segment [seg0, seg0 + kDfSeg) ----------------------------------------------------------------
// link[p] = distance to the nearest earlier position whose three bytes hash alike.  Two phases.
//   A, all waves, two tiles (64 positions) each, any order: the links INSIDE a tile.  Every lane signs owner[h] with
//      its lane number; a lane that reads back another signature may have company in its tile (or was overwritten
//      from another tile: then the ballot below finds it alone); one ballot per such hash gives every member its
//      predecessor.  A lane with a predecessor writes its link, the first of its hash writes kNeedsHead, and the tile's
//      mask of "last of its hash" lanes is kept.
//   B, wave 0, tile after tile: the links ACROSS tiles.  A kNeedsHead lane takes its link from head[h]; a last lane
//      becomes head[h].  The head read of a tile is in flight while the next tile's hashes are computed.
constexpr uint32_t kNeedsHead = 0xFFFFu; // (a genuine link is at most 65 534: longer ones are cut, as in the model)
// "No older position": a distance beyond any window (round 5; it was 0).  A walk ends at a link that leaves the window,
// so the end of a chain needs no test of its own -- one compare a link instead of two.  (A genuine link of 65 534 and more
// is stored as this: it was beyond the window anyway.)
constexpr uint32_t kNoLink = 0xFFFEu;
static_assert(kNoLink > kDfMaxDist, "the end of a chain fails the window test");

__device__ __forceinline__ void index_tile_inside(ChunkLds& L, uint8_t* __restrict__ owner, uint64_t n_in, uint64_t seg0, uint32_t t, uint64_t c1,
                                                  uint32_t lane)
{
    const uint64_t p = seg0 + (uint64_t)t * 64u + lane;
    const uint32_t pos = (uint32_t)p;
    const uint32_t h = df_hash(d32(L, pos));
    const bool valid = p < c1 && p + 3u <= n_in;
    const uint8_t me = (uint8_t)lane; // two lanes of ONE tile never carry the same signature: that is all the test below needs
    if (valid) owner[h] = me;
    __builtin_amdgcn_wave_barrier();
    const bool lost = valid && owner[h] != me;
    int prevlane = -1;
    bool is_last = valid;
    uint64_t todo = __ballot(lost);
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)leader);
        const uint64_t m = __ballot(valid && h == hl);
        if (valid && h == hl) {
            const uint64_t lower = m & ((1ull << lane) - 1ull);
            prevlane = lower ? 63 - (int)__builtin_clzll(lower) : -1;
            is_last = (m >> lane) >> 1 == 0ull;
        }
        todo &= ~m;
    }
    if (valid) L.ix.ring[pos & kRingMask] = (uint16_t)(prevlane >= 0 ? lane - (uint32_t)prevlane : kNeedsHead);
    const uint64_t lastmask = __ballot(is_last);
    if (lane == 0u) L.lastmask[t] = lastmask;
}

__device__ __forceinline__ void index_segment_across(ChunkLds& L, uint64_t n_in, uint64_t s0, uint64_t seg0, uint64_t c1, uint32_t lane)
{
    bool p_need = false, p_last = false;
    uint32_t p_h = 0, p_rel1 = 0, p_pos = 0, p_q1 = 0;
    for (uint32_t t = 0; t <= kSegTiles; ++t) {
        bool need = false, last = false;
        uint32_t h = 0, rel1 = 0, pos = 0;
        if (t < kSegTiles) {
            const uint64_t p = seg0 + (uint64_t)t * 64u + lane;
            pos = (uint32_t)p;
            h = df_hash(d32(L, pos));
            const bool valid = p < c1 && p + 3u <= n_in;
            rel1 = (uint32_t)(p - s0) + 1u; // this position + 1, relative to the first indexed one
            need = valid && L.ix.ring[pos & kRingMask] == kNeedsHead;
            last = (L.lastmask[t] >> lane) & 1ull;
        }
        if (p_need) { // the previous tile: its head values have arrived
            uint32_t d = p_q1 ? p_rel1 - p_q1 : kNoLink;
            if (d > kNoLink) d = kNoLink;
            L.ix.ring[p_pos & kRingMask] = (uint16_t)d;
        }
        if (p_last) L.ix.head[p_h] = p_rel1;
        if (t >= 1u && lane == 0u) __hip_atomic_store(&L.across_done, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // tiles < t are linked
        // this tile's head reads: behind the previous tile's head writes (one wave's LDS operations keep their order)
        uint32_t q1 = 0;
        if (need) q1 = L.ix.head[h];
        p_need = need; p_last = last; p_h = h; p_rel1 = rel1; p_pos = pos; p_q1 = q1;
    }
}

// How many bytes at cand equal those at p (at most maxl).  The first eight alone -- most candidates of prose end there --
// then THIRTY-TWO a step: the four 8-byte compares of a step are independent, so a step costs one LDS round trip where
// round 3's eight-bytes-a-step loop cost four, and the walk is bound by exactly those round trips (a wave issues an
// instruction every ~6 cycles with four waves per SIMD: it waits).  Reads run up to 31 bytes past the match's end: inside
// the data ring (the look-ahead covers 258 + 8; what lies beyond only ever raises a length that is cut to maxl).
#if !defined(SNAPHASH_DF_NARROW_EXTEND)
// index of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32 as the hardware has it: __builtin_ctz of 0 is undefined)
__device__ __forceinline__ uint32_t ffbl_or_ones(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t extend_match(const ChunkLds& L, uint32_t p, uint32_t cand, uint32_t maxl)
{
    // the first eight bytes without a branch (round 5): equal bytes = lowest differing bit / 8; of the upper word `| 32` adds
    // the lower word's 32 bits and leaves "no bit set" what it is, so that eight equal bytes come out as a huge count
    const uint64_t x0 = d64(L, p) ^ d64(L, cand);
    const uint32_t lo_ = ffbl_or_ones((uint32_t)x0), hi_ = ffbl_or_ones((uint32_t)(x0 >> 32)) | 32u;
    uint32_t l = (lo_ < hi_ ? lo_ : hi_) >> 3;
    if (l < 8u) return l < maxl ? l : maxl;
    l = 8u;
    while (l < maxl) {
        // (one masked address a side: the mirror behind the ring's end covers the 32 bytes from any address inside it)
        const uint8_t* pa = L.data + ((p + l) & kDataMask);
        const uint8_t* ca = L.data + ((cand + l) & kDataMask);
        const uint64_t a0 = *reinterpret_cast<const u64_unaligned*>(pa) ^ *reinterpret_cast<const u64_unaligned*>(ca),
                       a1 = *reinterpret_cast<const u64_unaligned*>(pa + 8) ^ *reinterpret_cast<const u64_unaligned*>(ca + 8),
                       a2 = *reinterpret_cast<const u64_unaligned*>(pa + 16) ^ *reinterpret_cast<const u64_unaligned*>(ca + 16),
                       a3 = *reinterpret_cast<const u64_unaligned*>(pa + 24) ^ *reinterpret_cast<const u64_unaligned*>(ca + 24);
        if ((a0 | a1) | (a2 | a3)) {
            if (a0) l += (uint32_t)__builtin_ctzll(a0) >> 3;
            else if (a1) l += 8u + ((uint32_t)__builtin_ctzll(a1) >> 3);
            else if (a2) l += 16u + ((uint32_t)__builtin_ctzll(a2) >> 3);
            else l += 24u + ((uint32_t)__builtin_ctzll(a3) >> 3);
            break;
        }
        l += 32u;
    }
    return l < maxl ? l : maxl;
}
#else // round 3's form, for A/B (make narrow)
__device__ __forceinline__ uint32_t extend_match(const ChunkLds& L, uint32_t p, uint32_t cand, uint32_t maxl)
{
    uint32_t l = 0;
    while (l < maxl) { // eight bytes a step
        const uint64_t x = d64(L, p + l) ^ d64(L, cand + l);
        if (x) { l += (uint32_t)__builtin_ctzll(x) >> 3; break; }
        l += 8u;
    }
    return l < maxl ? l : maxl;
}
#endif

// ---- searcher: the best match of one position ------------------------------------------------------------------
__device__ __forceinline__ uint32_t search_position(const ChunkLds& L, const uint8_t* __restrict__ in, uint64_t n_in, uint64_t p64, uint64_t c1, uint32_t depth)
{
    if (p64 >= c1) return 0u;
    const uint32_t p = (uint32_t)p64;
    const uint32_t byte = L.data[p & kDataMask];
    const uint32_t maxl = (c1 - p64 < 258u) ? (uint32_t)(c1 - p64) : 258u;
    if (maxl < kDfMinMatch || p64 + 3u > n_in) return res_pack(0u, 0u, byte);
    // The walk in batches of four links: the links of a batch come out of LDS one after the other, the four candidates'
    // check words (the four bytes ending at the best length the batch started with -- a candidate that differs there
    // cannot be longer) travel together, then the survivors are extended.  Exactly the serial walk's result: the
    // check only ever spares work.
    uint32_t best = kDfMinMatch - 1u, bdist = 0u, left = depth;
    uint32_t cur = p;
    bool more = true;
#if defined(SNAPHASH_DF_SHARED_EXTEND) && !defined(SNAPHASH_DF_BRANCHY_WALK)
    // Round 5 experiment (make sharedext; MEASURED SLOWER, profiles/r05_deflate_experiments.txt): ONE copy of the extension
    // per batch, shared by the lanes' survivors.  The shipped walk writes "check, extend, update" four times a batch (a
    // copy per candidate), and a wave runs a copy whenever ANY of its lanes has a candidate that passes its check, which on
    // prose is nearly always; 3.35 G vector instructions a launch at four cycles each are half of all SIMD cycles
    // (profiles/r04_targz_text_pmc.json), so fewer copies looked like the lever.  Here a batch is: the four links, the four
    // check words, the four checks (against the best the batch STARTED with: a check only ever spares work, so an older
    // best means at most an extension that did not have to be -- the result is the serial walk's), and then a loop in
    // which every lane extends its next survivor, in order: as many rounds as the lane with the most survivors has.  The
    // budget is counted as the serial walk counts it: a candidate visited costs one link whether it passed its check or
    // not, and a cut (good / nice) ends the visit of the rest.  Byte-identical output -- and 6-11 % slower on all three
    // corpora at every depth (text 9.9 against 9.3 ms per 64 MiB at depth 32, 15.5 against 14.8 at 96): a batch's first
    // candidates mostly pass (there is no best to check against yet), so the rounds are four anyway, each with its ballot,
    // its bit scan and its selects on top.
    while (more && left) {
        uint32_t c0_ = 0, c1_ = 0, c2_ = 0, c3_ = 0, ncand = 0;
#define SNAPHASH_DF_LINK(dst)                                                       \
        {                                                                           \
            const bool want_ = more && ncand < left;                                \
            const uint32_t d_ = L.ix.ring[cur & kRingMask];                         \
            const bool ok_ = want_ && d_ != 0u && p - (cur - d_) <= kDfMaxDist;     \
            more = want_ ? ok_ : more;                                              \
            cur = ok_ ? cur - d_ : cur;                                             \
            dst = ok_ ? cur : 0u;                                                   \
            ncand += ok_ ? 1u : 0u;                                                 \
        }
        SNAPHASH_DF_LINK(c0_) SNAPHASH_DF_LINK(c1_) SNAPHASH_DF_LINK(c2_) SNAPHASH_DF_LINK(c3_)
#undef SNAPHASH_DF_LINK
        const uint32_t off = best >= 3u ? best - 3u : 0u;
        const uint32_t mine = d32(L, p + off);
        const uint8_t* gb = in + (p64 - p); // the check words come through L1/L2, beside the LDS pipe (round 3)
        const uint32_t k0 = *reinterpret_cast<const u32_unaligned*>(gb + c0_ + off), k1 = *reinterpret_cast<const u32_unaligned*>(gb + c1_ + off),
                       k2 = *reinterpret_cast<const u32_unaligned*>(gb + c2_ + off), k3 = *reinterpret_cast<const u32_unaligned*>(gb + c3_ + off);
        // who passes (bit k: candidate k); without a best of three bytes or more there is nothing to check yet
        const bool chk = best >= 3u;
        uint32_t gomask = ((!chk || k0 == mine) ? 1u : 0u) | ((!chk || k1 == mine) ? 2u : 0u) | ((!chk || k2 == mine) ? 4u : 0u) | ((!chk || k3 == mine) ? 8u : 0u);
        gomask &= (1u << ncand) - 1u;
        uint32_t kpos = 0; // candidates of the batch this lane has visited
        bool open = ncand != 0u && left != 0u;
        while (__builtin_amdgcn_ballot_w64(open) != 0ull) { // (uniform: a round of the shared extension)
            uint32_t cand = 0, l = 0;
            bool ext = false;
            if (open) {
                const uint32_t rest = gomask >> kpos;
                const uint32_t skip = rest ? (uint32_t)__builtin_ctz(rest) : ncand - kpos; // candidates that failed their check, up to the next survivor
                if (rest == 0u || left <= skip) { // no survivor left in the batch, or the budget ends in front of it
                    left -= left < skip ? left : skip;
                    open = false;
                } else {
                    const uint32_t sidx = kpos + skip;
                    left -= skip + 1u;
                    kpos = sidx + 1u;
                    cand = sidx == 0u ? c0_ : (sidx == 1u ? c1_ : (sidx == 2u ? c2_ : c3_));
                    ext = true;
                }
            }
            if (ext) {
                l = extend_match(L, p, cand, maxl);
                const bool better = l > best;
                const uint32_t cut = (l >= kDfNice || l >= maxl) ? 0u : ((l >= kDfGood && left > depth / 4u) ? depth / 4u : left);
                bdist = better ? p - cand : bdist;
                left = better ? cut : left;
                best = better ? l : best;
                open = kpos < ncand && left != 0u;
            }
        }
    }
#elif !defined(SNAPHASH_DF_BRANCHY_WALK) && !defined(SNAPHASH_DF_UNPIPELINED) // the shipped walk
    // Round 5, the walk as a two-stage pipeline.  A batch is four links -- four DEPENDENT reads of the ring -- then four
    // check words from L2, then the survivors' extensions.  The links of the NEXT batch do not need the candidates of this
    // one (only the budget does, and a link walked in vain costs nothing but itself), so they are walked while this
    // batch's check words are on their way:
    //     link 0 | links(1) extend(0) | loads(1) links(2) evaluate(1) | loads(2) links(3) evaluate(2) | ...
    // What the kernel is bound by (profiles/r05_deflate_experiments.txt): a wave issues ONE instruction of any kind every
    // four cycles, and the counters say its waves do exactly that -- 298 000 instructions in 1 157 000 cycles, 40 % of them
    // scalar exec-mask bookkeeping.  So every instruction counts, the scalar ones like the vector ones:
    //   * what a batch's links leave behind is the candidates a0..a3 and, as lane masks, whether each is DEAD -- the chain
    //     had left the window before it (kNoLink does; `dead` of link k implies `dead` of every later one: one s_or a link).
    //     The window test is the BORROW of a subtraction: `room` is how far the chain may still go back, a link's distance
    //     comes off it, and a link that leaves the window borrows (behind a borrow `room` and `cur` are garbage, and
    //     nothing of the lane's walk is used again);
    //   * the budget is a LIMIT, not a countdown: candidate k of a batch is visited iff it is alive and k < lim (lim = the
    //     budget at the batch's start); a match that cuts the budget at candidate k re-bases the limit -- lim = the cut
    //     budget + k + 1 -- and the batch's end takes the visits off in one go.  (The serial walk counts a visit at a time:
    //     same visits, same cuts.)
    //   * the first candidate is taken out of the loop: there is no best to check it against, so it is extended without a
    //     check word -- and finds the best the others are checked against;
    //   * every candidate of a batch is checked against the best the batch STARTED with (a check only ever spares work, so
    //     an older best means at most an extension that did not have to be).
    // Exactly the serial walk's result.
    (void)more;
    uint32_t room = kDfMaxDist;
    bool last_dead; // ... of the batch in work: the chain ends inside it
    {
        const uint32_t d_ = L.ix.ring[cur & kRingMask];
        cur -= d_;
        last_dead = __builtin_sub_overflow(room, d_, &room);
    }
    const uint32_t first = cur; // (garbage where last_dead)
    // (the check word is loaded whatever `dead` says: a safe address, 0, where it is)
    const uint8_t* gb = in + ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((p64 - p) >> 32)) << 32);
    uint32_t a0, a1, a2, a3;
    bool x0, x1, x2, x3; // dead
#define SNAPHASH_DF_LINK(dst, dd, prev)                                             \
    {                                                                               \
        const uint32_t d_ = L.ix.ring[cur & kRingMask];                             \
        cur -= d_; /* (the ring's index is masked: any value reads) */              \
        const bool out_ = __builtin_sub_overflow(room, d_, &room);                  \
        dd = (prev) | out_;                                                         \
        dst = dd ? 0u : cur;                                                        \
    }
#define SNAPHASH_DF_UPDATE(k, cand, l)                                                                          \
    if (l > best) { /* (a branch: most extensions end short of the best, and the update is a dozen instructions) */ \
        best = l;                                                                                               \
        bdist = p - cand;                                                                                       \
        const uint32_t rest_ = lim - ((k) + 1u); /* the budget behind this visit */                             \
        lim = ((l >= kDfNice || l >= maxl) ? 0u : ((l >= kDfGood && rest_ > depth / 4u) ? depth / 4u : rest_)) + ((k) + 1u); \
    }
    uint32_t lim = left; // the batch in work: candidate k is visited iff alive and k < lim
    SNAPHASH_DF_LINK(a0, x0, last_dead) SNAPHASH_DF_LINK(a1, x1, x0) SNAPHASH_DF_LINK(a2, x2, x1) SNAPHASH_DF_LINK(a3, x3, x2)
    if (!last_dead && lim != 0u) { // the first candidate
        const uint32_t l = extend_match(L, p, first, maxl);
        SNAPHASH_DF_UPDATE(0u, first, l)
    }
    left = (last_dead || lim == 0u) ? 0u : lim - 1u;
    bool open = !x0 && left != 0u;
    last_dead = x3;
    while (open) {
        lim = left;
        const bool nochk = best < 3u;
        const uint32_t off = nochk ? 0u : best - 3u;
        const uint32_t mine = d32(L, p + off);
        // the candidates' check words come through L1/L2 (the texture path), not from the data ring: the ring's LDS pipe is
        // what the links and the extensions use, and the two paths run side by side; a scalar base and ONE 32-bit offset a
        // load (the tile's positions share the upper half of their 64-bit position)
        const uint32_t k0 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(a0 + off)), k1 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(a1 + off)),
                       k2 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(a2 + off)), k3 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(a3 + off));
        // the next batch's links, while those are in flight
        uint32_t n0, n1, n2, n3;
        bool y0, y1, y2, y3;
        SNAPHASH_DF_LINK(n0, y0, last_dead) SNAPHASH_DF_LINK(n1, y1, y0) SNAPHASH_DF_LINK(n2, y2, y1) SNAPHASH_DF_LINK(n3, y3, y2)
        bool v0, v1, v2, v3; // visited
#define SNAPHASH_DF_EVAL(k, dd, vis, cand, chk)                                                                 \
        vis = !(dd) & ((k) < lim);                                                                              \
        if (vis & (nochk | (chk == mine))) {                                                                    \
            const uint32_t l = extend_match(L, p, cand, maxl);                                                  \
            SNAPHASH_DF_UPDATE(k, cand, l)                                                                      \
        }
        SNAPHASH_DF_EVAL(0u, x0, v0, a0, k0) SNAPHASH_DF_EVAL(1u, x1, v1, a1, k1) SNAPHASH_DF_EVAL(2u, x2, v2, a2, k2) SNAPHASH_DF_EVAL(3u, x3, v3, a3, k3)
#undef SNAPHASH_DF_EVAL
        left = lim - (v3 ? 4u : (v2 ? 3u : (v1 ? 2u : (v0 ? 1u : 0u)))); // (`visited` of candidate k implies it of the ones before)
        // on iff the chain goes on behind this batch (its last link is alive) and budget is left
        open = !x3 && left != 0u;
        a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        last_dead = y3;
    }
#undef SNAPHASH_DF_LINK
#undef SNAPHASH_DF_UPDATE
#elif !defined(SNAPHASH_DF_BRANCHY_WALK) // round 5's walk before the pipeline (make unpipelined, for A/B): check, extend, update written out per candidate
    // (The inner decisions are selects, not branches: every `if` of a divergent wave costs scalar instructions for the
    // exec mask -- the kernel issued as many of those as vector instructions -- and only the ones that skip real work
    // (a candidate that fails its check word, the extension) are worth them.)
    // (Round 5: the links' bookkeeping in lane masks.  A link is a candidate when the one before it was, the budget wants
    // it (link k of a batch: k < left) and it leads to a position inside the window -- `ok` of link k implies `ok` of link
    // k - 1, so "candidate k exists" is ok_k itself and the chain goes on iff the LAST wanted link was there.  The earlier
    // form carried `more` and a candidate count through every link as vector registers -- a 0/1 select, its conversion
    // back into a mask, an add and a compare a link: sixteen vector instructions a link where this takes ten.)
#if defined(SNAPHASH_DF_FRESH_CHECKS)
    const bool first_alone = false;
#else
    bool first_alone = true; // the first batch is ONE link: it finds a best to check the others against (below)
#endif
    while (more && left) {
        const uint32_t lim = first_alone ? 1u : left; // links this batch may take (the first `lim` of its four)
        uint32_t c0_, c1_, c2_, c3_;
        bool ok0, ok1, ok2, ok3;
#define SNAPHASH_DF_LINK(k, dst, okv, prev)                                         \
        {                                                                           \
            const uint32_t d_ = L.ix.ring[cur & kRingMask];                         \
            const uint32_t nxt_ = cur - d_;                                         \
            okv = (prev) && (k) < lim && p - nxt_ <= kDfMaxDist; /* (kNoLink fails it) */ \
            cur = nxt_; /* (behind a link that was not taken nothing of this lane's walk is used again: the chain has ended, or the \
                           budget has -- lim = left, and every candidate visited costs one -- and the ring's index is masked; the \
                           one-link first batch puts its `cur` back below) */ \
            dst = okv ? nxt_ : 0u; /* (the check word is loaded whatever ok says: a safe address) */ \
        }
        SNAPHASH_DF_LINK(0u, c0_, ok0, true)
        const uint32_t cur_after0 = cur;
        SNAPHASH_DF_LINK(1u, c1_, ok1, ok0) SNAPHASH_DF_LINK(2u, c2_, ok2, ok1) SNAPHASH_DF_LINK(3u, c3_, ok3, ok2)
#undef SNAPHASH_DF_LINK
        cur = first_alone ? cur_after0 : cur;
        more = ok3 || (lim == 3u && ok2) || (lim == 2u && ok1) || (lim == 1u && ok0); // (ok3 implies lim >= 4)
        const uint32_t off = best >= 3u ? best - 3u : 0u;
        const uint32_t mine = d32(L, p + off);
        // the candidates' check words come through L1/L2 (the texture path), not from the data ring: the ring's LDS
        // pipe is what bounds the walk (links, this position's words, the extensions), and the two paths run side by
        // side (12.6 instead of 14.5 ms per 64 MiB of text)
        // (round 5: a scalar base and ONE 32-bit offset a load -- the tile's positions share the upper half of their 64-bit
        // position, a tile never straddles 4 GiB -- where `gb + c + off` was two 64-bit vector additions a candidate)
        const uint8_t* gb = in + ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((p64 - p) >> 32)) << 32);
        const uint32_t k0 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(c0_ + off)), k1 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(c1_ + off)),
                       k2 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(c2_ + off)), k3 = *reinterpret_cast<const u32_unaligned*>(gb + (uint32_t)(c3_ + off));
        // Round 5: every candidate of a batch is checked against the best the batch STARTED with -- a check only ever spares
        // work, so an older best means at most an extension that did not have to be -- where a candidate behind one that
        // raised the best used to fetch two fresh check words from LDS, under three nested branches.  The batch that would
        // pay for that is the first (no best yet: all four extended), so the first batch is one link.
#if defined(SNAPHASH_DF_FRESH_CHECKS)
#define SNAPHASH_DF_GO(cand, chk) (best < 3u || ((off == best - 3u) ? (chk == mine) : (d32(L, cand + best - 3u) == d32(L, p + best - 3u))))
#else
        const bool nochk = best < 3u;
        first_alone = false;
#define SNAPHASH_DF_GO(cand, chk) (nochk || chk == mine)
#endif
#define SNAPHASH_DF_EVAL(okv, cand, chk)                                                                        \
        if (okv && left) {                                                                                      \
            --left;                                                                                             \
            const bool go = SNAPHASH_DF_GO(cand, chk);                                                          \
            if (go) {                                                                                           \
                const uint32_t l = extend_match(L, p, cand, maxl);                                              \
                if (l > best) { /* (a branch: most extensions end short of the best, and the update is a dozen instructions) */ \
                    best = l;                                                                                   \
                    bdist = p - cand;                                                                           \
                    left = (l >= kDfNice || l >= maxl) ? 0u : ((l >= kDfGood && left > depth / 4u) ? depth / 4u : left); \
                }                                                                                               \
            }                                                                                                   \
        }
        SNAPHASH_DF_EVAL(ok0, c0_, k0) SNAPHASH_DF_EVAL(ok1, c1_, k1) SNAPHASH_DF_EVAL(ok2, c2_, k2) SNAPHASH_DF_EVAL(ok3, c3_, k3)
#undef SNAPHASH_DF_EVAL
#undef SNAPHASH_DF_GO
    }
#else // round 3's branches (make branchy, for A/B)
    while (more && left) {
        uint32_t c0_ = 0, c1_ = 0, c2_ = 0, c3_ = 0, ncand = 0;
#define SNAPHASH_DF_LINK(dst)                                                       \
        if (more && ncand < left) {                                                 \
            const uint32_t d_ = L.ix.ring[cur & kRingMask];                         \
            if (d_ == 0u || p - (cur - d_) > kDfMaxDist) more = false;              \
            else { cur -= d_; dst = cur; ++ncand; }                                 \
        }
        SNAPHASH_DF_LINK(c0_) SNAPHASH_DF_LINK(c1_) SNAPHASH_DF_LINK(c2_) SNAPHASH_DF_LINK(c3_)
#undef SNAPHASH_DF_LINK
        const uint32_t off = best >= 3u ? best - 3u : 0u;
        const uint32_t mine = d32(L, p + off);
        // the candidates' check words come through L1/L2 (the texture path), not from the data ring: the ring's LDS
        // pipe is what bounds the walk (links, this position's words, the extensions), and the two paths run side by
        // side (12.6 instead of 14.5 ms per 64 MiB of text)
        const uint8_t* gb = in + (p64 - p);
        const uint32_t k0 = *reinterpret_cast<const u32_unaligned*>(gb + c0_ + off), k1 = *reinterpret_cast<const u32_unaligned*>(gb + c1_ + off),
                       k2 = *reinterpret_cast<const u32_unaligned*>(gb + c2_ + off), k3 = *reinterpret_cast<const u32_unaligned*>(gb + c3_ + off);
#define SNAPHASH_DF_EVAL(k, cand, chk)                                                                         \
        if (k < ncand && left) {                                                                                \
            --left;                                                                                             \
            bool go = true;                                                                                     \
            if (best >= 3u) go = (off == best - 3u) ? (chk == mine) : (d32(L, cand + best - 3u) == d32(L, p + best - 3u)); \
            if (go) {                                                                                           \
                const uint32_t l = extend_match(L, p, cand, maxl);                                              \
                if (l > best) {                                                                                 \
                    best = l;                                                                                   \
                    bdist = p - cand;                                                                           \
                    if (l >= kDfNice || l >= maxl) left = 0u;                                                   \
                    else if (l >= kDfGood && left > depth / 4u) left = depth / 4u;                              \
                }                                                                                               \
            }                                                                                                   \
        }
        SNAPHASH_DF_EVAL(0u, c0_, k0) SNAPHASH_DF_EVAL(1u, c1_, k1) SNAPHASH_DF_EVAL(2u, c2_, k2) SNAPHASH_DF_EVAL(3u, c3_, k3)
#undef SNAPHASH_DF_EVAL
    }
#endif
    if (best == 3u && bdist > kDfTooFar) best = 0u;
    return best >= kDfMinMatch ? res_pack(best, bdist, byte) : res_pack(0u, 0u, byte);
}

// ---- searcher, round 5 experiment (make pairtiles): TWO tiles per wave at a time, a lane walking the chains of two
// positions (p and p + 64) side by side.  The walk of a position is unchanged -- same links, same candidates, same
// result -- but its dependent LDS reads (a link, the next link, ...) now have an independent twin in flight: the kernel
// is latency-bound (a SIMD issues one instruction in ~7 cycles with its four waves: LDS holds one workgroup per CU), so
// memory-level parallelism inside a wave is what is left to add.
#if defined(SNAPHASH_DF_PAIR_TILES)
__device__ __forceinline__ void search_pair(const ChunkLds& L, const uint8_t* __restrict__ in, uint64_t n_in, uint64_t pa64, uint64_t pb64, uint64_t c1,
                                            uint32_t depth, uint32_t& ra, uint32_t& rb)
{
    const uint64_t p64[2] = {pa64, pb64};
    uint32_t p[2], byte[2], maxl[2], best[2], bdist[2], left[2], cur[2], res[2];
    bool more[2], live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        p[u] = (uint32_t)p64[u];
        const bool inside = p64[u] < c1;
        byte[u] = inside ? L.data[p[u] & kDataMask] : 0u;
        maxl[u] = inside ? ((c1 - p64[u] < 258u) ? (uint32_t)(c1 - p64[u]) : 258u) : 0u;
        live[u] = inside && maxl[u] >= kDfMinMatch && p64[u] + 3u <= n_in;
        res[u] = inside ? res_pack(0u, 0u, byte[u]) : 0u;
        best[u] = kDfMinMatch - 1u; bdist[u] = 0u; left[u] = live[u] ? depth : 0u; cur[u] = p[u]; more[u] = live[u];
    }
    const uint8_t* gb = in + (p64[0] - p[0]); // (both positions lie in one staged piece: the same base)
    while ((more[0] && left[0]) || (more[1] && left[1])) {
        uint32_t c[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, ncand[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool want = more[u] && ncand[u] < left[u];
                const uint32_t d = L.ix.ring[cur[u] & kRingMask];
                const bool ok = want && d != 0u && p[u] - (cur[u] - d) <= kDfMaxDist;
                more[u] = want ? ok : more[u];
                cur[u] = ok ? cur[u] - d : cur[u];
                c[u][k] = ok ? cur[u] : 0u;
                ncand[u] += ok ? 1u : 0u;
            }
        }
        uint32_t off[2], mine[2], kw[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            off[u] = best[u] >= 3u ? best[u] - 3u : 0u;
            mine[u] = d32(L, p[u] + off[u]);
#pragma unroll
            for (int k = 0; k < 4; ++k) kw[u][k] = *reinterpret_cast<const u32_unaligned*>(gb + c[u][k] + off[u]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if ((uint32_t)k < ncand[u] && left[u]) {
                    --left[u];
                    bool go = true;
                    if (best[u] >= 3u) go = (off[u] == best[u] - 3u) ? (kw[u][k] == mine[u]) : (d32(L, c[u][k] + best[u] - 3u) == d32(L, p[u] + best[u] - 3u));
                    if (go) {
                        const uint32_t l = extend_match(L, p[u], c[u][k], maxl[u]);
                        const bool better = l > best[u];
                        const uint32_t cut = (l >= kDfNice || l >= maxl[u]) ? 0u : ((l >= kDfGood && left[u] > depth / 4u) ? depth / 4u : left[u]);
                        bdist[u] = better ? p[u] - c[u][k] : bdist[u];
                        left[u] = better ? cut : left[u];
                        best[u] = better ? l : best[u];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (live[u]) {
            if (best[u] == 3u && bdist[u] > kDfTooFar) best[u] = 0u;
            res[u] = best[u] >= kDfMinMatch ? res_pack(best[u], bdist[u], byte[u]) : res_pack(0u, 0u, byte[u]);
        }
    }
    ra = res[0];
    rb = res[1];
}
#endif // SNAPHASH_DF_PAIR_TILES

// ---- searcher, round 4 experiment (make stream; MEASURED SLOWER, profiles/r04_deflate_stream_search.txt): the lanes of a
// wave STREAM through positions.  search_position above gives a wave one tile and the tile costs what its slowest lane
// costs: on content with long matches a few lanes walk 8 batches while most are done after one or two (sources: 17 ms per
// 64 MiB against 10 for text).  Here a lane that is done with its position takes the next one of the
// wave's pool (a tile from the step's queue, as before) at the top of the next batch: the wave stays full until the queue
// is empty, and a step costs the lanes' average instead of a maximum per tile.  The walk of a position is unchanged and the
// output is byte for byte the same -- and the kernel is 10-20 % SLOWER on all three corpora (12.1 / 18.9 / 12.3 ms against
// 10.0 / 16.8 / 11.4): in a tile adjacent lanes hold adjacent positions, whose chains visit adjacent candidates (one
// cache line of check words, neighbouring LDS banks), and a wave of unrelated positions gives that up for its fuller lanes.
#if defined(SNAPHASH_DF_STREAM_SEARCH)
__device__ __forceinline__ void search_segment_stream(ChunkLds& L, const uint8_t* __restrict__ in, uint64_t n_in, uint64_t seg64, uint64_t c1,
                                                      uint32_t* __restrict__ res, uint32_t lane, uint32_t depth)
{
    uint32_t pool = 0, pool_end = 0; // the wave's pool: positions [pool, pool_end) of the segment (uniform)
    bool tiles_left = true;
    bool active = false;             // this lane is in the middle of a position's walk
    uint32_t rel = 0, p = 0, byte = 0, maxl = 0, best = 0, bdist = 0, left = 0, cur = 0;
    bool more = false;
    const uint8_t* gb = in;
    for (;;) {
        uint64_t idle = __ballot(!active);
        while (idle) { // every idle lane takes the next position of the pool; an empty pool takes the next tile of the queue
            if (pool == pool_end) {
                if (!tiles_left) break;
                uint32_t item = 0;
                if (lane == 0u) item = atomicAdd(&L.queue[1], 1u);
                item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
                if (item >= kSegTiles) { tiles_left = false; break; }
                while (__hip_atomic_load(&L.across_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= item) __builtin_amdgcn_s_sleep(1);
                pool = item * 64u;
                pool_end = pool + 64u;
            }
            const uint32_t navail = pool_end - pool;
            const uint32_t rank = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
            if (!active && rank < navail) {
                rel = pool + rank;
                const uint64_t p64 = seg64 + rel;
                p = (uint32_t)p64;
                gb = in + (p64 - p);
                if (p64 >= c1) {
                    res[rel] = 0u;
                } else {
                    byte = L.data[p & kDataMask];
                    maxl = (c1 - p64 < 258u) ? (uint32_t)(c1 - p64) : 258u;
                    if (maxl < kDfMinMatch || p64 + 3u > n_in) {
                        res[rel] = res_pack(0u, 0u, byte);
                    } else {
                        best = kDfMinMatch - 1u; bdist = 0u; left = depth; cur = p; more = true;
                        active = true;
                    }
                }
            }
            const uint32_t nidle = (uint32_t)__builtin_popcountll(idle);
            pool += nidle < navail ? nidle : navail;
            idle = __ballot(!active); // (a position that needs no walk leaves its lane idle: it takes another)
        }
        if (__ballot(active) == 0ull) break;
        if (active) { // one batch of the walk: search_position's loop body
            uint32_t c0_ = 0, c1_ = 0, c2_ = 0, c3_ = 0, ncand = 0;
#define SNAPHASH_DF_LINK(dst)                                                       \
            if (more && ncand < left) {                                             \
                const uint32_t d_ = L.ix.ring[cur & kRingMask];                     \
                if (d_ == 0u || p - (cur - d_) > kDfMaxDist) more = false;          \
                else { cur -= d_; dst = cur; ++ncand; }                             \
            }
            SNAPHASH_DF_LINK(c0_) SNAPHASH_DF_LINK(c1_) SNAPHASH_DF_LINK(c2_) SNAPHASH_DF_LINK(c3_)
#undef SNAPHASH_DF_LINK
            const uint32_t off = best >= 3u ? best - 3u : 0u;
            const uint32_t mine = d32(L, p + off);
            const uint32_t k0 = *reinterpret_cast<const u32_unaligned*>(gb + c0_ + off), k1 = *reinterpret_cast<const u32_unaligned*>(gb + c1_ + off),
                           k2 = *reinterpret_cast<const u32_unaligned*>(gb + c2_ + off), k3 = *reinterpret_cast<const u32_unaligned*>(gb + c3_ + off);
#define SNAPHASH_DF_EVAL(k, cand, chk)                                                                              \
            if (k < ncand && left) {                                                                                \
                --left;                                                                                             \
                bool go = true;                                                                                     \
                if (best >= 3u) go = (off == best - 3u) ? (chk == mine) : (d32(L, cand + best - 3u) == d32(L, p + best - 3u)); \
                if (go) {                                                                                           \
                    const uint32_t l = extend_match(L, p, cand, maxl);                                              \
                    if (l > best) {                                                                                 \
                        best = l;                                                                                   \
                        bdist = p - cand;                                                                           \
                        if (l >= kDfNice || l >= maxl) left = 0u;                                                   \
                        else if (l >= kDfGood && left > depth / 4u) left = depth / 4u;                              \
                    }                                                                                               \
                }                                                                                                   \
            }
            SNAPHASH_DF_EVAL(0u, c0_, k0) SNAPHASH_DF_EVAL(1u, c1_, k1) SNAPHASH_DF_EVAL(2u, c2_, k2) SNAPHASH_DF_EVAL(3u, c3_, k3)
#undef SNAPHASH_DF_EVAL
            if (!(more && left)) { // the walk is over: the position's result
                if (best == 3u && bdist > kDfTooFar) best = 0u;
                res[rel] = best >= kDfMinMatch ? res_pack(best, bdist, byte) : res_pack(0u, 0u, byte);
                active = false;
            }
        }
    }
}
#endif // SNAPHASH_DF_STREAM_SEARCH

// ---- parser (waves 1-4): the price parse of one window of a segment (deflate_core.h; tests/deflate_model.h is its
// serial form).  In: the search results of the segment.  Out: startbits / matchbits / per-tile match counts of its
// tiles, and in the result word of every position that begins a match the length the parse chose (it may be shorter
// than the one found).  L.ntok / L.nmatch: tokens / matches of the chunk's earlier segments (L.freq holds exactly
// their symbols).
//
// One step of the parse's window: w = min(w, cand) in lanes offset .. offset + size - 1 (uniform; the size is the low
// six bits of `size`: s_bfm_b64 builds the mask straight into EXEC), then every lane takes its upper neighbour's value
// (wave_shl:1; lane 63 keeps its own).  Only where the whole wave is active: EXEC is all ones behind it.  (The s_nop:
// a DPP operation reads a VGPR no sooner than two wait states after a VALU wrote it.)
__device__ __forceinline__ uint32_t min_and_shift(uint32_t w, uint32_t cand, uint32_t size, uint32_t offset)
{
    asm volatile("s_bfm_b64 exec, %2, %3\n\tv_min_u32 %0, %0, %1\n\ts_mov_b64 exec, -1\n\ts_nop 0\n\t"
                 "v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(w) : "v"(cand), "s"(size), "s"(offset));
    return w;
}
// One window [wa, wb) of segment [seg_rel0, seg_rel0 + m): every wave that parses writes the same price tables
// (same inputs, same values: whichever write lands last changes nothing), then works on positions of its own.
// Where window wi of the segment's m positions begins (deflate_core.h): at its nominal place, or up to kDfCutSpan - 1
// in front of it at the last position no match from further in front reaches across: six tiles, a running maximum of
// position + reach.  Worked out at the top of the step, before any window is parsed (a parse writes the lengths it
// chose over the ones found).
__device__ __forceinline__ uint32_t window_bound(const uint32_t* res, uint32_t m, uint32_t wi, uint32_t lane)
{
    if (wi == 0u) return 0u;
    const uint32_t b = df_window_begin(wi);
    if (b >= m) return m;
    static_assert(kDfCutSpan == 64u, "the candidates are the positions of one tile");
    const uint32_t t1 = b >> 6; // the nominal place is a whole tile; a match is at most 258 long: tiles t1 - 6 .. t1 - 1 say it all
    uint32_t running = 0, v = 0;
    for (uint32_t t = t1 - 6u; t < t1; ++t) {
        const uint32_t ppos = t * 64u + lane, ml = res[ppos] & 0x1ffu, room = m - ppos;
        const uint32_t mlc = ml < room ? ml : room;
        v = ppos + (mlc ? mlc : 1u);
        if (t + 1u < t1) {
            const uint32_t top = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max_incl(v), 63);
            running = running > top ? running : top;
        }
    }
    const uint32_t incl = wave_scan_max_incl(v); // the last tile: positions b - 64 .. b - 1
    const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xf, 0xf, false); // wave_shr:1: the lanes in front
    const uint32_t far = running > below ? running : below;
    const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if ((running > all ? running : all) <= b) return b;
    const uint64_t ok = __ballot(lane >= 1u && far <= (t1 - 1u) * 64u + lane);
    return ok ? (t1 - 1u) * 64u + 63u - (uint32_t)__builtin_clzll(ok) : b;
}

__device__ __forceinline__ void parse_window(ChunkLds& L, uint32_t* res, uint32_t seg_rel0, uint32_t m, uint32_t wi, uint32_t lane)
{
    const uint32_t tile0 = seg_rel0 >> 6;
    const uint32_t wa = L.bounds[wi], wb = L.bounds[wi + 1u];
    const uint32_t ntok = L.ntok, nmatch = L.nmatch;
    if (ntok < kDfPriceWarm) {
        for (uint32_t k = lane; k < 288u; k += 64u) L.price_ll[k] = (uint8_t)(k < 256u ? kDfLitPrice0 : kDfLenPrice0);
        if (lane < 32u) L.price_d[lane] = (uint8_t)kDfDistPrice0;
    } else {
        const uint32_t lt = df_ilog(ntok + 1u), dt = df_ilog(nmatch + 1u);
        for (uint32_t k = lane; k < 288u; k += 64u) L.price_ll[k] = (uint8_t)df_price(k < (uint32_t)kNumLL ? L.freq[k] : 0u, lt, kDfLLCap);
        if (lane < 32u) L.price_d[lane] = (uint8_t)(nmatch ? df_price(lane < (uint32_t)kNumD ? L.freq[288u + lane] : 0u, dt, kDfDistCap) : kDfDistPrice0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wa >= wb) return;
    uint32_t lenkey = 0xffffff00u; // this lane's token length is its number: the price of that length and 255 - it
    if (lane >= 3u) {
        uint32_t sy, eb, ev;
        len_symbol(lane, sy, eb, ev);
        lenkey = df_key((uint32_t)L.price_ll[sy] + eb * kDfPriceUnit, lane);
    }
    const uint32_t tile_end = (wb + 63u) >> 6;

    uint32_t pos = wa;
    while (pos < wb) { // a run: from pos to the window's end or to the first match of kDfLongMatch or more
        const uint32_t run0 = pos;
        uint32_t w = lane == 0u ? 0u : 0xffffffffu; // lane l: the key of the cheapest way found so far to position (current) + l
        uint32_t lit_key = 0xffffffffu;             // ... except the way through the previous position's literal (the same in every lane)
        asm volatile("" : "+v"(lit_key));           // (a VGPR: see the step)
        uint32_t end = wb, long_len = 0;
        for (uint32_t t = pos >> 6; t < tile_end; ++t) {
            const uint32_t r = res[t * 64u + lane];
            const uint32_t ml = r & 0x1ffu, dist = (r >> 9) & 0x7fffu, byte = r >> 24;
            const uint32_t ppos = t * 64u + lane;
            const uint32_t room = ppos < wb ? wb - ppos : 0u;
            const uint32_t mlc = ml < room ? ml : room;
            const uint32_t min_l = dist > kDfTooFar ? 4u : 3u;
            const uint32_t hi = mlc < kDfLongMatch - 1u ? mlc : kDfLongMatch - 1u;
            const uint32_t size = hi >= min_l ? hi - min_l + 1u : 0u;
            uint32_t ds, de, dv;
            dist_symbol(dist ? dist : 1u, ds, de, dv);
            const uint32_t dp = (uint32_t)L.price_d[ds] + de * kDfPriceUnit;
            const uint32_t lp = L.price_ll[byte];
            // what a step needs of its position, one word each so that nothing has to be taken apart on the way: the
            // literal's price in key form, the distance's price in key form, the lanes its lengths go to
            const uint32_t lit_word = (lp << 8) | 254u, dist_word = dp << 8, lanes_word = size | (min_l << 6);
            const uint32_t k0 = (t == (pos >> 6)) ? (pos & 63u) : 0u;
            const uint32_t tile_n = (wb - t * 64u < 64u) ? wb - t * 64u : 64u;
            const uint64_t fm = __ballot(mlc >= kDfLongMatch && lane >= k0 && lane < tile_n);
            const uint32_t k1 = fm ? (uint32_t)__builtin_ctzll(fm) : tile_n;
            uint32_t f = 0;
            for (uint32_t k = k0; k < k1; ++k) {
                // The step's arithmetic is vector arithmetic on uniform values (every lane computes the same key): measured
                // 6 % faster than the same chain on the scalar unit, which has to take the packed word apart first.
                const uint32_t s_lit = (uint32_t)__builtin_amdgcn_readlane((int)lit_word, (int)k);
                const uint32_t s_dist = (uint32_t)__builtin_amdgcn_readlane((int)dist_word, (int)k);
                const uint32_t s_lanes = (uint32_t)__builtin_amdgcn_readlane((int)lanes_word, (int)k);
                const uint32_t s_w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
                const uint32_t key = s_w0 < lit_key ? s_w0 : lit_key; // lit_key lives in a VGPR: v_min_u32
                f = lane == k ? key : f;
                const uint32_t key_hi = key & 0xffffff00u;
                lit_key = key_hi + s_lit;
                const uint32_t cand = key_hi + s_dist + lenkey;
                w = min_and_shift(w, cand, s_lanes, s_lanes >> 6); // lane 63 keeps its value: no token is longer than 62, so nothing ever lowers it from "no way yet"
            }
            if (ppos > run0 && lane < k1) L.from8[ppos] = (uint8_t)(255u - (f & 255u)); // (the run's first position belongs to whoever ended there)
            if (fm) {
                end = t * 64u + k1;
                long_len = (uint32_t)__builtin_amdgcn_readlane((int)mlc, (int)k1);
                break;
            }
        }
        if (lane == 0u && end > run0) L.from8[end] = (uint8_t)(255u - ((w < lit_key ? w : lit_key) & 255u)); // lane 0 is position end now
        // the way back: from end to run0, a hop per token
        {
            uint32_t p = end, from_tile = 0xffffffffu, mask_tile = 0xffffffffu, ft = 0;
            uint64_t cur = 0;
            while (p > run0) {
                const uint32_t tp = p >> 6;
                if (tp != from_tile) { ft = L.from8[tp * 64u + lane]; from_tile = tp; }
                p -= (uint32_t)__builtin_amdgcn_readlane((int)ft, (int)(p & 63u));
                const uint32_t ts = p >> 6;
                if (ts != mask_tile) {
                    if (cur != 0ull && lane == 0u) atomicOr(&L.startbits[tile0 + mask_tile], cur);
                    mask_tile = ts;
                    cur = 0;
                }
                cur |= 1ull << (p & 63u);
            }
            if (cur != 0ull && lane == 0u) atomicOr(&L.startbits[tile0 + mask_tile], cur);
        }
        if (end < wb) { // the long match
            if (lane == 0u) atomicOr(&L.startbits[tile0 + (end >> 6)], 1ull << (end & 63u));
            pos = end + long_len;
        } else {
            pos = wb;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // The tokens of the window.  A token ends where the next begins or where the window ends: lengths, which tokens are
    // matches, the chosen lengths into the result words.  A window's first and last tile may be shared with its
    // neighbours, which are at work at the same time: only the bits of this window's positions are looked at, and what
    // goes into the per-tile words goes there atomically.
    const auto starts_of = [&](uint32_t t) -> uint64_t {
        uint64_t v = t < tile_end ? L.startbits[tile0 + t] : 0ull;
        if ((wb >> 6) == t) v = (v & ((1ull << (wb & 63u)) - 1ull)) | (1ull << (wb & 63u)); // the window's end counts as a start here; behind it is the neighbour's
        return v;
    };
    uint64_t sm_v = starts_of(wa >> 6);
    for (uint32_t t = wa >> 6; t < tile_end; ++t) {
        const uint64_t nx_v = starts_of(t + 1u);
        const uint32_t r = res[t * 64u + lane];
        const uint32_t ppos = t * 64u + lane;
        const bool my_start = ppos >= wa && ppos < wb && ((sm_v >> lane) & 1ull);
        const uint32_t room = ppos < wb ? wb - ppos : 0u;
        const uint32_t ml = r & 0x1ffu;
        const uint32_t mlc = ml < room ? ml : room;
        const uint64_t above = lane < 63u ? sm_v >> (lane + 1u) : 0ull;
        uint32_t tl = above ? (uint32_t)__builtin_ctzll(above) + 1u : (64u - lane) + (nx_v ? (uint32_t)__builtin_ctzll(nx_v) : 0u);
        if (mlc >= kDfLongMatch) tl = mlc; // a long match was taken whole
        const bool my_match = my_start && tl >= kDfMinMatch;
        const uint64_t mk = __ballot(my_match);
        if (my_match) res[t * 64u + lane] = (r & ~0x1ffu) | tl;
        if (lane == 0u && mk != 0ull) {
            atomicOr(&L.matchbits[tile0 + t], mk);
            atomicAdd(&L.match_base[tile0 + t], (uint32_t)__builtin_popcountll(mk)); // a count: token_counts makes it the base
        }
        sm_v = nx_v;
    }
}

// A step later (wave 0; every window is done): the segment's matches per tile become bases into the token scratch,
// its tokens and matches join the chunk's counts.
__device__ __forceinline__ void token_counts(ChunkLds& L, uint32_t seg_rel0, uint32_t m, uint32_t lane)
{
    const uint32_t ntile = (m + 63u) >> 6, tile0 = seg_rel0 >> 6;
    const uint32_t cnt = lane < ntile ? L.match_base[tile0 + lane] : 0u;
    const uint32_t sts = lane < ntile ? (uint32_t)__builtin_popcountll(L.startbits[tile0 + lane]) : 0u;
    const uint32_t incl = wave_scan_incl(cnt, lane);
    const uint32_t nmatch = L.nmatch, ntok = L.ntok;
    if (lane < ntile) L.match_base[tile0 + lane] = nmatch + incl - cnt;
    const uint32_t all_m = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), all_t = wave_sum(sts, lane);
    if (lane == 0u) { L.nmatch = nmatch + all_m; L.ntok = ntok + all_t; }
}

// ---- code construction by a whole wave: the same results as the serial routines of deflate_core.h ----------------
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// huff_lengths(): the stable sort becomes a rank count (lane = up to five symbols, every lane reads every key), the
// two-queue merge stays one lane's job, the depths are walked leaf by leaf in parallel.  n <= 320.
__device__ __forceinline__ void huff_lengths_wave(const uint32_t* freq, uint32_t n, uint32_t max_bits, uint8_t* len, uint32_t* w, uint16_t* parent,
                                                  uint16_t* order, uint32_t lane)
{
    constexpr uint32_t kPer = 5;
    for (uint32_t shift = 0;; ++shift) {
        uint32_t key[kPer], rank[kPer];
        uint32_t zeros = 0;
#pragma unroll
        for (uint32_t s = 0; s < kPer; ++s) {
            const uint32_t i = lane + 64u * s;
            uint32_t f = 0;
            if (i < n) {
                f = freq[i] >> shift;
                if (f == 0u && freq[i] != 0u) f = 1u;
                if (f > 0xffffu) f = 0xffffu;
                w[n + i] = f; // (the upper half of w: the merge's internal nodes come later)
                zeros += f == 0u ? 1u : 0u;
            }
            key[s] = f;
            rank[s] = 0;
        }
        wave_lds_sync();
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t kj = w[n + j];
#pragma unroll
            for (uint32_t s = 0; s < kPer; ++s) rank[s] += (kj < key[s] || (kj == key[s] && j < lane + 64u * s)) ? 1u : 0u;
        }
        const uint32_t z = wave_sum(zeros, lane); // unused symbols sort first
        const uint32_t m = n - z;
#pragma unroll
        for (uint32_t s = 0; s < kPer; ++s) {
            const uint32_t i = lane + 64u * s;
            if (i < n) {
                if (key[s] == 0u) len[i] = 0u;
                else if (m == 1u) len[i] = 1u;
                else { order[rank[s]] = (uint16_t)i; w[rank[s]] = key[s]; }
            }
        }
        if (m <= 1u) return;
        wave_lds_sync();
        const uint32_t root = n + m - 2u;
        if (lane == 0u) { // two-queue merge: leaves z .. n-1 in sorted order, internal nodes n .. n+m-2.  The heads of both
                          // queues are kept in registers (an empty queue's head weighs more than anything): a pick costs one
                          // LDS read, and the leaf queue's is issued a pick ahead
            constexpr uint32_t kNone = 0xffffffffu;
            uint32_t li = z, ii = n, nn = n;
            uint32_t wl = w[z], wl1 = z + 1u < n ? w[z + 1u] : kNone, wi = kNone;
            for (uint32_t k = 0; k + 1u < m; ++k) {
                uint32_t a, b, wa, wb;
                if (wl <= wi) { a = li++; wa = wl; wl = wl1; wl1 = li + 1u < n ? w[li + 1u] : kNone; }
                else { a = ii++; wa = wi; wi = ii < nn ? w[ii] : kNone; }
                if (wl <= wi) { b = li++; wb = wl; wl = wl1; wl1 = li + 1u < n ? w[li + 1u] : kNone; }
                else { b = ii++; wb = wi; wi = ii < nn ? w[ii] : kNone; }
                const uint32_t sum = wa + wb;
                w[nn] = sum;
                parent[a] = (uint16_t)nn;
                parent[b] = (uint16_t)nn;
                if (ii == nn) wi = sum; // the internal queue was empty: the new node is its head
                ++nn;
            }
        }
        wave_lds_sync();
        uint32_t depth[kPer];
        bool deep = false;
#pragma unroll
        for (uint32_t s = 0; s < kPer; ++s) {
            const uint32_t i = z + lane + 64u * s; // a leaf
            uint32_t d = 0;
            if (i < n) {
                uint32_t node = i;
                while (node != root && d <= max_bits) { node = parent[node]; ++d; }
                deep = deep || d > max_bits;
            }
            depth[s] = d;
        }
        if (__ballot(deep) != 0ull) continue; // deeper than the limit: halve the weights and build again
#pragma unroll
        for (uint32_t s = 0; s < kPer; ++s) {
            const uint32_t i = z + lane + 64u * s;
            if (i < n) len[order[i]] = (uint8_t)depth[s];
        }
        return;
    }
}
// huff_codes(): a symbol's code is the first code of its length plus the symbols of that length in front of it
__device__ __forceinline__ void huff_codes_wave(const uint8_t* len, uint32_t n, uint32_t* out, uint32_t lane)
{
    constexpr uint32_t kPer = 5;
    uint32_t l[kPer], code[kPer];
#pragma unroll
    for (uint32_t s = 0; s < kPer; ++s) { const uint32_t i = lane + 64u * s; l[s] = i < n ? len[i] : 0u; code[s] = 0; }
    uint32_t first = 0, before = 0; // first code of the length in work, symbols of the previous length
    for (uint32_t b = 1; b <= (uint32_t)kMaxBits; ++b) {
        first = (first + before) << 1;
        uint32_t seen = 0;
#pragma unroll
        for (uint32_t s = 0; s < kPer; ++s) {
            const uint64_t mk = __ballot(l[s] == b);
            if (l[s] == b) code[s] = first + seen + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
            seen += (uint32_t)__builtin_popcountll(mk);
        }
        before = seen;
    }
#pragma unroll
    for (uint32_t s = 0; s < kPer; ++s) {
        const uint32_t i = lane + 64u * s;
        if (i < n) out[i] = l[s] ? ((rev_bits(code[s], l[s]) << 8) | l[s]) : 0u;
    }
}

// ---- finisher: the tokens of one parsed tile -> symbol counts, match tokens to the scratch, block prices -------------
__device__ __forceinline__ void finish_tile(ChunkLds& L, const uint32_t* __restrict__ res, uint32_t t, uint32_t tile, uint32_t lane,
                                            uint32_t& fixed_lane, uint32_t& extra_lane, uint32_t* __restrict__ tok)
{
    const uint32_t r = res[t * 64u + lane];
    const uint32_t mlen = r & 0x1ffu, dist = (r >> 9) & 0x7fffu, byte = r >> 24;
    const uint64_t start_mask = L.startbits[tile], match_mask = L.matchbits[tile];
    const bool my_start = (start_mask >> lane) & 1ull;
    const bool my_match = (match_mask >> lane) & 1ull;
    if (!my_start) return;
    uint32_t ls = byte, le = 0, lv = 0, ds = 0, de = 0, dv = 0;
    if (my_match) { len_symbol(mlen, ls, le, lv); dist_symbol(dist, ds, de, dv); }
    atomicAdd(&L.freq[ls], 1u);
    fixed_lane += fixed_ll_bits(ls);
    if (my_match) {
        atomicAdd(&L.freq[288u + ds], 1u);
        fixed_lane += 5u;
        extra_lane += le + de;
        const uint32_t rank = (uint32_t)__builtin_popcountll(match_mask & ((1ull << lane) - 1ull));
        tok[L.match_base[tile] + rank] = mlen | (dist << 16);
    }
}

// code and length of the token that starts at this lane's position (part A = literal/length code + extra bits,
// part B = distance code + extra bits; each at most 28 bits)
__device__ __forceinline__ void token_bits(const ChunkLds& L, bool dynamic, bool my_start, bool my_match, uint32_t byte, uint32_t t, uint32_t& ba,
                                           uint32_t& na, uint32_t& bb, uint32_t& nb)
{
    ba = na = bb = nb = 0u;
    if (!my_start) return;
    if (dynamic) {
        uint32_t ls = byte, le = 0, lv = 0, ds = 0, de = 0, dv = 0;
        if (my_match) { len_symbol(t & 0x1ffu, ls, le, lv); dist_symbol(t >> 16, ds, de, dv); }
        const uint32_t ca = L.freq[ls];
        ba = (ca >> 8) | (lv << (ca & 0xffu));
        na = (ca & 0xffu) + le;
        if (my_match) {
            const uint32_t cb = L.freq[288u + ds];
            bb = (cb >> 8) | (dv << (cb & 0xffu));
            nb = (cb & 0xffu) + de;
        }
    } else if (my_match) {
        enc_match(t & 0x1ffu, t >> 16, ba, na);
    } else {
        enc_literal(byte, ba, na);
    }
}

} // namespace

// One workgroup per chunk.  in: the staged stream (readable up to n_in + 8); slots: nchunks * kDeflateSlot bytes;
// sizes[c]: bytes chunk c produced; toks: nchunks * kDeflateTokWords words of scratch (the chunk's match tokens).
__global__ __launch_bounds__(1024) void deflate_chunks_kernel(const uint8_t* __restrict__ in, uint64_t n_in, uint8_t* __restrict__ slots,
                                                              uint32_t* __restrict__ sizes, uint32_t* __restrict__ toks, uint32_t chunk0,
                                                              uint32_t nchunks, uint32_t nx, uint32_t depth)
{
    __shared__ ChunkLds L;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    // Workgroups go round the device's nx XCDs (workgroup i to XCD i % nx; nx = 8 in SPX mode, asked of the device by the
    // host), each with an L2 of its own; a chunk's window is the 28 KiB the chunk in front of it also stages.  So an XCD
    // takes a RUN of consecutive chunks, not every nx-th: the window is then in its L2 already (or on its way there) when
    // the neighbour asks for it.  The mapping is a bijection for any nx, so a wrong nx costs locality, never output.
    const uint32_t per = gridDim.x / nx, extra = gridDim.x % nx;
    const uint32_t xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const uint32_t c = chunk0 + xcd * per + (xcd < extra ? xcd : extra) + slot;
    if (c >= nchunks) return;
    const uint64_t c0 = (uint64_t)c * kDfChunk;
    const uint32_t len = (uint32_t)((n_in - c0 < kDfChunk) ? (n_in - c0) : kDfChunk);
    const uint64_t c1 = c0 + len;
    const uint32_t nwin = c0 >= kDfMaxDist ? kWinSegs : 0u; // c0 is a multiple of 64 KiB: all of the window or none
    const uint64_t s0 = c0 - (uint64_t)nwin * kDfSeg;
    const uint32_t nseg = (len + kDfSeg - 1u) / kDfSeg;
    uint8_t* dst = slots + (uint64_t)c * kDeflateSlot;
    uint32_t* tok = toks + (uint64_t)c * kDeflateTokWords;

    for (uint32_t i = threadIdx.x; i < kHeadN; i += kThreads) L.ix.head[i] = 0u;
    for (uint32_t i = threadIdx.x; i < 320u; i += kThreads) L.freq[i] = 0u;
    if (threadIdx.x < kLook / 16u) { // the first bytes in front of the staging's stride
        const uint64_t p = s0 + (uint64_t)threadIdx.x * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p + 16u <= n_in) v = *reinterpret_cast<const uint4*>(in + p);
        else for (uint32_t k = 0; k < 16u; ++k) if (p + k < n_in) reinterpret_cast<uint8_t*>(&v)[k] = in[p + k];
        const uint32_t at = (uint32_t)p & kDataMask;
        *reinterpret_cast<uint4*>(L.data + at) = v;
        if (at < kMirror) *reinterpret_cast<uint4*>(L.data + kDataRing + at) = v;
    }
    for (uint32_t i = threadIdx.x; i < kChunkTiles; i += kThreads) { // the parse ORs / adds a tile's starts, matches and match count together
        L.startbits[i] = 0ull;
        L.matchbits[i] = 0ull;
        L.match_base[i] = 0u;
    }
    if (threadIdx.x == 0u) { L.queue[0] = 0u; L.queue[1] = 0u; L.ntok = 0u; L.nmatch = 0u; }
    __syncthreads();

    // ---------------- the pipeline, per step i ----------------
    //   all waves   stage the bytes of segment i (kLook ahead) | barrier | index it inside its tiles (two tiles per wave) | barrier
    //   then        wave 0: the links across tiles; everybody: finish segment j - 2 (tiles from a queue); waves 1-4:
    //               a window (a quarter) of segment j - 1 each, once the finishers are through; everybody: search chunk
    //               segment j = i - nwin (30 tiles from a queue, a tile as soon as its links are complete) | barrier
    //   Wave 0 turns a parsed segment's per-tile match counts into token bases at the top of the next step
    //   (token_counts), in front of the barriers the finishers wait behind.
    uint32_t fixed_lane = 0, extra_lane = 0;
    const uint32_t steps = nwin + nseg + 2u;
#if defined(SNAPHASH_DEFLATE_STAMPS)
    uint64_t t_ins = 0, t_par = 0, t_sea = 0, t_fin = 0, t_wait = 0, t_all0 = __builtin_amdgcn_s_memtime();
#define STAMP(acc, code) { const uint64_t s_ = __builtin_amdgcn_s_memtime(); code; acc += __builtin_amdgcn_s_memtime() - s_; }
    uint64_t t_bar[5] = {0, 0, 0, 0, 0};
#define SYNC_STAMPED(n) { const uint64_t s_ = __builtin_amdgcn_s_memtime(); __syncthreads(); if (chunk_step) t_bar[n] += __builtin_amdgcn_s_memtime() - s_; }
#else
#define STAMP(acc, code) { code; }
#define SYNC_STAMPED(n) __syncthreads()
#endif
    for (uint32_t i = 0; i < steps; ++i) {
        const bool chunk_step = i >= nwin;
        const uint32_t j = i - nwin; // the chunk segment this step indexes and searches (meaningful when chunk_step)
        const bool do_finish = chunk_step && j >= 2u && j - 2u < nseg;
        const uint32_t fin_rel0 = (j - 2u) * kDfSeg; // (meaningful when do_finish)
        const uint32_t fin_m = do_finish ? ((len - fin_rel0 < kDfSeg) ? len - fin_rel0 : kDfSeg) : 0u;
        if (do_finish && wave == 0u) STAMP(t_par, token_counts(L, fin_rel0, fin_m, lane));
        if (chunk_step && j >= 1u && j - 1u < nseg && wave >= 1u && wave <= kDfParseWaves) { // the windows of the segment parsed in this step
            const uint32_t rel0 = (j - 1u) * kDfSeg;
            const uint32_t m = (len - rel0 < kDfSeg) ? len - rel0 : kDfSeg;
            const uint32_t b = window_bound(L.res[(j - 1u) % 3u], m, wave, lane);
            if (lane == 0u) { L.bounds[wave] = b; if (wave == 1u) L.bounds[0] = 0u; }
        }
        if (threadIdx.x == 0u) { L.across_done = 0u; L.finish_done = 0u; }
        const bool do_index = i < nwin + nseg;
        if (do_index) {
            const uint64_t seg0 = s0 + (uint64_t)i * kDfSeg;
            stage_bytes(L, in, n_in, seg0 + kLook, threadIdx.x);
            SYNC_STAMPED(0);
            // the result buffer this step's search will fill is free until then: the owners' signatures live there
            uint8_t* owner = reinterpret_cast<uint8_t*>(L.res[(chunk_step ? j : i) % 3u]);
            static_assert(sizeof(L.res[0]) >= kHeadN, "the signatures of one segment fit a result buffer");
            STAMP(t_ins, for (uint32_t t = wave; t < kSegTiles; t += kWaves) index_tile_inside(L, owner, n_in, seg0, t, c1, lane));
        }
        SYNC_STAMPED(1);
        // From here to the step's last barrier nothing waits for everybody: the links across tiles announce their
        // progress tile by tile (a searcher waits for its own tile), the finishers count their tiles (the parse waits
        // for the last: its prices are the counts).
        if (do_index && wave == 0u) {
            const uint64_t seg0 = s0 + (uint64_t)i * kDfSeg;
            __builtin_amdgcn_s_setprio(3);
            STAMP(t_ins, index_segment_across(L, n_in, s0, seg0, c1, lane));
            __builtin_amdgcn_s_setprio(0);
        }
        if (do_finish) { // the tokens of segment j - 2
            for (;;) {
                uint32_t item = 0;
                if (lane == 0u) item = atomicAdd(&L.queue[0], 1u);
                item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
                if (item * 64u >= fin_m) break;
                STAMP(t_fin, finish_tile(L, L.res[(j - 2u) % 3u], item, (fin_rel0 >> 6) + item, lane, fixed_lane, extra_lane, tok));
                if (lane == 0u) __hip_atomic_fetch_add(&L.finish_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (chunk_step && wave >= 1u && wave <= kDfParseWaves && j >= 1u && j - 1u < nseg) { // a window of segment j - 1 (wave 0 has the step's other serial job)
            // (waves 1-4 sit on four different SIMDs -- a workgroup's waves go round them -- and that matters: four parsing
            // waves on ONE SIMD take a fifth longer each, 12.7 instead of 10.4 ms per 64 MiB)
            const uint32_t rel0 = (j - 1u) * kDfSeg;
            const uint32_t m = (len - rel0 < kDfSeg) ? len - rel0 : kDfSeg;
            const uint32_t fin_tiles = (fin_m + 63u) >> 6;
            STAMP(t_wait, while (__hip_atomic_load(&L.finish_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < fin_tiles) __builtin_amdgcn_s_sleep(1));
            __builtin_amdgcn_s_setprio(3);
            STAMP(t_par, parse_window(L, L.res[(j - 1u) % 3u], rel0, m, wave - 1u, lane));
            __builtin_amdgcn_s_setprio(0);
        }
        if (chunk_step && j < nseg) {
#if defined(SNAPHASH_DF_PAIR_TILES) // round 5 experiment: two tiles per wave at a time (search_pair)
            static_assert(kSegTiles % 2u == 0u, "whole pairs of tiles");
            for (;;) {
                uint32_t item = 0;
                if (lane == 0u) item = atomicAdd(&L.queue[1], 1u);
                item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
                if (item >= kSegTiles / 2u) break;
                STAMP(t_wait, while (__hip_atomic_load(&L.across_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= 2u * item + 1u) __builtin_amdgcn_s_sleep(1));
                uint32_t ra, rb;
                const uint64_t pa = c0 + (uint64_t)j * kDfSeg + item * 128u + lane;
                STAMP(t_sea, search_pair(L, in, n_in, pa, pa + 64u, c1, depth, ra, rb));
                L.res[j % 3u][item * 128u + lane] = ra;
                L.res[j % 3u][item * 128u + 64u + lane] = rb;
            }
#elif !defined(SNAPHASH_DF_STREAM_SEARCH) // a tile per wave at a time: a tile costs its slowest lane, and adjacent positions walk adjacent candidates
            for (;;) {
                uint32_t item = 0;
                if (lane == 0u) item = atomicAdd(&L.queue[1], 1u);
                item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
                if (item >= kSegTiles) break;
                STAMP(t_wait, while (__hip_atomic_load(&L.across_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= item) __builtin_amdgcn_s_sleep(1));
                STAMP(t_sea, L.res[j % 3u][item * 64u + lane] = search_position(L, in, n_in, c0 + (uint64_t)j * kDfSeg + item * 64u + lane, c1, depth));
            }
#else
            STAMP(t_sea, search_segment_stream(L, in, n_in, c0 + (uint64_t)j * kDfSeg, c1, L.res[j % 3u], lane, depth));
#endif
        }
        SYNC_STAMPED(3);
        if (threadIdx.x == 0u) { L.queue[0] = 0u; L.queue[1] = 0u; } // handed out again only behind the next step's barriers
    }
    // the block prices: every wave holds a share
    {
        const uint32_t f = wave_sum(fixed_lane, lane), e = wave_sum(extra_lane, lane);
        if (threadIdx.x == 0u) { L.fixed_bits = 3u + 7u; L.extra_bits = 0u; }
        __syncthreads();
        if (lane == 0u) { atomicAdd(&L.fixed_bits, f); atomicAdd(&L.extra_bits, e); }
    }
#if defined(SNAPHASH_DEFLATE_STAMPS)
    const uint64_t t_pipe = __builtin_amdgcn_s_memtime() - t_all0;
    if (c == 37u && lane == 0u && (wave <= 1u || wave == 5u))
        printf("chunk %u wave %u: pipeline %llu cycles; insert %llu parse %llu search %llu finish %llu spinning %llu; waiting at the step's barriers %llu %llu %llu %llu\n", c, wave, (unsigned long long)t_pipe,
               (unsigned long long)t_ins, (unsigned long long)t_par, (unsigned long long)t_sea, (unsigned long long)t_fin, (unsigned long long)t_wait, (unsigned long long)t_bar[0], (unsigned long long)t_bar[1], (unsigned long long)t_bar[2], (unsigned long long)t_bar[3]);
    const uint64_t t_tail0 = __builtin_amdgcn_s_memtime();
#endif
    __threadfence_block(); // the match tokens in the scratch are read back by every wave of this workgroup
    __syncthreads();

    // ---------------- the codes and the header: wave 0 the literal/length tree, wave 1 the distance tree, then wave 0 the
    // header.  Deterministic, and the same lengths as the serial routines the CPU model runs (deflate_core.h). ----------------
    HuffScratch& S = L.em.hs;
    if (wave == 0u) {
        if (lane == 0u) L.freq[256] += 1u; // end of block
        wave_lds_sync();
        huff_lengths_wave(L.freq, (uint32_t)kNumLL, (uint32_t)kMaxBits, L.len, S.w, S.parent, S.order, lane);
    } else if (wave == 1u) {
        HuffScratch& SD = L.em.hs_d;
        if (lane == 0u) {
            const bool ghost0 = L.freq[288] == 0u, ghost1 = L.freq[289] == 0u;
            if (ghost0) L.freq[288] = 1u; // at least two distance codes, as zlib sends
            if (ghost1) L.freq[289] = 1u;
            L.ghosts = (ghost0 ? 1u : 0u) | (ghost1 ? 2u : 0u);
        }
        wave_lds_sync();
        huff_lengths_wave(L.freq + 288, (uint32_t)kNumD, (uint32_t)kMaxBits, L.len + 288, SD.w, SD.parent, SD.order, lane);
    }
    __syncthreads();
    if (wave == 0u) {
        if (lane == 0u) dyn_header_tokens(L.len, L.len + 288, S.rle, S.clfreq, S.hdr);
        wave_lds_sync();
        huff_lengths_wave(S.clfreq, (uint32_t)kNumCL, (uint32_t)kMaxCLBits, S.cllen, S.w, S.parent, S.order, lane);
        wave_lds_sync();
        if (lane == 0u) {
            huff_codes(S.cllen, kNumCL, S.clcode, S.cnt);
            uint32_t ncl = kNumCL;
            while (ncl > 4u && S.cllen[cl_order(ncl - 1u)] == 0) --ncl;
            S.hdr.ncl = ncl;
        }
        wave_lds_sync();
        uint32_t part = 0;
        for (uint32_t t = lane; t < S.hdr.ntok; t += 64u) { const uint32_t sym = S.rle[t] & 0xffu; part += (uint32_t)S.cllen[sym] + cl_extra_bits(sym); }
        for (uint32_t k = lane; k < 318u; k += 64u)
            if (k < (uint32_t)kNumLL || k >= 288u) part += L.freq[k] * (uint32_t)L.len[k];
        const uint32_t sum = wave_sum(part, lane);
        // header bits alone (for the emission's placement), then the whole block
        uint32_t hpart = 0;
        for (uint32_t t = lane; t < S.hdr.ntok; t += 64u) { const uint32_t sym = S.rle[t] & 0xffu; hpart += (uint32_t)S.cllen[sym] + cl_extra_bits(sym); }
        const uint32_t hsum = wave_sum(hpart, lane);
        if (lane == 0u) {
            S.hdr.bits = 3u + 5u + 5u + 4u + 3u * S.hdr.ncl + hsum;
            const uint32_t db = 3u + 5u + 5u + 4u + 3u * S.hdr.ncl + sum + L.extra_bits;
            L.use_dynamic = db < L.fixed_bits + L.extra_bits ? 1u : 0u; // (priced with the two codes that may never be sent, as the model does)
            // what the block will really take: a distance code that exists only to complete the code is never emitted
            L.dyn_bits = db - ((L.ghosts & 1u) ? (uint32_t)L.len[288] : 0u) - ((L.ghosts & 2u) ? (uint32_t)L.len[289] : 0u);
        }
    }
    __syncthreads();
#if defined(SNAPHASH_DEFLATE_STAMPS)
    if (c == 37u && threadIdx.x == 0u) printf("chunk %u: codes %llu cycles\n", c, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_tail0));
#endif
    const bool dynamic = L.use_dynamic != 0u;
    const uint32_t body_bits = dynamic ? L.dyn_bits : L.fixed_bits + L.extra_bits; // header, tokens and end-of-block
    // + the empty stored block: 3 header bits, pad to a byte, LEN = 0, NLEN = 0xFFFF
    const uint32_t nbytes = ((body_bits + 3u + 7u) >> 3) + 4u;
    const uint32_t stored_bytes = deflate_stored_size(len);
    const bool stored = nbytes >= stored_bytes;
    if (stored) { // did not shrink: stored blocks (BFINAL=0, BTYPE=00 in a whole byte; LEN; ~LEN; the bytes).  LEN is 16 bits:
                  // a full 64 KiB chunk goes out as two blocks of 32 KiB
        const uint32_t first = len > 65535u ? 32768u : len;
        if (threadIdx.x == 0u) {
            dst[0] = 0u;
            dst[1] = (uint8_t)first; dst[2] = (uint8_t)(first >> 8);
            dst[3] = (uint8_t)~first; dst[4] = (uint8_t)(~first >> 8);
            if (first < len) {
                const uint32_t rest = len - first;
                uint8_t* h2 = dst + 5u + first;
                h2[0] = 0u;
                h2[1] = (uint8_t)rest; h2[2] = (uint8_t)(rest >> 8);
                h2[3] = (uint8_t)~rest; h2[4] = (uint8_t)(~rest >> 8);
            }
            sizes[c] = stored_bytes;
        }
        const uint8_t* src = in + c0; // 16-byte aligned (c0 is a multiple of 64 KiB)
        const uint32_t nw = len >> 2;
        for (uint32_t i = threadIdx.x; i < nw; i += kThreads) {
            const uint32_t at = 4u * i;
            *reinterpret_cast<u32_unaligned*>(dst + 5u + at + (at >= first ? 5u : 0u)) = *reinterpret_cast<const uint32_t*>(src + at);
        }
        for (uint32_t i = (nw << 2) + threadIdx.x; i < len; i += kThreads) dst[5u + i + (i >= first ? 5u : 0u)] = src[i];
        return;
    }

    // ---------------- emission: price the tiles, place them, encode into the LDS image, copy out ----------------
    const uint32_t ntiles = (len + 63u) >> 6;
    if (dynamic) { // the counts are spent: the codes go where they were
        if (wave == 0u) huff_codes_wave(L.len, (uint32_t)kNumLL, L.freq, lane);
        else if (wave == 1u) huff_codes_wave(L.len + 288, (uint32_t)kNumD, L.freq + 288, lane);
    }
    const uint32_t img_words = (nbytes + 3u) >> 2;
    for (uint32_t i = threadIdx.x; i < img_words + 2u; i += kThreads) L.image[i] = 0u; // the ring and the heads are spent
    __syncthreads();
    for (uint32_t tile = wave; tile < ntiles; tile += kWaves) { // pass A: bits per tile
        const uint64_t sm = L.startbits[tile], mk = L.matchbits[tile];
        const bool my_start = (sm >> lane) & 1ull, my_match = (mk >> lane) & 1ull;
        const uint32_t pos = tile * 64u + lane;
        const uint32_t byte = pos < len ? in[c0 + pos] : 0u;
        uint32_t t = 0;
        if (my_match) t = tok[L.match_base[tile] + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull))];
        uint32_t ba, na, bb, nb;
        token_bits(L, dynamic, my_start, my_match, byte, t, ba, na, bb, nb);
        const uint32_t total = wave_sum(na + nb, lane);
        if (lane == 0u) L.em.tile_bits[tile] = total;
    }
    __syncthreads();
    if (wave == 0u) { // exclusive scan of the tile sizes, the header's bits in front
        uint32_t run = dynamic ? S.hdr.bits : 3u;
        for (uint32_t t0 = 0; t0 < ntiles; t0 += 64u) {
            const uint32_t v = (t0 + lane < ntiles) ? L.em.tile_bits[t0 + lane] : 0u;
            const uint32_t incl = wave_scan_incl(v, lane);
            if (t0 + lane < ntiles) L.em.tile_bits[t0 + lane] = run + incl - v;
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0u) L.total_bits = run; // where the end-of-block code goes
    }
    if (threadIdx.x == 64u) { // meanwhile: the block header
        if (dynamic) {
            uint32_t at = 0;
            write_dyn_header(S.hdr, S.rle, S.cllen, S.clcode, [&](uint32_t bits, uint32_t nb) { put_bits(L.image, at, bits, nb); at += nb; });
        } else {
            put_bits(L.image, 0u, 2u, 3u); // BFINAL=0, BTYPE=01
        }
    }
    __syncthreads();
    for (uint32_t tile = wave; tile < ntiles; tile += kWaves) { // pass B: encode
        const uint64_t sm = L.startbits[tile], mk = L.matchbits[tile];
        const bool my_start = (sm >> lane) & 1ull, my_match = (mk >> lane) & 1ull;
        const uint32_t pos = tile * 64u + lane;
        const uint32_t byte = pos < len ? in[c0 + pos] : 0u;
        uint32_t t = 0;
        if (my_match) t = tok[L.match_base[tile] + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull))];
        uint32_t ba, na, bb, nb;
        token_bits(L, dynamic, my_start, my_match, byte, t, ba, na, bb, nb);
        const uint32_t incl = wave_scan_incl(na + nb, lane);
        const uint32_t at = L.em.tile_bits[tile] + incl - (na + nb);
        put_bits(L.image, at, ba, na);
        put_bits(L.image, at + na, bb, nb);
    }
    __syncthreads();
    if (threadIdx.x == 0u) { // end of block, then the empty stored block that byte-aligns the chunk
        uint32_t at = L.total_bits;
        if (dynamic) { const uint32_t ce = L.freq[256]; put_bits(L.image, at, ce >> 8, ce & 0xffu); at += ce & 0xffu; }
        else at += 7u;
        at += 3u;
        at = (at + 7u) & ~7u;
        put_bits(L.image, at, 0xFFFF0000u, 32u);
        sizes[c] = (at >> 3) + 4u;
    }
    __syncthreads();
    uint32_t* dstw = reinterpret_cast<uint32_t*>(dst); // slots are 64-byte aligned
    for (uint32_t i = threadIdx.x; i < img_words; i += kThreads) dstw[i] = L.image[i];
#if defined(SNAPHASH_DEFLATE_STAMPS)
    if (c == 37u && threadIdx.x == 0u) printf("chunk %u: codes + emission %llu ticks\n", c, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_tail0));
#endif
}

// Concatenates the chunk outputs: chunk c's sizes[c] bytes go to out + prefix[c].
__global__ __launch_bounds__(256) void deflate_compact_kernel(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ sizes,
                                                              const uint64_t* __restrict__ prefix, uint8_t* __restrict__ out,
                                                              uint32_t chunk0)
{
    const uint32_t c = chunk0 + blockIdx.x;
    const uint8_t* src = slots + (uint64_t)c * kDeflateSlot;
    uint8_t* dst = out + prefix[c];
    const uint32_t n = sizes[c];
    const uint32_t nw = n >> 2;
    for (uint32_t i = threadIdx.x; i < nw; i += 256u)
        *reinterpret_cast<u32_unaligned*>(dst + 4u * i) = *reinterpret_cast<const uint32_t*>(src + 4u * i);
    for (uint32_t i = (nw << 2) + threadIdx.x; i < n; i += 256u) dst[i] = src[i];
}

hipError_t launch_deflate_chunks(const uint8_t* d_in, uint64_t n_in, uint8_t* d_slots, uint32_t* d_sizes, uint32_t* d_toks,
                                 uint32_t chunk0, uint32_t count, uint32_t nchunks, uint32_t n_xcd, uint32_t depth, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    if (n_xcd == 0 || n_xcd > 64) n_xcd = 8;
    if (depth == 0) depth = kDfDepth;
    hipLaunchKernelGGL(deflate_chunks_kernel, dim3(count), dim3(1024), 0, s, d_in, n_in, d_slots, d_sizes, d_toks, chunk0, nchunks, n_xcd, depth);
    return hipGetLastError();
}

hipError_t launch_deflate_compact(const uint8_t* d_slots, const uint32_t* d_sizes, const uint64_t* d_prefix, uint8_t* d_out,
                                  uint32_t chunk0, uint32_t count, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(deflate_compact_kernel, dim3(count), dim3(256), 0, s, d_slots, d_sizes, d_prefix, d_out, chunk0);
    return hipGetLastError();
}

} // namespace snaphash
