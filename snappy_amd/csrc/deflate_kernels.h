// deflate_kernels.h -- internal interface of the block-parallel DEFLATE kernels (row f3).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "deflate_core.h"

namespace snaphash {

constexpr uint32_t kDeflateChunk = kDfChunk;                                // input bytes per chunk (one workgroup, one DEFLATE block)
constexpr uint32_t kDeflateSlot = kDeflateChunk + kDeflateChunk / 8 + 64;   // output room per chunk (a multiple of 64)
constexpr uint32_t kDeflateTokWords = kDeflateChunk / 3 + 40;               // match tokens a chunk can hold (every match covers >= 3 bytes)
static_assert(kDeflateSlot % 64 == 0, "slots are copied out in whole dwords");

// d_in must be readable up to n_in + 8 bytes.  sizes[c] <= kDeflateSlot.  d_toks: scratch, nchunks * kDeflateTokWords
// words: the match tokens of the parse, re-read by the passes that price and emit them.
// chunks [chunk0, chunk0 + count) of the nchunks the stream of n_in bytes has: a piece may be compressed in several
// launches (the bytes of the first launches travel on while the later ones run) and comes out as if in one.
hipError_t launch_deflate_chunks(const uint8_t* d_in, uint64_t n_in, uint8_t* d_slots, uint32_t* d_sizes, uint32_t* d_toks,
                                 uint32_t chunk0, uint32_t count, uint32_t nchunks, uint32_t n_xcd /* L2 domains the grid goes round: hipDeviceAttributeNumberOfXccs */,
                                 uint32_t depth /* links walked per position: 0 = kDfDepth */, hipStream_t s);
hipError_t launch_deflate_compact(const uint8_t* d_slots, const uint32_t* d_sizes, const uint64_t* d_prefix, uint8_t* d_out,
                                  uint32_t chunk0, uint32_t count, hipStream_t s);

} // namespace snaphash
