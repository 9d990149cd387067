// deflate_kernels.h -- internal interface of the block-parallel DEFLATE kernels (row f3).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace snaphash {

constexpr uint32_t kDeflateChunk = 16384;                                  // input bytes per chunk (one wave)
constexpr uint32_t kDeflateSlot = kDeflateChunk + kDeflateChunk / 8 + 64;  // output room per chunk: 9 bits per byte + framing

// d_in must be readable up to n_in + 8 bytes.  sizes[c] <= kDeflateSlot.  d_toks: scratch, one word per input byte
// (nchunks * kDeflateChunk words): the parse of the first pass, re-read by the pass that emits.
hipError_t launch_deflate_chunks(const uint8_t* d_in, uint64_t n_in, uint8_t* d_slots, uint32_t* d_sizes, uint32_t* d_toks,
                                 uint32_t nchunks, hipStream_t s);
hipError_t launch_deflate_compact(const uint8_t* d_slots, const uint32_t* d_sizes, const uint64_t* d_prefix, uint8_t* d_out,
                                  uint32_t nchunks, hipStream_t s);

} // namespace snaphash
