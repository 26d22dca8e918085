// mkt_sortlib.h -- the building blocks of mkt_sort.hip that mkt_bam.hip uses as well: newline index, 16-byte records, stable LSD
// radix passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mkt {

struct SortRec { uint64_t hi; uint32_t lo; uint32_t idx; };

// starts[0] = 0, starts[k + 1] = 1 + position of newline k (hipMalloc'ed here, nlines + 2 entries; the caller frees).
// The text must end with a newline.
hipError_t sort_line_index(const uint8_t* d_text, uint64_t n, hipStream_t st, uint64_t** d_starts, uint64_t* nlines);
// stable LSD passes (4 bits each) over bits [lo_bit, lo_bit + nbits) of rec.hi (which = 1) or rec.lo (which = 0); the sorted
// records end up in rA (the two buffers are swapped as needed).  d_hist: 16 * 1024 + 64 uint32.
void sort_radix_passes(SortRec*& rA, SortRec*& rB, uint64_t n, uint32_t* d_hist, int which, int lo_bit, int nbits, hipStream_t st);
constexpr size_t kSortHistBytes = (size_t)16 * 1024 * 4 + 256;

}  // namespace mkt
