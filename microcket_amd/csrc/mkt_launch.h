// mkt_launch.h -- kernel argument block and launch wrappers (mkt_kernels.hip <-> mkt_capi.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include "mkt_host.h"
#include "mkt_fast.h"
#include "mkt_synth.h"

namespace mkt {

enum { CFG_FAST = 0, CFG_SMALL = 1 };      // kernels: lean + generic for what it defers (tile bytes chosen at run time) / generic only, 256-byte tiles

// totals of the blocks a context has finished, kept on the device so that resident blocks chain
// without a host round trip
struct DevRun { uint64_t groups, sc, emitted; };

// One output region's cursors, alone on a 128-byte line so that the per-tile atomics of different
// regions never meet.  a: .pairs bytes (low 36 bits) | emitted lines (high 28); b: .sam bytes | self-circles.
// A block is < 2 GiB of text and both outputs are shorter than their input, so 36 bits hold any byte count; a reported
// pair or self-circle needs a group of >= 12 input bytes, so 2^28 lines cannot be exceeded inside one block either.
struct alignas(128) RegionCur { unsigned long long a, b; unsigned long long pad[14]; };
constexpr int kMaxRegions = 16;
constexpr int kCurShift = 36;
constexpr unsigned long long kCurLow = (1ull << kCurShift) - 1ull;

struct KArgs {
    const uint8_t* text;        // block text, 16-byte aligned
    uint32_t n;                 // block bytes (< 2^31)
    uint32_t ntiles;
    TileDims dims;              // bytes per tile / back halo / forward halo (multiples of 16, within the kernels' capacities)
    Params P;
    uint64_t* descA;            // per-tile look-back words: groups | emitted
    uint64_t* descB;            //                            pair_bytes | self-circles
    uint64_t* descC;            //                            sam_bytes
    uint32_t* ticket;
    uint64_t* tile_groups;      // per tile: groups opened (low 32) | pairs emitted (high 32); k_finish_scan turns both into exclusive prefixes
    int32_t ordered;            // 1: outputs in input order (decoupled look-back); 0: one atomic range per tile
    TileLast* tile_last;
    uint32_t* defer_list;       // tiles the lean kernel left to the generic one
    uint32_t* defer_count;
    int* last_tile;             // 1 + highest tile index that opened a group (0: none)
    uint64_t* scan_desc;        // look-back words of k_finish_scan, one per 1024 tiles
    uint32_t* scan_ticket;      // ... and the ticket that hands its chunks out
    int32_t use_list;           // generic kernel: 1 = walk defer_list[0, *defer_count) instead of all tiles
    BlockResult* res;
    DevRun* run;
    RegionCur* cur;             // [nregions]
    int32_t nregions;           // 1 (ordered) or kMaxRegions
    uint64_t pairs_rcap, sam_rcap, sc_rcap;   // capacity of one region in out.pairs / out.sam / out.sc
    uint64_t* sc_list;          // resolved self-circle group indices of the whole run (drained by the host)
    uint64_t sc_list_cap;
    uint64_t keys_rcap;         // extension: raw key records per region (out.keys), 0 = off
    KeyRec* key_list;           // extension: the run's key records in input order
    uint64_t key_list_cap;
    OutPtrs out;                // whole buffers; per-tile limits are derived from the region
    unsigned long long* stamps; // diagnostic builds only (MKT_STAMPS), else null
    int32_t debug_stop;         // diagnostic builds only: leave every tile after phase k (timing ladder; outputs are wrong)
};

TileDims small_dims();                       // the 256-byte tiles of CFG_SMALL
TileDims max_dims();                         // the largest geometry the lean kernel (and the generic kernel behind it) takes
uint32_t fast_max_workgroups(int cfg);      // grid of the lean kernel: the workgroups that are resident at once on 256 CUs
// newlines in text[0, n): the host's line-length probe on device-resident text (one small launch; *out zeroed by the caller)
hipError_t launch_count_newlines(const uint8_t* text, size_t n, unsigned long long* out, hipStream_t s);
hipError_t launch_tiles(const KArgs& a, int cfg, int grid, hipStream_t s);
hipError_t launch_fast(const KArgs& a, int cfg, int grid, hipStream_t s);
uint32_t finish_chunk_tiles();
hipError_t launch_finish(const KArgs& a, hipStream_t s);
// streaming path: one output of the block (which: 0 .pairs, 1 .sam) from its region slices to one contiguous range, last group at the end
hipError_t launch_gather(const BlockResult* d_res, const uint8_t* src, uint64_t rcap, uint8_t* dst, uint64_t dst_cap, int which, uint64_t max_bytes, hipStream_t s);
hipError_t launch_sc_logged(const uint64_t* list, uint64_t n, uint64_t drop_group, uint64_t group_offset, uint64_t K_total, uint32_t ref_threads,
                            unsigned long long* out, hipStream_t s);

// extensions (duplicate marking, per-chromosome counts) over the run's key list
struct DedupResult { uint64_t total, dups; };
hipError_t launch_dedup(const KeyRec* keys, uint64_t n, uint8_t* flags, void* work, size_t work_bytes, DedupResult* d_res, hipStream_t s);
size_t dedup_work_bytes(uint64_t n);
size_t partition_work_bytes();
hipError_t launch_partition(const KeyRec* keys, uint64_t n, const uint16_t* d_lut, uint32_t world, uint32_t* hist, KeyRec* send, uint32_t* perm, uint32_t* G_out, hipStream_t s);
hipError_t launch_unpermute(const uint8_t* in, const uint32_t* perm, uint64_t n, uint8_t* out, unsigned long long* dups, hipStream_t s);
hipError_t launch_chrstat(const KeyRec* keys, uint64_t n, const uint16_t* dense_of_slot, uint32_t ndense, unsigned long long* counts, hipStream_t s);

hipError_t launch_synth_sizes(const SynParams& p, uint64_t first, uint64_t n, uint64_t* sizes, hipStream_t s);
hipError_t launch_exscan(uint64_t* v, uint64_t n, uint64_t* total, hipStream_t s);
hipError_t launch_synth_write(const SynParams& p, uint64_t first, uint64_t n, const uint64_t* offs, char* out, hipStream_t s);
hipError_t launch_synth_tail(const SynParams& p, char* out, hipStream_t s);
size_t synth_tail_bytes(const SynParams& p);

}  // namespace mkt
