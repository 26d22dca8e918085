// mkt_launch.h -- kernel argument block and launch wrappers (mkt_kernels.hip <-> mkt_capi.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include "mkt_host.h"
#include "mkt_synth.h"

namespace mkt {

enum { CFG_FAST = 0, CFG_SMALL = 1 };

// totals of the blocks a context has finished, kept on the device so that resident blocks chain
// without a host round trip
struct DevRun { uint64_t groups, sc; };

struct KArgs {
    const uint8_t* text;        // block text, 16-byte aligned
    uint32_t n;                 // block bytes (< 2^31)
    uint32_t ntiles;
    Params P;
    uint64_t* descA;            // per-tile look-back words: groups | emitted
    uint64_t* descB;            //                            pair_bytes | self-circles
    uint64_t* descC;            //                            sam_bytes
    uint32_t* ticket;
    uint32_t* tile_groups;      // per tile: groups opened in it; k_finish turns it into the exclusive prefix
    int32_t ordered;            // 1: outputs in input order (decoupled look-back); 0: one atomic range per tile
    TileLast* tile_last;
    BlockResult* res;
    DevRun* run;
    OutPtrs out;                // sc_base is filled by the kernel from *run
    unsigned long long* stamps; // diagnostic builds only (MKT_STAMPS), else null
};

uint32_t tile_bytes(int cfg);
hipError_t launch_tiles(const KArgs& a, int cfg, int grid, hipStream_t s);
hipError_t launch_finish(const KArgs& a, hipStream_t s);

hipError_t launch_synth_sizes(const SynParams& p, uint64_t first, uint64_t n, uint64_t* sizes, hipStream_t s);
hipError_t launch_exscan(uint64_t* v, uint64_t n, uint64_t* total, hipStream_t s);
hipError_t launch_synth_write(const SynParams& p, uint64_t first, uint64_t n, const uint64_t* offs, char* out, hipStream_t s);
hipError_t launch_synth_tail(const SynParams& p, char* out, hipStream_t s);
size_t synth_tail_bytes(const SynParams& p);

}  // namespace mkt
