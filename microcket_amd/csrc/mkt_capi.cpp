// mkt_capi.cpp -- the C ABI of include/mkt.h: contexts, block scheduling, host bookkeeping.
// Compiled with hipcc into libmkt_hip.so together with mkt_kernels.hip.  There is no CPU path:
// every entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <utility>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mkt.h"
#include "mkt_launch.h"

using namespace mkt;

// block offsets are 32 bit, the ordered kernels pack byte counts into 31-bit fields
static const size_t kMaxBlock = ((size_t)1 << 31) - 65536;

static thread_local std::string g_create_error;

struct mkt_ctx {
    mkt_params p;
    Params P;
    int cfg = CFG_FAST;
    TileDims dims = {0, 0, 0};          // bytes per tile / halos of the blocks to come (MKT_TILES_AUTO: from the input's line length)
    bool dims_probed = false;           // the line length of this input has been looked at
    unsigned long long* d_probe = nullptr; unsigned long long* h_probe = nullptr;      // resident path: newline count of the first MiB
    TileDims last_dims = {0, 0, 0};     // geometry of the newest resident block (mkt_fetch_last_block)
    const uint8_t* last_text = nullptr; // its text (valid until the next sync: a failed probe block is re-run)
    bool probing = true;                // the next resident block is looked at before more are queued
    hipStream_t stream = nullptr;
    size_t block_cap = 0;
    // device
    uint8_t* d_in = nullptr;
    uint8_t* d_pairs = nullptr; size_t pairs_cap = 0;
    uint8_t* d_sam = nullptr; size_t sam_cap = 0;
    uint64_t* d_sc = nullptr; size_t sc_cap = 0;          // the run's resolved self-circle list (drained at syncs)
    unsigned long long* d_sc_logged = nullptr;            // result word of k_sc_logged
    uint8_t* h_chr_stage = nullptr; uint16_t* d_dense = nullptr;              // mkt_ext_chrstat: pinned staging, slot -> dense id
    uint8_t* d_dd_flags = nullptr; size_t dd_flags_cap = 0; void* d_dd_work = nullptr; size_t dd_work_cap = 0;      // mkt_ext_dedup
    DedupResult* d_dd_res = nullptr; DedupResult* h_dd_res = nullptr;
    uint32_t* d_perm = nullptr; size_t perm_cap = 0; uint32_t* d_part_hist = nullptr; uint16_t* d_lut = nullptr; uint64_t part_n = 0;      // sharded duplicate marking
    unsigned long long* d_chr_counts = nullptr; unsigned long long* h_chr_counts = nullptr; size_t chr_counts_cap = 0;
    double sc_density = 0;                                // most self-circles per input byte seen between two syncs (0: nothing seen yet)
    uint64_t* d_sc_tmp = nullptr; size_t sc_tmp_cap = 0;  // per block: raw (tile, ordinal) entries, one slice per region
    // extensions (MKT_EXT_KEYS)
    KeyRec* d_keys_raw = nullptr; size_t keys_raw_cap = 0; // per block, one slice per region
    KeyRec* d_key_list = nullptr; size_t key_list_cap = 0; // the run's keys in input order
    ChrTab* d_chr = nullptr;
    uint8_t* d_ws = nullptr; size_t ws_cap = 0;
    DevRun* d_run = nullptr;
    // host (pinned)
    size_t h_len = 0;                   // bytes in the input slot being filled
    BlockResult* h_res = nullptr; size_t res_slots = 0, res_used = 0, res_folded = 0;
    std::vector<void*> uploads;          // mkt_device_text buffers (freed with the context)
    std::vector<const uint8_t*> res_text; std::vector<size_t> res_n;      // resident path: the text of every queued block (a failed one is re-run)
    // ---- streaming pipeline (mkt_submit / mkt_input_window): the caller fills pinned input slots and queues GPU work
    // without waiting; one worker thread takes the results in order, copies the outputs back and hands them to the
    // consumer (mkt_drain / mkt_drain_wait).  reader || H2D || kernels || D2H || writer all overlap.
    static constexpr int kIn = 3, kOut = 2;
    static constexpr size_t kHead = 65536;                // room in front of a staged output for the held-back group of the block before
    struct InSlot { uint8_t* h = nullptr; uint8_t* d = nullptr; bool busy = false; hipEvent_t h2d = nullptr, k0 = nullptr, k1 = nullptr, done = nullptr; };
    struct OutSlot { uint8_t* d_pairs = nullptr; size_t d_pairs_cap = 0; uint8_t* d_sam = nullptr; size_t d_sam_cap = 0;
                     uint8_t* h = nullptr; size_t h_cap = 0; bool dev_busy = false, host_busy = false; };
    struct Job { int in_slot, out_slot; size_t n; int cfg; TileDims dims; int attempts; };
    struct Chunk { const char* pairs = nullptr; size_t pairs_len = 0; const char* sam = nullptr; size_t sam_len = 0; int out_slot = -1;
                   std::vector<char> own_pairs, own_sam; };
    InSlot in[kIn];
    OutSlot outs[kOut];
    std::deque<Job> jobs;                                 // queued on the GPU, results not yet taken (front = oldest)
    std::deque<Chunk> ready;                              // final output bytes waiting for the consumer
    Chunk handed; bool handed_valid = false;              // what mkt_drain_wait returned last (its staging slot is released by the next call)
    std::mutex mu;
    std::condition_variable cv;
    std::thread worker;
    bool worker_started = false, stop = false;
    int async_rc = MKT_OK;                                // first error the worker met; every later call reports it
    bool consumer_async = false;                          // mkt_drain_wait in use: another thread takes the outputs, so a full staging slot means WAIT (back-pressure)
    int cur = 0;                                          // input slot the caller is filling
    uint64_t seq = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<char> tail_pairs, tail_sam, drained_pairs, drained_sam;
    RunAccum acc;
    bool input_done = false, finished = false;
    uint64_t bytes_in = 0, blocks = 0;
    size_t last_n = 0;                   // bytes of the last resident block
    double key_density = 0;              // extension: most reported pairs per input byte seen between two syncs (0: nothing seen yet)
    uint64_t emitted_unfolded = 0;
    uint64_t sc_unfolded = 0;            // self-circle entries of the blocks folded at the last sync (for the density estimate)
    uint64_t bytes_unsynced = 0;         // resident bytes enqueued since the last sync
    // timing
    std::vector<hipEvent_t> ev;          // start/stop pairs of the tile kernel
    std::vector<uint64_t> ev_bytes;
    double folded_ms = 0; uint64_t folded_launches = 0, folded_bytes = 0;
    uint64_t tiles_total = 0, tiles_deferred = 0;      // lean-kernel tiles / those it left to the generic kernel
    // synth
    char* d_syn = nullptr; size_t syn_cap = 0;
    uint64_t* d_syn_sizes = nullptr; size_t syn_sizes_cap = 0;
    unsigned long long* d_stamps = nullptr;   // diagnostic builds (MKT_STAMPS) only
    bool no_lean = false;                     // MKT_NO_LEAN=1: generic kernel only (debugging aid)
    int halo_widened = 0;                     // times adapt_geometry widened the halos of this input (at most twice)
    std::string err;
};

static int fail(mkt_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}
#define HIPCHK(c, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail((c), MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int mkt_abi_version(void) { return MKT_ABI_VERSION; }

const char* mkt_strerror(int code) {
    switch (code) {
    case MKT_OK: return "ok";
    case MKT_E_ARG: return "bad argument";
    case MKT_E_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case MKT_E_HIP: return "HIP runtime error";
    case MKT_E_NOMEM: return "out of memory";
    case MKT_E_CAPACITY: return "a QNAME group does not fit the block buffer";
    case MKT_E_KERNEL: return "kernel reported an internal error";
    case MKT_E_STATE: return "call order violated";
    default: return "unknown error";
    }
}
const char* mkt_last_error(const mkt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mkt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// workspace of one block: descA | descB | descC | tile_last | tile_groups | defer_list |
//                         region cursors (16 x 128 B) | ticket, defer_count, ticket of the deferred pass (256 B) | BlockResult
static size_t ws_tiles_bytes(uint32_t ntiles) {
    size_t b = (size_t)ntiles * (3 * sizeof(uint64_t) + sizeof(TileLast) + sizeof(uint64_t) + sizeof(uint32_t));
    return (b + 127) & ~(size_t)127;
}
// fixed part: region cursors | 256 B of counters | up to 1024 scan words (one per 1024 tiles)
static size_t ws_fixed_bytes() { return kMaxRegions * sizeof(RegionCur) + 256 + 1024 * sizeof(uint64_t); }
static size_t ws_bytes_for(uint32_t ntiles) { return ws_tiles_bytes(ntiles) + ws_fixed_bytes() + sizeof(BlockResult); }
// every stream of the context idle (before a device buffer that queued work may still use is freed)
static int sync_all(mkt_ctx* c) {
    if (c->s_in) HIPCHK(c, hipStreamSynchronize(c->s_in));
    if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->s_out) HIPCHK(c, hipStreamSynchronize(c->s_out));
    return MKT_OK;
}
static int ensure_ws(mkt_ctx* c, uint32_t ntiles) {
    size_t need = ws_bytes_for(ntiles);
    if (need <= c->ws_cap) return MKT_OK;
    need += need / 4;
    if (c->d_ws) { int rc = sync_all(c); if (rc) return rc; HIPCHK(c, hipFree(c->d_ws)); }
    c->d_ws = nullptr; c->ws_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_ws, need));
    c->ws_cap = need;
    return MKT_OK;
}
static int ensure_dev(mkt_ctx* c, uint8_t** p, size_t* cap, size_t need) {
    if (need <= *cap) return MKT_OK;
    if (*p) { int rc = sync_all(c); if (rc) return rc; HIPCHK(c, hipFree(*p)); }
    *p = nullptr; *cap = 0;
    need += need / 8 + 4096;
    HIPCHK(c, hipMalloc((void**)p, need));
    *cap = need;
    return MKT_OK;
}

int mkt_create(const mkt_params* p, mkt_ctx** out) {
    if (!p || !out) return fail(nullptr, MKT_E_ARG, "null argument");
    *out = nullptr;
    if (p->mode != MKT_MODE_FLASH && p->mode != MKT_MODE_UNC) return fail(nullptr, MKT_E_ARG, "mode must be MKT_MODE_FLASH or MKT_MODE_UNC");
    if (p->ref_threads < 2) return fail(nullptr, MKT_E_ARG, "ref_threads must be >= 2 (sam2pairs.cpp:36)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, MKT_E_NO_DEVICE, "no HIP device: libmkt_hip has no CPU path");
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, MKT_E_ARG, "device %d out of range (have %d)", p->device, ndev);
    mkt_ctx* c = new mkt_ctx();
    c->p = *p;
    c->P.mode = p->mode; c->P.ratio = p->min_mapped_ratio; c->P.min_mapq = (uint32_t)p->min_mapq; c->P.write_sam = p->write_sam ? 1 : 0;
    c->cfg = p->tiles == MKT_TILES_SMALL ? CFG_SMALL : CFG_FAST;
    c->dims = c->cfg == CFG_SMALL ? small_dims() : max_dims();
    c->dims_probed = p->tiles != MKT_TILES_AUTO || p->ordered;
    { const char* e = getenv("MKT_NO_LEAN"); c->no_lean = e && e[0] == '1'; }
    size_t bc = p->block_bytes ? (size_t)p->block_bytes : ((size_t)64 << 20);
    if (bc < 4096) bc = 4096;
    if (bc >= kMaxBlock) bc = kMaxBlock - 4096;
    bc = (bc + 15) & ~(size_t)15;
    c->block_cap = bc;
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(nullptr, MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); mkt_destroy(c); return e_ == hipErrorOutOfMemory ? MKT_E_NOMEM : MKT_E_HIP; } } while (0)
    CK(hipSetDevice(p->device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipMalloc((void**)&c->d_run, sizeof(DevRun)));
    CK(hipMemsetAsync(c->d_run, 0, sizeof(DevRun), c->stream));
    c->res_slots = 1024;
    CK(hipHostMalloc((void**)&c->h_res, c->res_slots * sizeof(BlockResult), hipHostMallocDefault));
    CK(hipStreamSynchronize(c->stream));
#undef CK
    *out = c;
    return MKT_OK;
}

void mkt_destroy(mkt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->worker_started) {
        { std::lock_guard<std::mutex> g(c->mu); c->stop = true; }
        c->cv.notify_all();
        c->worker.join();
    }
    (void)sync_all(c);
    for (int i = 0; i < mkt_ctx::kIn; ++i) {
        mkt_ctx::InSlot& s = c->in[i];
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
        if (s.h2d) { (void)hipEventDestroy(s.h2d); (void)hipEventDestroy(s.k0); (void)hipEventDestroy(s.k1); (void)hipEventDestroy(s.done); }
    }
    for (int i = 0; i < mkt_ctx::kOut; ++i) {
        mkt_ctx::OutSlot& o = c->outs[i];
        if (o.d_pairs) (void)hipFree(o.d_pairs);
        if (o.d_sam) (void)hipFree(o.d_sam);
        if (o.h) (void)hipHostFree(o.h);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (void* u : c->uploads) (void)hipFree(u);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_pairs) (void)hipFree(c->d_pairs);
    if (c->d_sam) (void)hipFree(c->d_sam);
    if (c->d_sc) (void)hipFree(c->d_sc);
    if (c->d_sc_tmp) (void)hipFree(c->d_sc_tmp);
    if (c->d_keys_raw) (void)hipFree(c->d_keys_raw);
    if (c->d_key_list) (void)hipFree(c->d_key_list);
    if (c->d_chr) (void)hipFree(c->d_chr);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_run) (void)hipFree(c->d_run);
    if (c->d_syn) (void)hipFree(c->d_syn);
    if (c->d_syn_sizes) (void)hipFree(c->d_syn_sizes);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->d_sc_logged) (void)hipFree(c->d_sc_logged);
    if (c->d_probe) (void)hipFree(c->d_probe);
    if (c->h_probe) (void)hipHostFree(c->h_probe);
    if (c->h_chr_stage) (void)hipHostFree(c->h_chr_stage);
    if (c->d_dd_flags) (void)hipFree(c->d_dd_flags);
    if (c->d_dd_work) (void)hipFree(c->d_dd_work);
    if (c->d_dd_res) (void)hipFree(c->d_dd_res);
    if (c->d_perm) (void)hipFree(c->d_perm);
    if (c->d_part_hist) (void)hipFree(c->d_part_hist);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->h_dd_res) (void)hipHostFree(c->h_dd_res);
    if (c->d_dense) (void)hipFree(c->d_dense);
    if (c->d_chr_counts) (void)hipFree(c->d_chr_counts);
    if (c->h_chr_counts) (void)hipHostFree(c->h_chr_counts);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int ensure_sc_list(mkt_ctx* c, size_t need);

// MKT_TILES_AUTO.  The bytes per tile follow the input's line length, looked at BEFORE the first launch (lines of the first MiB:
// counted by the host on the streaming path, by one small kernel on the resident path); lean_dims() turns it into a window
// of ~122 lines.  Should a block all the same leave more than one tile in eight to the generic kernel (the lines got shorter on
// the way), the following blocks use tiles of 0.6 x the bytes.
static void set_dims_from_avg(mkt_ctx* c, double bytes_per_line) {
    c->dims_probed = true;
    c->halo_widened = 0;
    if (c->cfg == CFG_SMALL || c->p.tiles != MKT_TILES_AUTO) return;
    c->dims = lean_dims(bytes_per_line);
    if (getenv("MKT_DEBUG_SYNC")) fprintf(stderr, "tile geometry: %.1f bytes per line -> tile %u, halos %u / %u\n", bytes_per_line, c->dims.tile, c->dims.hb, c->dims.hf);
}
static bool shrink_dims(mkt_ctx* c) {
    if (c->cfg == CFG_SMALL || c->dims.tile <= 2048u) return false;
    auto f = [](uint32_t x, uint32_t lo) { uint32_t y = ((uint32_t)(x * 0.6) + 15u) & ~15u; return y < lo ? lo : y; };
    c->dims.tile = f(c->dims.tile, 2048u); c->dims.hb = f(c->dims.hb, 256u); c->dims.hf = f(c->dims.hf, 512u);
    return true;
}
// halos half as wide again, the window as it was (the tile gives the bytes): for an input whose groups reach further than the
// default halos (many lines per read name), seen as tiles left to the generic kernel
static bool widen_halos(mkt_ctx* c) {
    if (c->cfg == CFG_SMALL || c->halo_widened >= 2) return false;
    const TileDims mx = max_dims();
    auto f = [](uint32_t x, uint32_t hi) { uint32_t y = ((uint32_t)(x * 1.5) + 15u) & ~15u; return y > hi ? hi : y; };
    const uint32_t hb = f(c->dims.hb, mx.hb), hf = f(c->dims.hf, mx.hf);
    const uint32_t delta = (hb - c->dims.hb) + (hf - c->dims.hf);
    if (delta == 0 || c->dims.tile < delta + 2048u) return false;
    c->dims.tile -= delta; c->dims.hb = hb; c->dims.hf = hf;
    ++c->halo_widened;
    if (getenv("MKT_DEBUG_SYNC")) fprintf(stderr, "tile geometry: halos widened -> tile %u, halos %u / %u\n", c->dims.tile, c->dims.hb, c->dims.hf);
    return true;
}
static bool adapt_geometry(mkt_ctx* c, const BlockResult& r) {
    if (c->p.tiles != MKT_TILES_AUTO || c->p.ordered) return false;
    if (r.tiles >= 1000 && (uint64_t)r.pad2 * 500 > r.tiles && widen_halos(c)) return true;      // more than 0.2 % of the tiles deferred for their halos
    if (r.tiles >= 8 && (uint64_t)(r.pad - r.pad2) * 8 > r.tiles) return shrink_dims(c);
    return false;
}
// bytes per line of device-resident text (its first MiB); the context's stream is idle afterwards.  0: could not tell
static double probe_device_lines(mkt_ctx* c, const uint8_t* d_text, size_t n) {
    const size_t look = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);
    if (!look) return 0;
    if (!c->d_probe) {
        if (hipMalloc((void**)&c->d_probe, sizeof(unsigned long long)) != hipSuccess) return 0;
        if (hipHostMalloc((void**)&c->h_probe, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return 0;
    }
    if (hipMemsetAsync(c->d_probe, 0, sizeof(unsigned long long), c->stream) != hipSuccess) return 0;
    if (launch_count_newlines(d_text, look, c->d_probe, c->stream) != hipSuccess) return 0;
    if (hipMemcpyAsync(c->h_probe, c->d_probe, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    return *c->h_probe ? (double)look / (double)*c->h_probe : (double)look;
}
// a block came back with its line table overflown (E_LINE_TABLE: its lines are shorter than the geometry was chosen for): tiles
// for ITS line length if that makes them smaller, else 0.6 x the bytes, in the end the 256-byte tiles
static void smaller_geometry(mkt_ctx* c, const uint8_t* d_text, size_t n) {
    if (c->cfg != CFG_SMALL && d_text) {
        const double avg = probe_device_lines(c, d_text, n);
        const TileDims d = avg > 0 ? lean_dims(avg) : c->dims;
        if (d.tile < c->dims.tile) { c->dims = d; return; }
    }
    if (!shrink_dims(c)) { c->cfg = CFG_SMALL; c->dims = small_dims(); }
}

// ---------------------------------------------------------------------------------------------
// enqueue one block: memset workspace, tile kernel (timed), finish kernel, result D2H into slot.
// so (streaming path): the block's outputs are also gathered into the contiguous buffers of an output slot, the timing
// events are the input slot's own, and `done` is recorded behind everything.
struct StreamOut { uint8_t* gp; size_t gp_cap; uint8_t* gs; size_t gs_cap; hipEvent_t k0, k1, done; };
static int enqueue_block(mkt_ctx* c, const uint8_t* d_text, size_t n, int cfg, const TileDims& dims, size_t slot, const StreamOut* so = nullptr) {
    if (((uintptr_t)d_text & 15u) != 0) return fail(c, MKT_E_ARG, "block text must be 16-byte aligned");
    if (n >= kMaxBlock) return fail(c, MKT_E_ARG, "block of %zu bytes: must be < 2 GiB - 64 KiB", n);
    const uint32_t ntiles = num_tiles((uint32_t)n, dims.tile);
    int rc = ensure_ws(c, ntiles);
    if (rc) return rc;
    // output capacities: .sam is at most the block (+1 for a missing final newline); .pairs is
    // checked in-kernel and grown on demand (the result carries the exact size)
    if ((rc = ensure_dev(c, &c->d_pairs, &c->pairs_cap, n / 3 + 65536))) return rc;
    if (c->P.write_sam && (rc = ensure_dev(c, &c->d_sam, &c->sam_cap, n + n / 4 + 65536))) return rc;
    if (!c->d_sc && (rc = ensure_sc_list(c, 1))) return rc;
    KArgs a;
    memset(&a, 0, sizeof a);
    uint8_t* w = c->d_ws;
    a.text = d_text; a.n = (uint32_t)n; a.ntiles = ntiles; a.dims = dims; a.P = c->P;
    a.descA = (uint64_t*)w; w += (size_t)ntiles * 8;
    a.descB = (uint64_t*)w; w += (size_t)ntiles * 8;
    a.descC = (uint64_t*)w; w += (size_t)ntiles * 8;
    a.tile_last = (TileLast*)w; w += (size_t)ntiles * sizeof(TileLast);
    a.tile_groups = (uint64_t*)w; w += (size_t)ntiles * sizeof(uint64_t);
    a.defer_list = (uint32_t*)w;
    w = c->d_ws + ws_tiles_bytes(ntiles);
    a.cur = (RegionCur*)w; w += kMaxRegions * sizeof(RegionCur);
    a.ticket = (uint32_t*)w;
    a.defer_count = (uint32_t*)(w + 64);
    uint32_t* ticket2 = (uint32_t*)(w + 128);
    a.last_tile = (int*)(w + 192);
    a.scan_ticket = (uint32_t*)(w + 224);
    w += 256;
    a.scan_desc = (uint64_t*)w; w += 1024 * sizeof(uint64_t);
    if ((ntiles + finish_chunk_tiles() - 1) / finish_chunk_tiles() > 1024) return fail(c, MKT_E_ARG, "block has too many tiles for the finish scan");
    a.res = (BlockResult*)w;
    a.ordered = c->p.ordered ? 1 : 0;
    a.run = c->d_run;
    // any-order mode: outputs in kMaxRegions equal slices (one cursor line each); ordered mode: one region
    a.nregions = c->p.ordered ? 1 : kMaxRegions;
    {
        const size_t need = (size_t)a.nregions * ((n / 256 / (size_t)a.nregions) * 2 + 1024);
        if (c->sc_tmp_cap < need) {
            if (c->d_sc_tmp) { if ((rc = sync_all(c))) return rc; HIPCHK(c, hipFree(c->d_sc_tmp)); }
            c->d_sc_tmp = nullptr; c->sc_tmp_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_sc_tmp, need * sizeof(uint64_t)));
            c->sc_tmp_cap = need;
        }
    }
    a.pairs_rcap = (c->pairs_cap / a.nregions) & ~(uint64_t)15;
    a.sam_rcap = c->P.write_sam ? ((c->sam_cap / a.nregions) & ~(uint64_t)15) : 0;
    a.sc_rcap = c->sc_tmp_cap / a.nregions;
    a.out.pairs = c->d_pairs; a.out.pairs_cap = c->pairs_cap;
    a.out.sam = c->d_sam; a.out.sam_cap = c->P.write_sam ? c->sam_cap : 0;
    a.out.sc = c->d_sc_tmp; a.out.sc_cap = c->sc_tmp_cap;
    a.sc_list = c->d_sc; a.sc_list_cap = c->sc_cap;
    if (c->p.extensions & MKT_EXT_KEYS) {
        if (!c->d_chr) { HIPCHK(c, hipMalloc((void**)&c->d_chr, sizeof(ChrTab))); HIPCHK(c, hipMemsetAsync(c->d_chr, 0, sizeof(ChrTab), c->stream)); }
        // at most one reported pair per two 32-byte lines; twice that per region for imbalance
        const size_t per = (n / 64 / (size_t)a.nregions) * 2 + 4096, need = per * a.nregions;
        if (c->keys_raw_cap < need) {
            if (c->d_keys_raw) { if ((rc = sync_all(c))) return rc; HIPCHK(c, hipFree(c->d_keys_raw)); }
            c->d_keys_raw = nullptr; c->keys_raw_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_keys_raw, need * sizeof(KeyRec)));
            c->keys_raw_cap = need;
        }
        // the run's list grows by doubling (the stream is idle whenever it has to: growth syncs)
        // room for the records of the blocks in flight: one pair per 64 input bytes until a sync has shown this input's
        // density, afterwards twice the highest density seen (k_finish checks the real count: too small is an error)
        const double per_byte = c->key_density > 0 ? (c->key_density * 2 < 1.0 / 64 ? c->key_density * 2 : 1.0 / 64) : 1.0 / 64;
        const size_t want = (size_t)c->acc.emitted + (size_t)((double)(c->bytes_unsynced + n) * per_byte) + 65536;
        if (c->key_list_cap < want) {
            size_t ncap = c->key_list_cap ? c->key_list_cap * 2 : ((size_t)1 << 22);
            while (ncap < want) ncap *= 2;
            KeyRec* nl = nullptr;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipMalloc((void**)&nl, ncap * sizeof(KeyRec)));
            if (c->d_key_list) { HIPCHK(c, hipMemcpy(nl, c->d_key_list, c->key_list_cap * sizeof(KeyRec), hipMemcpyDeviceToDevice)); HIPCHK(c, hipFree(c->d_key_list)); }
            c->d_key_list = nl; c->key_list_cap = ncap;
        }
        a.keys_rcap = c->keys_raw_cap / a.nregions;
        a.out.keys = c->d_keys_raw; a.out.keys_cap = c->keys_raw_cap; a.out.chr = c->d_chr;
        a.out.key_lanes = (c->p.extensions & MKT_EXT_LANES) ? 1u : 0u;
        a.key_list = c->d_key_list; a.key_list_cap = c->key_list_cap;
    }
#if defined(MKT_STAMPS)
    if (!c->d_stamps) { HIPCHK(c, hipMalloc((void**)&c->d_stamps, 16 * sizeof(unsigned long long))); HIPCHK(c, hipMemset(c->d_stamps, 0, 16 * sizeof(unsigned long long))); }
    a.stamps = getenv("MKT_NO_STAMPS") ? nullptr : c->d_stamps;
    { const char* e = getenv("MKT_DEBUG_STOP"); a.debug_stop = e ? atoi(e) : 0; }
#endif
    HIPCHK(c, hipMemsetAsync(c->d_ws, 0, ws_bytes_for(ntiles), c->stream));
    hipEvent_t e0, e1;
    if (so) { e0 = so->k0; e1 = so->k1; }
    else {
        HIPCHK(c, hipEventCreate(&e0));
        HIPCHK(c, hipEventCreate(&e1));
        c->ev.push_back(e0); c->ev.push_back(e1); c->ev_bytes.push_back(n);
    }
    const uint32_t max_wgs = fast_max_workgroups(cfg);       // every workgroup resident, tiles dealt statically
    int grid = (int)(ntiles < max_wgs ? ntiles : max_wgs);
    const bool lean = !c->p.ordered && cfg != CFG_SMALL && !c->no_lean;
    HIPCHK(c, hipEventRecord(e0, c->stream));
    if (lean) {
        // lean kernel over all tiles, then the generic kernel over the tiles it deferred
        HIPCHK(c, launch_fast(a, cfg, grid, c->stream));
        HIPCHK(c, hipEventRecord(e1, c->stream));
        KArgs b = a;
        b.use_list = 1; b.ticket = ticket2;
        HIPCHK(c, launch_tiles(b, cfg, ntiles < 96u ? (int)ntiles : 96, c->stream));
    } else {
        HIPCHK(c, launch_tiles(a, cfg, grid, c->stream));
        HIPCHK(c, hipEventRecord(e1, c->stream));
    }
    HIPCHK(c, launch_finish(a, c->stream));
    if (so) {
        const uint64_t prc = a.nregions > 1 ? a.pairs_rcap : 0, src_ = a.nregions > 1 ? a.sam_rcap : 0;
        HIPCHK(c, launch_gather(a.res, c->d_pairs, prc, so->gp, so->gp_cap, 0, n / 8, c->stream));
        if (c->P.write_sam) HIPCHK(c, launch_gather(a.res, c->d_sam, src_, so->gs, so->gs_cap, 1, n, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(&c->h_res[slot], a.res, sizeof(BlockResult), hipMemcpyDeviceToHost, c->stream));
    if (so) HIPCHK(c, hipEventRecord(so->done, c->stream));
    return MKT_OK;
}

static void fold_timing(mkt_ctx* c) {       // stream must be idle
    for (size_t k = 0; k + 1 < c->ev.size(); k += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]) == hipSuccess) { c->folded_ms += ms; ++c->folded_launches; c->folded_bytes += c->ev_bytes[k / 2]; }
        (void)hipEventDestroy(c->ev[k]); (void)hipEventDestroy(c->ev[k + 1]);
    }
    c->ev.clear(); c->ev_bytes.clear();
}

// The run's resolved self-circle indices stay on the device until the end of the input (k_sc_logged).
// Stream idle: make room for `need` entries, keeping the c->acc.sc entries already there.
static int ensure_sc_list(mkt_ctx* c, size_t need) {
    if (need <= c->sc_cap) return MKT_OK;
    size_t ncap = c->sc_cap ? c->sc_cap : ((size_t)1 << 20);       // 1 Mi entries (8 MiB) at least; the resident path asks for its first 2 GB block's worth at once
    while (ncap < need) ncap *= 2;
    uint64_t* nl = nullptr;
    HIPCHK(c, hipMalloc((void**)&nl, ncap * sizeof(uint64_t)));
    if (c->d_sc) {
        if (c->acc.sc) HIPCHK(c, hipMemcpy(nl, c->d_sc, (size_t)c->acc.sc * sizeof(uint64_t), hipMemcpyDeviceToDevice));
        HIPCHK(c, hipFree(c->d_sc));
    }
    c->d_sc = nl; c->sc_cap = ncap;
    return MKT_OK;
}
// after a sync: what this input's self-circle density looks like (entries per input byte, highest seen)
static void note_sc_density(mkt_ctx* c) {
    if (c->bytes_unsynced) {
        const double d = (double)c->sc_unfolded / (double)c->bytes_unsynced;
        if (d > c->sc_density) c->sc_density = d;
        if (c->sc_density == 0) c->sc_density = 1e-12;   // seen, none so far
        const double k = (double)c->emitted_unfolded / (double)c->bytes_unsynced;
        if (k > c->key_density) c->key_density = k;
        if (c->key_density == 0) c->key_density = 1e-12;
    }
    c->sc_unfolded = 0; c->bytes_unsynced = 0; c->emitted_unfolded = 0;
}
// entries the next `bytes` of input may add at most, as far as the host can tell: one group per 64 bytes until a sync
// has shown this input's density, afterwards 4 x the highest density seen and at least one per 65536 bytes.  k_finish
// checks the real count against the capacity: a wrong guess is an error (E_SC_CAP), never a silent loss.
static size_t sc_estimate(const mkt_ctx* c, size_t bytes) {
    const double per_byte = c->sc_density > 0 ? (c->sc_density * 4 > 1.0 / 65536 ? c->sc_density * 4 : 1.0 / 65536) : 1.0 / 64;
    return (size_t)((double)bytes * per_byte) + 4096;
}

static int check_result(mkt_ctx* c, const BlockResult& r) {
    if (r.err == 0) return MKT_OK;
    return fail(c, MKT_E_KERNEL, "kernel error bits 0x%x%s%s%s%s%s%s", r.err,
                (r.err & E_LINE_TABLE) ? " [line table overflow: use MKT_TILES_SMALL]" : "",
                (r.err & E_LOOKBACK) ? " [look-back timeout]" : "",
                (r.err & E_PAIRS_CAP) ? " [.pairs buffer]" : "", (r.err & E_SAM_CAP) ? " [.sam buffer]" : "",
                (r.err & E_SC_CAP) ? " [self-circle buffer]" : "", (r.err & E_FIELD_RANGE) ? " [field > 65535 bytes]" : "");
}

// ---------------------------------------------------------------------------------------------
// Streaming path.  Caller thread: fills input slot `cur`, cuts it on a group boundary, queues H2D + kernels + gather for
// the prefix and moves on to the next slot with the carry.  Worker thread: per job, in order -- wait for the result, fold
// it into the run, copy the gathered outputs into a pinned staging slot, hold back the block's last group (quirk Q1),
// publish the rest.  Everything shared is guarded by c->mu; blocking waits on the GPU happen outside it.

static int stream_start(mkt_ctx* c) {
    if (!c->s_in) HIPCHK(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    if (!c->s_out) HIPCHK(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    return MKT_OK;
}
static int stream_alloc_in(mkt_ctx* c, int i) {
    mkt_ctx::InSlot& s = c->in[i];
    if (!s.h) HIPCHK(c, hipHostMalloc((void**)&s.h, c->block_cap + 64, hipHostMallocDefault));
    return MKT_OK;
}
static int stream_alloc_dev(mkt_ctx* c, int i) {
    mkt_ctx::InSlot& s = c->in[i];
    if (!s.d) HIPCHK(c, hipMalloc((void**)&s.d, c->block_cap + 64));
    if (!s.h2d) {
        HIPCHK(c, hipEventCreateWithFlags(&s.h2d, hipEventDisableTiming));
        HIPCHK(c, hipEventCreate(&s.k0));
        HIPCHK(c, hipEventCreate(&s.k1));
        HIPCHK(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    return MKT_OK;
}
static void worker_main(mkt_ctx* c);

// queue the GPU work of one job (c->mu held; the job's input is in d_in already or on its way on s_in)
static int stream_launch(mkt_ctx* c, const mkt_ctx::Job& j) {
    mkt_ctx::InSlot& is = c->in[j.in_slot];
    mkt_ctx::OutSlot& os = c->outs[j.out_slot];
    int rc;
    // the region buffers are sized here once for a whole block (never regrown in flight except by a replay, which is idle)
    if ((rc = ensure_dev(c, &c->d_pairs, &c->pairs_cap, c->block_cap / 3 + 65536))) return rc;
    if (c->P.write_sam && (rc = ensure_dev(c, &c->d_sam, &c->sam_cap, c->block_cap + c->block_cap / 4 + 65536))) return rc;
    if (os.d_pairs_cap < c->pairs_cap) { if ((rc = ensure_dev(c, &os.d_pairs, &os.d_pairs_cap, c->pairs_cap))) return rc; }
    if (c->P.write_sam && os.d_sam_cap < c->sam_cap) { if ((rc = ensure_dev(c, &os.d_sam, &os.d_sam_cap, c->sam_cap))) return rc; }
    StreamOut so;
    so.gp = os.d_pairs; so.gp_cap = os.d_pairs_cap; so.gs = os.d_sam; so.gs_cap = os.d_sam_cap;
    so.k0 = is.k0; so.k1 = is.k1; so.done = is.done;
    return enqueue_block(c, is.d, j.n, j.cfg, j.dims, c->res_slots - mkt_ctx::kIn + (size_t)j.in_slot, &so);
}

// c->mu held by lk.  Hands input slot `slot` (n bytes) to the GPU.
static int stream_enqueue(mkt_ctx* c, std::unique_lock<std::mutex>& lk, int slot, size_t n) {
    int rc;
    if ((rc = stream_start(c))) return rc;
    const int oslot = (int)(c->seq % mkt_ctx::kOut);
    // the output slot's device buffers are free once the job two back has been copied out
    c->cv.wait(lk, [&] { return c->async_rc != MKT_OK || !c->outs[oslot].dev_busy; });
    if (c->async_rc) return c->async_rc;
    // room in the run's self-circle list for every block in flight at one group per 64 input bytes; growing it needs the
    // pipeline idle (only the entries of folded blocks are carried over)
    const size_t per_block = c->block_cap / 64 + 4096;
    if (!c->d_sc || (size_t)c->acc.sc + (c->jobs.size() + 1) * per_block > c->sc_cap) {
        c->cv.wait(lk, [&] { return c->async_rc != MKT_OK || c->jobs.empty(); });
        if (c->async_rc) return c->async_rc;
        if ((rc = ensure_sc_list(c, 2 * (size_t)c->acc.sc + (size_t)(mkt_ctx::kOut + 1) * per_block))) return rc;
    }
    if ((rc = stream_alloc_dev(c, slot))) return rc;
    mkt_ctx::InSlot& is = c->in[slot];
    HIPCHK(c, hipMemcpyAsync(is.d, is.h, n, hipMemcpyHostToDevice, c->s_in));
    HIPCHK(c, hipEventRecord(is.h2d, c->s_in));
    HIPCHK(c, hipStreamWaitEvent(c->stream, is.h2d, 0));
    if (!c->dims_probed) {                                 // the input's line length, from the first MiB (in pinned host memory)
        const size_t look = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);
        size_t lines = 0;
        for (const uint8_t* q = is.h, *e = is.h + look; q < e && (q = (const uint8_t*)memchr(q, '\n', (size_t)(e - q))); ++q) ++lines;
        set_dims_from_avg(c, lines ? (double)look / (double)lines : (double)look);
    }
    mkt_ctx::Job j;
    j.in_slot = slot; j.out_slot = oslot; j.n = n; j.cfg = c->cfg; j.dims = c->dims; j.attempts = 0;
    c->bytes_unsynced = 0;
    for (const mkt_ctx::Job& q : c->jobs) c->bytes_unsynced += q.n;     // extension: key-list reservation covers the blocks in flight
    if ((rc = stream_launch(c, j))) return rc;
    is.busy = true; c->outs[oslot].dev_busy = true;
    c->jobs.push_back(j);
    ++c->seq;
    if (!c->worker_started) { c->worker_started = true; c->worker = std::thread(worker_main, c); }
    c->cv.notify_all();
    return MKT_OK;
}

// A job came back with error bits (c->mu held, worker thread): fix the cause the way a synchronous run would -- next
// smaller tile geometry, bigger output buffers -- and run it again, followed by every job queued behind it (they ran on
// top of run totals that the failed block never advanced).  Inputs are still in their device slots.
static int stream_replay(mkt_ctx* c, const BlockResult& r) {
    int rc;
    if ((rc = sync_all(c))) return rc;
    mkt_ctx::Job& j0 = c->jobs.front();
    if (++j0.attempts > 4) return check_result(c, r);
    bool fixed = false;
    if ((r.err & (E_LINE_TABLE | E_OVF_SLOTS)) && j0.cfg != CFG_SMALL && c->p.tiles == MKT_TILES_AUTO) {
        const int keep_cfg = c->cfg; const TileDims keep = c->dims;
        c->cfg = j0.cfg; c->dims = j0.dims;
        smaller_geometry(c, c->in[j0.in_slot].d, j0.n);
        for (mkt_ctx::Job& q : c->jobs) { q.cfg = c->cfg; q.dims = c->dims; }
        if (c->cfg == CFG_SMALL) { c->cfg = keep_cfg; c->dims = keep; }      // (the 256-byte tiles are a last resort per block: the stream keeps its lean geometry)
        fixed = true;
    } else {
        const uint32_t nr = r.nregions ? r.nregions : 1;
        if (r.err & E_PAIRS_CAP) {
            uint64_t mx = 0; for (uint32_t q = 0; q < nr; ++q) if (r.rpair[q] > mx) mx = r.rpair[q];
            if ((rc = ensure_dev(c, &c->d_pairs, &c->pairs_cap, (size_t)(mx * nr) + mx / 4 * nr + 65536))) return rc;
            fixed = true;
        }
        if (r.err & E_SAM_CAP) {
            uint64_t mx = 0; for (uint32_t q = 0; q < nr; ++q) if (r.rsam[q] > mx) mx = r.rsam[q];
            if ((rc = ensure_dev(c, &c->d_sam, &c->sam_cap, (size_t)(mx * nr) + mx / 4 * nr + 65536))) return rc;
            fixed = true;
        }
        if (r.err & E_SC_CAP) {            // per-block raw entries: grow the slices, and the run's list with them
            const size_t need = (size_t)r.sc * 4 + 65536;
            if (c->d_sc_tmp) HIPCHK(c, hipFree(c->d_sc_tmp));
            c->d_sc_tmp = nullptr; c->sc_tmp_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_sc_tmp, need * kMaxRegions * sizeof(uint64_t)));
            c->sc_tmp_cap = need * kMaxRegions;
            if ((rc = ensure_sc_list(c, (size_t)c->acc.sc + (c->jobs.size() + 1) * need))) return rc;
            c->key_density = 0;            // extension: the key list may be what overflowed: back to the worst-case reservation
            fixed = true;
        }
    }
    if (!fixed) return check_result(c, r);
    // the run totals on the device go back to what the folded blocks left
    DevRun dr;
    dr.groups = c->acc.groups; dr.sc = c->acc.sc; dr.emitted = c->acc.emitted;
    HIPCHK(c, hipMemcpy(c->d_run, &dr, sizeof dr, hipMemcpyHostToDevice));
    c->bytes_unsynced = 0;
    for (const mkt_ctx::Job& q : c->jobs) {
        if ((rc = stream_launch(c, q))) return rc;
        c->bytes_unsynced += q.n;
    }
    return MKT_OK;
}

static void worker_fail(mkt_ctx* c, int rc) {       // c->mu held
    if (c->async_rc == MKT_OK) c->async_rc = rc ? rc : MKT_E_HIP;
    c->cv.notify_all();
}

static void worker_main(mkt_ctx* c) {
    (void)hipSetDevice(c->p.device);
    std::unique_lock<std::mutex> lk(c->mu);
    for (;;) {
        c->cv.wait(lk, [&] { return c->stop || (!c->jobs.empty() && c->async_rc == MKT_OK); });
        if (c->stop) return;
        const mkt_ctx::Job j = c->jobs.front();
        hipEvent_t done = c->in[j.in_slot].done;
        lk.unlock();
        hipError_t he = hipEventSynchronize(done);
        lk.lock();
        if (c->stop) return;
        if (he != hipSuccess) { fail(c, MKT_E_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(he)); worker_fail(c, MKT_E_HIP); continue; }
        const BlockResult r = c->h_res[c->res_slots - mkt_ctx::kIn + (size_t)j.in_slot];
        if (r.err) {
            const int rc = stream_replay(c, r);
            if (rc) worker_fail(c, rc);
            continue;                                  // wait for the re-run of the same job
        }
        // ---- fold the block into the run
        c->acc.add_block(r); c->tiles_total += r.tiles; c->tiles_deferred += r.pad; adapt_geometry(c, r);
        ++c->blocks;
        c->sc_unfolded += r.sc; c->emitted_unfolded += r.emitted; c->bytes_unsynced = j.n;
        note_sc_density(c);
        {
            float ms = 0;
            mkt_ctx::InSlot& is = c->in[j.in_slot];
            if (hipEventElapsedTime(&ms, is.k0, is.k1) == hipSuccess) { c->folded_ms += ms; ++c->folded_launches; c->folded_bytes += j.n; }
        }
        const size_t pb = (size_t)r.pair_bytes, sb = c->P.write_sam ? (size_t)r.sam_bytes : 0;
        mkt_ctx::OutSlot& os = c->outs[j.out_slot];
        // ---- copy the gathered outputs out (the staging slot must be back from the consumer)
        while (os.host_busy && !c->stop) {
            if (!c->consumer_async) {
                // single-threaded caller (submit ... drain later): nobody will release the slot while the caller is inside
                // mkt_submit, so its published chunk moves to the heap instead
                for (mkt_ctx::Chunk& q : c->ready)
                    if (q.out_slot == j.out_slot) {
                        q.own_pairs.assign(q.pairs, q.pairs + q.pairs_len); q.own_sam.assign(q.sam, q.sam + q.sam_len);
                        q.pairs = q.own_pairs.data(); q.sam = q.own_sam.data(); q.out_slot = -1;
                    }
                if (!(c->handed_valid && c->handed.out_slot == j.out_slot)) { os.host_busy = false; break; }
            }
            c->cv.wait(lk);
        }
        if (c->stop) return;
        const size_t H = mkt_ctx::kHead;
        const size_t sam_at = ((H + pb + 4095) & ~(size_t)4095) + H, need = sam_at + sb + 64;
        bool bad = false;
        if (pb + sb) {
            if (os.h_cap < need) {
                if (os.h) (void)hipHostFree(os.h);
                os.h = nullptr; os.h_cap = 0;
                const size_t want = need + need / 4;
                if (hipHostMalloc((void**)&os.h, want, hipHostMallocDefault) != hipSuccess) { fail(c, MKT_E_NOMEM, "pinned staging of %zu bytes", want); worker_fail(c, MKT_E_NOMEM); bad = true; }
                else os.h_cap = want;
            }
            if (!bad) {
                lk.unlock();
                hipError_t e1 = pb ? hipMemcpyAsync(os.h + H, os.d_pairs, pb, hipMemcpyDeviceToHost, c->s_out) : hipSuccess;
                hipError_t e2 = sb ? hipMemcpyAsync(os.h + sam_at, os.d_sam, sb, hipMemcpyDeviceToHost, c->s_out) : hipSuccess;
                hipError_t e3 = hipStreamSynchronize(c->s_out);
                lk.lock();
                if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { fail(c, MKT_E_HIP, "output copy failed"); worker_fail(c, MKT_E_HIP); bad = true; }
            }
        }
        if (c->stop) return;
        if (bad) continue;
        // ---- quirk Q1: the newest group stays back until a later group supersedes it
        mkt_ctx::Chunk ch;
        ch.out_slot = j.out_slot;
        const char* hp = (const char*)os.h + H;
        const char* hs = (const char*)os.h + sam_at;
        if (r.last.valid) {
            const size_t tp = r.last.pair_bytes, ts = c->P.write_sam ? r.last.sam_bytes : 0;      // gathered layout: [ the rest | last group ]
            const size_t bp = pb - tp, bs = sb - ts;
            ch.pairs = hp; ch.pairs_len = bp; ch.sam = hs; ch.sam_len = bs;
            if (!c->tail_pairs.empty() || !c->tail_sam.empty()) {
                if (pb + sb && c->tail_pairs.size() <= H && c->tail_sam.size() <= H) {             // in front of this block's bytes
                    if (!c->tail_pairs.empty()) { memcpy(os.h + H - c->tail_pairs.size(), c->tail_pairs.data(), c->tail_pairs.size()); ch.pairs = hp - c->tail_pairs.size(); ch.pairs_len += c->tail_pairs.size(); }
                    if (!c->tail_sam.empty()) { memcpy(os.h + sam_at - c->tail_sam.size(), c->tail_sam.data(), c->tail_sam.size()); ch.sam = hs - c->tail_sam.size(); ch.sam_len += c->tail_sam.size(); }
                } else {                                                                           // as a chunk of its own
                    mkt_ctx::Chunk t;
                    t.own_pairs.swap(c->tail_pairs); t.own_sam.swap(c->tail_sam);
                    c->ready.push_back(std::move(t));
                }
            }
            if (tp) c->tail_pairs.assign(hp + bp, hp + bp + tp); else c->tail_pairs.clear();
            if (ts) c->tail_sam.assign(hs + bs, hs + bs + ts); else c->tail_sam.clear();
        } else {
            ch.pairs = hp; ch.pairs_len = pb; ch.sam = hs; ch.sam_len = sb;                        // a block without any group reports nothing
        }
        if (ch.pairs_len + ch.sam_len) { os.host_busy = true; c->ready.push_back(std::move(ch)); }
        os.dev_busy = false;
        c->in[j.in_slot].busy = false;
        c->jobs.pop_front();
        c->cv.notify_all();
    }
}

// c->mu held by lk: every queued job folded (or an error)
static int stream_wait_idle(mkt_ctx* c, std::unique_lock<std::mutex>& lk) {
    c->cv.wait(lk, [&] { return c->async_rc != MKT_OK || c->jobs.empty(); });
    return c->async_rc;
}

// the input slot is full (or the input ended): queue its group-aligned prefix, carry the rest into the next slot
static int stream_flush(mkt_ctx* c, bool everything) {
    std::unique_lock<std::mutex> lk(c->mu);
    if (c->async_rc) return c->async_rc;
    mkt_ctx::InSlot& is = c->in[c->cur];
    if (everything) {
        if (c->h_len) { int rc = stream_enqueue(c, lk, c->cur, c->h_len); if (rc) return rc; }
        c->cur = (c->cur + 1) % mkt_ctx::kIn;
        c->h_len = 0;
        return MKT_OK;
    }
    size_t end = 0;
    const size_t cut = group_aligned_prefix((const char*)is.h, c->h_len, c->P.min_mapq, &end);
    if (cut == 0) {
        // one group (or none closed) in the whole slot: lines that the filter drops influence nothing, squeeze them out
        const size_t nl = compact_carry((char*)is.h, c->h_len, c->P.min_mapq);
        if (nl + 4096 > c->h_len || nl + 4096 > c->block_cap)
            return fail(c, MKT_E_CAPACITY, "no QNAME-group boundary inside a %zu-byte block: raise block_bytes", c->block_cap);
        c->h_len = nl;
        return MKT_OK;
    }
    const int next = (c->cur + 1) % mkt_ctx::kIn;
    c->cv.wait(lk, [&] { return c->async_rc != MKT_OK || !c->in[next].busy; });
    if (c->async_rc) return c->async_rc;
    int rc = stream_alloc_in(c, next);
    if (rc) return rc;
    size_t carry = c->h_len - cut;
    memcpy(c->in[next].h, is.h + cut, carry);
    if (carry > c->block_cap / 2) carry = compact_carry((char*)c->in[next].h, carry, c->P.min_mapq);
    if ((rc = stream_enqueue(c, lk, c->cur, cut))) return rc;
    c->cur = next;
    c->h_len = carry;
    return MKT_OK;
}

static int stream_check_open(mkt_ctx* c) {
    if (c->input_done || c->finished) return fail(c, MKT_E_STATE, "input after the end of input");
    { std::lock_guard<std::mutex> g(c->mu); if (c->async_rc) return c->async_rc; }
    HIPCHK(c, hipSetDevice(c->p.device));
    return stream_alloc_in(c, c->cur);
}

int mkt_submit(mkt_ctx* c, const char* bytes, size_t n, int last) {
    if (!c) return MKT_E_ARG;
    if (n && !bytes) return fail(c, MKT_E_ARG, "null bytes");
    int rc = stream_check_open(c);
    if (rc) return rc;
    size_t pos = 0;
    c->bytes_in += n;
    for (;;) {
        size_t space = c->block_cap - c->h_len;
        size_t take = n - pos < space ? n - pos : space;
        if (take) { memcpy(c->in[c->cur].h + c->h_len, bytes + pos, take); c->h_len += take; pos += take; }
        const bool all_in = pos == n;
        if (c->h_len == c->block_cap && !(all_in && last)) {
            if ((rc = stream_flush(c, false))) return rc;
            if ((rc = stream_alloc_in(c, c->cur))) return rc;
            continue;
        }
        if (all_in) break;
    }
    if (last) {
        if ((rc = stream_flush(c, true))) return rc;
        c->input_done = true;
    }
    return MKT_OK;
}

int mkt_input_window(mkt_ctx* c, char** buf, size_t* cap) {
    if (!c || !buf || !cap) return MKT_E_ARG;
    int rc = stream_check_open(c);
    if (rc) return rc;
    if (c->h_len == c->block_cap) {
        if ((rc = stream_flush(c, false))) return rc;
        if ((rc = stream_alloc_in(c, c->cur))) return rc;
    }
    *buf = (char*)c->in[c->cur].h + c->h_len;
    *cap = c->block_cap - c->h_len;
    return MKT_OK;
}
int mkt_submit_window(mkt_ctx* c, size_t n, int last) {
    if (!c) return MKT_E_ARG;
    if (c->input_done || c->finished) return fail(c, MKT_E_STATE, "submit after the end of input");
    if (!c->in[c->cur].h || n > c->block_cap - c->h_len) return fail(c, MKT_E_ARG, "more bytes than the input window holds");
    HIPCHK(c, hipSetDevice(c->p.device));
    c->h_len += n;
    c->bytes_in += n;
    int rc;
    if (last) {
        if ((rc = stream_flush(c, true))) return rc;
        c->input_done = true;
    } else if (c->h_len == c->block_cap) {
        if ((rc = stream_flush(c, false))) return rc;
    }
    return MKT_OK;
}

// the staging slot of the chunk handed out last goes back to the worker (c->mu held)
static void release_handed(mkt_ctx* c) {
    if (c->handed_valid) {
        if (c->handed.out_slot >= 0) c->outs[c->handed.out_slot].host_busy = false;
        c->handed = mkt_ctx::Chunk();
        c->handed_valid = false;
        c->cv.notify_all();
    }
}
static void chunk_ptrs(mkt_ctx::Chunk& ch) {        // a heap chunk's pointers follow its vectors (they move with the chunk)
    if (ch.out_slot < 0) { ch.pairs = ch.own_pairs.data(); ch.pairs_len = ch.own_pairs.size(); ch.sam = ch.own_sam.data(); ch.sam_len = ch.own_sam.size(); }
}

int mkt_drain(mkt_ctx* c, mkt_out* out) {
    if (!c || !out) return MKT_E_ARG;
    std::unique_lock<std::mutex> lk(c->mu);
    release_handed(c);
    c->drained_pairs.clear(); c->drained_sam.clear();
    while (!c->ready.empty()) {
        mkt_ctx::Chunk& ch = c->ready.front();
        chunk_ptrs(ch);
        c->drained_pairs.insert(c->drained_pairs.end(), ch.pairs, ch.pairs + ch.pairs_len);
        c->drained_sam.insert(c->drained_sam.end(), ch.sam, ch.sam + ch.sam_len);
        if (ch.out_slot >= 0) c->outs[ch.out_slot].host_busy = false;
        c->ready.pop_front();
    }
    c->cv.notify_all();
    out->pairs = c->drained_pairs.data(); out->pairs_len = c->drained_pairs.size();
    out->sam = c->drained_sam.data(); out->sam_len = c->drained_sam.size();
    return c->async_rc;
}

int mkt_drain_wait(mkt_ctx* c, mkt_out* out, int* done) {
    if (!c || !out || !done) return MKT_E_ARG;
    std::unique_lock<std::mutex> lk(c->mu);
    release_handed(c);
    memset(out, 0, sizeof *out);
    *done = 0;
    c->consumer_async = true;
    c->cv.wait(lk, [&] { return c->async_rc != MKT_OK || !c->ready.empty() || c->finished; });
    if (!c->ready.empty()) {
        c->handed = std::move(c->ready.front());
        c->ready.pop_front();
        c->handed_valid = true;
        chunk_ptrs(c->handed);
        out->pairs = c->handed.pairs; out->pairs_len = c->handed.pairs_len; out->sam = c->handed.sam; out->sam_len = c->handed.sam_len;
        return MKT_OK;
    }
    *done = 1;
    return c->async_rc;
}

int mkt_submit_device(mkt_ctx* c, const void* d_text, size_t n) {
    if (!c) return MKT_E_ARG;
    if (c->finished) return fail(c, MKT_E_STATE, "submit after finish");
    if (!d_text && n) return fail(c, MKT_E_ARG, "null device pointer");
    HIPCHK(c, hipSetDevice(c->p.device));
    // The self-circle list must have room for what the blocks in flight may add (their counts are known at the next sync)
    // ... and the first block of an input is a probe: its result (self-circle density, tiles the lean kernel could not
    // take) is looked at before the second block is queued
    const bool probe = c->probing && c->res_used >= 1;
    if (probe || c->res_used == c->res_slots - mkt_ctx::kIn || (size_t)c->acc.sc + sc_estimate(c, c->bytes_unsynced + n) > c->sc_cap) {
        if (getenv("MKT_DEBUG_SYNC")) fprintf(stderr, "submit_device: sync before block (slots %zu/%zu, unsynced %.1f GB, density %.3g /B, list %llu of %zu)\n",
                                              c->res_used, c->res_slots, (double)c->bytes_unsynced / 1e9, c->sc_density, (unsigned long long)c->acc.sc, c->sc_cap);
        int rc = mkt_sync(c);
        if (rc) return rc;
        // room for as much again as the run holds now, and for this block at the very least
        if ((rc = ensure_sc_list(c, 2 * (size_t)c->acc.sc + sc_estimate(c, n)))) return rc;
    }
    if (!c->dims_probed) {                                 // the input's line length: newlines of the first MiB, counted on the device
        const double avg = probe_device_lines(c, (const uint8_t*)d_text, n);
        if (avg <= 0 && n) return fail(c, MKT_E_HIP, "line-length probe failed: %s", hipGetErrorString(hipGetLastError()));
        set_dims_from_avg(c, avg);
    }
    c->bytes_unsynced += n;
    const int cfg_used = c->cfg;
    const TileDims dims_used = c->dims;
    int rc = enqueue_block(c, (const uint8_t*)d_text, n, cfg_used, dims_used, c->res_used);
    if (rc) return rc;
    if (c->res_text.size() < c->res_slots) { c->res_text.resize(c->res_slots); c->res_n.resize(c->res_slots); }
    c->res_text[c->res_used] = (const uint8_t*)d_text; c->res_n[c->res_used] = n;
    ++c->res_used;
    c->last_n = n; c->last_dims = dims_used; c->last_text = (const uint8_t*)d_text;
    c->bytes_in += n;
    return MKT_OK;
}

int mkt_sync(mkt_ctx* c) {
    if (!c) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    { std::unique_lock<std::mutex> lk(c->mu); const int arc = stream_wait_idle(c, lk); if (arc) return arc; }
    const bool dbg = getenv("MKT_DEBUG_SYNC") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = dbg ? now() : 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double t1 = dbg ? now() : 0;
    fold_timing(c);
    const double t2 = dbg ? now() : 0;
    int rc = MKT_OK;
    // A block whose line table overflowed (shorter lines than the geometry was chosen for: the first block of an input, or
    // a stream whose read length shrinks on the way) is run again with the next smaller geometry -- together with every
    // block queued behind it, which ran on top of run totals the failed block never advanced.  The texts are still resident.
    bool changed = false;
    int replays = 0;
    for (size_t k = c->res_folded; k < c->res_used; ++k) {
        const BlockResult& r = c->h_res[k];
        if (r.err) {
            if ((r.err & (E_LINE_TABLE | E_OVF_SLOTS)) && c->p.tiles == MKT_TILES_AUTO && c->cfg != CFG_SMALL && replays < 4 && c->res_text.size() > k) {
                ++replays;
                smaller_geometry(c, c->res_text[k], c->res_n[k]);
                c->last_dims = c->dims;
                changed = true;
                DevRun dr;
                dr.groups = c->acc.groups; dr.sc = c->acc.sc; dr.emitted = c->acc.emitted;
                HIPCHK(c, hipMemcpy(c->d_run, &dr, sizeof dr, hipMemcpyHostToDevice));
                for (size_t j = k; j < c->res_used; ++j) if ((rc = enqueue_block(c, c->res_text[j], c->res_n[j], c->cfg, c->dims, j))) return rc;
                HIPCHK(c, hipStreamSynchronize(c->stream));
                fold_timing(c);
                --k;                                     // look at the same block again
                continue;
            }
            rc = check_result(c, r);                     // fail loudly
            break;
        }
        c->acc.add_block(r); c->tiles_total += r.tiles; c->tiles_deferred += r.pad;
        changed = adapt_geometry(c, r) || changed;
        c->sc_unfolded += r.sc; c->emitted_unfolded += r.emitted;
        ++c->blocks;
    }
    if (c->res_used > c->res_folded) c->probing = changed;  // a new geometry is checked on one block before queueing ahead
    const size_t nres = c->res_used;
    c->res_used = 0; c->res_folded = 0;
    if (rc == MKT_OK) note_sc_density(c);
    if (dbg) fprintf(stderr, "mkt_sync: %zu blocks, wait %.2f ms, timing fold %.2f ms, results %.2f ms\n", nres, t1 - t0, t2 - t1, now() - t2);
    return rc;
}

int mkt_fetch_last_block(mkt_ctx* c, char* pairs, size_t pairs_cap, size_t* pairs_len, char* sam, size_t sam_cap, size_t* sam_len) {
    if (!c) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the last folded result is not kept per block; re-read it from the device workspace
    BlockResult r;
    const uint32_t ntiles = num_tiles((uint32_t)c->last_n, c->last_dims.tile ? c->last_dims.tile : c->dims.tile);
    const uint8_t* w = c->d_ws + ws_tiles_bytes(ntiles) + ws_fixed_bytes();
    HIPCHK(c, hipMemcpy(&r, w, sizeof r, hipMemcpyDeviceToHost));
    if (pairs_len) *pairs_len = (size_t)r.pair_bytes;
    if (sam_len) *sam_len = c->P.write_sam ? (size_t)r.sam_bytes : 0;
    const uint32_t nreg = r.nregions ? r.nregions : 1;
    const size_t prc = (c->pairs_cap / nreg) & ~(size_t)15, src_ = c->P.write_sam ? ((c->sam_cap / nreg) & ~(size_t)15) : 0;
    if (pairs && r.pair_bytes) {
        if (pairs_cap < r.pair_bytes) return fail(c, MKT_E_ARG, "pairs buffer too small (%llu needed)", (unsigned long long)r.pair_bytes);
        size_t acc = 0;
        for (uint32_t q = 0; q < nreg; ++q) { if (r.rpair[q]) HIPCHK(c, hipMemcpy(pairs + acc, c->d_pairs + (size_t)q * prc, (size_t)r.rpair[q], hipMemcpyDeviceToHost)); acc += (size_t)r.rpair[q]; }
    }
    if (sam && c->P.write_sam && r.sam_bytes) {
        if (sam_cap < r.sam_bytes) return fail(c, MKT_E_ARG, "sam buffer too small (%llu needed)", (unsigned long long)r.sam_bytes);
        size_t acc = 0;
        for (uint32_t q = 0; q < nreg; ++q) { if (r.rsam[q]) HIPCHK(c, hipMemcpy(sam + acc, c->d_sam + (size_t)q * src_, (size_t)r.rsam[q], hipMemcpyDeviceToHost)); acc += (size_t)r.rsam[q]; }
    }
    return MKT_OK;
}

int mkt_finish(mkt_ctx* c, int drop_last, uint64_t group_offset, uint64_t total_groups, mkt_stats* st) {
    if (!c || !st) return MKT_E_ARG;
    int rc = mkt_sync(c);
    if (rc) return rc;
    const uint64_t K = total_groups ? total_groups : c->acc.groups;
    unsigned long long logged = 0;
    if (c->acc.sc) {
        if (!c->d_sc_logged) HIPCHK(c, hipMalloc((void**)&c->d_sc_logged, sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->d_sc_logged, 0, sizeof(unsigned long long), c->stream));
        const uint64_t drop_group = c->acc.drops(drop_last != 0) ? c->acc.groups - 1 : ~0ull;
        HIPCHK(c, launch_sc_logged(c->d_sc, c->acc.sc, drop_group, group_offset, K, (uint32_t)c->p.ref_threads, c->d_sc_logged, c->stream));
        HIPCHK(c, hipMemcpyAsync(&logged, c->d_sc_logged, sizeof logged, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    RunStats s = c->acc.finish_logged(drop_last != 0, logged);
    memset(st, 0, sizeof *st);
    st->lowMap = s.counters[C_LOWMAP]; st->manyHits = s.counters[C_MANYHITS]; st->unpaired = s.counters[C_UNPAIRED];
    st->selfCircle = s.counters[C_SELFCIRCLE]; st->trans = s.counters[C_TRANS];
    st->cis10K = s.counters[C_CIS10K]; st->cis1K = s.counters[C_CIS1K]; st->cis0 = s.counters[C_CIS0];
    st->selfCircle_all = s.selfcircle_all;
    st->groups = s.groups; st->pairs = s.pairs; st->pair_bytes = s.pair_bytes; st->sam_bytes = s.sam_bytes;
    st->bytes_in = c->bytes_in; st->blocks = c->blocks;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (!c->finished) {
            if (!drop_last && (!c->tail_pairs.empty() || !c->tail_sam.empty())) {      // the newest group is final after all: release it
                mkt_ctx::Chunk t;
                t.own_pairs.swap(c->tail_pairs); t.own_sam.swap(c->tail_sam);
                c->ready.push_back(std::move(t));
            }
            c->tail_pairs.clear(); c->tail_sam.clear();
            c->finished = true;
        }
    }
    c->cv.notify_all();
    return MKT_OK;
}

// ---- extensions ---------------------------------------------------------------------------------
static uint64_t ext_key_count(mkt_ctx* c, int drop_last) {
    uint64_t n = c->acc.emitted;
    if (drop_last && c->acc.pending.valid && c->acc.pending.pair_bytes && n) --n;     // quirk Q1: the input's last group reported a pair
    return n;
}
static int ensure_dedup_work(mkt_ctx* c, uint64_t n);
int mkt_ext_dedup(mkt_ctx* c, int drop_last, uint64_t* total, uint64_t* dups, uint8_t* flags, size_t flags_cap) {
    if (!c) return MKT_E_ARG;
    if (!(c->p.extensions & MKT_EXT_KEYS)) return fail(c, MKT_E_STATE, "context created without MKT_EXT_KEYS");
    int rc = mkt_sync(c);
    if (rc) return rc;
    const uint64_t n = ext_key_count(c, drop_last);
    if (total) *total = n;
    if (dups) *dups = 0;
    if (n == 0) return MKT_OK;
    if (flags && flags_cap < n) return fail(c, MKT_E_ARG, "flags buffer too small (%llu needed)", (unsigned long long)n);
    const size_t wb = dedup_work_bytes(n);
    if ((rc = ensure_dedup_work(c, n))) return rc;
    HIPCHK(c, launch_dedup(c->d_key_list, n, c->d_dd_flags, c->d_dd_work, wb, c->d_dd_res, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_dd_res, c->d_dd_res, sizeof(DedupResult), hipMemcpyDeviceToHost, c->stream));
    if (flags) HIPCHK(c, hipMemcpyAsync(flags, c->d_dd_flags, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (dups) *dups = c->h_dd_res->dups;
    return MKT_OK;
}
int mkt_ext_chr_names(mkt_ctx* c, char* out, size_t cap, size_t* len) {
    if (!c || !len) return MKT_E_ARG;
    if (!(c->p.extensions & MKT_EXT_KEYS)) return fail(c, MKT_E_STATE, "context created without MKT_EXT_KEYS");
    int rc = mkt_sync(c);
    if (rc) return rc;
    *len = 0;
    if (!c->d_chr) return MKT_OK;
    std::vector<unsigned long long> hh(kChrSlots);
    std::vector<uint8_t> names((size_t)kChrSlots * 64);
    HIPCHK(c, hipMemcpy(hh.data(), c->d_chr->hash, kChrSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(names.data(), c->d_chr->name, names.size(), hipMemcpyDeviceToHost));
    std::string txt;
    char num[16];
    for (uint32_t s2 = 0; s2 < kChrSlots; ++s2) if (hh[s2]) {
        snprintf(num, sizeof num, "%u", s2);
        txt += num; txt += '\t'; txt.append((const char*)&names[(size_t)s2 * 64], names[(size_t)s2 * 64 + 63]); txt += '\n';
    }
    *len = txt.size();
    if (out) { if (cap < txt.size()) return fail(c, MKT_E_ARG, "buffer too small (%zu needed)", txt.size()); memcpy(out, txt.data(), txt.size()); }
    return MKT_OK;
}
int mkt_ext_keys_fetch(mkt_ctx* c, int drop_last, void* keys, size_t cap_bytes, uint64_t* n) {
    if (!c || !n) return MKT_E_ARG;
    if (!(c->p.extensions & MKT_EXT_KEYS)) return fail(c, MKT_E_STATE, "context created without MKT_EXT_KEYS");
    int rc = mkt_sync(c);
    if (rc) return rc;
    *n = ext_key_count(c, drop_last);
    if (keys && *n) {
        if (cap_bytes < *n * sizeof(KeyRec)) return fail(c, MKT_E_ARG, "key buffer too small (%llu bytes needed)", (unsigned long long)(*n * sizeof(KeyRec)));
        HIPCHK(c, hipMemcpy(keys, c->d_key_list, (size_t)*n * sizeof(KeyRec), hipMemcpyDeviceToHost));
    }
    return MKT_OK;
}
// work buffers of the duplicate marking, kept between calls (GB-sized hipMalloc / hipFree pairs cost more than the marking)
static int ensure_dedup_work(mkt_ctx* c, uint64_t n) {
    const size_t wb = dedup_work_bytes(n);
    if (c->dd_flags_cap < n) {
        if (c->d_dd_flags) HIPCHK(c, hipFree(c->d_dd_flags));
        c->d_dd_flags = nullptr; c->dd_flags_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_dd_flags, n + n / 8 + 4096));
        c->dd_flags_cap = n + n / 8 + 4096;
    }
    if (c->dd_work_cap < wb) {
        if (c->d_dd_work) HIPCHK(c, hipFree(c->d_dd_work));
        c->d_dd_work = nullptr; c->dd_work_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_dd_work, wb + wb / 8));
        c->dd_work_cap = wb + wb / 8;
    }
    if (!c->d_dd_res) {
        HIPCHK(c, hipMalloc((void**)&c->d_dd_res, sizeof(DedupResult)));
        HIPCHK(c, hipHostMalloc((void**)&c->h_dd_res, sizeof(DedupResult), hipHostMallocDefault));
    }
    return MKT_OK;
}
int mkt_ext_dedup_device(mkt_ctx* c, const void* d_keys, uint64_t n, uint8_t* d_flags, uint64_t* dups) {
    if (!c || (n && (!d_keys || !d_flags))) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    if (dups) *dups = 0;
    if (n == 0) return MKT_OK;
    int rc = ensure_dedup_work(c, n);
    if (rc) return rc;
    HIPCHK(c, launch_dedup((const KeyRec*)d_keys, n, d_flags, c->d_dd_work, dedup_work_bytes(n), c->d_dd_res, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_dd_res, c->d_dd_res, sizeof(DedupResult), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (dups) *dups = c->h_dd_res->dups;
    return MKT_OK;
}
int mkt_ext_dedup_keys(mkt_ctx* c, const void* keys, uint64_t n, uint8_t* flags, uint64_t* dups) {
    if (!c || (n && (!keys || !flags))) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    if (dups) *dups = 0;
    if (n == 0) return MKT_OK;
    int rc = ensure_dedup_work(c, n);
    if (rc) return rc;
    KeyRec* d_keys = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_keys, (size_t)n * sizeof(KeyRec)));
    hipError_t e = hipMemcpyAsync(d_keys, keys, (size_t)n * sizeof(KeyRec), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_dedup(d_keys, n, c->d_dd_flags, c->d_dd_work, dedup_work_bytes(n), c->d_dd_res, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_dd_res, c->d_dd_res, sizeof(DedupResult), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, c->d_dd_flags, n, hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_keys);                                   // on every path
    if (e != hipSuccess || e2 != hipSuccess) return fail(c, MKT_E_HIP, "duplicate marking of %llu host keys failed: %s", (unsigned long long)n, hipGetErrorString(e != hipSuccess ? e : e2));
    if (dups) *dups = c->h_dd_res->dups;
    return MKT_OK;
}
int mkt_ext_keys_device(mkt_ctx* c, int drop_last, const void** d_keys, uint64_t* n) {
    if (!c || !d_keys || !n) return MKT_E_ARG;
    if (!(c->p.extensions & MKT_EXT_KEYS)) return fail(c, MKT_E_STATE, "context created without MKT_EXT_KEYS");
    int rc = mkt_sync(c);
    if (rc) return rc;
    *n = ext_key_count(c, drop_last);
    *d_keys = c->d_key_list;
    return MKT_OK;
}
int mkt_ext_partition(mkt_ctx* c, int drop_last, const uint16_t* lut, uint32_t world, void* d_send, uint64_t* counts) {
    if (!c || !counts || world == 0 || world > 16) return MKT_E_ARG;
    if (!(c->p.extensions & MKT_EXT_KEYS)) return fail(c, MKT_E_STATE, "context created without MKT_EXT_KEYS");
    int rc = mkt_sync(c);
    if (rc) return rc;
    const uint64_t n = ext_key_count(c, drop_last);
    for (uint32_t d = 0; d < world; ++d) counts[d] = 0;
    c->part_n = n;
    if (n == 0) return MKT_OK;
    if (!d_send) return MKT_E_ARG;
    if (c->perm_cap < n) {
        if (c->d_perm) HIPCHK(c, hipFree(c->d_perm));
        c->d_perm = nullptr; c->perm_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_perm, (n + n / 8 + 1024) * sizeof(uint32_t)));
        c->perm_cap = n + n / 8 + 1024;
    }
    if (!c->d_part_hist) HIPCHK(c, hipMalloc((void**)&c->d_part_hist, partition_work_bytes()));
    if (lut) {
        if (!c->d_lut) HIPCHK(c, hipMalloc((void**)&c->d_lut, kChrSlots * sizeof(uint16_t)));
        HIPCHK(c, hipMemcpyAsync(c->d_lut, lut, kChrSlots * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    }
    uint32_t G = 1;
    HIPCHK(c, launch_partition(c->d_key_list, n, lut ? c->d_lut : nullptr, world, c->d_part_hist, (KeyRec*)d_send, c->d_perm, &G, c->stream));
    std::vector<uint32_t> hh((size_t)16 * G);
    HIPCHK(c, hipMemcpyAsync(hh.data(), c->d_part_hist, hh.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t d = 0; d < world; ++d) {
        const uint64_t lo = hh[(size_t)d * G], hi = d + 1 < 16 ? hh[(size_t)(d + 1) * G] : n;      // starts of the destinations in `send` (exclusive scan)
        counts[d] = (d + 1 < world ? hi : n) - lo;
    }
    return MKT_OK;
}
int mkt_ext_unpartition(mkt_ctx* c, const uint8_t* d_flags_part, uint8_t* flags, size_t flags_cap, uint64_t* dups) {
    if (!c) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    const uint64_t n = c->part_n;
    if (dups) *dups = 0;
    if (n == 0) return MKT_OK;
    if (!d_flags_part || !c->d_perm) return MKT_E_ARG;
    if (flags && flags_cap < n) return fail(c, MKT_E_ARG, "flags buffer too small (%llu needed)", (unsigned long long)n);
    int rc = ensure_dedup_work(c, n);
    if (rc) return rc;
    HIPCHK(c, launch_unpermute(d_flags_part, c->d_perm, n, c->d_dd_flags, (unsigned long long*)&c->d_dd_res->dups, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_dd_res, c->d_dd_res, sizeof(DedupResult), hipMemcpyDeviceToHost, c->stream));
    if (flags) HIPCHK(c, hipMemcpyAsync(flags, c->d_dd_flags, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (dups) *dups = c->h_dd_res->dups;
    return MKT_OK;
}
// Duplicate marking across the contexts of ONE process (one context per GPU, contiguous shards of the input in rank order): the
// in-process form of microcket_amd/shard.py's exchange.  Every key record travels to the context mix64(key) % world -- device to
// device, hipMemcpyPeerAsync: between two GPUs of one node that is one xGMI hop, nothing passes through the host --, is marked
// there together with the equal keys of all other shards (segments are laid down in source-rank order and the partition is
// stable, so "first in input order wins" holds globally), and one byte per record travels back the same way.
int mkt_ext_dedup_multi(mkt_ctx** cs, uint32_t world, uint32_t last_rank, uint64_t* totals, uint64_t* dups, uint8_t** flags, const size_t* flags_cap) {
    if (!cs || world == 0 || world > 16 || !totals || !dups) return MKT_E_ARG;
    for (uint32_t r = 0; r < world; ++r) {
        if (!cs[r]) return MKT_E_ARG;
        if (!(cs[r]->p.extensions & MKT_EXT_KEYS)) return fail(cs[r], MKT_E_STATE, "context created without MKT_EXT_KEYS");
        if ((cs[r]->p.extensions ^ cs[0]->p.extensions) & MKT_EXT_LANES) return fail(cs[r], MKT_E_ARG, "contexts disagree on MKT_EXT_LANES");
    }
    mkt_ctx* c0 = cs[0];
    // chromosome slots are per context: every slot -> the rank of its name in the sorted union of all tables
    std::vector<std::vector<std::pair<uint32_t, std::string>>> tabs(world);
    std::vector<std::string> uni;
    for (uint32_t r = 0; r < world; ++r) {
        size_t len = 0;
        int rc = mkt_ext_chr_names(cs[r], nullptr, 0, &len);
        if (rc) return rc;
        std::string txt(len, '\0');
        if (len && (rc = mkt_ext_chr_names(cs[r], &txt[0], len, &len))) return rc;
        size_t p0 = 0;
        while (p0 < txt.size()) {
            const size_t nl = txt.find('\n', p0), tb = txt.find('\t', p0);
            if (nl == std::string::npos || tb == std::string::npos || tb > nl) break;
            tabs[r].push_back({(uint32_t)atoi(txt.substr(p0, tb - p0).c_str()), txt.substr(tb + 1, nl - tb - 1)});
            uni.push_back(tabs[r].back().second);
            p0 = nl + 1;
        }
    }
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    if (uni.size() > kChrSlots) return fail(c0, MKT_E_CAPACITY, "more than %u chromosome names over all shards", kChrSlots);
    struct Side { uint8_t* d_send = nullptr; uint8_t* d_recv = nullptr; uint8_t* d_flags = nullptr; uint8_t* d_back = nullptr; uint64_t n = 0, nrecv = 0; uint64_t cnt[16]; };
    std::vector<Side> sd(world);
    auto cleanup = [&]() {
        for (uint32_t r = 0; r < world; ++r) {
            (void)hipSetDevice(cs[r]->p.device);
            if (sd[r].d_send) (void)hipFree(sd[r].d_send);
            if (sd[r].d_recv) (void)hipFree(sd[r].d_recv);
            if (sd[r].d_flags) (void)hipFree(sd[r].d_flags);
            if (sd[r].d_back) (void)hipFree(sd[r].d_back);
        }
    };
#define MCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail((c), MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    // partition every shard's keys by destination (stable), slots rewritten to the shared ids
    for (uint32_t r = 0; r < world; ++r) {
        mkt_ctx* c = cs[r];
        std::vector<uint16_t> lut(kChrSlots, 0);
        for (const auto& e : tabs[r]) lut[e.first & (kChrSlots - 1)] = (uint16_t)(std::lower_bound(uni.begin(), uni.end(), e.second) - uni.begin());
        int rc = mkt_sync(c);
        if (rc) { cleanup(); return rc; }
        sd[r].n = ext_key_count(c, r == last_rank);
        totals[r] = sd[r].n;
        for (uint32_t d = 0; d < 16; ++d) sd[r].cnt[d] = 0;
        if (sd[r].n) MCHK(c, hipMalloc((void**)&sd[r].d_send, (size_t)sd[r].n * sizeof(KeyRec)));
        rc = mkt_ext_partition(c, r == last_rank, lut.data(), world, sd[r].d_send, sd[r].cnt);
        if (rc) { cleanup(); return rc; }
    }
    for (uint32_t a = 0; a < world; ++a)                         // direct device-to-device copies where the hardware offers them (best effort)
        for (uint32_t b = 0; b < world; ++b)
            if (cs[a]->p.device != cs[b]->p.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, cs[a]->p.device, cs[b]->p.device) == hipSuccess && can) {
                    (void)hipSetDevice(cs[a]->p.device);
                    const hipError_t e = hipDeviceEnablePeerAccess(cs[b]->p.device, 0);
                    if (e != hipSuccess) (void)hipGetLastError();      // (already enabled: fine)
                }
            }
    // the exchange: rank r receives, in source-rank order, what every rank s partitioned for it
    for (uint32_t r = 0; r < world; ++r) {
        mkt_ctx* c = cs[r];
        MCHK(c, hipSetDevice(c->p.device));
        sd[r].nrecv = 0;
        for (uint32_t s2 = 0; s2 < world; ++s2) sd[r].nrecv += sd[s2].cnt[r];
        if (!sd[r].nrecv) continue;
        MCHK(c, hipMalloc((void**)&sd[r].d_recv, (size_t)sd[r].nrecv * sizeof(KeyRec)));
        MCHK(c, hipMalloc((void**)&sd[r].d_flags, (size_t)sd[r].nrecv));
        uint64_t at = 0;
        for (uint32_t s2 = 0; s2 < world; ++s2) {
            uint64_t soff = 0;
            for (uint32_t d = 0; d < r; ++d) soff += sd[s2].cnt[d];
            const uint64_t k = sd[s2].cnt[r];
            if (k) MCHK(c, hipMemcpyPeerAsync(sd[r].d_recv + at * sizeof(KeyRec), c->p.device, sd[s2].d_send + soff * sizeof(KeyRec), cs[s2]->p.device, (size_t)k * sizeof(KeyRec), c->stream));
            at += k;
        }
    }
    uint64_t all_dups = 0;
    for (uint32_t r = 0; r < world; ++r) {
        mkt_ctx* c = cs[r];
        MCHK(c, hipSetDevice(c->p.device));
        MCHK(c, hipStreamSynchronize(c->stream));
        uint64_t d = 0;
        const int rc = mkt_ext_dedup_device(c, sd[r].d_recv, sd[r].nrecv, sd[r].d_flags, &d);
        if (rc) { cleanup(); return rc; }
        all_dups += d;
    }
    // one byte per record back to where the record came from
    for (uint32_t s2 = 0; s2 < world; ++s2) {
        mkt_ctx* c = cs[s2];
        MCHK(c, hipSetDevice(c->p.device));
        if (!sd[s2].n) { dups[s2] = 0; continue; }
        MCHK(c, hipMalloc((void**)&sd[s2].d_back, (size_t)sd[s2].n));
        uint64_t soff = 0;
        for (uint32_t r = 0; r < world; ++r) {
            uint64_t roff = 0;
            for (uint32_t q = 0; q < s2; ++q) roff += sd[q].cnt[r];
            const uint64_t k = sd[s2].cnt[r];
            if (k) MCHK(c, hipMemcpyPeerAsync(sd[s2].d_back + soff, c->p.device, sd[r].d_flags + roff, cs[r]->p.device, (size_t)k, c->stream));
            soff += k;
        }
        MCHK(c, hipStreamSynchronize(c->stream));
        const int rc = mkt_ext_unpartition(c, sd[s2].d_back, flags ? flags[s2] : nullptr, flags_cap ? flags_cap[s2] : 0, &dups[s2]);
        if (rc) { cleanup(); return rc; }
    }
#undef MCHK
    cleanup();
    uint64_t sum = 0;
    for (uint32_t r = 0; r < world; ++r) sum += dups[r];
    if (sum != all_dups) return fail(c0, MKT_E_KERNEL, "duplicate counts disagree after the exchange (%llu marked, %llu returned)", (unsigned long long)all_dups, (unsigned long long)sum);
    return MKT_OK;
}
int mkt_ext_chrstat(mkt_ctx* c, int drop_last, char* out, size_t cap, size_t* len) {
    if (!c || !len) return MKT_E_ARG;
    if (!(c->p.extensions & MKT_EXT_KEYS)) return fail(c, MKT_E_STATE, "context created without MKT_EXT_KEYS");
    int rc = mkt_sync(c);
    if (rc) return rc;
    *len = 0;
    const uint64_t n = ext_key_count(c, drop_last);
    if (n == 0 || !c->d_chr) return MKT_OK;
    // the name table -> dense ids in bytewise name order (through pinned staging: pageable copies cost milliseconds each)
    const size_t name_bytes = (size_t)kChrSlots * 64, hash_bytes = kChrSlots * sizeof(unsigned long long);
    if (!c->h_chr_stage) HIPCHK(c, hipHostMalloc((void**)&c->h_chr_stage, hash_bytes + name_bytes + kChrSlots * sizeof(uint16_t), hipHostMallocDefault));
    unsigned long long* hh = (unsigned long long*)c->h_chr_stage;
    uint8_t* names = c->h_chr_stage + hash_bytes;
    uint16_t* dense = (uint16_t*)(c->h_chr_stage + hash_bytes + name_bytes);
    HIPCHK(c, hipMemcpyAsync(hh, c->d_chr->hash, hash_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(names, c->d_chr->name, name_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<std::pair<std::string, uint32_t>> used;
    for (uint32_t s2 = 0; s2 < kChrSlots; ++s2) if (hh[s2]) used.emplace_back(std::string((const char*)&names[(size_t)s2 * 64], names[(size_t)s2 * 64 + 63]), s2);
    std::sort(used.begin(), used.end());
    const uint32_t nd = (uint32_t)used.size();
    memset(dense, 0, kChrSlots * sizeof(uint16_t));
    for (uint32_t d = 0; d < nd; ++d) dense[used[d].second] = (uint16_t)d;
    const size_t cnt_bytes = (size_t)nd * nd * sizeof(unsigned long long);
    if (!c->d_dense) HIPCHK(c, hipMalloc((void**)&c->d_dense, kChrSlots * sizeof(uint16_t)));
    if (c->chr_counts_cap < cnt_bytes) {
        if (c->d_chr_counts) { HIPCHK(c, hipFree(c->d_chr_counts)); HIPCHK(c, hipHostFree(c->h_chr_counts)); }
        c->d_chr_counts = nullptr; c->h_chr_counts = nullptr; c->chr_counts_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_chr_counts, cnt_bytes));
        HIPCHK(c, hipHostMalloc((void**)&c->h_chr_counts, cnt_bytes, hipHostMallocDefault));
        c->chr_counts_cap = cnt_bytes;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_dense, dense, kChrSlots * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_chr_counts, 0, cnt_bytes, c->stream));
    HIPCHK(c, launch_chrstat(c->d_key_list, n, c->d_dense, nd, c->d_chr_counts, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_chr_counts, c->d_chr_counts, cnt_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const unsigned long long* counts = c->h_chr_counts;
    std::string txt;
    char num[32];
    for (uint32_t a2 = 0; a2 < nd; ++a2)
        for (uint32_t b2 = 0; b2 < nd; ++b2)
            if (counts[(size_t)a2 * nd + b2]) {
                snprintf(num, sizeof num, "%llu", counts[(size_t)a2 * nd + b2]);
                txt += used[a2].first; txt += '\t'; txt += used[b2].first; txt += '\t'; txt += num; txt += '\n';
            }
    *len = txt.size();
    if (out) { if (cap < txt.size()) return fail(c, MKT_E_ARG, "chrstat buffer too small (%zu needed)", txt.size()); memcpy(out, txt.data(), txt.size()); }
    return MKT_OK;
}

int mkt_reset(mkt_ctx* c) {
    if (!c) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    {   // the streaming pipeline idle (an earlier asynchronous error is forgotten with the input it belonged to)
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv.wait(lk, [&] { return c->async_rc != MKT_OK || c->jobs.empty(); });
        (void)sync_all(c);
        c->jobs.clear(); c->ready.clear(); c->handed = mkt_ctx::Chunk(); c->handed_valid = false;
        for (int i = 0; i < mkt_ctx::kIn; ++i) c->in[i].busy = false;
        for (int i = 0; i < mkt_ctx::kOut; ++i) { c->outs[i].dev_busy = false; c->outs[i].host_busy = false; }
        c->async_rc = MKT_OK; c->cur = 0; c->seq = 0; c->consumer_async = false;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fold_timing(c);
    HIPCHK(c, hipMemsetAsync(c->d_run, 0, sizeof(DevRun), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->acc = RunAccum();
    c->sc_unfolded = 0; c->bytes_unsynced = 0; c->emitted_unfolded = 0;
    // a new input is probed afresh for its densities; the tile geometry learned on the previous input is where its probe
    // starts (a context usually sees one kind of data; a fresh context starts from the largest tiles)
    c->sc_density = 0; c->key_density = 0; c->probing = true;
    c->dims_probed = c->p.tiles != MKT_TILES_AUTO || c->p.ordered || c->cfg == CFG_SMALL;      // a new input: its own line length
    if (c->d_chr) HIPCHK(c, hipMemsetAsync(c->d_chr, 0, sizeof(ChrTab), c->stream));
    c->res_used = c->res_folded = 0;
    c->h_len = 0;
    c->tail_pairs.clear(); c->tail_sam.clear(); c->drained_pairs.clear(); c->drained_sam.clear();
    c->input_done = c->finished = false;
    c->bytes_in = 0; c->blocks = 0;
    return MKT_OK;
}

int mkt_format_log(const mkt_stats* st, char* out, size_t cap) {
    if (!st || !out) return MKT_E_ARG;
    return snprintf(out, cap, "lowMap\t%u\nmanyHits\t%u\nunpaired\t%u\nselfCircle\t%u\ntrans\t%u\ncis10K\t%u\ncis1K\t%u\ncis0\t%u\n",
                    st->lowMap, st->manyHits, st->unpaired, st->selfCircle, st->trans, st->cis10K, st->cis1K, st->cis0);
}

int mkt_get_timing(const mkt_ctx* c, mkt_timing* t) {
    if (!c || !t) return MKT_E_ARG;
    t->tile_kernel_ms = c->folded_ms; t->tile_launches = c->folded_launches; t->tile_bytes = c->folded_bytes; t->other_ms = 0;
    t->tiles = c->tiles_total; t->deferred_tiles = c->tiles_deferred;
    return MKT_OK;
}
int mkt_reset_timing(mkt_ctx* c) {
    if (!c) return MKT_E_ARG;
    c->folded_ms = 0; c->folded_launches = 0; c->folded_bytes = 0; c->tiles_total = 0; c->tiles_deferred = 0;
    return MKT_OK;
}

int mkt_synth_device(mkt_ctx* c, uint64_t seed, int profile, int genome, int read_len, int lanes, uint64_t first_group,
                     uint64_t n_groups, int tail_group, const void** d_text, size_t* n_bytes) {
    if (!c || !d_text || !n_bytes) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    SynParams sp;
    sp.seed = seed; sp.profile = profile; sp.genome = genome; sp.read_len = read_len; sp.lanes = lanes;
    if (c->syn_sizes_cap < n_groups + 2) {
        if (c->d_syn_sizes) HIPCHK(c, hipFree(c->d_syn_sizes));
        c->d_syn_sizes = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_syn_sizes, (n_groups + 2) * sizeof(uint64_t)));
        c->syn_sizes_cap = n_groups + 2;
    }
    uint64_t* d_total = c->d_syn_sizes + n_groups;
    HIPCHK(c, launch_synth_sizes(sp, first_group, n_groups, c->d_syn_sizes, c->stream));
    HIPCHK(c, launch_exscan(c->d_syn_sizes, n_groups, d_total, c->stream));
    uint64_t total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t tail = tail_group ? synth_tail_bytes(sp) : 0;
    size_t need = (size_t)total + tail + 64;
    if (c->syn_cap < need) {
        if (c->d_syn) HIPCHK(c, hipFree(c->d_syn));
        c->d_syn = nullptr; c->syn_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_syn, need));
        c->syn_cap = need;
    }
    HIPCHK(c, launch_synth_write(sp, first_group, n_groups, c->d_syn_sizes, c->d_syn, c->stream));
    if (tail) HIPCHK(c, launch_synth_tail(sp, c->d_syn + total, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *d_text = c->d_syn;
    *n_bytes = (size_t)total + tail;
    return MKT_OK;
}

struct mkt_dataset {
    mkt_ctx* ctx;
    char* arena = nullptr;
    std::vector<uint64_t> off, len, groups;
    uint64_t total_bytes = 0, total_groups = 0;
};

int mkt_dataset_create(mkt_ctx* c, uint64_t seed, int profile, int genome, int read_len, int lanes, uint64_t first_group,
                       uint64_t n_groups, uint64_t gpb, int tail_group, mkt_dataset** out) {
    if (!c || !out || gpb == 0) return MKT_E_ARG;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->p.device));
    SynParams sp;
    sp.seed = seed; sp.profile = profile; sp.genome = genome; sp.read_len = read_len; sp.lanes = lanes;
    if (c->syn_sizes_cap < gpb + 2) {
        if (c->d_syn_sizes) HIPCHK(c, hipFree(c->d_syn_sizes));
        c->d_syn_sizes = nullptr;
        HIPCHK(c, hipMalloc((void**)&c->d_syn_sizes, (gpb + 2) * sizeof(uint64_t)));
        c->syn_sizes_cap = gpb + 2;
    }
    mkt_dataset* ds = new mkt_dataset();
    ds->ctx = c;
    const uint64_t nb = (n_groups + gpb - 1) / gpb;
    const size_t tail = tail_group ? synth_tail_bytes(sp) : 0;
    uint64_t* d_total = c->d_syn_sizes + gpb;
    uint64_t cursor = 0;
    for (uint64_t b = 0; b < nb; ++b) {          // pass 1: block sizes
        const uint64_t g0 = b * gpb, g = (g0 + gpb <= n_groups) ? gpb : n_groups - g0;
        HIPCHK(c, launch_synth_sizes(sp, first_group + g0, g, c->d_syn_sizes, c->stream));
        HIPCHK(c, launch_exscan(c->d_syn_sizes, g, d_total, c->stream));
        uint64_t total = 0;
        HIPCHK(c, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (b + 1 == nb) total += tail;
        if (total >= kMaxBlock) { delete ds; return fail(c, MKT_E_ARG, "block %llu would hold %llu bytes (>= 2 GiB - 64 KiB): lower groups_per_block", (unsigned long long)b, (unsigned long long)total); }
        ds->off.push_back(cursor); ds->len.push_back(total); ds->groups.push_back(g + ((b + 1 == nb && tail_group) ? 1 : 0));
        cursor += (total + 15) & ~(uint64_t)15;
    }
    ds->total_bytes = 0;
    for (uint64_t l : ds->len) ds->total_bytes += l;
    ds->total_groups = n_groups + (tail_group ? 1 : 0);
    hipError_t e = hipMalloc((void**)&ds->arena, cursor + 64);
    if (e != hipSuccess) { delete ds; return fail(c, MKT_E_NOMEM, "hipMalloc of %llu bytes for the data set failed: %s", (unsigned long long)cursor, hipGetErrorString(e)); }
    for (uint64_t b = 0; b < nb; ++b) {          // pass 2: bytes
        const uint64_t g0 = b * gpb, g = (g0 + gpb <= n_groups) ? gpb : n_groups - g0;
        HIPCHK(c, launch_synth_sizes(sp, first_group + g0, g, c->d_syn_sizes, c->stream));
        HIPCHK(c, launch_exscan(c->d_syn_sizes, g, d_total, c->stream));
        HIPCHK(c, launch_synth_write(sp, first_group + g0, g, c->d_syn_sizes, ds->arena + ds->off[b], c->stream));
        if (b + 1 == nb && tail) HIPCHK(c, launch_synth_tail(sp, ds->arena + ds->off[b] + ds->len[b] - tail, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *out = ds;
    return MKT_OK;
}
int mkt_dataset_info(const mkt_dataset* ds, uint64_t* n_blocks, uint64_t* total_bytes, uint64_t* total_groups) {
    if (!ds) return MKT_E_ARG;
    if (n_blocks) *n_blocks = ds->off.size();
    if (total_bytes) *total_bytes = ds->total_bytes;
    if (total_groups) *total_groups = ds->total_groups;
    return MKT_OK;
}
int mkt_dataset_block(const mkt_dataset* ds, uint64_t i, const void** d_text, size_t* n_bytes, uint64_t* n_groups) {
    if (!ds || i >= ds->off.size()) return MKT_E_ARG;
    if (d_text) *d_text = ds->arena + ds->off[i];
    if (n_bytes) *n_bytes = (size_t)ds->len[i];
    if (n_groups) *n_groups = ds->groups[i];
    return MKT_OK;
}
void mkt_dataset_destroy(mkt_dataset* ds) {
    if (!ds) return;
    if (ds->arena) { (void)hipSetDevice(ds->ctx->p.device); (void)hipFree(ds->arena); }
    delete ds;
}

int mkt_group_count(mkt_ctx* c, uint64_t* groups) {
    if (!c || !groups) return MKT_E_ARG;
    int rc = mkt_sync(c);
    if (rc) return rc;
    *groups = c->acc.groups;
    return MKT_OK;
}

#if defined(MKT_STAMPS)
// diagnostic build only: per-phase shader-clock sums of k_tiles (see STAMP in mkt_kernels.hip)
int mkt_debug_stamps(mkt_ctx* c, unsigned long long* out16) {
    if (!c || !out16) return MKT_E_ARG;
    memset(out16, 0, 16 * sizeof(unsigned long long));
    if (!c->d_stamps) return MKT_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out16, c->d_stamps, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemset(c->d_stamps, 0, 16 * sizeof(unsigned long long)));
    return MKT_OK;
}
#endif

int mkt_device_text(mkt_ctx* c, const char* bytes, size_t n, const void** d_text) {
    if (!c || !d_text || (n && !bytes)) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, n + 64));
    c->uploads.push_back(d);
    if (n) HIPCHK(c, hipMemcpy(d, bytes, n, hipMemcpyHostToDevice));
    *d_text = d;
    return MKT_OK;
}

int mkt_copy_to_host(mkt_ctx* c, const void* d_src, void* dst, size_t n) {
    if (!c || !d_src || !dst) return MKT_E_ARG;
    HIPCHK(c, hipSetDevice(c->p.device));
    HIPCHK(c, hipMemcpy(dst, d_src, n, hipMemcpyDeviceToHost));
    return MKT_OK;
}

}  // extern "C"
