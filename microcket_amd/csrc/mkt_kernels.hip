// mkt_kernels.hip -- gfx950 kernels of the sam2pairs hot path.
//
// k_tiles: ONE fused pass over a block of SAM text.  A workgroup draws tiles from an atomic
// ticket, stages tile + halos in LDS with 16-byte loads, finds line starts with a SWAR newline
// mask + workgroup scan, then runs the phases of mkt_tile.h one work item per lane (parse six
// fields + CIGAR, group-start test, group walk + classification, in-tile exclusive sums), obtains
// its output offsets from a decoupled look-back over per-tile descriptors (single 8-byte words
// carrying their own flag, relaxed agent-scope atomics) and writes .pairs bytes (LDS-staged),
// the pass-through .sam bytes, self-circle group indices and the 8 counters.
// Text is read from HBM once (plus halos); outputs are written once; everything else is on chip.
//
// No MFMA: this is byte/integer work bounded by HBM bandwidth (DESIGN.md has the byte budget).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mkt_launch.h"

namespace mkt {

constexpr int NT = 256;                      // 4 waves per workgroup
constexpr uint64_t FLAG_AGG = 1ull << 62, FLAG_PREFIX = 2ull << 62, PAYLOAD = (1ull << 62) - 1;
constexpr uint32_t LOOKBACK_SPIN_LIMIT = 1u << 22;

struct ScanScratch { uint64_t a[NT / 64], b[NT / 64]; };

// Diagnostic build only (-DMKT_STAMPS): per-phase shader-clock shares, accumulated by lane 0 of
// every tile into a.stamps[phase].  Never compiled into the shipped library.
#if defined(MKT_STAMPS)
// (accumulated in registers of lane 0 and added to a.stamps once per workgroup: an atomic per phase and tile would itself be
//  the longest thing on lane 0's path)
#define STAMP_DECL() unsigned long long stamp_prev_ = 0, stamp_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k)                                                                        \
    do {                                                                                \
        if (tid == 0 && a.stamps) {                                                     \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();               \
            if ((k) > 0) stamp_acc_[(k)] += now_ - stamp_prev_;                         \
            stamp_prev_ = now_;                                                         \
        }                                                                               \
    } while (0)
#define STAMP_FLUSH(t0_)                                                                \
    do {                                                                                \
        if ((t0_) == 0 && a.stamps) for (int k_ = 1; k_ < 16; ++k_) if (stamp_acc_[k_]) atomicAdd(&a.stamps[k_], stamp_acc_[k_]); \
    } while (0)
#define STOP_AFTER(k) if (a.debug_stop == (k)) { __syncthreads(); continue; }
#else
#define STAMP_DECL() do { } while (0)
#define STAMP(k) do { } while (0)
#define STAMP_FLUSH(t0_) do { } while (0)
#define STOP_AFTER(k)
#endif

__device__ inline uint64_t shfl_up64(uint64_t v, int d) { return (uint64_t)__shfl_up((long long)v, d, 64); }
__device__ inline uint64_t shfl_xor64(uint64_t v, int d) { return (uint64_t)__shfl_xor((long long)v, d, 64); }

// exclusive scan of one u32 per thread over the workgroup; the total returned to every thread
__device__ inline void block_exscan1(uint32_t& a, uint32_t& total, ScanScratch& sc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t ia = a;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)ia, d, 64);
        if (lane >= d) ia += x;
    }
    if (lane == 63) sc.a[wv] = ia;
    __syncthreads();
    uint32_t pre = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t x = (uint32_t)sc.a[w];
        if (w < wv) pre += x;
        total += x;
    }
    a = pre + ia - a;
    __syncthreads();
}

// inclusive scan of one u32 over the wave by data-parallel-primitive adds (row shifts 1, 2, 4, 8 inside each row of 16 lanes, then
// lane 15 / 31 of a row broadcast into the rows behind): six VALU instructions, no LDS crossbar
__device__ inline uint32_t wave_iscan32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return v;
}
// exclusive scan of three u32 per thread over the workgroup (the tile sums of k_fast: every total stays far below 2^32)
// (no barrier at the end: the caller reaches one of its own before the scratch is used again)
template <int NW>
__device__ inline void block_exscan3(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& ta, uint32_t& tb, uint32_t& tc, ScanScratch& sc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ia = wave_iscan32(a), ib = wave_iscan32(b), ic = wave_iscan32(c);
    if (lane == 63) { sc.a[wv] = (uint64_t)ia | ((uint64_t)ib << 32); sc.b[wv] = ic; }
    __syncthreads();
    uint32_t pa = 0, pb = 0, pc = 0;
    ta = 0; tb = 0; tc = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint64_t x = sc.a[w];
        const uint32_t xa = (uint32_t)x, xb = (uint32_t)(x >> 32), xc = (uint32_t)sc.b[w];
        if (w < wv) { pa += xa; pb += xb; pc += xc; }
        ta += xa; tb += xb; tc += xc;
    }
    a = pa + ia - a; b = pb + ib - b; c = pc + ic - c;
}

// exclusive scan of two u64 lanes-values over the workgroup; totals returned to every thread
__device__ inline void block_exscan2(uint64_t& a, uint64_t& b, uint64_t& ta, uint64_t& tb, ScanScratch& sc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t xa = shfl_up64(ia, d), xb = shfl_up64(ib, d);
        if (lane >= d) { ia += xa; ib += xb; }
    }
    if (lane == 63) { sc.a[wv] = ia; sc.b[wv] = ib; }
    __syncthreads();
    uint64_t pa = 0, pb = 0;
    ta = 0; tb = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wv) { pa += sc.a[w]; pb += sc.b[w]; }
        ta += sc.a[w]; tb += sc.b[w];
    }
    a = pa + ia - a;
    b = pb + ib - b;
    __syncthreads();
}

// 0x80 in every byte of x that is zero (exact, no cross-byte borrow)
__device__ inline uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }
// 0x80 in every byte of x that is >= n (1 <= n <= 128), bytes >= 0x80 included
__device__ inline uint32_t ge_bytes(uint32_t x, uint32_t n) { return (((x & 0x7F7F7F7Fu) + (0x80u - n) * 0x01010101u) | x) & 0x80808080u; }
// 0x80 flags of the four bytes -> bits 0..3 (one multiply gathers them: no two partial products share a bit)
__device__ inline uint32_t pack_msb(uint32_t z) { return (((z >> 7) * 0x00204081u) >> 21) & 0xFu; }
// per-byte masks of a dword: newline, whitespace (space, \t \n \v \f \r), one bit per byte
__device__ inline uint32_t nl_bits(uint32_t x) { return pack_msb(zero_bytes(x ^ 0x0A0A0A0Au)); }
__device__ inline uint32_t ws_bits(uint32_t x) {
    const uint32_t ctl = ge_bytes(x, 9u) & ~ge_bytes(x, 14u);        // 9..13
    return pack_msb(ctl | zero_bytes(x ^ 0x20202020u));
}

// lean kernel: the same two classifications with shared sub-expressions; 0x80 flags per byte
__device__ inline uint32_t nl_flags(uint32_t x) {                 // byte == '\n'
    const uint32_t y = x & 0x7F7F7F7Fu;
    return ~((y ^ 0x0A0A0A0Au) + 0x7F7F7F7Fu) & ~x & 0x80808080u;
}
// nonzero iff some byte of x is '\n' (exact as a yes/no; which byte is not)
__device__ inline uint32_t has_nl(uint32_t x) {
    const uint32_t t = x ^ 0x0A0A0A0Au;
    return (t - 0x01010101u) & ~t & 0x80808080u;
}
__device__ inline uint32_t sep_flags(uint32_t x) {                // byte <= 0x20: tab, newline, the other whitespace and control bytes
    const uint32_t y = x & 0x7F7F7F7Fu;
    return ~((y + 0x5F5F5F5Fu) | x) & 0x80808080u;               // bit 7 of (y + 0x5F) <=> y >= 0x21
}
// sixteen 0x80 byte flags (four dwords) -> one bit per byte.  v_dot4_u32_u8 does the gathering: the flags are 128 x {0, 1}, the
// weights 1, 2, 4, 8 (16 .. 128 for the second dword), so two dot products per byte of result and one shift (the multiply-
// gather of pack_msb costs a quarter-rate v_mul_lo_u32 per dword).
__device__ inline uint32_t pack16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    const uint32_t lo = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(f3, 0x80402010u, __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false), false);
    return (lo >> 7) | ((hi >> 7) << 8);
}

// Decoupled look-back on one descriptor word per tile.  Executed by one full wave.
__device__ inline uint64_t lookback(uint64_t* desc, uint32_t t, uint64_t agg, uint32_t* err_word) {
    const int lane = threadIdx.x & 63;
    if (t == 0) {
        if (lane == 0) __hip_atomic_store(&desc[0], FLAG_PREFIX | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&desc[t], FLAG_AGG | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t excl = 0;
    int64_t j = (int64_t)t - 1;
    uint32_t spins = 0;
    for (;;) {
        const int64_t idx = j - lane;
        uint64_t w = FLAG_PREFIX;                           // tiles before the first have prefix 0
        if (idx >= 0) w = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t flag = (uint32_t)(w >> 62);
        const uint64_t m_prefix = __ballot(flag == 2);
        const uint64_t m_invalid = __ballot(flag == 0);
        const int first = m_prefix ? __builtin_ctzll(m_prefix) : 64;
        const uint64_t needed = first >= 63 ? ~0ull : ((1ull << (first + 1)) - 1);
        if (m_invalid & needed) {
            if (++spins > LOOKBACK_SPIN_LIMIT) { if (lane == 0) atomicOr(err_word, (uint32_t)E_LOOKBACK); break; }
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        uint64_t c = lane <= first ? (w & PAYLOAD) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += shfl_xor64(c, d);
        excl += c;
        if (first < 64) break;
        j -= 64;
    }
    if (lane == 0) __hip_atomic_store(&desc[t], FLAG_PREFIX | ((excl + agg) & PAYLOAD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// Any-order mode: the tile claims its output ranges from the cursors of its region with one 64-bit
// atomic (.pairs bytes + emitted lines) and, only when needed, a second one (.sam bytes + self-circles).
// Executed by ONE thread.  Returns error bits.
__device__ inline uint32_t claim_ranges(const KArgs& a, uint32_t region, const TileSums& sums, TileSums& base,
                                        uint32_t& region_pair0, uint32_t& region_sam0, OutPtrs& lim) {
    RegionCur* cur = a.cur + region;
    uint32_t err = 0;
    const uint64_t p0 = (uint64_t)region * a.pairs_rcap, s0 = (uint64_t)region * a.sam_rcap, c0 = (uint64_t)region * a.sc_rcap;
    unsigned long long oa = 0, ob = 0;
    if (sums.pair_bytes | sums.emitted) oa = atomicAdd(&cur->a, (unsigned long long)sums.pair_bytes | ((unsigned long long)sums.emitted << kCurShift));
    if (sums.sam_bytes | sums.sc) ob = atomicAdd(&cur->b, (unsigned long long)sums.sam_bytes | ((unsigned long long)sums.sc << kCurShift));
    const uint64_t op = oa & kCurLow, os = ob & kCurLow, oc = ob >> kCurShift;
    if (op + sums.pair_bytes > a.pairs_rcap) err |= E_PAIRS_CAP;
    if (os + sums.sam_bytes > a.sam_rcap) err |= E_SAM_CAP;
    if (oc + sums.sc > a.sc_rcap) err |= E_SC_CAP;
    base.pair_bytes = (uint32_t)(p0 + op); base.sam_bytes = s0 + os; base.sc = (uint32_t)(c0 + oc);
    base.emitted = (uint32_t)((uint64_t)region * a.keys_rcap + (oa >> kCurShift));           // extension: slot of the tile's first key record
    lim.keys_cap = a.keys_rcap ? ((uint64_t)region * a.keys_rcap + ((oa >> kCurShift) + sums.emitted <= a.keys_rcap ? a.keys_rcap : 0)) : 0;
    region_pair0 = (uint32_t)p0; region_sam0 = (uint32_t)s0;
    lim.pairs_cap = err & E_PAIRS_CAP ? 0 : p0 + a.pairs_rcap;     // an overflowing tile writes nothing
    lim.sam_cap = err & E_SAM_CAP ? 0 : s0 + a.sam_rcap;
    lim.sc_cap = err & E_SC_CAP ? 0 : c0 + a.sc_rcap;
    return err;
}

template <class Cfg>
__global__ __launch_bounds__(NT) void k_tiles(KArgs a) {
    __shared__ TileState<Cfg> st;
    __shared__ ScanScratch scan;
    __shared__ uint32_t s_tile;
    constexpr int NVEC = (Cfg::W + 15) / 16;
    constexpr int NM16 = Cfg::MW * 4;                  // 16-bit mask slots (incl. zero padding)
    constexpr int VPT = (NVEC + NT - 1) / NT;          // newline-mask words per thread
    constexpr int IPT = (Cfg::LCAP + NT - 1) / NT;     // line-table items per thread in the sums
    const int tid = threadIdx.x;
    const Params P = a.P;
    const uint32_t n = a.n;
    __shared__ OutPtrs s_out;             // this tile's output limits
    const uint32_t region = a.nregions > 1 ? (blockIdx.x & (uint32_t)(a.nregions - 1)) : 0u;
    uint16_t* nlmask = reinterpret_cast<uint16_t*>(st.u.m.nlm);
    uint16_t* wsmask = reinterpret_cast<uint16_t*>(st.u.m.wsm);
    STAMP_DECL();

    for (;;) {
        if (tid == 0) {
            s_out = a.out;
            uint32_t k = atomicAdd(a.ticket, 1u);
            if (a.use_list) k = k < *a.defer_count ? a.defer_list[k] : 0xFFFFFFFFu;     // only the tiles the lean kernel deferred
            s_tile = k;
            tile_reset(st);
        }
        __syncthreads();
        const uint32_t t = s_tile;
        if (t >= a.ntiles) break;
        STAMP(0);
        const TileGeom G = tile_geom(t, n, a.dims);             // (window <= Cfg::W: the launcher checks)
        const uint32_t wlen = G.w1 - G.w0;
        const uint32_t nvec = (wlen + 15u) >> 4;
        const TextView tv = tile_view(st, a.text, n, G);

        // ---- stage window in LDS, build the newline / whitespace bitmaps ------------------------
        // (all loads of a lane first: one memory round trip for the window instead of one per iteration)
        constexpr int SPT = (NM16 + NT - 1) / NT;
        uint4 xs[SPT];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t v = (uint32_t)tid + (uint32_t)q * NT;
            const uint32_t go = G.w0 + (v << 4);
            xs[q] = *reinterpret_cast<const uint4*>(a.text + ((v < nvec && go + 16u <= n) ? go : 0u));      // (no conditional load: vector 0 again, unused)
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t v = (uint32_t)tid + (uint32_t)q * NT;
            if (v >= (uint32_t)NM16) continue;
            uint32_t mnl = 0, mws = 0;
            if (v < nvec) {
                const uint32_t go = G.w0 + (v << 4);
                uint4 x;
                if (go + 16u <= n) {
                    x = xs[q];
                } else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (uint32_t b = 0; go + b < n; ++b) w[b >> 2] |= (uint32_t)a.text[go + b] << ((b & 3u) * 8u);
                    x = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *reinterpret_cast<uint4*>(&st.win[v << 4]) = x;
                mnl = nl_bits(x.x) | (nl_bits(x.y) << 4) | (nl_bits(x.z) << 8) | (nl_bits(x.w) << 12);
                mws = ws_bits(x.x) | (ws_bits(x.y) << 4) | (ws_bits(x.z) << 8) | (ws_bits(x.w) << 12);
                const uint32_t r0 = v << 4;
                if (r0 + 16u > wlen) {                    // bytes past the window end read as "nothing"
                    const uint32_t keep = wlen - r0;
                    mnl &= (1u << keep) - 1u; mws &= (1u << keep) - 1u;
                }
            } else if (v == nvec && (v << 4) < (uint32_t)(Cfg::W + 16)) {
                *reinterpret_cast<uint4*>(&st.win[v << 4]) = make_uint4(0, 0, 0, 0);     // zero pad for win_load4
            }
            nlmask[v] = (uint16_t)mnl;
            wsmask[v] = (uint16_t)mws;
        }
        __syncthreads();
        STAMP(1);

        // ---- line table: exclusive scan of newline counts ------------------------------------
        {
            uint64_t cnt = 0, dummy = 0, total, td;
            const uint32_t v0 = tid * VPT;
            // a '\n' at window byte r opens a line at r+1: only r <= wlen-2 counts
            auto line_bits = [&](uint32_t v) -> uint32_t {
                uint32_t m = nlmask[v];
                const uint32_t r0 = v << 4;
                if (r0 + 16u > wlen - 1u) { const uint32_t keep = wlen - 1u > r0 ? wlen - 1u - r0 : 0u; m &= keep >= 16u ? 0xFFFFu : ((1u << keep) - 1u); }
                return m;
            };
            for (uint32_t k = 0; k < (uint32_t)VPT; ++k) if (v0 + k < nvec) cnt += __popc(line_bits(v0 + k));
            uint64_t ex = cnt;
            block_exscan2(ex, dummy, total, td, scan);
            const uint32_t lead = G.w0 == 0 ? 1u : 0u;
            const uint32_t NL = (uint32_t)total + lead;
            if (NL > (uint32_t)Cfg::LCAP) {
                if (tid == 0) { lds_or(&st.err, E_LINE_TABLE); st.NL = 0; st.first_idx = 0; st.end_idx = 0; }
            } else {
                uint32_t idx = (uint32_t)ex + lead;
                for (uint32_t k = 0; k < (uint32_t)VPT; ++k) {
                    if (v0 + k >= nvec) break;
                    uint32_t m = line_bits(v0 + k);
                    while (m) {
                        const uint32_t b = __builtin_ctz(m);
                        st.off[idx++] = G.w0 + ((v0 + k) << 4) + b + 1u;
                        m &= m - 1u;
                    }
                }
                if (tid == 0) { if (lead) st.off[0] = 0; st.NL = NL; st.first_idx = NL; st.end_idx = NL; }
            }
        }
        __syncthreads();
        const uint32_t NL = st.NL;

        STAMP(2);
        for (uint32_t i = tid; i < NL; i += NT) ph_parse(st, tv, P, G, i);
        __syncthreads();
        if (tid == 0) tile_trim(st);
        __syncthreads();
        STAMP(3);
        const uint32_t first_idx = st.first_idx, end_idx = st.end_idx;
        for (uint32_t i = first_idx + tid; i < end_idx; i += NT) ph_group(st, tv, P, G, i);
        __syncthreads();
        STAMP(4);
        STAMP(5);

        // ---- exclusive sums over the tile's groups ---------------------------------------------
        {
            auto& g = st.u.g;
            uint64_t ca = 0, cb = 0;                       // ca: groups | emitted<<16 | sc<<32 ; cb: pair | sam<<32
            const uint32_t i0 = first_idx + tid * IPT;
            for (uint32_t k = 0; k < (uint32_t)IPT; ++k) {
                const uint32_t i = i0 + k;
                if (i >= end_idx) break;
                const uint32_t info = g.g_info[i];
                if (info & GI_START) ca += 1ull;
                if (info & GI_EMIT) ca += 1ull << 16;
                if ((info & GI_START) && (info & GI_COUNTER) == C_SELFCIRCLE) ca += 1ull << 32;
                cb += (uint64_t)g.g_plen[i] | ((uint64_t)g.g_slen[i] << 32);
            }
            uint64_t ea = ca, eb = cb, ta, tb;
            block_exscan2(ea, eb, ta, tb, scan);
            for (uint32_t k = 0; k < (uint32_t)IPT; ++k) {
                const uint32_t i = i0 + k;
                if (i >= end_idx) break;
                g.x_grp[i] = (uint16_t)(ea & 0xFFFFu); g.x_sc[i] = (uint16_t)((ea >> 32) & 0xFFFFu); g.x_emit[i] = (uint16_t)((ea >> 16) & 0xFFFFu);
                g.x_pair[i] = (uint32_t)eb; g.x_sam[i] = (uint32_t)(eb >> 32);
                const uint32_t info = g.g_info[i];
                if (info & GI_EMIT) g.em_idx[(ea >> 16) & 0xFFFFu] = (uint16_t)i;
                if (info & GI_START) ea += 1ull;
                if (info & GI_EMIT) ea += 1ull << 16;
                if ((info & GI_START) && (info & GI_COUNTER) == C_SELFCIRCLE) ea += 1ull << 32;
                eb += (uint64_t)g.g_plen[i] | ((uint64_t)g.g_slen[i] << 32);
            }
            if (tid == 0) {
                st.sums.groups = (uint32_t)(ta & 0xFFFFu); st.sums.emitted = (uint32_t)((ta >> 16) & 0xFFFFu); st.sums.sc = (uint32_t)((ta >> 32) & 0xFFFFu);
                st.sums.pair_bytes = (uint32_t)tb; st.sums.sam_bytes = tb >> 32;
            }
        }
        __syncthreads();
        STAMP(6);
        // ---- where do this tile's outputs go? ----------------------------------------------------
        if (a.ordered) {
            // input order: decoupled look-back over three descriptor words, one wave each (single region)
            const int wv = tid >> 6;
            if (wv == 0) {
                uint64_t ex = lookback(a.descA, t, ((uint64_t)st.sums.groups << 31) | st.sums.emitted, &a.res->err);
                if ((tid & 63) == 0) { st.base.groups = (uint32_t)(ex >> 31); st.base.emitted = (uint32_t)(ex & 0x7FFFFFFFu); }
            } else if (wv == 1) {
                uint64_t ex = lookback(a.descB, t, ((uint64_t)st.sums.pair_bytes << 31) | st.sums.sc, &a.res->err);
                if ((tid & 63) == 0) { st.base.pair_bytes = (uint32_t)(ex >> 31); st.base.sc = (uint32_t)(ex & 0x7FFFFFFFu); }
            } else if (wv == 2) {
                uint64_t ex = lookback(a.descC, t, st.sums.sam_bytes, &a.res->err);
                if ((tid & 63) == 0) st.base.sam_bytes = ex;
            } else if (tid == 192) {      // totals (k_finish reads them from the region cursors in both modes)
                if (st.sums.pair_bytes | st.sums.emitted) atomicAdd(&a.cur[0].a, (unsigned long long)st.sums.pair_bytes | ((unsigned long long)st.sums.emitted << kCurShift));
                if (st.sums.sam_bytes | st.sums.sc) atomicAdd(&a.cur[0].b, (unsigned long long)st.sums.sam_bytes | ((unsigned long long)st.sums.sc << kCurShift));
            }
        } else if (tid == 0) {
            const uint32_t e = claim_ranges(a, region, st.sums, st.base, st.region_pair0, st.region_sam0, s_out);
            st.region_id = region;
            if (e) lds_or(&st.err, e);
        }
        if (tid == 5) a.tile_groups[t] = (uint64_t)st.sums.groups | ((uint64_t)st.sums.emitted << 32);
        __syncthreads();
        STAMP(7);

        // ---- emit ------------------------------------------------------------------------------
        for (uint32_t i = first_idx + tid; i < end_idx; i += NT) {
            ph_account(st, tv, P, s_out, t, i);
            ph_last(st, &a.tile_last[t], i);
        }
        {   // .pairs bytes: one lane per reported pair writes its line (a lane per output BYTE, with a search for the pair and a walk
            // over its fields behind every byte, was 40 % of a tile's time here)
            const uint32_t total = st.sums.pair_bytes;
            const uint64_t go = st.base.pair_bytes;
            if (go + total <= s_out.pairs_cap) { for (uint32_t e = tid; e < st.sums.emitted; e += NT) tile_emit_line(st, tv, e, s_out.pairs + go); }
            else if (tid == 0 && total) lds_or(&st.err, E_PAIRS_CAP);
        }
        if (P.write_sam) {   // contiguous groups: straight byte-range copies
            for (uint32_t i = first_idx; i < end_idx; ++i) {
                const uint32_t info = st.u.g.g_info[i];
                if ((info & (GI_EMIT | GI_CONTIG)) != (GI_EMIT | GI_CONTIG)) continue;
                const uint32_t len = st.u.g.g_slen[i], src = st.off[i];
                const uint64_t go = st.base.sam_bytes + st.u.g.x_sam[i];
                if (go + len <= s_out.sam_cap) { for (uint32_t k = tid; k < len; k += NT) s_out.sam[go + k] = tv.at(src + k); }
                else if (tid == 0) lds_or(&st.err, E_SAM_CAP);
            }
        }
        __syncthreads();
        STAMP(8);
        if (tid < (int)C_COUNT && st.cnt[tid]) atomicAdd(&a.res->counters[tid], st.cnt[tid]);
        if (tid == 0 && st.err) atomicOr(&a.res->err, st.err);
        __syncthreads();
    }
    STAMP_FLUSH(tid);
}

// ---------------------------------------------------------------------------------------------
// k_fast: the lean tile kernel (mkt_fast.h).  Same phases as k_tiles without any generic path; a
// tile that needs one is appended to the defer list and produces nothing here.
// NTW threads per workgroup; RR waves share the group phase of a tile (lines dealt round-robin: 2 beats 1, 3 and 4 at
// 48 KiB tiles); WPS = waves per SIMD the kernel is compiled for (profiles/r02_kernel_variants.txt has the measurements
// behind these choices and behind the variants that are gone from the source).
// Newline scan of (a part of) tile tt's window by nws waves: wave slot ws (0 .. nws-1) takes the 64-vector groups k * nws + ws
// for k in [kb, ke).  The text streams through registers only, up to fourteen 16-byte vectors of a lane in flight at once (a part
// of the largest window is one batch: one memory round trip); per vector a 3-op-per-dword "some byte == '\n'" test and one
// ballot: bit l of hit[g] <-> vector 64 g + l of the window holds a newline.
// FULL: the window is a whole number of 64-vector groups and every group of the part lies inside it (all tiles but those at the
// two ends of a block when hb + tile + hf is a multiple of 1024, as lean_dims() makes it): a running scalar pointer, no
// per-lane bounds.  Otherwise groups and lanes past the window read its last group / vector again -- no branch, no exec mask,
// no zero fill in front of the loads (a conditional load costs a wait right behind it) -- and what they read is ignored.
constexpr int kLoadBatch = 14;
// "some byte of y is a newline": with t = y ^ 0x0A0A0A0A, (t - 0x01010101) & ~t & 0x80808080 != 0.  Eleven VALU instructions per
// 16-byte vector: per dword ONE v_xad_u32 ((y ^ nl) + (-0x01010101); the compiler splits it into xor + add) and ONE three-input
// v_bitop3 (z & ~(y ^ nl): truth table 0xF0 & ~(0xCC ^ 0xAA) = 0x90), then the four dwords ORed and masked by two more, one compare.
__device__ inline uint32_t nl_xad(uint32_t y, uint32_t nl, uint32_t m1) {
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(y), "s"(nl), "v"(m1));
    return r;
}
struct NlConst { uint32_t nl, m1; };                     // 0x0A0A0A0A in a scalar, -0x01010101 in a vector register: one instruction cannot take two literals
__device__ inline NlConst nl_const() {
    NlConst c;
    c.nl = 0x0A0A0A0Au; c.m1 = 0xFEFEFEFFu;
    asm volatile("" : "+s"(c.nl));
    asm volatile("" : "+v"(c.m1));
    return c;
}
__device__ inline uint64_t has_newline_ballot(const uint4& y, bool valid, const NlConst& c) {
    const uint32_t nl = c.nl, m1 = c.m1;
    const uint32_t r0 = __builtin_amdgcn_bitop3_b32(nl_xad(y.x, nl, m1), y.x, nl, 0x90), r1 = __builtin_amdgcn_bitop3_b32(nl_xad(y.y, nl, m1), y.y, nl, 0x90);
    const uint32_t r2 = __builtin_amdgcn_bitop3_b32(nl_xad(y.z, nl, m1), y.z, nl, 0x90), r3 = __builtin_amdgcn_bitop3_b32(nl_xad(y.w, nl, m1), y.w, nl, 0x90);
    const uint32_t z = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(r0, r1, r2, 0xFE), r3, 0x80808080u, 0xA8);      // (a | b | c), then (a | b) & c
    return __ballot(z != 0u && valid);
}
template <class Cfg, uint32_t nws>
__device__ inline void fast_scan(const KArgs& a, uint32_t tt, uint64_t* hit, uint32_t ws_, uint32_t kb, uint32_t ke, int lane) {
    const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane((int)ws_);      // (wave-uniform: the group arithmetic below is scalar)
    const NlConst nlc = nl_const();
    const uint32_t n = a.n;
    const TileGeom G = tile_geom(tt, n, a.dims);
    const uint32_t wlen = G.w1 - G.w0, nvec = (wlen + 15u) >> 4;
    const uint8_t* wbase = a.text + G.w0;
    const uint32_t loff = (uint32_t)lane << 4;
    const uint32_t ngr = nvec >> 6;                                            // whole groups of the window
    if (!(wlen & 1023u) && kb < ke && (ke - 1u) * nws + ws < ngr && ke - kb <= (uint32_t)kLoadBatch) {
        const uint32_t cnt = ke - kb, stride = nws << 10;
        const uint8_t* p0 = wbase + ((size_t)(kb * nws + ws) << 10);
        uint4 x[kLoadBatch];
#pragma unroll
        for (int k = 0; k < kLoadBatch; ++k) {         // (past the part's last group: that group again)
            const uint32_t kk = (uint32_t)k < cnt ? (uint32_t)k : cnt - 1u;
            x[k] = *reinterpret_cast<const uint4*>(p0 + (size_t)kk * stride + loff);
        }
#pragma unroll
        for (int k = 0; k < kLoadBatch; ++k) {
            if ((uint32_t)k < cnt) {                   // (no break: the array must stay in registers, every index a constant)
                const uint64_t bm = has_newline_ballot(x[k], true, nlc);
                const uint32_t g = (kb + (uint32_t)k) * nws + ws;
                if (lane == 0 && g < (uint32_t)Cfg::HMW) hit[g] = bm;
            }
        }
        return;
    }
    const uint32_t tail = (G.w1 >= n) ? (wlen & 15u) : 0u;      // bytes of a partial last vector (only the block's last window has one)
    for (uint32_t k0 = kb; k0 < ke; k0 += kLoadBatch) {
        uint4 x[kLoadBatch];
#pragma unroll
        for (int k = 0; k < kLoadBatch; ++k) {         // all loads of the batch first ...
            // (the text buffer is readable up to the next multiple of 16: include/mkt.h)
            const uint32_t g0 = (k0 + k) * nws + ws, glast = (nvec - 1u) >> 6, g = g0 < glast ? g0 : glast;
            const uint32_t lim = nvec - (g << 6), l = (uint32_t)lane < lim ? (uint32_t)lane : lim - 1u;
            x[k] = *reinterpret_cast<const uint4*>(wbase + ((size_t)g << 10) + (l << 4));
        }
#pragma unroll
        for (int k = 0; k < kLoadBatch; ++k) {         // ... then the math
            const uint32_t g = (k0 + k) * nws + ws;
            if (k0 + k >= ke || (g << 6) >= nvec) continue;
            uint4 y = x[k];
            if (tail && (g << 6) + 64u >= nvec && (g << 6) + (uint32_t)lane == nvec - 1u) {      // bytes past the end of the block are not text
                uint32_t* w = reinterpret_cast<uint32_t*>(&y);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t lo = (uint32_t)d * 4u;
                    w[d] = tail >= lo + 4u ? w[d] : (tail > lo ? (w[d] & ((1u << ((tail - lo) * 8u)) - 1u)) : 0u);
                }
            }
            const uint64_t bm = has_newline_ballot(y, (uint32_t)lane < nvec - (g << 6), nlc);
            if (lane == 0 && g < (uint32_t)Cfg::HMW) hit[g] = bm;
        }
    }
}

template <class Cfg, int NTW, int RR, int WPS>
__global__ __launch_bounds__(NTW, WPS) void k_fast(KArgs a) {
    __shared__ FastState<Cfg> st;
    __shared__ ScanScratch scan;
    static_assert(Cfg::LCAP <= NTW, "one line per thread in the sums");
    static_assert(NTW % 64 == 0 && NTW <= NT && RR * 64 <= NTW, "whole waves");
    const int tid0 = threadIdx.x;
    const Params P = a.P;
    const uint32_t n = a.n;
    __shared__ OutPtrs s_out;
    const uint32_t region = blockIdx.x & (uint32_t)(a.nregions - 1);
    STAMP_DECL();

    __shared__ uint32_t wg_cnt[C_COUNT];                  // this workgroup's share of the block's counters
    // Newline bitmaps of two windows: while waves 0 and 1 parse and classify tile t, the waves that have nothing to do in those
    // phases (NSW of them, the last ones) scan the window of the workgroup's NEXT tile into the other bitmap
    __shared__ uint64_t s_hit[2][Cfg::HMW + 1];
    constexpr int NWV = NTW / 64;
    constexpr int NSW = NWV >= 4 ? 2 : 0;                 // scanning waves of the pipeline (0: every tile is scanned by all waves at its start)
    // 64-vector groups of a full window: per wave when all waves scan / per scanning wave of the pipeline, in two parts
    const uint32_t ngrp = (a.dims.hb + a.dims.tile + a.dims.hf + 1023u) >> 10;
    const uint32_t lpt_all = (ngrp + NWV - 1) / NWV, lpt_sw = NSW ? (ngrp + NSW - 1) / NSW : 0u, lpt_a = (lpt_sw + 1u) >> 1;
    // (a wave whose last group would lie past a full window stops one short: ngrp need not be a multiple of the wave count)
    auto own = [&](uint32_t ws, uint32_t nws, uint32_t ke) { const uint32_t c = ws < ngrp ? (ngrp - ws + nws - 1u) / nws : 0u; return ke < c ? ke : c; };
    if (tid0 < (int)C_COUNT) wg_cnt[tid0] = 0;
    fast_init(st, (uint32_t)tid0);
    // static tile assignment: no ticket atomic (30 k tiles per block would saturate one address)
    // the output pointers once per workgroup: the per-tile limits in s_out are all rewritten by every claim (the argument block
    // lives in scratch memory by now -- five dependent loads that used to sit in front of lane 0's scan of EVERY tile)
    if (tid0 == 0) s_out = a.out;
    uint32_t gdim = gridDim.x;                                   // (read once: the dispatch packet is a scalar memory load away)
    asm volatile("" : "+s"(gdim));
    uint32_t cur = 0;                                             // bitmap of the current tile
    bool scanned = false;                                         // ... already filled by the previous iteration
    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gdim) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                   // per-lane addresses are recomputed per tile, not kept live (and spilled) across the loop
        // (no barrier: the previous tile ended on one, and nothing below reads what lane 0 resets here before the barrier
        // that ends the scan phase)
        if (tid == 0) fast_reset(st);
        STAMP(0);
        const TileGeom G = tile_geom(t, n, a.dims);             // (window <= Cfg::W: the launcher checks)
        const TextView tv = fast_view(st, a.text, n, G);

        // ---- scan: which 16-byte vectors of the window hold a newline (the workgroup's first tile, and every tile of a build
        //      without the pipeline: all waves; afterwards the bitmap is there already)
        // (the wave number as a SCALAR: a wave that does not scan must branch around the scan's scalar code, not run it with EXEC = 0)
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
        if (!scanned) fast_scan<Cfg, NWV>(a, t, s_hit[cur], wave, 0u, own(wave, NWV, lpt_all), tid & 63);
        scanned = false;
        __syncthreads();
        STAMP(1);
        STOP_AFTER(1)

        // ---- line table: one entry per newline vector, in window order (one wave; a bitmap word per lane)
        if (tid < 64) {
            static_assert(Cfg::HMW <= 64, "one bitmap word per lane of one wave");
            uint64_t m = (uint32_t)tid < ((G.w1 - G.w0 + 1023u) >> 10) ? s_hit[cur][tid] : 0ull;      // (words past the window hold other tiles' bits)
            const uint32_t cnt = (uint32_t)__popcll(m);
            const uint32_t inc = wave_iscan32(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            const uint32_t lead = G.w0 == 0 ? 1u : 0u;             // the block's first line has no newline in front of it
            const uint32_t NLt = total + lead;
            if (NLt > (uint32_t)Cfg::LCAP) {
                if (tid == 0) { st.abn = AB_LCAP; st.NL = 0; }
            } else {
                uint32_t idx = inc - cnt + lead;
                while (m) {
                    st.hv16[idx++] = (uint16_t)(((uint32_t)tid << 6) + (uint32_t)__builtin_ctzll(m));
                    m &= m - 1ull;
                }
                if (tid == 0) {
                    if (lead) st.hv16[0] = 0;
                    st.NL = NLt; st.first_idx = NLt; st.end_idx = NLt;
                    st.last_line_end = G.w1 >= n ? n : kUnknown;   // the last entry's line runs to the block end / out of the window
                }
            }
        }
        __syncthreads();
        const uint32_t NL = st.NL;
        STAMP(2);
        STOP_AFTER(2)
        // ---- line heads: eight aligned 16-byte chunks per line, from the vector that holds the newline in front of it (the
        //      fabric delivers most of them a second time), with their separator bits (byte <= 0x20).  Only the block's last windows can
        //      reach past the end of the text: every other tile takes the path without the per-lane end checks.
        {
            constexpr int HPT = (Cfg::LCAP * Cfg::HCH + NTW - 1) / NTW;    // chunks per lane
            auto heads = [&](auto edge_c) {
                constexpr bool EDGE = decltype(edge_c)::value;
                uint4 q[HPT];
#pragma unroll
                for (int k = 0; k < HPT; ++k) {                                // all loads first ...
                    const uint32_t it = (uint32_t)tid + (uint32_t)k * NTW;
                    q[k] = make_uint4(0, 0, 0, 0);
                    if (it < NL * (uint32_t)Cfg::HCH) {
                        const uint32_t i = it / (uint32_t)Cfg::HCH, c = it % (uint32_t)Cfg::HCH;
                        const uint32_t g0 = G.w0 + (((uint32_t)st.hv16[i] + c) << 4);       // < 2^31 + 64 Ki
                        if (!EDGE || g0 < n) q[k] = *reinterpret_cast<const uint4*>(a.text + g0);
                    }
                }
#pragma unroll
                for (int k = 0; k < HPT; ++k) {                                // ... then the stores and the whitespace bits
                    const uint32_t it = (uint32_t)tid + (uint32_t)k * NTW;
                    if (it >= NL * (uint32_t)Cfg::HCH) continue;
                    const uint32_t i = it / (uint32_t)Cfg::HCH, c = it % (uint32_t)Cfg::HCH;
                    uint4 x = q[k];
                    uint32_t keep = 16u;
                    if (EDGE) {
                        const uint32_t g0 = G.w0 + (((uint32_t)st.hv16[i] + c) << 4);
                        keep = g0 >= n ? 0u : (n - g0 < 16u ? n - g0 : 16u);
                        if (keep != 0u && keep < 16u) {                        // last vector of the block: clear the bytes past the end
                            uint32_t* w = reinterpret_cast<uint32_t*>(&x);
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const uint32_t lo = (uint32_t)d * 4u;
                                w[d] = keep >= lo + 4u ? w[d] : (keep > lo ? (w[d] & ((1u << ((keep - lo) * 8u)) - 1u)) : 0u);
                            }
                        }
                    }
                    {   // rows are 33 dwords apart (bank spread for the parse lanes): four dword stores
                        uint32_t* row = reinterpret_cast<uint32_t*>(&st.win[mul24(i, (uint32_t)Cfg::HSTRIDE) + (c << 4)]);
                        row[0] = x.x; row[1] = x.y; row[2] = x.z; row[3] = x.w;
                    }
                    uint32_t m = pack16(sep_flags(x.x), sep_flags(x.y), sep_flags(x.z), sep_flags(x.w));
                    if (EDGE && keep < 16u) m &= (1u << keep) - 1u;            // cleared bytes are no separators
                    st.u.m.hmask[i][c] = (uint16_t)m;
                }
            };
            if ((uint64_t)G.w1 + 16u * ((uint32_t)Cfg::HCH + 2u) > (uint64_t)n) heads(std::true_type{});
            else heads(std::false_type{});
        }
        __syncthreads();
        STAMP(9);
        STOP_AFTER(9)

        // group phase: lines are dealt round-robin to RR waves (fewer divergent classifier paths per wave)
        const uint32_t ltid = (uint32_t)tid;
        const uint32_t rr_id = ((uint32_t)tid >> 6) < (uint32_t)RR ? ((uint32_t)tid & 63u) * RR + ((uint32_t)tid >> 6) : 0xFFFFFFu;
        {   // one lane per line (LCAP <= NTW); the line masks are the waves' ballots
            const uint32_t i = ltid;
            uint32_t lb = 0, fl = 0;
            if (i < NL) { fast_parse(st, tv, P, G, i); lb = st.bits[i]; fl = st.flag[i]; }
            const bool sv = (lb & LB_SURVIVE) != 0;
            const uint64_t b_s = __ballot(sv), b_e = __ballot((lb & LB_EQPREV) != 0);
            const uint64_t b_1 = __ballot(sv && (fl & 64u)), b_2 = __ballot(sv && !(fl & 64u) && (fl & 128u));
            // lines are in text order: the tile's own lines are [NL - #(start >= t0), NL - #(start >= t1))
            const uint32_t gs = i < NL ? (uint32_t)st.goff[i] : 0u;
            const uint64_t b_t0 = __ballot(i < NL && G.w0 + gs >= G.t0), b_t1 = __ballot(i < NL && G.w0 + gs >= G.t1);
            if ((tid & 63) == 0) {
                const uint32_t w = ltid >> 6; st.m_surv[w] = b_s; st.m_eqp[w] = b_e; st.m_r1[w] = b_1; st.m_r2[w] = b_2;
                if (b_t0) atomicAdd(&st.c_t0, (uint32_t)__popcll(b_t0));
                if (b_t1) atomicAdd(&st.c_t1, (uint32_t)__popcll(b_t1));
            }
            if (NTW < 256 && tid < 4 && tid >= NTW / 64) { st.m_surv[tid] = 0; st.m_eqp[tid] = 0; st.m_r1[tid] = 0; st.m_r2[tid] = 0; st.m_start[tid] = 0; }
        }
        // the scanning waves: first half of the next tile's window (behind their own few lines, if the window has more than 128)
        if (NSW && wave >= (uint32_t)(NWV - NSW) && t + gdim < a.ntiles)
            fast_scan<Cfg, (NSW ? NSW : 1)>(a, t + gdim, s_hit[cur ^ 1u], wave - (uint32_t)(NWV - NSW), 0u, own(wave - (uint32_t)(NWV - NSW), NSW, lpt_a), tid & 63);
        __syncthreads();
        STAMP(3);
        STOP_AFTER(3)
        const uint32_t NLe = fast_nle(st);
        const uint32_t first_idx = NL - st.c_t0 < NLe ? NL - st.c_t0 : NLe;
        const uint32_t end_idx = NL - st.c_t1 < NLe ? NL - st.c_t1 : NLe;
        {   // which surviving lines open a group
            const uint32_t i = ltid;
            bool s0 = false;
            if (!st.abn && i >= first_idx && i < NLe && mask_bit(st.m_surv, i)) s0 = fast_is_start(st, tv, G, i);
            const uint64_t b = __ballot(s0);
            if ((tid & 63) == 0) st.m_start[ltid >> 6] = b;
        }
        __syncthreads();
        STAMP(10);
#if defined(MKT_STAMPS)
        // diagnostic (debug_stop 10): what a kernel that ENDS here would cost -- the front half of a split into "parse" and "classify +
        // emit" kernels.  Every own line leaves a 32-byte record (in the .sam buffer: run it with the .sam output on), the scan of the
        // next window goes on as usual.  Outputs are wrong.
        if (a.debug_stop == 10) {
            if (a.out.sam && !st.abn) for (uint32_t i = first_idx + (uint32_t)tid; i < end_idx; i += NTW) {
                const uint4 r0 = make_uint4(st.rc.f.pos[i], st.rc.f.lclip[i] | (st.rc.f.rclip[i] << 16), st.rc.f.mappable[i] | ((uint32_t)st.flag[i] << 16), st.rc.f.right0[i]);
                const uint4 r1 = make_uint4(st.rc.f.left1[i], st.rc.f.right1[i], G.w0 + st.goff[i], st.qn_len[i] | ((uint32_t)st.rn_off[i] << 8) | ((uint32_t)st.rn_len[i] << 16) | ((uint32_t)st.bits[i] << 24));
                uint4* d = reinterpret_cast<uint4*>(a.out.sam + ((size_t)t * Cfg::LCAP + i) * 32u);
                if (((size_t)t * Cfg::LCAP + i) * 32u + 32u <= a.out.sam_cap) { d[0] = r0; d[1] = r1; }
            }
            if (NSW && wave >= (uint32_t)(NWV - NSW) && t + gdim < a.ntiles)
                fast_scan<Cfg, (NSW ? NSW : 1)>(a, t + gdim, s_hit[cur ^ 1u], wave - (uint32_t)(NWV - NSW), lpt_a, own(wave - (uint32_t)(NWV - NSW), NSW, lpt_sw), tid & 63);
            if (NSW && t + gdim < a.ntiles) { scanned = true; cur ^= 1u; }
            __syncthreads();
            continue;
        }
#endif
        if (!st.abn) for (uint32_t i = first_idx + rr_id; i < end_idx; i += 64 * RR) fast_group(st, tv, P, G, i);
        // the scanning waves: the rest of the next tile's window
        if (NSW && wave >= (uint32_t)(NWV - NSW) && t + gdim < a.ntiles)
            fast_scan<Cfg, (NSW ? NSW : 1)>(a, t + gdim, s_hit[cur ^ 1u], wave - (uint32_t)(NWV - NSW), lpt_a, own(wave - (uint32_t)(NWV - NSW), NSW, lpt_sw), tid & 63);
        if (NSW && t + gdim < a.ntiles) { scanned = true; cur ^= 1u; }
        __syncthreads();
        STAMP(4);
        STOP_AFTER(4)
        STAMP(5);

        // ---- tile sums: groups / emitted / self-circles / .pairs bytes per group, .sam bytes per line --
        {
            auto& g = st.u.g;
            const uint32_t i = first_idx + tid;
            // counts packed 10 bits apiece (<= LCAP each), .pairs bytes, .sam bytes: three 32-bit scans
            uint32_t cc = 0, cp = 0, cs = 0;
            uint32_t info = 0;
            if (i < NLe && !st.abn) {
                if (i < end_idx) {
                    info = g.g_info[i];
                    if (info & GI_START) cc += 1u;
                    if (info & GI_EMIT) cc += 1u << 10;
                    if ((info & GI_START) && (info & GI_COUNTER) == C_SELFCIRCLE) cc += 1u << 20;
                    cp = g.g_plen[i];
                }
                if (P.write_sam) cs = fast_line_sam(st, G, i);
            }
            uint32_t ec = cc, ep = cp, es = cs, tc, tp, ts;
            block_exscan3<NTW / 64>(ec, ep, es, tc, tp, ts, scan);
            if (tid == 0) {
                // the tile's totals, and with them at once its output ranges: the claiming atomic is on its way while the other
                // lanes store their offsets (one barrier fewer than a phase of its own)
                st.sums.groups = tc & 0x3FFu; st.sums.emitted = (tc >> 10) & 0x3FFu; st.sums.sc = (tc >> 20) & 0x3FFu;
                st.sums.pair_bytes = tp; st.sums.sam_bytes = ts;
                if (tp > 0xFFFFu) st.abn = AB_PAIR_BYTES;                    // 16-bit in-tile offsets
                else if (!st.abn) {
                    const uint32_t e = claim_ranges(a, region, st.sums, st.base, st.region_pair0, st.region_sam0, s_out);
                    st.region_id = region;
                    if (e) lds_or(&st.abn, e << 8);
                    a.tile_groups[t] = (uint64_t)st.sums.groups | ((uint64_t)st.sums.emitted << 32);
                }
            }
            if (i < NLe && !(st.abn & 0xFFu)) {                  // (bits 8.. are lane 0's claim errors, possibly set by now)
                g.x_sam[i] = es;
                if (i < end_idx) {
                    g.x_grp[i] = (uint8_t)(ec & 0xFFu); g.x_sc[i] = (uint8_t)((ec >> 20) & 0xFFu); g.x_emit[i] = (uint8_t)((ec >> 10) & 0xFFu);
                    g.x_pair[i] = (uint16_t)(ep & 0xFFFFu);
                    if (info & GI_EMIT) g.em_idx[(ec >> 10) & 0xFFu] = (uint8_t)i;
                }
            }
        }
        __syncthreads();
        STAMP(6);
        STOP_AFTER(6)
        if (st.abn & 0xFFu) {                                 // leave the whole tile to the generic kernel
            if (tid == 0) {
                a.defer_list[atomicAdd(a.defer_count, 1u)] = t;
                // (those a wider halo would have kept: counted apart, the host widens the halos of the following blocks)
                const uint32_t why = st.abn & 0xFFu;
                if (why == AB_NO_PREV || why == AB_OPEN_GROUP) atomicAdd(a.defer_count + 1, 1u);
            }
#if defined(MKT_STAMPS)
            // diagnostic build: why tiles are deferred -- counts per reason code AB_* (1 .. 12), 16 bits apiece, in stamps[12 .. 14]
            if (tid == 0 && a.stamps) { const uint32_t r = (st.abn & 0xFFu) - 1u; if (r < 12u) atomicAdd(&a.stamps[12 + r / 4u], 1ull << (16u * (r & 3u))); }
#endif
            __syncthreads();
            continue;
        }
        // ---- output ranges: claimed by lane 0 at the end of the sums phase (above)
        STAMP(7);
        STOP_AFTER(7)

        // ---- emit ----------------------------------------------------------------------------------
        const uint32_t total = st.sums.pair_bytes;
        uint8_t* const dst = s_out.pairs + st.base.pair_bytes;
        for (uint32_t i = first_idx + ltid; i < end_idx; i += NTW) {
            fast_account(st, s_out, t, i);
            fast_last(st, G, &a.tile_last[t], i);
        }
        // .pairs: one lane per reported pair writes its whole line (fast_emit_line), on the waves the accounting left idle
        if (total && st.base.pair_bytes + total <= s_out.pairs_cap) {
            const auto& g = st.u.g;
            for (uint32_t e = (ltid + (uint32_t)(NTW / 2)) & (uint32_t)(NTW - 1); e < st.sums.emitted; e += NTW) {
                const uint32_t i = g.em_idx[e];
                fast_emit_line(st, g.g_slot[i], g.g_plen[i], dst + g.x_pair[i]);
            }
        }
        __syncthreads();
        STAMP(11);
        // counters and error bits of the tile are final here; flushed once per workgroup (9 global atomics per TILE on one
        // cache line would queue up behind each other)
        if (tid < (int)C_COUNT) wg_cnt[tid] += st.cnt[tid];
        if (tid == 0 && (st.abn >> 8)) atomicOr(&a.res->err, st.abn >> 8);
        // The tile's .sam bytes are ONE contiguous range of the output (the emitting lines in order): copied destination first --
        // every thread takes aligned 16-byte vectors of that range (full lanes, coalesced stores) and finds the line a vector
        // lies in by bisection over the emitting lines' offsets; the text comes straight from global memory (L2).
        const bool sam_copy = P.write_sam && st.sums.sam_bytes && st.base.sam_bytes + st.sums.sam_bytes <= s_out.sam_cap;      // (uniform)
        uint32_t* eo = st.rc.f.pos;                                 // (the parse records are dead since the classifier)
        uint32_t* es = st.rc.f.rclip;                               // (eo[E] may be pos[LCAP] = lclip[0], unused here)
        uint32_t E = 0;
        if (sam_copy) {
            E = (uint32_t)(__popcll(st.m_emit[0]) + __popcll(st.m_emit[1]) + __popcll(st.m_emit[2]) + __popcll(st.m_emit[3]));
            for (uint32_t i = first_idx + (uint32_t)tid; i < NLe; i += NTW) {
                if (!mask_bit(st.m_emit, i)) continue;
                uint32_t k = (uint32_t)__popcll(st.m_emit[i >> 6] & ((1ull << (i & 63u)) - 1ull));
                for (uint32_t w = 0; w < (i >> 6); ++w) k += (uint32_t)__popcll(st.m_emit[w]);
                eo[k] = st.u.g.x_sam[i];
                es[k] = st.goff[i];
            }
            if (tid == 0) eo[E] = (uint32_t)st.sums.sam_bytes;
        }
        if (P.write_sam) __syncthreads();                           // (kernel-uniform)
        if (sam_copy && E) {
            const uint32_t total = (uint32_t)st.sums.sam_bytes;
            uint8_t* dbase = s_out.sam + st.base.sam_bytes;
            const uint8_t* tbase = a.text + G.w0;
            auto line_of = [&](uint32_t d) {                        // the emitting line that holds output byte d
                uint32_t lo = 0, hi = E - 1u;
                while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (eo[mid] <= d) lo = mid; else hi = mid - 1u; }
                return lo;
            };
            const uint32_t head0 = (uint32_t)((16u - ((uintptr_t)dbase & 15u)) & 15u);
            const uint32_t head = head0 < total ? head0 : total;
            const uint32_t nv = (total - head) >> 4, tail0 = head + (nv << 4);
            if ((uint32_t)tid < head) { const uint32_t k = line_of((uint32_t)tid); *MKT_GLOBAL(uint8_t, dbase + tid) = tbase[es[k] + (uint32_t)tid - eo[k]]; }
            if ((uint32_t)tid < total - tail0) { const uint32_t d = tail0 + (uint32_t)tid, k = line_of(d); *MKT_GLOBAL(uint8_t, dbase + d) = tbase[es[k] + d - eo[k]]; }
            for (uint32_t v = (uint32_t)tid; v < nv; v += NTW) {
                const uint32_t d = head + (v << 4);
                const uint32_t k = line_of(d);
                const uint32_t o = d - eo[k], left = eo[k + 1u] - d;               // bytes of line k from here on
                const uint32_t so = es[k] + o;
                uint4 x;
                if (left >= 16u) __builtin_memcpy(&x, tbase + so, 16);
                else if (k + 2u <= E && eo[k + 2u] - eo[k + 1u] >= 16u - left && (uint64_t)G.w0 + so + 16u <= (uint64_t)n) {
                    // the vector ends in the next emitting line: `left` bytes from here, the rest from that line's start
                    uint4 xa, xb;
                    __builtin_memcpy(&xa, tbase + so, 16);
                    __builtin_memcpy(&xb, tbase + es[k + 1u] - left, 16);
                    const uint32_t* pa = reinterpret_cast<const uint32_t*>(&xa);
                    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&xb);
                    uint32_t* px = reinterpret_cast<uint32_t*>(&x);
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) {
                        const uint32_t m = left >= 4u * q + 4u ? 0xFFFFFFFFu : (left <= 4u * q ? 0u : (1u << (8u * (left - 4u * q))) - 1u);
                        px[q] = (pa[q] & m) | (pb[q] & ~m);
                    }
                } else {                                            // three or more lines in 16 bytes, or the end of the block: byte by byte
                    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                    uint32_t kk = k, oo = so, lim = left;
#pragma unroll
                    for (uint32_t b2 = 0; b2 < 16u; ++b2) {
                        while (lim == 0u) { ++kk; oo = es[kk]; lim = eo[kk + 1u] - eo[kk]; }
                        const uint32_t by = (uint32_t)tbase[oo++] << (8u * (b2 & 3u));
                        if (b2 < 4u) q0 |= by; else if (b2 < 8u) q1 |= by; else if (b2 < 12u) q2 |= by; else q3 |= by;
                        --lim;
                    }
                    x = make_uint4(q0, q1, q2, q3);
                }
                *MKT_GLOBAL(uint4, dbase + d) = x;
            }
        }
        __syncthreads();                                           // every lane is done with this tile's state
        STAMP(8);
    }
    if (tid0 < (int)C_COUNT && wg_cnt[tid0]) atomicAdd(&a.res->counters[tid0], wg_cnt[tid0]);
    STAMP_FLUSH(tid0);
}

// After the tiles, step 1 (one workgroup per 1024 tiles): exclusive scan of the per-tile group counts
// (chunk prefix by the same decoupled look-back, on descA which the any-order kernels leave unused...
// in ordered mode descA is busy, so the scan uses its own words in a.scan_desc), and the block's last
// tile that opened a group.
constexpr int NTF = 1024;
__global__ __launch_bounds__(NTF) void k_finish_scan(KArgs a) {
    __shared__ uint64_t s_wave[NTF / 64];
    __shared__ uint64_t s_pre;
    __shared__ uint32_t s_chunk;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // chunks are handed out by ticket, in the order workgroups actually start: a chunk only ever waits (look-back) for
    // chunks whose workgroups are already running, whatever the dispatch order of blockIdx
    if (tid == 0) s_chunk = atomicAdd(a.scan_ticket, 1u);
    __syncthreads();
    const uint32_t chunk = s_chunk;
    const uint32_t t = chunk * NTF + tid;
    const uint64_t x = t < a.ntiles ? a.tile_groups[t] : 0ull;       // groups | emitted << 32
    if (t < a.ntiles && a.tile_last[t].valid) atomicMax(a.last_tile, (int)t + 1);      // stored + 1: the zeroed workspace means "none"
    uint64_t inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint64_t y = shfl_up64(inc, d); if (lane >= d) inc += y; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint64_t pre = 0, tot = 0;
    for (int w = 0; w < NTF / 64; ++w) { if (w < wv) pre += s_wave[w]; tot += s_wave[w]; }
    if (wv == 0) {
        // payload: groups (31 bits) | emitted (31 bits)
        const uint64_t ex = lookback(a.scan_desc, chunk, ((tot & 0x7FFFFFFFull) << 31) | (tot >> 32), &a.res->err);
        if (lane == 0) s_pre = (ex >> 31) | ((ex & 0x7FFFFFFFull) << 32);
    }
    __syncthreads();
    if (t < a.ntiles) a.tile_groups[t] = s_pre + pre + inc - x;
    if (chunk == gridDim.x - 1 && tid == 0) a.res->groups = (s_pre + tot) & 0xFFFFFFFFull;
}

// step 2 (a few workgroups; all compute the same totals from the region cursors, workgroup 0 publishes them):
// self-circle entries resolved to global group indices, the block's last group (quirk Q1 bookkeeping happens
// on the host).  The run totals advance in k_run_advance, after every reader of the old values.
constexpr int kFinishGrid = 16;
__global__ __launch_bounds__(NTF) void k_finish(KArgs a) {
    const int tid = threadIdx.x;
    BlockResult* r = a.res;
    __shared__ uint64_t s_scpre[kMaxRegions + 1], s_ca[kMaxRegions], s_cb[kMaxRegions];
    __shared__ uint32_t s_err;
    if (tid < a.nregions) { s_ca[tid] = a.cur[tid].a; s_cb[tid] = a.cur[tid].b; }      // one lane per region: the loads overlap
    __syncthreads();
    if (tid == 0) {
        uint64_t pb = 0, em = 0, sb = 0, sc = 0;
        uint32_t err = 0;
        for (int q = 0; q < a.nregions; ++q) {
            const uint64_t ca = s_ca[q], cb = s_cb[q];
            const uint64_t p = ca & kCurLow, e = ca >> kCurShift, sm = cb & kCurLow, c = cb >> kCurShift;
            if (blockIdx.x == 0) { r->rpair[q] = p; r->rsam[q] = sm; }
            s_scpre[q] = sc;
            pb += p; em += e; sb += sm; sc += c;
            if (!a.ordered) {
                if (p > a.pairs_rcap) err |= E_PAIRS_CAP;
                if (sm > a.sam_rcap) err |= E_SAM_CAP;
                if (c > a.sc_rcap) err |= E_SC_CAP;
            }
        }
        s_scpre[a.nregions] = sc;
        if (a.ordered) { if (pb > a.out.pairs_cap) err |= E_PAIRS_CAP; if (sb > a.out.sam_cap) err |= E_SAM_CAP; if (sc > a.out.sc_cap) err |= E_SC_CAP; }
        if (a.run->sc + sc > a.sc_list_cap) err |= E_SC_CAP;
        if (a.keys_rcap && a.run->emitted + em > a.key_list_cap) err |= E_SC_CAP;
        s_err = err | r->err;                                     // r->err: what the tile kernels and the scan reported
        if (blockIdx.x == 0) {
            r->pair_bytes = pb; r->emitted = em; r->sam_bytes = sb; r->sc = sc; r->nregions = (uint32_t)a.nregions;
            r->pad = a.defer_count ? *a.defer_count : 0u;         // tiles the lean kernel deferred
            r->pad2 = a.defer_count ? a.defer_count[1] : 0u;      // ... those of them for a halo that was too narrow
            r->tiles = a.ntiles;
            const int last = *a.last_tile - 1;
            if (last >= 0) r->last = a.tile_last[last];
        }
    }
    __syncthreads();
    if (s_err == 0) {
        // self-circle entries (tile, ordinal) of every region -> global group indices, appended to the run's list
        const uint64_t sc_base = a.run->sc, g_base = a.run->groups;
        // the workgroups split the regions among themselves (and one region's entries when there are more workgroups than
        // regions): short independent loops instead of one pass over every region by everybody
        const int nr = a.nregions;
        const int parts = (int)gridDim.x >= nr ? (int)gridDim.x / nr : 1;
        for (int q = (int)blockIdx.x % nr; q < nr; q += (int)gridDim.x) {
            const int part = (int)blockIdx.x / nr;
            if (part >= parts) break;
            const uint64_t cnt = s_scpre[q + 1] - s_scpre[q];
            const uint64_t* src = a.out.sc + (uint64_t)q * (a.ordered ? 0 : a.sc_rcap);
            for (uint64_t k = (uint64_t)part * NTF + tid; k < cnt; k += (uint64_t)parts * NTF) {
                const uint64_t e = src[k];
                a.sc_list[sc_base + s_scpre[q] + k] = g_base + (a.tile_groups[(uint32_t)(e >> 32)] & 0xFFFFFFFFull) + (uint32_t)(e & 0xFFFFFFFFu);
            }
        }
    }
    __syncthreads();                                              // every read of r->err above precedes this write
    if (blockIdx.x == 0 && tid == 0 && (s_err & ~r->err)) r->err |= s_err;
}

// step 3 (extension, many workgroups): the block's raw key records -> the run's key list, at their
// emitted-pair ordinal in INPUT order (run total + tile prefix + ordinal in tile); then the run totals advance.
__global__ __launch_bounds__(256) void k_keys_place(KArgs a) {
    const BlockResult* r = a.res;
    if (r->err) return;
    // the regions' record counts as one index space (one pass over all records, every lane's loads independent of each other --
    // region after region, each with its own grid-stride loop, was sixteen short dependent phases: 39 us for 1.4 M records)
    __shared__ uint64_t pre[kMaxRegions + 1];
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        for (int q = 0; q < kMaxRegions; ++q) { pre[q] = acc; if (q < a.nregions) acc += a.cur[q].a >> kCurShift; }
        pre[kMaxRegions] = acc;
    }
    __syncthreads();
    const uint64_t base = a.run->emitted, total = pre[kMaxRegions];
    const uint64_t rstride = a.ordered ? 0 : a.keys_rcap;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t q = 0;
#pragma unroll
        for (uint32_t k = 1; k < (uint32_t)kMaxRegions; ++k) q += (x >= pre[k]) ? 1u : 0u;
        KeyRec rec = a.out.keys[(uint64_t)q * rstride + (x - pre[q])];
        const uint64_t ord = base + (a.tile_groups[(uint32_t)(rec.ord >> 16)] >> 32) + (rec.ord & 0xFFFFu);
        rec.ord = ord;
        a.key_list[ord] = rec;
    }
}
__global__ void k_run_advance(KArgs a) {
    BlockResult* r = a.res;
    if (threadIdx.x == 0 && r->err == 0) { a.run->groups += r->groups; a.run->sc += r->sc; a.run->emitted += r->emitted; }     // a failed block is re-run
}

// Streaming path, after k_finish: one output of the block (its .pairs or its .sam bytes, which sit in `nregions` slices of the
// region buffer) -> ONE contiguous range in `dst`, with the bytes of the block's last group moved to the very end:
//     dst = [ everything but the last group | last group ]
// The host then copies one range over PCIe and holds back only that tail (quirk Q1) instead of editing 16 slices.
// dst is 16-byte aligned: every lane stores one aligned 16-byte chunk, loaded from wherever its bytes come from.
struct GatherArgs {
    const BlockResult* res;
    const uint8_t* src; uint64_t rcap;          // region buffer, bytes per region slice
    uint8_t* dst; uint64_t dst_cap;
    int32_t which;                              // 0: .pairs, 1: .sam
};
__global__ __launch_bounds__(256) void k_gather(GatherArgs g) {
    __shared__ uint64_t pre[kMaxRegions + 1];
    __shared__ uint64_t s_hole, s_tp;
    const BlockResult* r = g.res;
    if (r->err) return;                                               // a failed block is re-run
    if (threadIdx.x == 0) {
        const uint32_t nr = r->nregions ? r->nregions : 1u;
        uint64_t acc = 0;
        for (uint32_t q = 0; q < (uint32_t)kMaxRegions; ++q) { pre[q] = acc; if (q < nr) acc += g.which ? r->rsam[q] : r->rpair[q]; }
        pre[kMaxRegions] = acc;
        uint64_t tp = 0, hole = acc;
        if (r->last.valid) {
            tp = g.which ? r->last.sam_bytes : r->last.pair_bytes;
            const uint32_t lr = r->last.region < nr ? r->last.region : 0u;
            hole = pre[lr] + (g.which ? r->last.sam_off : r->last.pair_off);
        }
        s_hole = hole; s_tp = tp;
    }
    __syncthreads();
    const uint64_t total = pre[kMaxRegions], tp = s_tp, hole = s_hole, body = total - tp;
    if (total > g.dst_cap) return;                                    // the host sizes dst from the region buffer: cannot happen
    auto src_of = [&](uint64_t x) -> const uint8_t* {                 // byte x of dst comes from ...
        const uint64_t y = x < body ? (x < hole ? x : x + tp) : hole + (x - body);      // ... byte y of the concatenated slices
        uint32_t q = 0;
#pragma unroll
        for (uint32_t k = 1; k < (uint32_t)kMaxRegions; ++k) q += (y >= pre[k]) ? 1u : 0u;      // slices are in order; empty ones share a prefix
        return g.src + (uint64_t)q * g.rcap + (y - pre[q]);
    };
    for (uint64_t x0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u; x0 < total; x0 += (uint64_t)gridDim.x * blockDim.x * 16u) {
        const uint64_t x1 = x0 + 15u < total ? x0 + 15u : total - 1u;
        const uint8_t* a = src_of(x0);
        const uint8_t* b = src_of(x1);
        if (x1 - x0 == 15u && (uint64_t)(b - a) == 15u) {            // the usual case: 16 contiguous source bytes
            uint4 v;
            __builtin_memcpy(&v, a, 16);
            *reinterpret_cast<uint4*>(g.dst + x0) = v;
        } else {
            for (uint64_t x = x0; x <= x1; ++x) g.dst[x] = *src_of(x);
        }
    }
}
hipError_t launch_gather(const BlockResult* d_res, const uint8_t* src, uint64_t rcap, uint8_t* dst, uint64_t dst_cap, int which, uint64_t max_bytes, hipStream_t s) {
    GatherArgs g;
    g.res = d_res; g.src = src; g.rcap = rcap; g.dst = dst; g.dst_cap = dst_cap; g.which = which;
    uint64_t chunks = (max_bytes + 15) / 16;
    unsigned grid = (unsigned)((chunks + 255) / 256 < 2048 ? (chunks + 255) / 256 : 2048);
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, s, g);
    return hipGetLastError();
}

// End of the input: how many of the run's self-circle groups the reference would have LOGGED (quirk Q2: only its
// thread 0's share of every batch reaches the .log).  The list never leaves the device.
__global__ void k_sc_logged(const uint64_t* list, uint64_t n, uint64_t drop_group, uint64_t group_offset, uint64_t K_total, uint32_t ref_threads,
                            unsigned long long* out) {
    unsigned long long mine = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g = list[k];
        if (g == drop_group) continue;                                  // the dropped last group (quirk Q1)
        if (selfcircle_logged(group_offset + g, K_total, ref_threads)) ++mine;
    }
    for (int d = 32; d >= 1; d >>= 1) mine += shfl_xor64(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}
hipError_t launch_sc_logged(const uint64_t* list, uint64_t n, uint64_t drop_group, uint64_t group_offset, uint64_t K_total, uint32_t ref_threads,
                            unsigned long long* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_sc_logged, dim3(grid), dim3(256), 0, s, list, n, drop_group, group_offset, K_total, ref_threads, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
typedef FastCfg<kLeanTile, kLeanHB, kLeanHF, kLeanLCAP> CfgLean;               // capacities of the lean kernel
typedef TileCfg<kLeanTile, kLeanHB, kLeanHF, 512, 4> CfgFast;                 // ... and of the generic kernel that takes what it defers
typedef TileCfg<256, 64, 192, 512, 4> CfgSmall;

TileDims small_dims() { return cfg_dims<CfgSmall>(); }
TileDims max_dims() { return cfg_dims<CfgLean>(); }
static bool dims_ok(const TileDims& d, const TileDims& mx) {
    return d.tile >= 16u && !((d.tile | d.hb | d.hf) & 15u) && d.tile <= mx.tile && d.hb <= mx.hb && d.hf <= mx.hf;
}

uint32_t fast_max_workgroups(int) { return 256u * 4u; }      // 4 workgroups of 256 threads per CU (k_fast is compiled for 4 waves per SIMD)

hipError_t launch_tiles(const KArgs& a, int cfg, int grid, hipStream_t s) {
    if (a.ntiles == 0) return hipSuccess;
    if (!dims_ok(a.dims, cfg == CFG_SMALL ? small_dims() : max_dims())) return hipErrorInvalidValue;
    if (cfg == CFG_SMALL) hipLaunchKernelGGL(k_tiles<CfgSmall>, dim3(grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL(k_tiles<CfgFast>, dim3(grid), dim3(NT), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fast(const KArgs& a, int, int grid, hipStream_t s) {
    if (a.ntiles == 0) return hipSuccess;
    if (!dims_ok(a.dims, max_dims())) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_fast<CfgLean, NT, 2, 4>), dim3(grid), dim3(NT), 0, s, a);
    return hipGetLastError();
}
// line-length probe: newlines in the first bytes of device-resident text
__global__ void k_count_nl(const uint8_t* text, size_t n, unsigned long long* out) {
    unsigned long long mine = 0;
    const size_t nv = n >> 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = reinterpret_cast<const uint4*>(text)[v];
        mine += __popc(nl_flags(x.x)) + __popc(nl_flags(x.y)) + __popc(nl_flags(x.z)) + __popc(nl_flags(x.w));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) for (size_t b = nv << 4; b < n; ++b) mine += text[b] == '\n';
    for (int d = 32; d >= 1; d >>= 1) mine += shfl_xor64(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}
hipError_t launch_count_newlines(const uint8_t* text, size_t n, unsigned long long* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_count_nl, dim3(64), dim3(256), 0, s, text, n, out);
    return hipGetLastError();
}
hipError_t launch_finish(const KArgs& a, hipStream_t s) {
    const unsigned chunks = (a.ntiles + NTF - 1) / NTF;
    if (chunks) hipLaunchKernelGGL(k_finish_scan, dim3(chunks), dim3(NTF), 0, s, a);
    hipLaunchKernelGGL(k_finish, dim3(kFinishGrid), dim3(NTF), 0, s, a);
    if (a.keys_rcap) hipLaunchKernelGGL(k_keys_place, dim3(2048), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_run_advance, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}
uint32_t finish_chunk_tiles() { return NTF; }

// ---------------------------------------------------------------------------------------------
// Extension A9: duplicate marking over the run's key list (input order).  A pair is a duplicate when
// an EARLIER pair has the same (chr1, pos1, chr2, pos2, strand1, strand2).  Hand-written LSD radix
// sort of one u64 per pair, (32-bit key hash << 32 | input index) -- 4-bit digits, ballot ranking,
// stable -- over only as many hash bits as the pair count needs (log2 n + 2: equal-digit runs then hold
// 1/4 element on average); every element then looks back inside its run for an equal FULL key (exact;
// hash ties only cost time, and within a run earlier in memory = earlier in the input).
constexpr int DD_WG = 256;
constexpr uint64_t kKeyMask1 = 0xFFFFFFFFC000FFFFull;          // posB, the two strand bits and the lane (0 unless MKT_EXT_LANES) of KeyRec::k1

__device__ inline bool key_eq(const KeyRec& x, const KeyRec& y) { return x.k0 == y.k0 && (x.k1 & kKeyMask1) == (y.k1 & kKeyMask1); }

__global__ void k_dd_init(const KeyRec* keys, uint64_t n, uint64_t* rec) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        rec[j] = (mix64(keys[j].k0 ^ mix64(keys[j].k1 & kKeyMask1)) & 0xFFFFFFFF00000000ull) | j;
}
__global__ __launch_bounds__(DD_WG) void k_dd_hist(const uint64_t* rec, uint64_t n, uint64_t per, int shift, uint32_t* hist, uint32_t G) {
    __shared__ uint32_t cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t j = b + threadIdx.x; j < e; j += DD_WG) atomicAdd(&cnt[(rec[j] >> shift) & 15u], 1u);
    __syncthreads();
    if (threadIdx.x < 16) hist[threadIdx.x * G + blockIdx.x] = cnt[threadIdx.x];
}
__global__ __launch_bounds__(NT) void k_dd_scan(uint32_t* hist, uint32_t m) {       // exclusive scan of m counters, one workgroup
    __shared__ ScanScratch sc;
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < m; base += NT) {
        const uint32_t i = base + threadIdx.x;
        uint64_t x = i < m ? hist[i] : 0, d = 0, tx, td, e = x;
        block_exscan2(e, d, tx, td, sc);
        if (i < m) hist[i] = (uint32_t)(carry + e);
        __syncthreads();
        if (threadIdx.x == 0) carry += tx;
        __syncthreads();
    }
}
__global__ __launch_bounds__(DD_WG) void k_dd_scatter(const uint64_t* rec, uint64_t n, uint64_t per, int shift, const uint32_t* hist, uint32_t G, uint64_t* rec2) {
    __shared__ uint32_t base[16];
    __shared__ uint32_t wcnt[DD_WG / 64][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 16) base[tid] = hist[tid * G + blockIdx.x];
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t j0 = b; j0 < e; j0 += DD_WG) {            // sub-tiles in input order keep the pass stable
        const uint64_t j = j0 + tid;
        const bool live = j < e;
        const uint64_t rv = live ? rec[j] : 0;
        const uint32_t d = live ? (uint32_t)((rv >> shift) & 15u) : 16u;
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t dd = 0; dd < 16; ++dd) {
            const uint64_t m = __ballot(d == dd);
            if (d == dd) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wv][dd] = __popcll(m);
        }
        __syncthreads();
        if (live) {
            uint32_t o = base[d] + rank;
            for (int w = 0; w < wv; ++w) o += wcnt[w][d];
            rec2[o] = rv;
        }
        __syncthreads();
        if (tid < 16) { uint32_t s = 0; for (int w = 0; w < DD_WG / 64; ++w) s += wcnt[w][tid]; base[tid] += s; }
        __syncthreads();
    }
}
// run = neighbours that agree on the sorted hash bits.  Keys are fetched (random 24-byte gathers) only on a full 32-bit
// hash match with an earlier element of the run, i.e. almost never unless it IS a duplicate; flags start out zeroed.
__global__ void k_dd_mark(const KeyRec* keys, const uint64_t* rec, uint64_t n, uint64_t run_mask, uint8_t* flags, DedupResult* res) {
    uint32_t mine = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = rec[j];
        bool dup = false, have = false;
        KeyRec me;
        for (uint64_t b = j; b-- > 0;) {
            const uint64_t o = rec[b];
            if ((o ^ r) & run_mask) break;                                  // left the run
            if ((o >> 32) != (r >> 32)) continue;
            if (!have) { me = keys[(uint32_t)r]; have = true; }
            if (key_eq(keys[(uint32_t)o], me)) { dup = true; break; }       // earlier in the run = earlier in the input
        }
        if (dup) { flags[(uint32_t)r] = 1; ++mine; }
    }
    if (mine) atomicAdd((unsigned long long*)&res->dups, (unsigned long long)mine);
    if (blockIdx.x == 0 && threadIdx.x == 0) res->total = n;
}
size_t dedup_work_bytes(uint64_t n) {
    const uint64_t G = 1024;
    return (size_t)(n * (8 + 8) + 16 * G * 4 + 4096);
}
hipError_t launch_dedup(const KeyRec* keys, uint64_t n, uint8_t* flags, void* work, size_t work_bytes, DedupResult* d_res, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_res, 0, sizeof(DedupResult), s);
    if (e != hipSuccess || n == 0) return e;
    if (work_bytes < dedup_work_bytes(n) || n >= (1ull << 32)) return hipErrorInvalidValue;
    uint8_t* w = (uint8_t*)work;
    uint64_t* rA = (uint64_t*)w; w += n * 8;
    uint64_t* rB = (uint64_t*)w; w += n * 8;
    uint32_t* hist = (uint32_t*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    uint32_t G = (uint32_t)((n + 8191) / 8192);
    if (G > 1024) G = 1024;
    if (G == 0) G = 1;
    const uint64_t per = (n + G - 1) / G;
    int bits = 2;                                             // log2(n) + 2, in whole digits, 12 .. 32
    while (bits < 34 && (1ull << (bits - 2)) < n) ++bits;
    bits = bits < 12 ? 12 : (bits > 32 ? 32 : bits);
    const int passes = (bits + 3) / 4;
    hipLaunchKernelGGL(k_dd_init, dim3(1024), dim3(256), 0, s, keys, n, rA);
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(k_dd_hist, dim3(G), dim3(DD_WG), 0, s, (const uint64_t*)rA, n, per, 32 + 4 * p, hist, G);
        hipLaunchKernelGGL(k_dd_scan, dim3(1), dim3(NT), 0, s, hist, 16u * G);
        hipLaunchKernelGGL(k_dd_scatter, dim3(G), dim3(DD_WG), 0, s, (const uint64_t*)rA, n, per, 32 + 4 * p, (const uint32_t*)hist, G, rB);
        uint64_t* t = rA; rA = rB; rB = t;
    }
    const uint64_t run_mask = ((passes * 4 >= 32 ? 0xFFFFFFFFull : ((1ull << (passes * 4)) - 1ull))) << 32;
    e = hipMemsetAsync(flags, 0, n, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dd_mark, dim3(1024), dim3(256), 0, s, keys, (const uint64_t*)rA, n, run_mask, flags, d_res);
    return hipGetLastError();
}

// ---- sharded duplicate marking (one context per GPU): hash-partitioned exchange of the key space ----------------
// Every rank sends each key record to rank mix64(key) % world (RCCL all_to_all over xGMI, microcket_amd/shard.py), so equal
// keys meet on one rank, which marks them with the kernels above.  k_part_* is the stable partition in front of that:
// chromosome slots are rewritten to ids that are the same on every rank (lut, from the exchanged name tables), records are
// grouped by destination IN INPUT ORDER (so that "first in input order wins" survives the exchange: ranks hold contiguous
// shards, segments arrive in rank order), and perm[j] remembers where record j went (the flags come back the same way).
constexpr uint32_t kMaxWorld = 16;
__device__ inline KeyRec part_remap(KeyRec r, const uint16_t* lut) {
    if (lut) {
        const uint64_t a = lut[(r.k0 >> 45) & (kChrSlots - 1u)], b = lut[(r.k0 >> 32) & (kChrSlots - 1u)];
        r.k0 = (a << 45) | (b << 32) | (r.k0 & 0xFFFFFFFFull);
    }
    return r;
}
__device__ inline uint32_t part_dest(const KeyRec& r, uint32_t world) { return (uint32_t)(mix64(r.k0 ^ mix64(r.k1 & kKeyMask1)) % world); }
__global__ __launch_bounds__(DD_WG) void k_part_hist(const KeyRec* keys, uint64_t n, uint64_t per, const uint16_t* lut, uint32_t world, uint32_t* hist, uint32_t G) {
    __shared__ uint32_t cnt[kMaxWorld];
    if (threadIdx.x < kMaxWorld) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t j = b + threadIdx.x; j < e; j += DD_WG) atomicAdd(&cnt[part_dest(part_remap(keys[j], lut), world)], 1u);
    __syncthreads();
    if (threadIdx.x < kMaxWorld) hist[threadIdx.x * G + blockIdx.x] = cnt[threadIdx.x];
}
__global__ __launch_bounds__(DD_WG) void k_part_scatter(const KeyRec* keys, uint64_t n, uint64_t per, const uint16_t* lut, uint32_t world, const uint32_t* hist,
                                                       uint32_t G, KeyRec* send, uint32_t* perm) {
    __shared__ uint32_t base[kMaxWorld];
    __shared__ uint32_t wcnt[DD_WG / 64][kMaxWorld];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < (int)kMaxWorld) base[tid] = hist[tid * G + blockIdx.x];
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t j0 = b; j0 < e; j0 += DD_WG) {            // sub-tiles in input order keep the partition stable
        const uint64_t j = j0 + tid;
        const bool live = j < e;
        KeyRec r;
        r.k0 = r.k1 = r.ord = 0;
        if (live) r = part_remap(keys[j], lut);
        const uint32_t d = live ? part_dest(r, world) : kMaxWorld;
        uint32_t rank = 0;
        for (uint32_t dd = 0; dd < world; ++dd) {
            const uint64_t m = __ballot(d == dd);
            if (d == dd) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wv][dd] = __popcll(m);
        }
        __syncthreads();
        if (live) {
            uint32_t o = base[d] + rank;
            for (int w = 0; w < wv; ++w) o += wcnt[w][d];
            send[o] = r;
            perm[j] = o;
        }
        __syncthreads();
        if (tid < (int)world) { uint32_t s2 = 0; for (int w = 0; w < DD_WG / 64; ++w) s2 += wcnt[w][tid]; base[tid] += s2; }
        __syncthreads();
    }
}
__global__ void k_unpermute_flags(const uint8_t* in, const uint32_t* perm, uint64_t n, uint8_t* out, unsigned long long* dups) {
    uint32_t mine = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t f = in[perm[j]];
        out[j] = f;
        mine += f ? 1u : 0u;
    }
    if (mine) atomicAdd(dups, (unsigned long long)mine);
}
size_t partition_work_bytes() { return (size_t)(kMaxWorld * 1024 + 64) * 4; }
// hist: kMaxWorld x G counters (device work buffer); after the call hist[d * G] is where destination d starts in `send`
hipError_t launch_partition(const KeyRec* keys, uint64_t n, const uint16_t* d_lut, uint32_t world, uint32_t* hist, KeyRec* send, uint32_t* perm, uint32_t* G_out, hipStream_t s) {
    if (world == 0 || world > kMaxWorld || n >= (1ull << 32)) return hipErrorInvalidValue;
    uint32_t G = (uint32_t)((n + 8191) / 8192);
    if (G > 1024) G = 1024;
    if (G == 0) G = 1;
    *G_out = G;
    const uint64_t per = (n + G - 1) / G;
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)kMaxWorld * G * 4, s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(k_part_hist, dim3(G), dim3(DD_WG), 0, s, keys, n, per, d_lut, world, hist, G);
    hipLaunchKernelGGL(k_dd_scan, dim3(1), dim3(NT), 0, s, hist, kMaxWorld * G);
    hipLaunchKernelGGL(k_part_scatter, dim3(G), dim3(DD_WG), 0, s, keys, n, per, d_lut, world, (const uint32_t*)hist, G, send, perm);
    return hipGetLastError();
}
hipError_t launch_unpermute(const uint8_t* in, const uint32_t* perm, uint64_t n, uint8_t* out, unsigned long long* dups, hipStream_t s) {
    hipError_t e = hipMemsetAsync(dups, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(k_unpermute_flags, dim3(1024), dim3(256), 0, s, in, perm, n, out, dups);
    return hipGetLastError();
}

// Extension A10: contact counts per (chrA, chrB) over the key list.  dense_of_slot maps name-table slots to
// dense ids (built on the host from the table); counts is ndense x ndense.  Ids < 32 go through an LDS
// histogram (the main chromosomes), the rest straight to global atomics.
__global__ __launch_bounds__(256) void k_chrstat(const KeyRec* keys, uint64_t n, const uint16_t* dense_of_slot, uint32_t ndense, unsigned long long* counts) {
    __shared__ uint32_t loc[32 * 32];
    for (int k = threadIdx.x; k < 1024; k += 256) loc[k] = 0;
    __syncthreads();
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k0 = keys[j].k0;
        const uint32_t a = dense_of_slot[(k0 >> 45) & (kChrSlots - 1u)], b = dense_of_slot[(k0 >> 32) & (kChrSlots - 1u)];
        if (a < 32u && b < 32u) atomicAdd(&loc[a * 32u + b], 1u);
        else atomicAdd(&counts[(uint64_t)a * ndense + b], 1ull);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) {
        const uint32_t a = k >> 5, b = k & 31;
        if (loc[k] && a < ndense && b < ndense) atomicAdd(&counts[(uint64_t)a * ndense + b], (unsigned long long)loc[k]);
    }
}
hipError_t launch_chrstat(const KeyRec* keys, uint64_t n, const uint16_t* dense_of_slot, uint32_t ndense, unsigned long long* counts, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_chrstat, dim3(512), dim3(256), 0, s, keys, n, dense_of_slot, ndense, counts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// synthetic SAM: sizes, then bytes (offsets from an exclusive scan done between the two kernels)
__global__ void k_synth_sizes(SynParams p, uint64_t first, uint64_t n, uint64_t* sizes) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SynCountSink c;
    synth_group(c, p, first + i);
    sizes[i] = c.n;
}
__global__ void k_synth_write(SynParams p, uint64_t first, uint64_t n, const uint64_t* offs, char* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SynMemSink m{out + offs[i]};
    synth_group(m, p, first + i);
}
// single-workgroup chained exclusive scan (generator only; not on the hot path)
__global__ void k_exscan_u64(uint64_t* v, uint64_t n, uint64_t* total) {
    __shared__ ScanScratch sc;
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += NT) {
        uint64_t i = base + threadIdx.x;
        uint64_t x = i < n ? v[i] : 0, d = 0, tx, td;
        uint64_t e = x;
        block_exscan2(e, d, tx, td, sc);
        if (i < n) v[i] = carry + e;
        __syncthreads();
        if (threadIdx.x == 0) carry += tx;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void k_synth_tail(SynParams p, char* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { SynMemSink m{out}; synth_tail_group(m, p); }
}

hipError_t launch_synth_sizes(const SynParams& p, uint64_t first, uint64_t n, uint64_t* sizes, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_synth_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, first, n, sizes);
    return hipGetLastError();
}
// long arrays (line tables of the sorter / krmdup / BAM writer): per-chunk sums, the single-workgroup scan over the sums, then the
// chunks again with their offsets
constexpr uint32_t XS_PER = 16 * NT;
__global__ __launch_bounds__(NT) void k_xs_sums(const uint64_t* v, uint64_t n, uint64_t* part) {
    __shared__ uint64_t sh[NT / 64];
    const uint64_t b = (uint64_t)blockIdx.x * XS_PER;
    uint64_t x = 0;
    for (uint32_t k = 0; k < XS_PER / NT; ++k) { const uint64_t i = b + k * NT + threadIdx.x; if (i < n) x += v[i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += shfl_xor64(x, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t t = 0; for (int w = 0; w < NT / 64; ++w) t += sh[w]; part[blockIdx.x] = t; }
}
__global__ __launch_bounds__(NT) void k_xs_apply(uint64_t* v, uint64_t n, const uint64_t* part) {
    __shared__ ScanScratch sc;
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = part[blockIdx.x];
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * XS_PER;
    for (uint32_t k = 0; k < XS_PER / NT; ++k) {
        const uint64_t i = b + k * NT + threadIdx.x;
        uint64_t e = i < n ? v[i] : 0, d = 0, tx, td;
        block_exscan2(e, d, tx, td, sc);
        if (i < n) v[i] = carry + e;
        __syncthreads();
        if (threadIdx.x == 0) carry += tx;
        __syncthreads();
    }
}
hipError_t launch_exscan(uint64_t* v, uint64_t n, uint64_t* total, hipStream_t s) {
    if (n <= 4 * XS_PER) {
        hipLaunchKernelGGL(k_exscan_u64, dim3(1), dim3(NT), 0, s, v, n, total);
        return hipGetLastError();
    }
    const uint64_t nb = (n + XS_PER - 1) / XS_PER;
    uint64_t* part = nullptr;
    hipError_t e = hipMallocAsync((void**)&part, (nb + 1) * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_xs_sums, dim3((unsigned)nb), dim3(NT), 0, s, (const uint64_t*)v, n, part);
    hipLaunchKernelGGL(k_exscan_u64, dim3(1), dim3(NT), 0, s, part, nb, total);
    hipLaunchKernelGGL(k_xs_apply, dim3((unsigned)nb), dim3(NT), 0, s, v, n, (const uint64_t*)part);
    e = hipGetLastError();
    const hipError_t e2 = hipFreeAsync(part, s);
    return e != hipSuccess ? e : e2;
}
hipError_t launch_synth_write(const SynParams& p, uint64_t first, uint64_t n, const uint64_t* offs, char* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_synth_write, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, first, n, offs, out);
    return hipGetLastError();
}
hipError_t launch_synth_tail(const SynParams& p, char* out, hipStream_t s) {
    hipLaunchKernelGGL(k_synth_tail, dim3(1), dim3(64), 0, s, p, out);
    return hipGetLastError();
}
size_t synth_tail_bytes(const SynParams& p) {
    SynCountSink c;
    synth_tail_group(c, p);
    return c.n;
}

}  // namespace mkt
