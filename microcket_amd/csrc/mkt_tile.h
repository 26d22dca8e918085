// mkt_tile.h -- the per-tile algorithm of the fused sam2pairs kernel.
//
// A block of SAM text (< 2 GiB, starting on a QNAME-group boundary) is cut into fixed-size byte
// tiles.  One workgroup owns one tile: it stages the tile plus a back halo and a forward halo in
// LDS, finds the line starts, parses the six leading fields of every line in the window, decides
// which surviving lines open a QNAME group (comparison with the previous surviving line, as
// pairutil.h:163 does), and classifies every group whose FIRST line starts inside the tile.  Bytes
// outside the window are reachable through TextView::at() (global memory), so halo sizes only
// affect speed, never results: lines or groups that leave the window take the cold loops below.
//
// The phases are plain functions over TileState so that the same code runs
//   * in the HIP kernel, one work item per lane with barriers between phases (mkt_kernels.hip);
//   * serially in the host emulation the CPU tests use to check this logic (tests/host/).
//
// Phases:  [load window + bitmaps] -> [line table] -> ph_parse (fields, CIGAR, "same QNAME as the
// previous line") -> ph_group (group-start test, member walk, classification) -> tile sums ->
// output offsets (atomic ranges, or ordered look-back) -> ph_account / per-byte .pairs / .sam copy.
#pragma once
#include "mkt_core.h"

namespace mkt {

template <int TILE_, int HB_, int HF_, int LCAP_, int OVF_>
struct TileCfg {
    static constexpr int TILE = TILE_;   // bytes owned by one workgroup
    static constexpr int HB = HB_;       // back halo (previous surviving line's QNAME)
    static constexpr int HF = HF_;       // forward halo (rest of the last group)
    static constexpr int W = HB_ + TILE_ + HF_;
    static constexpr int LCAP = LCAP_;   // line-table capacity (lines starting in the window)
    static constexpr int OVF = OVF_;     // kept records parsed beyond the window (one group: <= 4)
    static constexpr int RCAP = LCAP_ + OVF_;
    static constexpr int MW = (W + 63) / 64 + 3;   // bitmap words incl. zero padding
    static_assert(TILE_ % 16 == 0 && HB_ % 16 == 0 && HF_ % 16 == 0, "16-byte vector staging");
    static_assert(LCAP_ + OVF_ < 0xFFFF, "record indices are 16 bit");
};

enum ErrBits : uint32_t {
    E_LINE_TABLE = 1,     // more line starts in a window than LCAP: rerun with the small-tile config
    E_OVF_SLOTS = 2,      // more kept out-of-window records than OVF (cannot happen with OVF >= 4)
    E_LOOKBACK = 4,       // ordered mode: decoupled look-back spin bound hit (never expected)
    E_PAIRS_CAP = 8,      // .pairs output buffer too small (sizes in the result are still exact)
    E_SAM_CAP = 16,       // .sam output buffer too small
    E_SC_CAP = 32,        // self-circle index buffer too small
    E_FIELD_RANGE = 64,   // a QNAME/RNAME longer than 65535 bytes was dropped
};

struct TileGeom { uint32_t t0, t1, w0, w1; };
// Tile geometry of a block, chosen per input at run time (multiples of 16 bytes): the kernels are compiled for CAPACITIES (window
// bytes, lines per window), the host picks the bytes per tile so that a window holds about as many lines as the phases have lanes
struct TileDims { uint32_t tile, hb, hf; };
MKT_HD TileGeom tile_geom(uint32_t tile, uint32_t n, const TileDims& d) {
    TileGeom G;
    uint64_t t0 = (uint64_t)tile * d.tile;
    uint64_t t1 = t0 + d.tile;
    G.t0 = (uint32_t)t0;
    G.t1 = t1 < n ? (uint32_t)t1 : n;
    G.w0 = G.t0 >= d.hb ? G.t0 - d.hb : 0u;
    uint64_t w1 = (uint64_t)G.t1 + d.hf;
    G.w1 = w1 < n ? (uint32_t)w1 : n;
    return G;
}
template <class Cfg> MKT_HD TileDims cfg_dims() { TileDims d; d.tile = (uint32_t)Cfg::TILE; d.hb = (uint32_t)Cfg::HB; d.hf = (uint32_t)Cfg::HF; return d; }
template <class Cfg> MKT_HD TileGeom tile_geom(uint32_t tile, uint32_t n) { return tile_geom(tile, n, cfg_dims<Cfg>()); }
MKT_HD uint32_t num_tiles(uint32_t n, uint32_t tile_bytes) { return n == 0 ? 0u : (n + tile_bytes - 1) / tile_bytes; }

// line bits
constexpr uint8_t LB_SURVIVE = 1, LB_EQPREV = 2, LB_CUT = 4;   // EQPREV: same QNAME token as the line before (table index - 1);
                                                               // CUT: last halo line, fields cut by the window end (not parsed)
// per-group info word
constexpr uint32_t GI_COUNTER = 0xF, GI_EMIT = 1u << 4, GI_CONTIG = 1u << 5, GI_SA_MINUS = 1u << 6, GI_SB_MINUS = 1u << 7, GI_START = 1u << 8;

// What one tile contributes to the block-wide sums.
struct TileSums {
    uint32_t groups, emitted, sc, pair_bytes;
    uint64_t sam_bytes;
};

struct TileLast {          // the tile's last group (quirk Q1 bookkeeping on the host)
    uint32_t counter, pair_bytes, sam_bytes, valid;
    uint32_t pair_off, sam_off;      // where its bytes sit inside its output region
    uint32_t region, pad;
};

// LDS-resident state of one tile.
template <class Cfg>
struct TileState {
    alignas(16) uint8_t win[Cfg::W + 16];          // + zero pad for the dword loads
    // records: [0, LCAP) lines of the window, [LCAP, RCAP) kept records parsed beyond it
    uint32_t off[Cfg::RCAP];
    uint32_t pos[Cfg::RCAP];
    int32_t lclip[Cfg::RCAP], rclip[Cfg::RCAP], mappable[Cfg::RCAP];
    int32_t right0[Cfg::RCAP], left1[Cfg::RCAP], right1[Cfg::RCAP];     // left0 == pos
    uint16_t qn_off[Cfg::RCAP], qn_len[Cfg::RCAP], rn_off[Cfg::RCAP], rn_len[Cfg::RCAP];
    uint16_t flag[Cfg::RCAP];
    uint8_t segCnt[Cfg::RCAP];
    uint8_t bits[Cfg::LCAP];             // LB_*
    union alignas(16) Phase {
        struct {                         // while parsing: bitmaps over the window (bit r <-> byte r)
            uint64_t nlm[Cfg::MW];       // newline
            uint64_t wsm[Cfg::MW];       // whitespace
        } m;
        struct {                         // afterwards: group results (indexed by the group's first line) + tile-local sums
            uint32_t g_info[Cfg::LCAP], g_posA[Cfg::LCAP], g_posB[Cfg::LCAP];
            uint32_t g_chrA[Cfg::LCAP], g_chrB[Cfg::LCAP];
            uint16_t g_chrA_len[Cfg::LCAP], g_chrB_len[Cfg::LCAP];
            uint32_t g_plen[Cfg::LCAP], g_slen[Cfg::LCAP], g_last_end[Cfg::LCAP];
            uint32_t x_pair[Cfg::LCAP], x_sam[Cfg::LCAP];
            uint16_t x_grp[Cfg::LCAP], x_sc[Cfg::LCAP], x_emit[Cfg::LCAP];
            uint16_t em_idx[Cfg::LCAP];  // emit ordinal -> first line of the emitting group
        } g;
    } u;
    // scalars
    uint32_t NL, first_idx, end_idx, ovf_n, err, last_line_end;
    uint32_t cnt[C_COUNT];
    TileSums sums;            // this tile's totals
    TileSums base;            // where this tile's outputs start: ABSOLUTE positions in OutPtrs::pairs / sam / sc
    uint32_t region_pair0, region_sam0, region_id;   // the tile's output region (TileLast offsets are region relative)
};

#if defined(__HIPCC__)
#define MKT_COLD __host__ __device__ inline __attribute__((noinline))
#else
#define MKT_COLD inline __attribute__((noinline))
#endif

template <class Cfg> MKT_HD TextView tile_view(const TileState<Cfg>& st, const uint8_t* text, uint32_t n, const TileGeom& G) {
    TextView tv;
    tv.g = text; tv.n = n; tv.win = st.win; tv.w0 = G.w0; tv.wlen = G.w1 - G.w0; tv.nlm = st.u.m.nlm; tv.wsm = st.u.m.wsm;
    return tv;
}

template <class Cfg> MKT_HD void tile_reset(TileState<Cfg>& st) {
    st.NL = 0; st.first_idx = 0; st.end_idx = 0; st.ovf_n = 0; st.err = 0; st.last_line_end = kUnknown;
    st.region_pair0 = 0; st.region_sam0 = 0; st.region_id = 0;
    for (int k = 0; k < (int)C_COUNT; ++k) st.cnt[k] = 0;
}

// LDS read-modify-write helpers (atomics in the kernels, plain in the serial host emulation)
#if defined(__HIP_DEVICE_COMPILE__)
MKT_HD uint32_t lds_inc(uint32_t* p) { return atomicAdd(p, 1u); }
MKT_HD void lds_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
MKT_HD void lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
MKT_HD void lds_or64(uint64_t* p, uint64_t v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
#else
MKT_HD uint32_t lds_inc(uint32_t* p) { return (*p)++; }
MKT_HD void lds_add(uint32_t* p, uint32_t v) { *p += v; }
MKT_HD void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
MKT_HD void lds_or64(uint64_t* p, uint64_t v) { *p |= v; }
#endif

MKT_HD bool rec_in_range(const Rec& r) { return r.qn_off <= 0xFFFFu && r.qn_len <= 0xFFFFu && r.rn_off <= 0xFFFFu && r.rn_len <= 0xFFFFu; }

template <class Cfg> MKT_HD bool store_rec(TileState<Cfg>& st, uint32_t idx, const Rec& r) {
    bool s = r.survive;
    if (!rec_in_range(r)) {
        if (s) lds_or(&st.err, E_FIELD_RANGE);
        s = false;
    }
    st.off[idx] = r.off; st.pos[idx] = r.pos;
    st.lclip[idx] = r.lclip; st.rclip[idx] = r.rclip; st.mappable[idx] = r.mappable;
    st.right0[idx] = r.right0; st.left1[idx] = r.left1; st.right1[idx] = r.right1;
    st.qn_off[idx] = (uint16_t)r.qn_off; st.qn_len[idx] = (uint16_t)r.qn_len;
    st.rn_off[idx] = (uint16_t)r.rn_off; st.rn_len[idx] = (uint16_t)r.rn_len;
    st.flag[idx] = (uint16_t)(r.flag & 0xFFFFu);
    st.segCnt[idx] = (uint8_t)(r.segCnt > 4 ? 4 : r.segCnt);
    return s;
}
template <class Cfg> MKT_HD Seg load_seg(const TileState<Cfg>& st, uint32_t idx) {
    Seg s;
    s.segCnt = st.segCnt[idx]; s.lclip = st.lclip[idx]; s.rclip = st.rclip[idx]; s.mappable = st.mappable[idx];
    s.left0 = (int32_t)st.pos[idx]; s.left1 = st.left1[idx]; s.right0 = st.right0[idx]; s.right1 = st.right1[idx];
    s.rightLast = s.segCnt == 2 ? s.right1 : s.right0;       // right[segCnt-1]; only read when segCnt is 1 or 2
    s.flag = st.flag[idx]; s.pos = st.pos[idx];
    s.chr_off = st.off[idx] + st.rn_off[idx]; s.chr_len = st.rn_len[idx]; s.chr_key = 0;
    return s;
}
MKT_HD Seg seg_zero() {
    Seg s;
    s.segCnt = s.lclip = s.rclip = s.mappable = s.left0 = s.left1 = s.right0 = s.right1 = s.rightLast = 0;
    s.flag = s.pos = s.chr_off = s.chr_len = 0; s.chr_key = 0;
    return s;
}

MKT_HD uint32_t find_newline(const TextView& tv, uint32_t from) {     // offset of '\n' at/after `from`, or n
    uint32_t p = from;
    while (p < tv.n && tv.at(p) != '\n') ++p;
    return p;
}
// offset of line i's '\n' (or n); kUnknown when it lies past the window
template <class Cfg> MKT_HD uint32_t line_end(const TileState<Cfg>& st, uint32_t i) {
    return i + 1 < st.NL ? st.off[i + 1] - 1u : st.last_line_end;
}

// ---- phase: parse line i of the window -------------------------------------------------------
// first whitespace-delimited token of the line starting at `off` (generic; used for odd lines)
MKT_COLD void first_token(const TextView& tv, uint32_t off, uint32_t* ts, uint32_t* len) {
    uint32_t p = off;
    while (p < tv.n) { uint8_t c = tv.at(p); if (c == '\n' || !is_ws(c)) break; ++p; }
    *ts = p;
    while (p < tv.n) { uint8_t c = tv.at(p); if (c == '\n' || is_ws(c)) break; ++p; }
    *len = p - *ts;
}

template <class Cfg> MKT_HD void ph_parse(TileState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G, uint32_t i) {
    const uint32_t off = st.off[i];
    if (i + 1 == st.NL) {
        // last line of the table: a '\n' before the window's last byte would have opened another line
        uint32_t e;
        if (G.w1 - 1 >= off && tv.win[G.w1 - 1 - G.w0] == '\n') e = G.w1 - 1;
        else e = G.w1 >= tv.n ? tv.n : kUnknown;
        st.last_line_end = e;
    }
    Rec r;
    const int pf = parse_record_fast(tv, off, P, r);
    if (pf == PF_CUT && off >= G.t1) {
        // a halo line whose fields run past the window end: it is only ever needed by a group that
        // spans the whole forward halo; leave it to the out-of-window walk (group_walk_beyond)
        st.bits[i] = LB_CUT;
        return;
    }
    if (pf != PF_OK) r = parse_record(tv, off, P);
    uint8_t b = store_rec(st, i, r) ? LB_SURVIVE : 0;
    // same QNAME token as the line before?  (table lines are adjacent lines of the text)
    if (i > 0 && r.qn_len > 0) {
        const uint32_t poff = st.off[i - 1];
        const uint32_t qa = off + r.qn_off, ql = r.qn_len;
        bool eq;
        if (!is_ws(tv.at(poff)) && tv.inside(poff, ql + 1u)) {
            // the previous line's first token starts at its first byte: equal iff the bytes match
            // and the token ends right after them
            eq = text_eq(tv, qa, ql, poff, ql) && is_ws(tv.at(poff + ql));
        } else {
            uint32_t pts, plen;
            first_token(tv, poff, &pts, &plen);
            eq = text_eq(tv, qa, ql, pts, plen);
        }
        if (eq) b |= LB_EQPREV;
    }
    st.bits[i] = b;
    if (off >= G.t0 && (i == 0 || st.off[i - 1] < G.t0)) st.first_idx = i;
    if (off >= G.t1 && (i == 0 || st.off[i - 1] < G.t1)) st.end_idx = i;
}

// After ph_parse (one thread): a trailing LB_CUT line leaves the table; the line before it becomes
// the last one and its '\n' is the byte in front of the dropped line.
template <class Cfg> MKT_HD void tile_trim(TileState<Cfg>& st) {
    if (st.NL && (st.bits[st.NL - 1] & LB_CUT)) {
        st.last_line_end = st.off[st.NL - 1] - 1u;
        st.NL -= 1;
        if (st.end_idx > st.NL) st.end_idx = st.NL;
        if (st.first_idx > st.NL) st.first_idx = st.NL;
    }
}

// start of the line that contains block offset e (the byte after the last '\n' before e, or 0): walks back through
// global memory 16 aligned bytes at a time (the block base is 16-byte aligned)
MKT_HD uint32_t prev_line_start(const TextView& tv, uint32_t e) {
    uint32_t lim = e;                                  // bytes [.., lim) are searched
    while (lim > 0) {
        const uint32_t b = (lim - 1u) & ~15u;
        uint64_t w[2] = {0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_memcpy(w, __builtin_assume_aligned(tv.g + b, 16), 16);      // readable up to the next multiple of 16 (include/mkt.h)
#else
        __builtin_memcpy(w, tv.g + b, tv.n - b < 16u ? tv.n - b : 16u);
#endif
        for (int h = 1; h >= 0; --h) {
            const uint32_t base = b + 8u * (uint32_t)h;
            if (base >= lim) continue;
            const uint64_t x = w[h] ^ 0x0A0A0A0A0A0A0A0Aull;
            uint64_t z = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);   // 0x80 where byte == '\n'
            const uint32_t keep = lim - base;          // bytes of this word below lim (1..8, or more)
            if (keep < 8u) z &= (1ull << (8u * keep)) - 1ull;
            if (z) return base + (63u - clz64(z)) / 8u + 1u;
        }
        lim = b;
    }
    return 0;
}

// ---- does surviving line i open a group?  (pairutil.h:163: QNAME != the previous SURVIVING line's)
template <class Cfg> MKT_COLD bool start_vs_global(const TileState<Cfg>& st, const TextView& tv, const Params& P, uint32_t i) {
    // no surviving line precedes line i inside the window: walk back through global memory
    uint32_t q = st.off[0];
    const uint32_t qa = st.off[i] + st.qn_off[i], ql = st.qn_len[i];
    while (q > 0) {
        const uint32_t ls = prev_line_start(tv, q - 1);   // q - 1: the '\n' ending the previous line
        // the line's head comes over in eleven independent 16-byte loads and is parsed from that private copy
        // (the byte-at-a-time parser on global memory is a chain of dependent loads)
        alignas(16) uint8_t head[176];
        TextView hv = tv;
        hv.win = head; hv.w0 = ls & ~15u;
        hv.wlen = tv.n - hv.w0 < 176u ? tv.n - hv.w0 : 176u;
#if defined(__HIP_DEVICE_COMPILE__)
        for (uint32_t k = 0; k < 176u; k += 16u)
            if (hv.w0 + k < tv.n) __builtin_memcpy(head + k, __builtin_assume_aligned(tv.g + hv.w0 + k, 16), 16);
#else
        __builtin_memcpy(head, tv.g + hv.w0, hv.wlen);
#endif
        Rec r = parse_record(hv, ls, P);
        if (r.survive && rec_in_range(r)) {
            if (r.qn_len != ql) return true;
            for (uint32_t k = 0; k < ql; ++k) if (tv.at(qa + k) != hv.at(ls + r.qn_off + k)) return true;
            return false;
        }
        q = ls;
    }
    return true;                                      // first surviving line of the block
}
template <class Cfg> MKT_HD bool is_start(const TileState<Cfg>& st, const TextView& tv, const Params& P, uint32_t i) {
    // precondition: line i survives
    // usual case: the line before survives, and "same QNAME token as the line before" was settled while parsing
    if (i > 0 && (st.bits[i - 1] & LB_SURVIVE)) return !(st.bits[i] & LB_EQPREV);
    bool chain = true;                                // every line in (j, i] has the QNAME of its predecessor
    uint32_t j = i;
    for (;;) {
        if (j == 0) return start_vs_global(st, tv, P, i);
        chain = chain && (st.bits[j] & LB_EQPREV);
        --j;
        if (st.bits[j] & LB_SURVIVE) break;
    }
    if (chain) return false;                          // equal by transitivity
    // some line in between has another name: compare with the previous surviving line itself
    return !text_eq(tv, st.off[i] + st.qn_off[i], st.qn_len[i], st.off[j] + st.qn_off[j], st.qn_len[j]);
}

// ---- phase: walk the group opened by line i, classify it ---------------------------------------

struct GroupWalk {
    uint32_t nmem, n1, n2;
    uint32_t sa, sb, sc, sd;        // unc: r1a r1b r2a r2b; flash: a b (record indices, 0xFFFF = none)
    bool gap, contig;
    uint32_t last_end;              // one past the '\n' of the last member
    uint64_t sam_bytes;
    MKT_HD int wants(int mode, uint32_t f) const {
        if (mode == MODE_FLASH) return nmem < 2 ? (int)nmem : -1;
        if (f & 64u) return n1 < 2 ? (int)n1 : -1;
        if (f & 128u) return n2 < 2 ? 2 + (int)n2 : -1;
        return -1;
    }
    MKT_HD void take(int mode, uint32_t n, uint32_t idx, uint32_t f, uint32_t loff, uint32_t lend) {
        const int slot = wants(mode, f);
        if (slot == 0) sa = idx; else if (slot == 1) sb = idx; else if (slot == 2) sc = idx; else if (slot == 3) sd = idx;
        if (mode == MODE_UNC) { if (f & 64u) ++n1; else if (f & 128u) ++n2; }
        ++nmem;
        if (gap) contig = false;
        sam_bytes += (uint64_t)(lend - loff) + 1u;
        if (lend >= n) contig = false;                // last line without '\n': the copy adds one
        last_end = lend < n ? lend + 1 : n;
    }
};

// the group may continue past the window: walk line by line through global memory
template <class Cfg> MKT_COLD void group_walk_beyond(TileState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G,
                                                     uint32_t qa, uint32_t ql, GroupWalk& w) {
    uint32_t le = st.last_line_end;
    if (le == kUnknown) le = find_newline(tv, G.w1);
    uint32_t q = le + 1;
    while (q < tv.n) {
        Rec r = parse_record(tv, q, P);
        const uint32_t e = find_newline(tv, q);
        if (r.survive && rec_in_range(r)) {
            if (!text_eq(tv, qa, ql, q + r.qn_off, r.qn_len)) break;
            uint32_t idx = 0xFFFFu;
            if (w.wants(P.mode, r.flag) >= 0) {
                uint32_t k = lds_inc(&st.ovf_n);
                if (k < (uint32_t)Cfg::OVF) { idx = Cfg::LCAP + k; store_rec(st, idx, r); }
                else lds_or(&st.err, E_OVF_SLOTS);
            }
            w.take(P.mode, tv.n, idx, r.flag, q, e);
        } else w.gap = true;
        q = e + 1;
    }
}

template <class Cfg> MKT_HD void ph_group(TileState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G, uint32_t i) {
    auto& g = st.u.g;
    g.g_info[i] = 0; g.g_plen[i] = 0; g.g_slen[i] = 0;
    if (!(st.bits[i] & LB_SURVIVE) || !is_start(st, tv, P, i)) return;
    const uint32_t qa = st.off[i] + st.qn_off[i], ql = st.qn_len[i];
    GroupWalk w;
    w.nmem = w.n1 = w.n2 = 0; w.sa = w.sb = w.sc = w.sd = 0xFFFFu;
    w.gap = false; w.contig = true; w.last_end = 0; w.sam_bytes = 0;

    uint32_t j = i;
    bool closed = false;
    while (j < st.NL) {
        const uint8_t b = st.bits[j];
        if (b & LB_SURVIVE) {
            // a surviving line whose QNAME differs from the previous surviving line's closes the group
            if (j > i) {
                bool same;
                if ((b & LB_EQPREV) && (st.bits[j - 1] & LB_SURVIVE)) same = true;     // the usual case
                else same = !is_start(st, tv, P, j);
                if (!same) { closed = true; break; }
            }
            uint32_t e = line_end(st, j);
            if (e == kUnknown) e = find_newline(tv, G.w1);
            w.take(P.mode, tv.n, j, st.flag[j], st.off[j], e);
        } else w.gap = true;
        ++j;
    }
    if (!closed && G.w1 < tv.n) group_walk_beyond(st, tv, P, G, qa, ql, w);

    Verdict v;
    {
        Seg a = w.sa != 0xFFFFu ? load_seg(st, w.sa) : seg_zero();
        Seg b = w.sb != 0xFFFFu ? load_seg(st, w.sb) : seg_zero();
        if (P.mode == MODE_FLASH) v = classify_flash(tv, w.nmem, a, b, P.ratio);
        else {
            Seg c = w.sc != 0xFFFFu ? load_seg(st, w.sc) : seg_zero();
            Seg d = w.sd != 0xFFFFu ? load_seg(st, w.sd) : seg_zero();
            v = classify_unc(tv, w.n1, w.n2, a, b, c, d, P.ratio);
        }
    }
    uint32_t info = GI_START | (v.counter & GI_COUNTER);
    if (v.emit) {
        info |= GI_EMIT;
        if (w.contig) info |= GI_CONTIG;
        if (v.sA == '-') info |= GI_SA_MINUS;
        if (v.sB == '-') info |= GI_SB_MINUS;
        g.g_posA[i] = v.posA; g.g_posB[i] = v.posB;
        g.g_chrA[i] = v.chrA_off; g.g_chrB[i] = v.chrB_off;
        g.g_chrA_len[i] = (uint16_t)v.chrA_len; g.g_chrB_len[i] = (uint16_t)v.chrB_len;
        g.g_plen[i] = pair_line_len(ql, v);
        g.g_slen[i] = P.write_sam ? (uint32_t)w.sam_bytes : 0u;
        g.g_last_end[i] = w.last_end;
    }
    g.g_info[i] = info;
}

// ---- phase: account for / emit the group opened by line i ---------------------------------------
// Slot of a chromosome name in the run's table (claims an empty slot with one 64-bit CAS; no payload is ever read
// back on the device, so no fences are needed).  Two names are the same chromosome iff their 64-bit FNV-1a agree.
#if defined(__HIP_DEVICE_COMPILE__)
MKT_HD uint32_t chr_slot_h(ChrTab* tab, const TextView& tv, uint32_t off, uint32_t len, uint64_t h, uint32_t* err) {
    uint32_t s = (uint32_t)(h >> 17) & (kChrSlots - 1u);
    for (uint32_t probe = 0; probe < kChrSlots; ++probe) {
        unsigned long long cur = __hip_atomic_load(&tab->hash[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            cur = atomicCAS(&tab->hash[s], 0ull, (unsigned long long)h);
            if (cur == 0ull) {                         // claimed: leave the name for the host
                const uint32_t m = len < kChrNameMax ? len : kChrNameMax;
                for (uint32_t i = 0; i < m; ++i) tab->name[s][i] = tv.at(off + i);
                tab->name[s][63] = (uint8_t)m;
                if (len > kChrNameMax) atomicOr(err, (uint32_t)E_FIELD_RANGE);
                return s;
            }
        }
        if (cur == h) return s;
        s = (s + 1u) & (kChrSlots - 1u);
    }
    atomicOr(err, (uint32_t)E_FIELD_RANGE);
    return 0;
}
#else
MKT_HD uint32_t chr_slot_h(ChrTab* tab, const TextView& tv, uint32_t off, uint32_t len, uint64_t h, uint32_t* err) {
    uint32_t s = (uint32_t)(h >> 17) & (kChrSlots - 1u);
    for (uint32_t probe = 0; probe < kChrSlots; ++probe) {
        if (tab->hash[s] == 0ull) {
            tab->hash[s] = h;
            const uint32_t m = len < kChrNameMax ? len : kChrNameMax;
            for (uint32_t i = 0; i < m; ++i) tab->name[s][i] = tv.at(off + i);
            tab->name[s][63] = (uint8_t)m;
            if (len > kChrNameMax) *err |= E_FIELD_RANGE;
            return s;
        }
        if (tab->hash[s] == h) return s;
        s = (s + 1u) & (kChrSlots - 1u);
    }
    *err |= E_FIELD_RANGE;
    return 0;
}
#endif
MKT_HD uint32_t chr_slot(ChrTab* tab, const TextView& tv, uint32_t off, uint32_t len, uint32_t* err) {
    return chr_slot_h(tab, tv, off, len, fnv1a64(tv, off, len), err);
}
// The sequencing lane of a read: the fourth ':'-separated field of an Illumina-style QNAME (instrument:run:flowcell:LANE:tile:x:y),
// decimal; 0 when the name has no such field.  With MKT_EXT_LANES it joins the duplicate-marking key, which is the driver's
// -b ("inter-lane duplicates are kept": it runs one krmdup per lane, microcket:421-451).
MKT_HD uint32_t qname_lane(const TextView& tv, uint32_t off, uint32_t len) {
    uint32_t colons = 0, p = 0;
    while (p < len && colons < 3u) { if (tv.at(off + p) == ':') ++colons; ++p; }
    if (colons < 3u) return 0u;
    uint32_t v = 0, nd = 0;
    while (p < len) {
        const uint32_t d = (uint32_t)tv.at(off + p) - (uint32_t)'0';
        if (d > 9u) break;
        v = v * 10u + d; ++nd; ++p;
        if (v > 0xFFFFu) return 0xFFFFu;
    }
    return nd ? v : 0u;
}
MKT_HD KeyRec make_key(uint32_t slotA, uint32_t posA, uint32_t slotB, uint32_t posB, bool minusA, bool minusB, uint32_t tile, uint32_t ordinal, uint32_t lane = 0) {
    KeyRec k;
    k.k0 = ((uint64_t)slotA << 45) | ((uint64_t)slotB << 32) | posA;
    k.k1 = ((uint64_t)posB << 32) | ((uint64_t)(minusA ? 1u : 0u) << 31) | ((uint64_t)(minusB ? 1u : 0u) << 30) | (uint64_t)(lane & 0xFFFFu);
    k.ord = ((uint64_t)tile << 16) | ordinal;
    return k;
}

struct OutPtrs {                        // *_cap: end of the range this tile may write (its output region)
    uint8_t* pairs; uint64_t pairs_cap;
    uint8_t* sam; uint64_t sam_cap;
    uint64_t* sc; uint64_t sc_cap;      // self-circle groups: (tile << 32 | ordinal in tile), resolved to global indices by k_finish
    KeyRec* keys; uint64_t keys_cap;    // extension: raw key records of the block (null: extension off)
    ChrTab* chr;
    uint32_t key_lanes, pad_;           // extension MKT_EXT_LANES: the read's lane (QNAME field 4) is part of the key
};

// slow .sam copy: surviving lines of [first line, last member end), each followed by '\n'
template <class Cfg> MKT_COLD void sam_copy_slow(TileState<Cfg>& st, const TextView& tv, const Params& P, const OutPtrs& out, uint32_t i) {
    uint64_t go = st.base.sam_bytes + st.u.g.x_sam[i];
    if (go + st.u.g.g_slen[i] > out.sam_cap) { lds_or(&st.err, E_SAM_CAP); return; }
    uint32_t q = st.off[i];
    const uint32_t stop = st.u.g.g_last_end[i];
    while (q < stop) {
        Rec r = parse_record(tv, q, P);
        const uint32_t e = find_newline(tv, q);
        if (r.survive && rec_in_range(r)) {
            for (uint32_t p = q; p < e; ++p) out.sam[go++] = tv.at(p);
            out.sam[go++] = '\n';
        }
        q = e + 1;
    }
}

template <class Cfg> MKT_HD void ph_account(TileState<Cfg>& st, const TextView& tv, const Params& P, const OutPtrs& out, uint32_t tile, uint32_t i) {
    const uint32_t info = st.u.g.g_info[i];
    if (!(info & GI_START)) return;
    const uint32_t counter = info & GI_COUNTER;
    if (counter) lds_add(&st.cnt[counter], 1u);
    if (counter == C_SELFCIRCLE) {
        uint64_t k = (uint64_t)st.base.sc + st.u.g.x_sc[i];
        if (k < out.sc_cap) out.sc[k] = ((uint64_t)tile << 32) | st.u.g.x_grp[i];
        else lds_or(&st.err, E_SC_CAP);
    }
    if ((info & GI_EMIT) && out.keys) {
        const auto& g = st.u.g;
        const uint64_t k = (uint64_t)st.base.emitted + g.x_emit[i];
        if (k < out.keys_cap) {
            const uint32_t sa = chr_slot(out.chr, tv, g.g_chrA[i], g.g_chrA_len[i], &st.err);
            const uint32_t sb = chr_slot(out.chr, tv, g.g_chrB[i], g.g_chrB_len[i], &st.err);
            const uint32_t lane = out.key_lanes ? qname_lane(tv, st.off[i] + st.qn_off[i], st.qn_len[i]) : 0u;
            out.keys[k] = make_key(sa, g.g_posA[i], sb, g.g_posB[i], (info & GI_SA_MINUS) != 0, (info & GI_SB_MINUS) != 0, tile, g.x_emit[i], lane);
        } else lds_or(&st.err, E_SC_CAP);
    }
    if ((info & GI_EMIT) && P.write_sam && !(info & GI_CONTIG)) sam_copy_slow(st, tv, P, out, i);
}

// byte k (0 <= k < sums.pair_bytes) of the tile's .pairs output
template <class Cfg> MKT_HD uint8_t tile_pair_byte(const TileState<Cfg>& st, const TextView& tv, uint32_t k) {
    const auto& g = st.u.g;
    uint32_t lo = 0, hi = st.sums.emitted;            // largest ordinal whose offset is <= k
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.x_pair[g.em_idx[mid]] <= k) lo = mid; else hi = mid;
    }
    const uint32_t i = g.em_idx[lo];
    const uint32_t info = g.g_info[i];
    Verdict v;
    v.counter = info & GI_COUNTER; v.emit = true;
    v.chrA_off = g.g_chrA[i]; v.chrA_len = g.g_chrA_len[i]; v.chrB_off = g.g_chrB[i]; v.chrB_len = g.g_chrB_len[i];
    v.posA = g.g_posA[i]; v.posB = g.g_posB[i];
    v.sA = (info & GI_SA_MINUS) ? '-' : '+'; v.sB = (info & GI_SB_MINUS) ? '-' : '+';
    return pair_line_byte(tv, st.off[i] + st.qn_off[i], st.qn_len[i], v, k - g.x_pair[i]);
}

// the whole .pairs line of the e-th reported pair of the tile, by ONE lane (the same bytes as tile_pair_byte, without a search and a
// field walk per byte: the lanes of a workgroup write the tile's lines side by side)
template <class Cfg> MKT_HD void tile_emit_line(const TileState<Cfg>& st, const TextView& tv, uint32_t e, uint8_t* tile_out) {
    const auto& g = st.u.g;
    const uint32_t i = g.em_idx[e];
    const uint32_t info = g.g_info[i];
    Verdict v;
    v.counter = info & GI_COUNTER; v.emit = true;
    v.chrA_off = g.g_chrA[i]; v.chrA_len = g.g_chrA_len[i]; v.chrB_off = g.g_chrB[i]; v.chrB_len = g.g_chrB_len[i];
    v.posA = g.g_posA[i]; v.posB = g.g_posB[i];
    v.sA = (info & GI_SA_MINUS) ? '-' : '+'; v.sB = (info & GI_SB_MINUS) ? '-' : '+';
    const uint32_t qn = st.off[i] + st.qn_off[i], ql = st.qn_len[i], plen = g.g_plen[i];
    uint8_t* d = tile_out + g.x_pair[i];
    for (uint32_t k = 0; k < plen; ++k) *MKT_GLOBAL(uint8_t, d + k) = pair_line_byte(tv, qn, ql, v, k);
}

// the tile's last group, for the host's Q1 bookkeeping
template <class Cfg> MKT_HD void ph_last(const TileState<Cfg>& st, TileLast* tl, uint32_t i) {
    const auto& g = st.u.g;
    const uint32_t info = g.g_info[i];
    if (!(info & GI_START)) return;
    if ((uint32_t)g.x_grp[i] + 1u != st.sums.groups) return;
    tl->counter = info & GI_COUNTER;
    tl->pair_bytes = g.g_plen[i];
    tl->sam_bytes = g.g_slen[i];
    tl->pair_off = st.base.pair_bytes - st.region_pair0 + g.x_pair[i];
    tl->sam_off = (uint32_t)(st.base.sam_bytes - st.region_sam0 + g.x_sam[i]);
    tl->region = st.region_id; tl->pad = 0;
    tl->valid = 1;
}

}  // namespace mkt
