// mkt_tile.h -- the per-tile algorithm of the fused sam2pairs kernel.
//
// A block of SAM text (< 2 GiB, starting on a QNAME-group boundary) is cut into fixed-size byte
// tiles.  One workgroup owns one tile: it stages the tile plus a back halo and a forward halo in
// LDS, finds the line starts, parses the six leading fields of every line in the window, decides
// which surviving lines open a QNAME group (comparison with the previous surviving line, as
// pairutil.h:163 does), and classifies every group whose FIRST line starts inside the tile.  Bytes
// outside the window are reachable through TextView::at() (global memory), so halo sizes only
// affect speed, never results: lines or groups that leave the window take the slow loops below.
//
// The phases are plain functions over TileState so that the same code runs
//   * in the HIP kernel, one work item per lane with barriers between phases (mkt_kernels.hip);
//   * serially in the host emulation the CPU tests use to check this logic (tests/host/).
#pragma once
#include "mkt_core.h"

namespace mkt {

template <int TILE_, int HB_, int HF_, int LCAP_, int OVF_, int STG_>
struct TileCfg {
    static constexpr int TILE = TILE_;   // bytes owned by one workgroup
    static constexpr int HB = HB_;       // back halo (previous surviving line's QNAME)
    static constexpr int HF = HF_;       // forward halo (rest of the last group)
    static constexpr int W = HB_ + TILE_ + HF_;
    static constexpr int LCAP = LCAP_;   // line-table capacity (lines starting in the window)
    static constexpr int OVF = OVF_;     // records parsed beyond the window that must be kept
    static constexpr int RCAP = LCAP_ + OVF_;
    static constexpr int STG = STG_;     // LDS staging for the tile's .pairs bytes
    static_assert(TILE_ % 16 == 0 && HB_ % 16 == 0 && HF_ % 16 == 0, "16-byte vector staging");
};

enum ErrBits : uint32_t {
    E_LINE_TABLE = 1,     // more line starts in a window than LCAP: rerun with the small-tile config
    E_OVF_SLOTS = 2,      // more kept out-of-window records than OVF: rerun with the small-tile config
    E_LOOKBACK = 4,       // decoupled look-back spin bound hit (never expected)
    E_PAIRS_CAP = 8,      // .pairs output buffer too small (sizes in the result are still exact)
    E_SAM_CAP = 16,       // .sam output buffer too small
    E_SC_CAP = 32,        // self-circle index buffer too small
    E_FIELD_RANGE = 64,   // a QNAME/RNAME longer than 65535 bytes was dropped
};

struct TileGeom { uint32_t t0, t1, w0, w1; };
template <class Cfg> MKT_HD TileGeom tile_geom(uint32_t tile, uint32_t n) {
    TileGeom G;
    uint64_t t0 = (uint64_t)tile * Cfg::TILE;
    uint64_t t1 = t0 + Cfg::TILE;
    G.t0 = (uint32_t)t0;
    G.t1 = t1 < n ? (uint32_t)t1 : n;
    G.w0 = G.t0 >= (uint32_t)Cfg::HB ? G.t0 - Cfg::HB : 0u;
    uint64_t w1 = (uint64_t)G.t1 + Cfg::HF;
    G.w1 = w1 < n ? (uint32_t)w1 : n;
    return G;
}
MKT_HD uint32_t num_tiles(uint32_t n, uint32_t tile_bytes) { return n == 0 ? 0u : (n + tile_bytes - 1) / tile_bytes; }

// per-group info word
constexpr uint32_t GI_COUNTER = 0xF, GI_EMIT = 1u << 4, GI_CONTIG = 1u << 5, GI_SA_MINUS = 1u << 6, GI_SB_MINUS = 1u << 7, GI_START = 1u << 8;

// What one tile contributes to the block-wide exclusive sums.
struct TileSums {
    uint32_t groups, emitted, sc, pair_bytes;
    uint64_t sam_bytes;
};

struct TileLast {          // the tile's last group (quirk Q1 bookkeeping on the host)
    uint32_t counter, pair_bytes, sam_bytes, valid;
};

template <class Cfg>
struct TileState {
    alignas(16) uint8_t win[Cfg::W];
    alignas(16) uint8_t stg[Cfg::STG];
    // records: [0, LCAP) lines of the window, [LCAP, RCAP) kept records parsed beyond it
    uint32_t off[Cfg::RCAP];
    uint32_t pos[Cfg::RCAP];
    int32_t lclip[Cfg::RCAP], rclip[Cfg::RCAP], mappable[Cfg::RCAP];
    int32_t left0[Cfg::RCAP], left1[Cfg::RCAP], right0[Cfg::RCAP], right1[Cfg::RCAP], rightLast[Cfg::RCAP];
    uint16_t qn_off[Cfg::RCAP], qn_len[Cfg::RCAP], rn_off[Cfg::RCAP], rn_len[Cfg::RCAP];
    uint16_t flag[Cfg::RCAP];
    uint8_t segCnt[Cfg::RCAP];
    uint8_t bits[Cfg::LCAP];             // 1 = survives, 2 = opens a group
    uint32_t end[Cfg::LCAP];             // offset of the line's '\n' (or n), kUnknown if past the window
    // group results, indexed by the group's first line
    uint32_t g_info[Cfg::LCAP], g_posA[Cfg::LCAP], g_posB[Cfg::LCAP];
    uint32_t g_chrA[Cfg::LCAP], g_chrB[Cfg::LCAP];
    uint16_t g_chrA_len[Cfg::LCAP], g_chrB_len[Cfg::LCAP];
    uint32_t g_plen[Cfg::LCAP], g_slen[Cfg::LCAP], g_last_end[Cfg::LCAP];
    // exclusive sums inside the tile
    uint32_t x_pair[Cfg::LCAP], x_sam[Cfg::LCAP];
    uint16_t x_grp[Cfg::LCAP], x_emit[Cfg::LCAP], x_sc[Cfg::LCAP];
    // scalars
    uint32_t NL, first_idx, end_idx, ovf_n, err, stg_used;
    uint32_t cnt[C_COUNT];
    TileSums sums;            // this tile's totals
    TileSums base;            // exclusive prefix over earlier tiles
};

template <class Cfg> MKT_HD void tile_reset(TileState<Cfg>& st) {
    st.NL = 0; st.first_idx = 0; st.end_idx = 0; st.ovf_n = 0; st.err = 0; st.stg_used = 0;
    for (int k = 0; k < (int)C_COUNT; ++k) st.cnt[k] = 0;
}

template <class Cfg> MKT_HD void store_rec(TileState<Cfg>& st, uint32_t idx, const Rec& r, bool* survive) {
    bool s = r.survive;
    if (r.qn_off > 0xFFFFu || r.qn_len > 0xFFFFu || r.rn_off > 0xFFFFu || r.rn_len > 0xFFFFu) {
        if (s) st.err |= E_FIELD_RANGE;     // benign race: every writer ORs the same bit
        s = false;
    }
    st.off[idx] = r.off; st.pos[idx] = r.pos;
    st.lclip[idx] = r.lclip; st.rclip[idx] = r.rclip; st.mappable[idx] = r.mappable;
    st.left0[idx] = r.left0; st.left1[idx] = r.left1; st.right0[idx] = r.right0; st.right1[idx] = r.right1;
    st.rightLast[idx] = r.rightLast;
    st.qn_off[idx] = (uint16_t)r.qn_off; st.qn_len[idx] = (uint16_t)r.qn_len;
    st.rn_off[idx] = (uint16_t)r.rn_off; st.rn_len[idx] = (uint16_t)r.rn_len;
    st.flag[idx] = (uint16_t)(r.flag & 0xFFFFu);
    st.segCnt[idx] = (uint8_t)(r.segCnt > 4 ? 4 : r.segCnt);
    *survive = s;
}
template <class Cfg> MKT_HD Seg load_seg(const TileState<Cfg>& st, uint32_t idx) {
    Seg s;
    s.segCnt = st.segCnt[idx]; s.lclip = st.lclip[idx]; s.rclip = st.rclip[idx]; s.mappable = st.mappable[idx];
    s.left0 = st.left0[idx]; s.left1 = st.left1[idx]; s.right0 = st.right0[idx]; s.right1 = st.right1[idx];
    s.rightLast = st.rightLast[idx];
    s.flag = st.flag[idx]; s.pos = st.pos[idx];
    s.chr_off = st.off[idx] + st.rn_off[idx]; s.chr_len = st.rn_len[idx];
    return s;
}
MKT_HD Seg seg_zero() {
    Seg s;
    s.segCnt = s.lclip = s.rclip = s.mappable = s.left0 = s.left1 = s.right0 = s.right1 = s.rightLast = 0;
    s.flag = s.pos = s.chr_off = s.chr_len = 0;
    return s;
}

MKT_HD uint32_t find_newline(const TextView& tv, uint32_t from) {     // offset of '\n' at/after `from`, or n
    uint32_t p = from;
    while (p < tv.n && tv.at(p) != '\n') ++p;
    return p;
}

// ---- phase: parse line i of the window -------------------------------------------------------
template <class Cfg> MKT_HD void ph_parse(TileState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G, uint32_t i) {
    const uint32_t off = st.off[i];
    uint32_t e;
    if (i + 1 < st.NL) e = st.off[i + 1] - 1;
    else {
        e = kUnknown;
        for (uint32_t p = off; p < G.w1; ++p) if (tv.at(p) == '\n') { e = p; break; }
        if (e == kUnknown && G.w1 >= tv.n) e = tv.n;
    }
    st.end[i] = e;
    Rec r = parse_record(tv, off, P);
    bool s;
    store_rec(st, i, r, &s);
    st.bits[i] = s ? 1 : 0;
    if (off >= G.t0 && (i == 0 || st.off[i - 1] < G.t0)) st.first_idx = i;
    if (off >= G.t1 && (i == 0 || st.off[i - 1] < G.t1)) st.end_idx = i;
}

// ---- phase: does surviving line i open a group? ------------------------------------------------
template <class Cfg> MKT_HD bool resolve_back(const TileState<Cfg>& st, const TextView& tv, const Params& P, uint32_t i) {
    // no surviving line precedes line i inside the window: walk back through global memory
    uint32_t q = st.off[0];
    const uint32_t qa = st.off[i] + st.qn_off[i], ql = st.qn_len[i];
    while (q > 0) {
        uint32_t ls = q - 1;                          // the '\n' ending the previous line
        while (ls > 0 && tv.at(ls - 1) != '\n') --ls;
        Rec r = parse_record(tv, ls, P);
        if (r.survive && r.qn_off <= 0xFFFFu && r.qn_len <= 0xFFFFu && r.rn_off <= 0xFFFFu && r.rn_len <= 0xFFFFu)
            return !text_eq(tv, qa, ql, ls + r.qn_off, r.qn_len);
        q = ls;
    }
    return true;                                      // first surviving line of the block
}
template <class Cfg> MKT_HD void ph_start(TileState<Cfg>& st, const TextView& tv, const Params& P, uint32_t i) {
    if (!(st.bits[i] & 1)) return;
    int32_t j = (int32_t)i - 1;
    while (j >= 0 && !(st.bits[j] & 1)) --j;
    bool start;
    if (j >= 0) start = !text_eq(tv, st.off[i] + st.qn_off[i], st.qn_len[i], st.off[j] + st.qn_off[j], st.qn_len[j]);
    else start = resolve_back(st, tv, P, i);
    if (start) st.bits[i] |= 2;
}

// ---- phase: walk the group opened by line i, classify it ---------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
MKT_HD uint32_t lds_inc(uint32_t* p) { return atomicAdd(p, 1u); }
MKT_HD void lds_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
MKT_HD void lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
MKT_HD void lds_max(uint32_t* p, uint32_t v) { atomicMax(p, v); }
#else
MKT_HD uint32_t lds_inc(uint32_t* p) { return (*p)++; }
MKT_HD void lds_add(uint32_t* p, uint32_t v) { *p += v; }
MKT_HD void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
MKT_HD void lds_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
#endif

template <class Cfg> MKT_HD void ph_group(TileState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G, uint32_t i) {
    st.g_info[i] = 0; st.g_plen[i] = 0; st.g_slen[i] = 0; st.g_last_end[i] = 0;
    st.g_posA[i] = st.g_posB[i] = 0; st.g_chrA[i] = st.g_chrB[i] = 0; st.g_chrA_len[i] = st.g_chrB_len[i] = 0;
    if (!(st.bits[i] & 2)) return;
    const uint32_t qa = st.off[i] + st.qn_off[i], ql = st.qn_len[i];
    uint32_t nmem = 0, n1 = 0, n2 = 0;
    uint32_t sa = 0xFFFFu, sb = 0xFFFFu, sc = 0xFFFFu, sd = 0xFFFFu;    // unc: r1a r1b r2a r2b; flash: a b
    bool gap = false, contig = true;
    uint32_t last_end = 0;          // one past the '\n' of the last member
    uint64_t sam_bytes = 0;

    // returns true when record `idx` (flag f) must be kept for the classifier
    auto wants = [&](uint32_t f) -> int {
        if (P.mode == MODE_FLASH) return nmem < 2 ? (int)nmem : -1;
        if (f & 64u) return n1 < 2 ? (int)n1 : -1;
        if (f & 128u) return n2 < 2 ? 2 + (int)n2 : -1;
        return -1;
    };
    auto take = [&](uint32_t idx, uint32_t f, uint32_t loff, uint32_t lend) {
        int slot = wants(f);
        if (slot == 0) sa = idx; else if (slot == 1) sb = idx; else if (slot == 2) sc = idx; else if (slot == 3) sd = idx;
        if (P.mode == MODE_UNC) { if (f & 64u) ++n1; else if (f & 128u) ++n2; }
        ++nmem;
        if (gap) contig = false;
        sam_bytes += (uint64_t)(lend - loff) + 1u;
        if (lend >= tv.n) contig = false;             // last line without '\n': the copy adds one
        last_end = lend < tv.n ? lend + 1 : tv.n;
    };

    uint32_t j = i;
    for (;;) {
        if (j < st.NL) {
            const uint8_t b = st.bits[j];
            if (j > i && (b & 2)) break;
            if (b & 1) {
                uint32_t e = st.end[j];
                if (e == kUnknown) e = find_newline(tv, G.w1);
                take(j, st.flag[j], st.off[j], e);
            } else gap = true;
            ++j;
            continue;
        }
        if (G.w1 >= tv.n) break;                      // the window reaches the end of the block
        // the group may continue past the window: walk line by line through global memory
        uint32_t le = st.end[st.NL - 1];
        if (le == kUnknown) le = find_newline(tv, G.w1);
        uint32_t q = le + 1;
        while (q < tv.n) {
            Rec r = parse_record(tv, q, P);
            const uint32_t e = find_newline(tv, q);
            bool s = r.survive && r.qn_off <= 0xFFFFu && r.qn_len <= 0xFFFFu && r.rn_off <= 0xFFFFu && r.rn_len <= 0xFFFFu;
            if (s) {
                if (!text_eq(tv, qa, ql, q + r.qn_off, r.qn_len)) break;
                uint32_t idx = 0xFFFFu;
                if (wants(r.flag) >= 0) {
                    uint32_t k = lds_inc(&st.ovf_n);
                    if (k < (uint32_t)Cfg::OVF) { idx = Cfg::LCAP + k; bool dummy; store_rec(st, idx, r, &dummy); }
                    else lds_or(&st.err, E_OVF_SLOTS);
                }
                take(idx, r.flag, q, e);
            } else gap = true;
            q = e + 1;
        }
        break;
    }

    Verdict v;
    if (P.mode == MODE_FLASH) {
        Seg a = sa != 0xFFFFu ? load_seg(st, sa) : seg_zero();
        Seg b = sb != 0xFFFFu ? load_seg(st, sb) : seg_zero();
        v = classify_flash(tv, nmem, a, b, P.ratio);
    } else {
        Seg a = sa != 0xFFFFu ? load_seg(st, sa) : seg_zero();
        Seg b = sb != 0xFFFFu ? load_seg(st, sb) : seg_zero();
        Seg c = sc != 0xFFFFu ? load_seg(st, sc) : seg_zero();
        Seg d = sd != 0xFFFFu ? load_seg(st, sd) : seg_zero();
        v = classify_unc(tv, n1, n2, a, b, c, d, P.ratio);
    }
    uint32_t info = GI_START | (v.counter & GI_COUNTER);
    if (v.emit) {
        info |= GI_EMIT;
        if (contig) info |= GI_CONTIG;
        if (v.sA == '-') info |= GI_SA_MINUS;
        if (v.sB == '-') info |= GI_SB_MINUS;
        st.g_posA[i] = v.posA; st.g_posB[i] = v.posB;
        st.g_chrA[i] = v.chrA_off; st.g_chrB[i] = v.chrB_off;
        st.g_chrA_len[i] = (uint16_t)v.chrA_len; st.g_chrB_len[i] = (uint16_t)v.chrB_len;
        st.g_plen[i] = pair_line_len(ql, v);
        st.g_slen[i] = P.write_sam ? (uint32_t)sam_bytes : 0u;
        st.g_last_end[i] = last_end;
    }
    st.g_info[i] = info;
}

// ---- phase: emit the group opened by line i ----------------------------------------------------
struct StageSink {
    uint8_t* p;
    MKT_HD void put(uint8_t c) { *p++ = c; }
};
struct OutPtrs {
    uint8_t* pairs; uint64_t pairs_cap;
    uint8_t* sam; uint64_t sam_cap;
    uint64_t* sc; uint64_t sc_cap;      // global indices of self-circle groups (quirk Q2)
    uint64_t sc_base, group_base;       // totals of the blocks before this one
};

template <class Cfg> MKT_HD void ph_emit(TileState<Cfg>& st, const TextView& tv, const Params& P, const OutPtrs& out, uint32_t i) {
    const uint32_t info = st.g_info[i];
    if (!(info & GI_START)) return;
    const uint32_t counter = info & GI_COUNTER;
    if (counter) lds_add(&st.cnt[counter], 1u);
    if (counter == C_SELFCIRCLE) {
        uint64_t k = out.sc_base + st.base.sc + st.x_sc[i];
        if (k < out.sc_cap) out.sc[k] = out.group_base + st.base.groups + st.x_grp[i];
        else lds_or(&st.err, E_SC_CAP);
    }
    if (!(info & GI_EMIT)) return;
    Verdict v;
    v.counter = counter; v.emit = true;
    v.chrA_off = st.g_chrA[i]; v.chrA_len = st.g_chrA_len[i]; v.chrB_off = st.g_chrB[i]; v.chrB_len = st.g_chrB_len[i];
    v.posA = st.g_posA[i]; v.posB = st.g_posB[i];
    v.sA = (info & GI_SA_MINUS) ? '-' : '+'; v.sB = (info & GI_SB_MINUS) ? '-' : '+';
    const uint32_t qa = st.off[i] + st.qn_off[i], ql = st.qn_len[i];
    const uint32_t lo = st.x_pair[i], len = st.g_plen[i];
    if (lo + len <= (uint32_t)Cfg::STG) {
        StageSink s{st.stg + lo};
        format_pair(s, tv, qa, ql, v);
        lds_max(&st.stg_used, lo + len);
    } else {
        uint64_t go = (uint64_t)st.base.pair_bytes + lo;
        if (go + len <= out.pairs_cap) { StageSink s{out.pairs + go}; format_pair(s, tv, qa, ql, v); }
        else lds_or(&st.err, E_PAIRS_CAP);
    }
    if (P.write_sam && !(info & GI_CONTIG)) {
        // slow copy: surviving lines of [first line, last member end), each followed by '\n'
        uint64_t go = st.base.sam_bytes + st.x_sam[i];
        if (go + st.g_slen[i] > out.sam_cap) { lds_or(&st.err, E_SAM_CAP); return; }
        uint32_t q = st.off[i];
        const uint32_t stop = st.g_last_end[i];
        while (q < stop) {
            Rec r = parse_record(tv, q, P);
            const uint32_t e = find_newline(tv, q);
            bool s = r.survive && r.qn_off <= 0xFFFFu && r.qn_len <= 0xFFFFu && r.rn_off <= 0xFFFFu && r.rn_len <= 0xFFFFu;
            if (s) {
                for (uint32_t p = q; p < e; ++p) out.sam[go++] = tv.at(p);
                out.sam[go++] = '\n';
            }
            q = e + 1;
        }
    }
}

// the tile's last group, for the host's Q1 bookkeeping
template <class Cfg> MKT_HD void ph_last(const TileState<Cfg>& st, TileLast* tl, uint32_t i) {
    const uint32_t info = st.g_info[i];
    if (!(info & GI_START)) return;
    if ((uint32_t)st.x_grp[i] + 1u != st.sums.groups) return;
    tl->counter = info & GI_COUNTER;
    tl->pair_bytes = st.g_plen[i];
    tl->sam_bytes = st.g_slen[i];
    tl->valid = 1;
}

}  // namespace mkt
