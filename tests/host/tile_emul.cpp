// tests/host/tile_emul.cpp -- TEST TOOL, never shipped or loaded by the product.
//
// Runs the per-tile phases of microcket_amd/csrc/mkt_tile.h serially on the CPU, tile after tile,
// exactly as the HIP kernel sequences them (one phase = one parallel loop + barrier there), and
// the host bookkeeping of mkt_host.h on top.  It lets the CPU test-suite check the tile logic
// (halo handling, slow paths, block cuts, Q1/Q2) against the oracle for several tile geometries
// without a GPU.  The GPU tests check the real kernel; this only de-risks it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#include "../../microcket_amd/csrc/mkt_host.h"
#include "../../microcket_amd/csrc/mkt_fast.h"

using namespace mkt;

// The lean path (mkt_fast.h) for one tile, exactly as k_fast sequences it.  Returns false when the tile is
// abnormal (deferred to the generic path); otherwise appends its outputs.
struct LeanOut { TileSums run; };
template <class FC>
static bool lean_tile(FastState<FC>& st, const uint8_t* text, uint32_t n, const Params& P, uint32_t t, OutPtrs& out, uint64_t sc_base,
                      TileSums& run, std::vector<uint32_t>& tile_groups, BlockResult& res) {
    fast_reset(st);
    const TileGeom G = fast_geom<FC>(t, n);
    const uint32_t wlen = G.w1 - G.w0;
    memset(st.win, 0, sizeof st.win);
    memset(&st.u, 0, sizeof st.u);
    const TextView tv = fast_view(st, text, n, G);
    const uint8_t* win = text + G.w0;                          // the kernel streams these bytes through registers
    // scan + line table: one entry per 16-byte vector of the window that holds a newline (plus the block's first line)
    uint32_t NL = 0;
    if (G.w0 == 0) st.hv16[NL++] = 0;
    for (uint32_t v = 0; v * 16u < wlen; ++v) {
        bool hit = false;
        for (uint32_t b = 0; b < 16u && v * 16u + b < wlen; ++b) hit = hit || win[v * 16u + b] == '\n';
        if (hit) { if (NL == (uint32_t)FC::LCAP) return false; st.hv16[NL++] = (uint16_t)v; }
    }
    st.last_line_end = G.w1 >= n ? n : kUnknown;
    // line heads with their whitespace bits (k_fast: one lane per 16-byte chunk)
    for (uint32_t i = 0; i < NL; ++i)
        for (uint32_t c = 0; c < (uint32_t)FC::HCH; ++c) fast_head_chunk_ref(st, text, n, G, i, c);
    st.NL = NL;
    for (uint32_t i = 0; i < NL; ++i) fast_parse(st, tv, P, G, i);
    // the tile's own lines (the kernel: two ballots over "line starts at or after t0 / t1")
    st.first_idx = NL; st.end_idx = NL;
    for (uint32_t i = NL; i-- > 0;) {
        if (G.w0 + st.goff[i] >= G.t0) st.first_idx = i;
        if (G.w0 + st.goff[i] >= G.t1) st.end_idx = i;
    }
    const uint32_t NLe = fast_nle(st);
    const uint32_t first_idx = st.first_idx < NLe ? st.first_idx : NLe, end_idx = st.end_idx < NLe ? st.end_idx : NLe;
    fast_build_masks(st);                                      // the kernel: ballots after fast_parse
    for (int k = 0; k < 4; ++k) st.m_start[k] = 0;
    if (!st.abn)
        for (uint32_t i = 0; i < NLe; ++i)                     // the kernel: one lane per line, ballot
            if (mask_bit(st.m_surv, i) && i >= first_idx && fast_is_start(st, tv, G, i)) st.m_start[i >> 6] |= 1ull << (i & 63);
    if (!st.abn) for (uint32_t i = first_idx; i < end_idx; ++i) fast_group(st, tv, P, G, i);
    if (st.abn) { if (getenv("MKT_EMUL_DEBUG")) fprintf(stderr, "lean tile %u deferred: reason %u (NL %u)\n", t, st.abn, st.NL); return false; }
    TileSums s = {0, 0, 0, 0, 0};
    auto& g = st.u.g;
    for (uint32_t i = first_idx; i < NLe; ++i) {
        g.x_sam[i] = (uint32_t)s.sam_bytes;
        if (i < end_idx) {
            const uint32_t info = g.g_info[i];
            g.x_grp[i] = (uint8_t)s.groups; g.x_sc[i] = (uint8_t)s.sc; g.x_emit[i] = (uint8_t)s.emitted; g.x_pair[i] = (uint16_t)s.pair_bytes;
            if (info & GI_EMIT) g.em_idx[s.emitted] = (uint8_t)i;
            if (info & GI_START) ++s.groups;
            if (info & GI_EMIT) ++s.emitted;
            if ((info & GI_START) && (info & GI_COUNTER) == C_SELFCIRCLE) ++s.sc;
            s.pair_bytes += g.g_plen[i];
        }
        if (P.write_sam) s.sam_bytes += fast_line_sam(st, G, i);
    }
    if (s.pair_bytes > 0xFFFFu) return false;
    st.sums = s; st.base = run; st.base.sc += (uint32_t)sc_base;
    while (tile_groups.size() <= t) tile_groups.push_back(0);
    tile_groups[t] = run.groups;
    TileLast tl = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = first_idx; i < end_idx; ++i) { fast_account(st, out, t, i); fast_last(st, G, &tl, i); }
    // the kernel: one lane per reported pair writes its whole line (fast_emit_line: aligned dwords + single bytes at the ends);
    // here last line first, between sentinels, then checked byte by byte against the per-byte reference fast_pair_byte
    {
        uint8_t* o = out.pairs + run.pair_bytes;
        const uint8_t before = run.pair_bytes ? o[-1] : 0, after = o[s.pair_bytes];
        for (uint32_t k = 0; k < s.pair_bytes; ++k) o[k] = 0xEE;
        for (uint32_t e = s.emitted; e-- > 0;) {
            const uint32_t i = g.em_idx[e];
            fast_emit_line(st, g.g_slot[i], g.g_plen[i], o + g.x_pair[i]);
        }
        for (uint32_t k = 0; k < s.pair_bytes; ++k) if (o[k] != fast_pair_byte(st, k)) res.err |= 0x4000;
        if ((run.pair_bytes && o[-1] != before) || o[s.pair_bytes] != after) res.err |= 0x4000;
    }
    if (P.write_sam)
        for (uint32_t i = first_idx; i < NLe; ++i)
            if (mask_bit(st.m_emit, i)) memcpy(out.sam + run.sam_bytes + g.x_sam[i], text + G.w0 + st.goff[i], fast_line_sam(st, G, i));
    if (tl.valid) res.last = tl;
    for (int c = 0; c < (int)C_COUNT; ++c) res.counters[c] += st.cnt[c];
    run.groups += s.groups; run.emitted += s.emitted; run.sc += s.sc; run.pair_bytes += s.pair_bytes; run.sam_bytes += s.sam_bytes;
    return true;
}

template <class Cfg, int FLCAP = (Cfg::LCAP > 255 ? 255 : Cfg::LCAP)>
static void emul_block(const uint8_t* text, uint32_t n, const Params& P, std::vector<uint8_t>& pairs,
                       std::vector<uint8_t>& sam, std::vector<uint64_t>& sc, uint64_t group_base, BlockResult& res, bool lean = false) {
    memset(&res, 0, sizeof res);
    pairs.assign((size_t)n + 4096, 0);
    sam.assign((size_t)n + 4096, 0);
    const uint64_t sc_base = sc.size();
    sc.resize(sc_base + (size_t)n / 2 + 16, 0);
    OutPtrs out{pairs.data(), pairs.size(), sam.data(), sam.size(), sc.data(), sc.size(), nullptr, 0, nullptr};
    std::vector<uint32_t> tile_groups;
    std::unique_ptr<TileState<Cfg>> stp(new TileState<Cfg>);
    TileState<Cfg>& st = *stp;
    TileSums run = {0, 0, 0, 0, 0};
    const uint32_t nt = num_tiles(n, Cfg::TILE);
    res.tiles = nt;
    typedef FastCfg<Cfg::TILE, Cfg::HB, Cfg::HF, FLCAP> FC;
    std::unique_ptr<FastState<FC>> fstp(new FastState<FC>);
    for (uint32_t k = 0; k < 64; ++k) fast_init(*fstp, k);
    uint64_t lean_ok = 0;
    for (uint32_t t = 0; t < nt; ++t) {
        if (lean && lean_tile<FC>(*fstp, text, n, P, t, out, sc_base, run, tile_groups, res)) { ++lean_ok; continue; }
        tile_reset(st);
        TileGeom G = tile_geom<Cfg>(t, n);
        const uint32_t wlen = G.w1 - G.w0;
        memset(st.win, 0, sizeof st.win);
        memcpy(st.win, text + G.w0, wlen);
        memset(&st.u, 0, sizeof st.u);
        for (uint32_t r = 0; r < wlen; ++r) {
            if (st.win[r] == '\n') st.u.m.nlm[r >> 6] |= 1ull << (r & 63);
            if (is_ws(st.win[r])) st.u.m.wsm[r >> 6] |= 1ull << (r & 63);
        }
        TextView tv = tile_view(st, text, n, G);
        // line table
        uint32_t NL = 0;
        bool overflow = false;
        if (G.w0 == 0) st.off[NL++] = 0;
        for (uint32_t r = 0; r + 1 < wlen; ++r)
            if (st.win[r] == '\n') { if (NL == (uint32_t)Cfg::LCAP) { overflow = true; break; } st.off[NL++] = G.w0 + r + 1; }
        if (overflow) { st.err |= E_LINE_TABLE; NL = 0; }
        st.NL = NL; st.first_idx = NL; st.end_idx = NL;
        for (uint32_t i = 0; i < NL; ++i) ph_parse(st, tv, P, G, i);
        tile_trim(st);
        NL = st.NL;
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) ph_group(st, tv, P, G, i);
        // exclusive sums
        TileSums s = {0, 0, 0, 0, 0};
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) {
            auto& g = st.u.g;
            const uint32_t info = g.g_info[i];
            g.x_grp[i] = (uint16_t)s.groups; g.x_sc[i] = (uint16_t)s.sc;
            g.x_pair[i] = s.pair_bytes; g.x_sam[i] = (uint32_t)s.sam_bytes;
            if (info & GI_EMIT) g.em_idx[s.emitted] = (uint16_t)i;
            if (info & GI_START) ++s.groups;
            if (info & GI_EMIT) ++s.emitted;
            if ((info & GI_START) && (info & GI_COUNTER) == C_SELFCIRCLE) ++s.sc;
            s.pair_bytes += g.g_plen[i]; s.sam_bytes += g.g_slen[i];
        }
        st.sums = s; st.base = run; st.base.sc += (uint32_t)sc_base;
        while (tile_groups.size() <= t) tile_groups.push_back(0);
        tile_groups[t] = run.groups;            // exclusive prefix, as k_finish computes it
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) ph_account(st, tv, P, out, t, i);
        if (run.pair_bytes + (uint64_t)s.pair_bytes <= out.pairs_cap) {
            for (uint32_t k = 0; k < s.pair_bytes; ++k) out.pairs[run.pair_bytes + k] = tile_pair_byte(st, tv, k);
        } else st.err |= E_PAIRS_CAP;
        if (P.write_sam)
            for (uint32_t i = st.first_idx; i < st.end_idx; ++i)
                if ((st.u.g.g_info[i] & GI_EMIT) && (st.u.g.g_info[i] & GI_CONTIG)) {
                    uint64_t go = run.sam_bytes + st.u.g.x_sam[i];
                    if (go + st.u.g.g_slen[i] <= out.sam_cap) memcpy(out.sam + go, text + st.off[i], st.u.g.g_slen[i]);
                    else st.err |= E_SAM_CAP;
                }
        TileLast tl = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) ph_last(st, &tl, i);
        if (tl.valid) res.last = tl;
        for (int c = 0; c < (int)C_COUNT; ++c) res.counters[c] += st.cnt[c];
        res.err |= st.err;
        run.groups += s.groups; run.emitted += s.emitted; run.sc += s.sc; run.pair_bytes += s.pair_bytes; run.sam_bytes += s.sam_bytes;
    }
    res.pad = (uint32_t)lean_ok;
    res.groups = run.groups; res.emitted = run.emitted; res.sc = run.sc; res.pair_bytes = run.pair_bytes; res.sam_bytes = run.sam_bytes;
    for (uint64_t k = 0; k < run.sc; ++k) {       // what k_finish does: (tile, ordinal) -> global group index
        const uint64_t e = sc[sc_base + k];
        sc[sc_base + k] = group_base + tile_groups[(uint32_t)(e >> 32)] + (uint32_t)(e & 0xFFFFFFFFu);
    }
    pairs.resize(run.pair_bytes); sam.resize(run.sam_bytes); sc.resize(sc_base + run.sc);
}

typedef TileCfg<kLeanTile, kLeanHB, kLeanHF, 512, 4> CfgFast;          // the production geometry
typedef TileCfg<256, 64, 192, 512, 4> CfgSafe;
typedef TileCfg<1024, 128, 512, 96, 4> CfgMid;
typedef TileCfg<2048, 16, 16, 128, 4> CfgNoHalo;
typedef TileCfg<kDenseTile, kDenseHB, kDenseHF, 256, 4> CfgDense;          // the production short-line geometries
typedef TileCfg<kMidTile, kMidHB, kMidHF, 512, 4> CfgMid32;

// One emulated shard / input stream (mirrors a mkt_ctx of the library: feed, group count, finish).
struct EmulShard {
    Params P;
    int ref_threads, cfg;
    RunAccum acc;
    std::vector<uint8_t> pairs, sam;
    std::vector<uint64_t> sc;
    uint32_t err = 0;
    uint64_t blocks = 0, lean_tiles = 0, tiles = 0;
};

static void emul_feed(EmulShard& S, const char* text, size_t n, size_t block_bytes) {
    std::vector<uint8_t> bp, bs;
    size_t pos = 0;
    if (block_bytes == 0) block_bytes = n ? n : 1;
    while (pos < n) {
        size_t take = n - pos < block_bytes ? n - pos : block_bytes;
        if (pos + take < n) {                      // not the final block: cut on a group boundary
            size_t span = take;
            for (;;) {
                size_t cut = group_aligned_prefix(text + pos, span, S.P.min_mapq);
                if (cut) { take = cut; break; }
                if (pos + span >= n) { take = n - pos; break; }
                span = (span * 2 < n - pos) ? span * 2 : n - pos;
            }
        }
        BlockResult r;
        const uint8_t* b = (const uint8_t*)text + pos;
        const bool lean = S.cfg >= 10;              // cfg 10 + k: lean path first, generic path for the tiles it defers
        switch (S.cfg % 10) {
        case 1: emul_block<CfgSafe>(b, (uint32_t)take, S.P, bp, bs, S.sc, S.acc.groups, r, lean); break;
        case 2: emul_block<CfgMid>(b, (uint32_t)take, S.P, bp, bs, S.sc, S.acc.groups, r, lean); break;
        case 3: emul_block<CfgNoHalo>(b, (uint32_t)take, S.P, bp, bs, S.sc, S.acc.groups, r, lean); break;
        case 4: emul_block<CfgDense, kDenseLCAP>(b, (uint32_t)take, S.P, bp, bs, S.sc, S.acc.groups, r, lean); break;
        case 5: emul_block<CfgMid32, kMidLCAP>(b, (uint32_t)take, S.P, bp, bs, S.sc, S.acc.groups, r, lean); break;
        default: emul_block<CfgFast, kLeanLCAP>(b, (uint32_t)take, S.P, bp, bs, S.sc, S.acc.groups, r, lean); break;
        }
        S.lean_tiles += r.pad; S.tiles += r.tiles;
        S.err |= r.err;
        S.acc.add_block(r);
        S.pairs.insert(S.pairs.end(), bp.begin(), bp.end());
        S.sam.insert(S.sam.end(), bs.begin(), bs.end());
        pos += take;
        ++S.blocks;
    }
}

extern "C" {
void* emul_open(int mode, float ratio, int min_mapq, int write_sam, int ref_threads, int cfg) {
    EmulShard* S = new EmulShard();
    S->P = Params{mode, ratio, (uint32_t)min_mapq, write_sam};
    S->ref_threads = ref_threads; S->cfg = cfg;
    return S;
}
void emul_feed_bytes(void* h, const char* text, size_t n, size_t block_bytes) { emul_feed(*(EmulShard*)h, text, n, block_bytes); }
uint64_t emul_groups(void* h) { return ((EmulShard*)h)->acc.groups; }
// counters8: lowMap manyHits unpaired selfCircle trans cis10K cis1K cis0 ; stats: groups, pairs, err, blocks
int emul_finish(void* h, int drop_last, uint64_t group_offset, uint64_t total_groups, char** out_pairs, size_t* n_pairs,
                char** out_sam, size_t* n_sam, uint32_t* counters8, uint64_t* stats) {
    EmulShard& S = *(EmulShard*)h;
    RunStats s = S.acc.finish(drop_last != 0, (uint32_t)S.ref_threads, group_offset, total_groups ? total_groups : S.acc.groups, S.sc.data());
    // the dropped group's bytes are the tail of the (input-ordered) emulation output
    S.pairs.resize(s.pair_bytes);
    S.sam.resize(s.sam_bytes);
    *out_pairs = (char*)malloc(S.pairs.size() + 1); if (!S.pairs.empty()) memcpy(*out_pairs, S.pairs.data(), S.pairs.size()); *n_pairs = S.pairs.size();
    *out_sam = (char*)malloc(S.sam.size() + 1); if (!S.sam.empty()) memcpy(*out_sam, S.sam.data(), S.sam.size()); *n_sam = S.sam.size();
    const int order[8] = {C_LOWMAP, C_MANYHITS, C_UNPAIRED, C_SELFCIRCLE, C_TRANS, C_CIS10K, C_CIS1K, C_CIS0};
    for (int k = 0; k < 8; ++k) counters8[k] = s.counters[order[k]];
    stats[0] = s.groups; stats[1] = s.pairs; stats[2] = S.err; stats[3] = S.blocks | (S.lean_tiles << 20) | (S.tiles << 42);
    return 0;
}
void emul_close(void* h) { delete (EmulShard*)h; }

// whole input in one call (single stream)
int emul_run(const char* text, size_t n, int mode, float ratio, int min_mapq, int write_sam, int ref_threads, int cfg,
             size_t block_bytes, char** out_pairs, size_t* n_pairs, char** out_sam, size_t* n_sam, char* log,
             uint64_t* stats /* groups, pairs, err, blocks */) {
    void* h = emul_open(mode, ratio, min_mapq, write_sam, ref_threads, cfg);
    emul_feed_bytes(h, text, n, block_bytes);
    uint32_t c8[8];
    emul_finish(h, 1, 0, 0, out_pairs, n_pairs, out_sam, n_sam, c8, stats);
    snprintf(log, 256, "lowMap\t%u\nmanyHits\t%u\nunpaired\t%u\nselfCircle\t%u\ntrans\t%u\ncis10K\t%u\ncis1K\t%u\ncis0\t%u\n",
             c8[0], c8[1], c8[2], c8[3], c8[4], c8[5], c8[6], c8[7]);
    emul_close(h);
    return 0;
}
void emul_free(void* p) { free(p); }
// the run-time tile geometry for an average line length (mkt_fast.h: lean_dims) and the capacities it must stay inside
void emul_lean_dims(double avg_line_bytes, uint32_t* out /* tile, hb, hf, max tile, max hb, max hf */) {
    const TileDims d = lean_dims(avg_line_bytes);
    out[0] = d.tile; out[1] = d.hb; out[2] = d.hf; out[3] = (uint32_t)kLeanTile; out[4] = (uint32_t)kLeanHB; out[5] = (uint32_t)kLeanHF;
}
}
