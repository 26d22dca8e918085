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

using namespace mkt;

template <class Cfg>
static void emul_block(const uint8_t* text, uint32_t n, const Params& P, std::vector<uint8_t>& pairs,
                       std::vector<uint8_t>& sam, std::vector<uint64_t>& sc, uint64_t group_base, BlockResult& res) {
    memset(&res, 0, sizeof res);
    pairs.assign((size_t)n + 4096, 0);
    sam.assign((size_t)n + 4096, 0);
    const uint64_t sc_base = sc.size();
    sc.resize(sc_base + (size_t)n / 2 + 16, 0);
    OutPtrs out{pairs.data(), pairs.size(), sam.data(), sam.size(), sc.data(), sc.size(), sc_base, group_base};
    std::unique_ptr<TileState<Cfg>> stp(new TileState<Cfg>);
    TileState<Cfg>& st = *stp;
    TileSums run = {0, 0, 0, 0, 0};
    const uint32_t nt = num_tiles(n, Cfg::TILE);
    res.tiles = nt;
    for (uint32_t t = 0; t < nt; ++t) {
        tile_reset(st);
        TileGeom G = tile_geom<Cfg>(t, n);
        const uint32_t wlen = G.w1 - G.w0;
        memcpy(st.win, text + G.w0, wlen);
        TextView tv{text, n, st.win, G.w0, wlen};
        // line table
        uint32_t NL = 0;
        bool overflow = false;
        if (G.w0 == 0) st.off[NL++] = 0;
        for (uint32_t r = 0; r + 1 < wlen; ++r)
            if (st.win[r] == '\n') { if (NL == (uint32_t)Cfg::LCAP) { overflow = true; break; } st.off[NL++] = G.w0 + r + 1; }
        if (overflow) { st.err |= E_LINE_TABLE; NL = 0; }
        st.NL = NL; st.first_idx = NL; st.end_idx = NL;
        for (uint32_t i = 0; i < NL; ++i) ph_parse(st, tv, P, G, i);
        for (uint32_t i = st.first_idx; i < NL; ++i) ph_start(st, tv, P, i);
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) ph_group(st, tv, P, G, i);
        // exclusive sums
        TileSums s = {0, 0, 0, 0, 0};
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) {
            const uint32_t info = st.g_info[i];
            st.x_grp[i] = (uint16_t)s.groups; st.x_emit[i] = (uint16_t)s.emitted; st.x_sc[i] = (uint16_t)s.sc;
            st.x_pair[i] = s.pair_bytes; st.x_sam[i] = (uint32_t)s.sam_bytes;
            if (info & GI_START) ++s.groups;
            if (info & GI_EMIT) ++s.emitted;
            if ((info & GI_START) && (info & GI_COUNTER) == C_SELFCIRCLE) ++s.sc;
            s.pair_bytes += st.g_plen[i]; s.sam_bytes += st.g_slen[i];
        }
        st.sums = s; st.base = run;
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) ph_emit(st, tv, P, out, i);
        if (run.pair_bytes + (uint64_t)st.stg_used <= out.pairs_cap) memcpy(out.pairs + run.pair_bytes, st.stg, st.stg_used);
        else st.err |= E_PAIRS_CAP;
        if (P.write_sam)
            for (uint32_t i = st.first_idx; i < st.end_idx; ++i)
                if ((st.g_info[i] & GI_EMIT) && (st.g_info[i] & GI_CONTIG)) {
                    uint64_t go = run.sam_bytes + st.x_sam[i];
                    if (go + st.g_slen[i] <= out.sam_cap) memcpy(out.sam + go, text + st.off[i], st.g_slen[i]);
                    else st.err |= E_SAM_CAP;
                }
        TileLast tl = {0, 0, 0, 0};
        for (uint32_t i = st.first_idx; i < st.end_idx; ++i) ph_last(st, &tl, i);
        if (tl.valid) res.last = tl;
        for (int c = 0; c < (int)C_COUNT; ++c) res.counters[c] += st.cnt[c];
        res.err |= st.err;
        run.groups += s.groups; run.emitted += s.emitted; run.sc += s.sc; run.pair_bytes += s.pair_bytes; run.sam_bytes += s.sam_bytes;
    }
    res.groups = run.groups; res.emitted = run.emitted; res.sc = run.sc; res.pair_bytes = run.pair_bytes; res.sam_bytes = run.sam_bytes;
    pairs.resize(run.pair_bytes); sam.resize(run.sam_bytes); sc.resize(sc_base + run.sc);
}

typedef TileCfg<16384, 1024, 4096, 256, 4, 4096> CfgFast;
typedef TileCfg<256, 64, 192, 512, 4, 512> CfgSafe;
typedef TileCfg<1024, 128, 512, 96, 4, 256> CfgMid;
typedef TileCfg<2048, 16, 16, 128, 4, 64> CfgNoHalo;

extern "C" {
// Returns 0 on success.  out_* are malloc'ed; caller frees with emul_free.  log must hold >= 256 bytes.
int emul_run(const char* text, size_t n, int mode, float ratio, int min_mapq, int write_sam, int ref_threads, int cfg,
             size_t block_bytes, char** out_pairs, size_t* n_pairs, char** out_sam, size_t* n_sam, char* log,
             uint64_t* stats /* groups, pairs, err, blocks */) {
    Params P{mode, ratio, (uint32_t)min_mapq, write_sam};
    RunAccum acc;
    std::vector<uint8_t> all_pairs, all_sam, bp, bs;
    std::vector<uint64_t> sc;
    uint32_t err = 0;
    uint64_t blocks = 0;
    size_t pos = 0;
    if (block_bytes == 0) block_bytes = n ? n : 1;
    while (pos < n) {
        size_t take = n - pos < block_bytes ? n - pos : block_bytes;
        if (pos + take < n) {                      // not the final block: cut on a group boundary
            size_t span = take;
            for (;;) {
                size_t cut = group_aligned_prefix(text + pos, span);
                if (cut) { take = cut; break; }
                if (pos + span >= n) { take = n - pos; break; }
                span = (span * 2 < n - pos) ? span * 2 : n - pos;
            }
        }
        BlockResult r;
        const uint8_t* b = (const uint8_t*)text + pos;
        switch (cfg) {
        case 1: emul_block<CfgSafe>(b, (uint32_t)take, P, bp, bs, sc, acc.groups, r); break;
        case 2: emul_block<CfgMid>(b, (uint32_t)take, P, bp, bs, sc, acc.groups, r); break;
        case 3: emul_block<CfgNoHalo>(b, (uint32_t)take, P, bp, bs, sc, acc.groups, r); break;
        default: emul_block<CfgFast>(b, (uint32_t)take, P, bp, bs, sc, acc.groups, r); break;
        }
        err |= r.err;
        acc.add_block(r);
        all_pairs.insert(all_pairs.end(), bp.begin(), bp.end());
        all_sam.insert(all_sam.end(), bs.begin(), bs.end());
        pos += take;
        ++blocks;
    }
    RunStats s = acc.finish(true, (uint32_t)ref_threads, 0, acc.groups, sc.data());
    all_pairs.resize(s.pair_bytes);
    all_sam.resize(s.sam_bytes);
    *out_pairs = (char*)malloc(all_pairs.size() + 1); memcpy(*out_pairs, all_pairs.data(), all_pairs.size()); *n_pairs = all_pairs.size();
    *out_sam = (char*)malloc(all_sam.size() + 1); memcpy(*out_sam, all_sam.data(), all_sam.size()); *n_sam = all_sam.size();
    format_log(s, log, 256);
    stats[0] = s.groups; stats[1] = s.pairs; stats[2] = err; stats[3] = blocks;
    return 0;
}
void emul_free(void* p) { free(p); }
}
