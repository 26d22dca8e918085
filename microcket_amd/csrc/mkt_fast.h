// mkt_fast.h -- the lean tile path of the fused sam2pairs kernel (k_fast in mkt_kernels.hip).
//
// Same algorithm as mkt_tile.h, restricted to what well-formed name-grouped SAM looks like so that
// the hot kernel carries no generic code: every line's six fields end inside its first ~112 bytes,
// every group opened in the tile closes inside the window, and the previous surviving line sits in
// the back halo.  A tile that violates any of this sets `abn` and is left untouched: the host runs
// the generic kernel (k_tiles, mkt_tile.h) over the list of such tiles.  Outputs go to atomically
// allocated ranges, so the two kernels compose freely.
//
// LDS holds only what the parser reads: the HEAD of every line (the HEADB = 128 bytes of the eight
// aligned 16-byte chunks that start at the chunk holding the line's first byte).  SEQ / QUAL / tags,
// 60 % of the bytes, stream through registers once for the newline scan and never touch LDS, so a
// 48 KB tile (+ halos) costs 22 KB of head store instead of a 53 KB window and three times as many
// lines are in flight per CU.  `off16` are offsets into that compact head store (what all text code
// uses); `goff` are the true window-relative line starts (geometry, line lengths).  State is packed
// (16-bit offsets, 8-bit field lengths): one tile is 39 KB of LDS, four workgroups per CU.  The phase functions
// are host+device so that tests/host/tile_emul.cpp checks this logic on the CPU as well.
#pragma once
#include "mkt_tile.h"

namespace mkt {

// Capacities of the production lean kernel: a window (tile + halos) of up to kLeanTile + kLeanHB + kLeanHF bytes and kLeanLCAP
// lines (LDS: ~226 B per line).  The BYTES per tile are chosen per input at run time (TileDims, lean_dims below): the phases
// run one lane per line on two of the four waves, so a window should hold just under 128 lines whatever the read length; a
// window with more lines than kLeanLCAP is left to the generic kernel.  tests/host/tile_emul.cpp emulates fixed geometries
// (configs 0/10, 5/15, 4/14).
#if defined(MKT_LEAN_GEOM)        // experiment builds only (microcket_amd.build.build_variant): -DMKT_LEAN_GEOM=tile,back,forward,lines
constexpr int kLeanGeom[4] = {MKT_LEAN_GEOM};
constexpr int kLeanTile = kLeanGeom[0], kLeanHB = kLeanGeom[1], kLeanHF = kLeanGeom[2], kLeanLCAP = kLeanGeom[3];
#else
constexpr int kLeanTile = 49152, kLeanHB = 3072, kLeanHF = 4096, kLeanLCAP = 168;
#endif
constexpr int kMidTile = 32768, kMidHB = 2048, kMidHF = 3072, kMidLCAP = 160;            // (emulation configs)
constexpr int kDenseTile = 16384, kDenseHB = 1024, kDenseHF = 2048, kDenseLCAP = 160;
constexpr int kLeanLinesTarget = 126;      // lines per window the host aims at: measured 122 / 126 / 129 / 132 -> 0.838 / 0.819 / 0.819 / 0.868 ms per block at 150 bp (fewer tiles against more windows past 128 lines)

// tile dimensions for text of `avg` bytes per line: ~kLeanLinesTarget lines per window, a back halo of ~7.5 lines (the previous
// surviving line) and a forward halo of ~8.75 (the rest of the last group and the first surviving line of the next one), all within
// the kernel's capacities.  Halos are sized by what a deferred tile costs: a pass of the generic kernel takes 0.1 - 0.2 ms for a
// handful of tiles, a tenth of a block's time.  Measured on the bench data: back 5 / forward 7.5 lines: one tile in 21 000
// deferred (two per 2 GB block); 7.5 / 7.5: one in 170 000 (all "group open at the end of the window").
// halos in lines.  Narrower halos mean more own lines per window and fewer tiles (6 / 7 lines: k_fast -3.4 %), but a tile whose halo is
// too narrow goes to the generic kernel, which takes ~150 us for it while the GPU waits: at 6 / 7 lines one tile in 10^5 of the bench's
// data (0.75 per 45 000-tile block) made the whole step 4 % SLOWER; at 7.5 / 8.75 none of 6.5 M is deferred (profiles/r03_raw/
// halo_sweep.txt, r03_kernel_variants.txt).  An input whose groups are longer makes the host widen them (mkt_capi.cpp: adapt_geometry)
constexpr double kLeanHaloBackLines = 7.5, kLeanHaloFwdLines = 8.75;
MKT_HD TileDims lean_dims(double avg) {
    if (avg < 48.0) avg = 48.0;
    if (avg > 4096.0) avg = 4096.0;
    auto r16 = [](double x) { return ((uint32_t)x + 15u) & ~15u; };
    TileDims d;
    d.hb = r16(kLeanHaloBackLines * avg); d.hf = r16(kLeanHaloFwdLines * avg);
    if (d.hb < 256u) d.hb = 256u;
    if (d.hb > (uint32_t)kLeanHB) d.hb = (uint32_t)kLeanHB;
    if (d.hf < 512u) d.hf = 512u;
    if (d.hf > (uint32_t)kLeanHF) d.hf = (uint32_t)kLeanHF;
    const double w = (double)kLeanLinesTarget * avg;
    uint32_t t = w > (double)(d.hb + d.hf + 2048u) ? r16(w - (double)(d.hb + d.hf)) : 2048u;
    if (t > (uint32_t)kLeanTile) t = (uint32_t)kLeanTile;
    // the whole window a multiple of 1 KiB (64 vectors of 16 bytes, one ballot word: the scan of a full window needs no bounds)
    const uint32_t rem = (d.hb + t + d.hf) & 1023u;
    if (rem >= 512u && t + (1024u - rem) <= (uint32_t)kLeanTile) t += 1024u - rem; else t -= rem;
    d.tile = t;
    return d;
}

template <int TILE_, int HB_, int HF_, int LCAP_, int GCAP_ = 96>
struct FastCfg {
    static constexpr int TILE = TILE_, HB = HB_, HF = HF_, W = HB_ + TILE_ + HF_, LCAP = LCAP_;
    static constexpr int MW = (W + 63) / 64 + 3;
    static constexpr int GCAP = GCAP_;               // emitting groups per tile (more: generic kernel)
    static constexpr int NV16 = (W + 15) / 16;       // 16-byte vectors of the window
    static constexpr int HMW = (NV16 + 63) / 64;     // 64-bit words of the "vector holds a newline" bitmap
    static constexpr int HCH = 8, HEADB = HCH * 16;  // head store: 16-byte chunks / bytes per line
    static constexpr int HSTRIDE = HEADB + 4;        // row pitch: 33 dwords, so lanes reading the same column of their own rows hit 32 different banks
    static constexpr int HW = LCAP_ * HSTRIDE;       // bytes of the head store
    static_assert(TILE_ % 16 == 0 && HB_ % 16 == 0 && HF_ % 16 == 0, "16-byte vector staging");
    static_assert(W < 65536, "window-relative offsets are 16 bit");
    static_assert(LCAP_ <= 255, "line / group ordinals are 8 bit");
    static_assert(LCAP_ * 132 < 65536, "head-store offsets are 16 bit");
};

// why a tile is left to the generic kernel (low byte of FastState::abn; any value != 0 defers)
enum { AB_LCAP = 1, AB_LONG, AB_TAB, AB_PREV_WS, AB_PREV_HEAD, AB_NO_PREV, AB_OPEN_GROUP, AB_LAST_LINE, AB_GCAP, AB_PAIR_BYTES, AB_SHORT_LINE };

constexpr uint8_t LB_EMIT = 8;          // line belongs to an emitting group (its bytes go to the .sam)

template <class Cfg>
struct FastState {
    alignas(16) uint8_t win[Cfg::HW + 16];           // line heads, HEADB bytes per line, rows HSTRIDE apart
    // per line of the window
    struct Recs {
        struct { uint32_t pos[Cfg::LCAP], lclip[Cfg::LCAP], rclip[Cfg::LCAP], mappable[Cfg::LCAP], right0[Cfg::LCAP], left1[Cfg::LCAP], right1[Cfg::LCAP]; } f;
    } rc;
    uint16_t hv16[Cfg::LCAP];            // window vector that holds the newline in front of line i (line table); the head row starts there
    uint16_t off16[Cfg::LCAP];           // line start in the head store: i * HSTRIDE + (1 + byte of that newline in its vector), set while parsing
    uint16_t goff[Cfg::LCAP];            // line start, window relative: 16 * hv16 + the same; may be >= the window length for the last entry
    uint16_t flag[Cfg::LCAP];
    uint8_t qn_off[Cfg::LCAP], qn_len[Cfg::LCAP], rn_off[Cfg::LCAP], rn_len[Cfg::LCAP], segCnt[Cfg::LCAP], bits[Cfg::LCAP];
    union alignas(16) Phase {
        struct {                                               // while parsing
            uint64_t hitmap[Cfg::HMW + 1];                     // bit v: window vector v (16 bytes) holds a newline (scan phase -> line table)
            alignas(8) uint16_t hmask[Cfg::LCAP][Cfg::HCH];    // per line: whitespace bits of its head chunks
        } m;
        struct {                                               // afterwards, indexed by the group's first line / by line
            uint32_t g_info[Cfg::LCAP], g_slen[Cfg::LCAP];
            uint32_t x_sam[Cfg::LCAP];                         // per LINE: offset of its bytes in the tile's .sam output
            uint16_t g_plen[Cfg::LCAP], x_pair[Cfg::LCAP];
            uint8_t x_grp[Cfg::LCAP], x_sc[Cfg::LCAP], x_emit[Cfg::LCAP], em_idx[Cfg::LCAP], g_slot[Cfg::LCAP];
            uint32_t l_posA[Cfg::GCAP], l_posB[Cfg::GCAP];     // extension (key records) only
            // emitting groups only (slot = g_slot[first line]): the .pairs line as five byte runs
            //   win[qa, +e0) | win[ca, +(e1-e0)) | litA | win[cb, +(e3-e2)) | litB
            // (QNAME and RNAME are copied together with the tab that follows them in the SAM line)
            uint16_t l_qa[Cfg::GCAP], l_ca[Cfg::GCAP], l_cb[Cfg::GCAP];
            uint16_t l_e0[Cfg::GCAP], l_e1[Cfg::GCAP], l_e2[Cfg::GCAP], l_e3[Cfg::GCAP];
            uint64_t l_litA[Cfg::GCAP][2], l_litB[Cfg::GCAP][2];   // "<posA>\t" and "<posB>\t<sA>\t<sB>\n", little-endian bytes
        } g;
    } u;
    // one bit per line of the window (ballots of the parse / start phases)
    uint64_t m_surv[4], m_eqp[4], m_r1[4], m_r2[4], m_start[4], m_emit[4];
    // extension: this workgroup's cache of the chromosome table, kept across its tiles.  One word per entry (name bytes
    // in bits 0..47, table slot in 48..60, valid in 63), so that a lane never pairs one entry's name with another's slot
    uint64_t cc[64];
    uint32_t NL, first_idx, end_idx, abn, last_line_end, nslot, c_t0, c_t1;
    uint32_t cnt[C_COUNT];
    TileSums sums, base;                  // base: ABSOLUTE positions in OutPtrs::pairs / sam / sc
    uint32_t region_pair0, region_sam0, region_id;
};

template <class Cfg> MKT_HD TileGeom fast_geom(uint32_t tile, uint32_t n) { return tile_geom(tile, n, cfg_dims<Cfg>()); }
template <class Cfg> MKT_HD TextView fast_view(const FastState<Cfg>& st, const uint8_t* text, uint32_t n, const TileGeom& G) {
    TextView tv;
    (void)G;                                          // text offsets of the lean path are head-store offsets
    tv.g = text; tv.n = n; tv.win = st.win; tv.w0 = 0; tv.wlen = Cfg::HW; tv.nlm = nullptr; tv.wsm = nullptr;
    return tv;
}
// once per workgroup (kernel) / per state object (emulation)
template <class Cfg> MKT_HD void fast_init(FastState<Cfg>& st, uint32_t k) { if (k < 64u) st.cc[k] = 0; }
template <class Cfg> MKT_HD uint32_t fast_chr_slot(FastState<Cfg>& st, ChrTab* tab, const TextView& tv, uint32_t off, uint32_t len, uint32_t* err) {
    static_assert(kChrSlots <= 8192, "slot in 13 bits");
    if (len == 0u || len > 6u) return chr_slot(tab, tv, off, len, err);           // long names: the table itself (hash + probe)
    // names of up to 6 bytes (chr1 .. chr22, chrX, chrY, chrM, chrEBV): the bytes themselves are the cache key
    const uint64_t name = ((uint64_t)win_load4(tv, off) | ((uint64_t)win_load4(tv, off + 4u) << 32)) & ((1ull << (8u * len)) - 1ull);
    // 64 entries, two ways (idx, idx ^ 1): with this multiplier the main chromosomes of hg38 / mm10 / Ensembl naming
    // (chr1..chr22 chrX chrY chrM, 1..22 X Y MT) all find a place, so after the first tiles every lookup is an LDS hit
    const uint32_t idx = (uint32_t)((name * 0x9E3779B1ull) >> 30) & 63u;
    const uint64_t e0 = st.cc[idx], e1 = st.cc[idx ^ 1u];
    if ((e0 >> 63) && (e0 & 0xFFFFFFFFFFFFull) == name) return (uint32_t)(e0 >> 48) & 8191u;
    if ((e1 >> 63) && (e1 & 0xFFFFFFFFFFFFull) == name) return (uint32_t)(e1 >> 48) & 8191u;
    const uint32_t s = chr_slot(tab, tv, off, len, err);
    st.cc[(e0 >> 63) && !(e1 >> 63) ? (idx ^ 1u) : idx] = name | ((uint64_t)s << 48) | (1ull << 63);
    return s;
}
template <class Cfg> MKT_HD void fast_reset(FastState<Cfg>& st) {
    st.NL = 0; st.first_idx = 0; st.end_idx = 0; st.abn = 0; st.last_line_end = kUnknown; st.nslot = 0; st.c_t0 = 0; st.c_t1 = 0;
    for (int k = 0; k < 4; ++k) st.m_emit[k] = 0;
    st.region_pair0 = 0; st.region_sam0 = 0; st.region_id = 0;
    for (int k = 0; k < (int)C_COUNT; ++k) st.cnt[k] = 0;
}
// lines that take part: a trailing window-cut halo line does not
template <class Cfg> MKT_HD uint32_t fast_nle(const FastState<Cfg>& st) {
    return (st.NL && (st.bits[st.NL - 1] & LB_CUT)) ? st.NL - 1u : st.NL;
}
template <class Cfg> MKT_HD uint32_t fast_line_end(const FastState<Cfg>& st, const TileGeom& G, uint32_t i) {
    return i + 1 < st.NL ? G.w0 + st.goff[i + 1] - 1u : st.last_line_end;
}

// the two 64-bit SEPARATOR words of line i's head (bit k <-> byte k of the line is <= 0x20), from its head-mask row
template <class Cfg> MKT_HD void fast_head_ws(const FastState<Cfg>& st, uint32_t i, uint64_t& ws0, uint64_t& ws1) {
    const uint64_t* row = reinterpret_cast<const uint64_t*>(st.u.m.hmask[i]);
    const uint64_t A = row[0], B = row[1];
    const uint32_t sh = (uint32_t)st.off16[i] - mul24(i, (uint32_t)Cfg::HSTRIDE);  // 0 (first line of a block) .. 16
    ws0 = sh ? ((A >> sh) | (B << (64u - sh))) : A;
    ws1 = B >> sh;
}
// one head chunk: the 16 text bytes at window offset r0 (multiple of 16); bytes at or past the block end read as 0 and are
// no separators.  Separator bits: byte <= 0x20 (tab, newline, the other whitespace and control bytes)
template <class Cfg> MKT_HD void fast_head_chunk_ref(FastState<Cfg>& st, const uint8_t* text, uint32_t n, const TileGeom& G, uint32_t i, uint32_t c) {
    const uint32_t r0 = 16u * ((uint32_t)st.hv16[i] + c);
    uint32_t m = 0;
    for (uint32_t b = 0; b < 16u; ++b) {
        const uint64_t g = (uint64_t)G.w0 + r0 + b;
        const uint8_t ch = g < n ? text[g] : 0;
        st.win[i * Cfg::HSTRIDE + 16u * c + b] = ch;
        if (g < n && ch <= 0x20u) m |= 1u << b;
    }
    st.u.m.hmask[i][c] = (uint16_t)m;
}

// ---- where line i starts inside its head row ------------------------------------------------------------
// Row i begins with the 16-byte vector that holds the newline in front of the line (the scan only recorded WHICH vectors
// hold one).  Returns 1 + the byte of that newline (1..16); 0 for the line that opens the block (nothing in front of
// it).  *multi: the vector holds a second newline, i.e. a line of fewer than 16 bytes that the table does not know.
MKT_HD uint32_t nl_flags_exact(uint32_t x) {                      // 0x80 in every byte of x that is '\n'
    const uint32_t y = (x ^ 0x0A0A0A0Au), z = y & 0x7F7F7F7Fu;
    return ~((z + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}
template <class Cfg> MKT_HD uint32_t fast_row_start(const FastState<Cfg>& st, const TileGeom& G, uint32_t i, bool* multi) {
    *multi = false;
    if (i == 0u && G.w0 == 0u) return 0u;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(&st.win[mul24(i, (uint32_t)Cfg::HSTRIDE)]);
    const uint32_t f0 = nl_flags_exact(row[0]), f1 = nl_flags_exact(row[1]), f2 = nl_flags_exact(row[2]), f3 = nl_flags_exact(row[3]);
    // one bit per byte (the flags sit at bit 7 of every byte; dword j shifted right by 7 - j: byte b of dword j -> bit 8 b + j)
    const uint32_t m = (f0 >> 7) | (f1 >> 6) | (f2 >> 5) | (f3 >> 4);
    *multi = (m & (m - 1u)) != 0u;
    const uint32_t p = f0 ? (ctz32(f0) >> 3) : (f1 ? 4u + (ctz32(f1) >> 3) : (f2 ? 8u + (ctz32(f2) >> 3) : 12u + (ctz32(f3 | 0x80000000u) >> 3)));
    return p + 1u;                                                             // first newline of the vector (none: cannot happen)
}

// ---- parse line i ------------------------------------------------------------------------------
// Straight-line record parser of the lean path (every lane of a wave runs the same few hundred instructions; the generic
// parsers of mkt_core.h loop per byte / per token with trip counts that differ from lane to lane).
//
// A line is LEAN when, up to the end of its sixth token: its first byte is no separator (byte <= 0x20), every token is
// followed by exactly one tab and a token byte, and the sixth token ends at a tab, at the line's newline or at the end of the
// block.  The six tokens are then exactly what `ss >> a >> b ...` (pairutil.h:152-161) extracts: the only bytes <= 0x20 the
// reference meets on its way are those tabs.  Everything else is LP_ODD / LP_LONG: the tile goes to the generic kernel.
enum { LP_OK = 1, LP_NOREC = 2, LP_LONG = 0, LP_ODD = -1 };

// four ASCII digits, text order = byte order (byte 0 the most significant digit) -> value; ok cleared when a byte is no digit
MKT_HD uint32_t dec4(uint32_t w, bool& ok) {
    const uint32_t d = w ^ 0x30303030u;
    if (((d + 0x76767676u) | d) & 0x80808080u) ok = false;                     // some byte > 9
    const uint32_t x = (d << 3) + (d << 1) + (d >> 8);                          // byte 0: 10 b0 + b1, byte 2: 10 b2 + b3 (no carries: <= 99)
    return mul24(x & 0xFFu, 100u) + ((x >> 16) & 0xFFu);
}
// unsigned decimal token of 1..4 bytes starting at head-store offset a
MKT_HD uint32_t lean_uint4(const TextView& tv, uint32_t a, uint32_t len, bool& ok) {
    const uint32_t sh = 8u * (4u - len);                                        // 0, 8, 16, 24
    const uint32_t w = (win_load4(tv, a) << sh) | (0x30303030u & ((1u << sh) - 1u));      // right aligned, '0' in front
    return dec4(w, ok);
}
// unsigned decimal token of 1..10 bytes ENDING at head-store offset e (exclusive); the 12 bytes in front of e are readable
MKT_HD uint32_t lean_uint12(const TextView& tv, uint32_t e, uint32_t len, bool& ok) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tv.win);
    const uint32_t a = e - 12u, k = a >> 2, sh = a & 3u;
    const uint32_t q0 = w[k], q1 = w[k + 1], q2 = w[k + 2], q3 = w[k + 3];
    uint32_t x0, x1, x2;
#if defined(__HIP_DEVICE_COMPILE__)
    x0 = __builtin_amdgcn_alignbyte(q1, q0, sh); x1 = __builtin_amdgcn_alignbyte(q2, q1, sh); x2 = __builtin_amdgcn_alignbyte(q3, q2, sh);
#else
    x0 = sh ? ((q0 >> (8u * sh)) | (q1 << (32u - 8u * sh))) : q0;
    x1 = sh ? ((q1 >> (8u * sh)) | (q2 << (32u - 8u * sh))) : q1;
    x2 = sh ? ((q2 >> (8u * sh)) | (q3 << (32u - 8u * sh))) : q2;
#endif
    const uint32_t lead = 12u - len;                                            // bytes in front of the token: read as '0'
    const uint32_t n0 = lead < 4u ? lead : 4u, n1 = lead < 4u ? 0u : (lead < 8u ? lead - 4u : 4u), n2 = lead < 8u ? 0u : lead - 8u;
    const uint32_t m0 = n0 >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n0)) - 1u), m1 = n1 >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n1)) - 1u),
                   m2 = (1u << (8u * n2)) - 1u;                                  // n2 <= 3
    x0 = (x0 & ~m0) | (0x30303030u & m0); x1 = (x1 & ~m1) | (0x30303030u & m1); x2 = (x2 & ~m2) | (0x30303030u & m2);
    const uint32_t A = dec4(x0, ok), B = dec4(x1, ok), C = dec4(x2, ok);
    const uint64_t v = (uint64_t)A * 100000000ull + (uint64_t)(mul24(B, 10000u) + C);
    if (v > 0xFFFFFFFFull) ok = false;
    return (uint32_t)v;
}
// len bytes at head-store offsets a and b equal?  (eight bytes per step; reads up to 11 bytes past the ranges)
MKT_HD bool lean_eq(const TextView& tv, uint32_t a, uint32_t b, uint32_t len) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tv.win);
    const uint32_t ka = a >> 2, sa = a & 3u, kb = b >> 2, sb = b & 3u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < len; i += 8u) {
        const uint32_t j = i >> 2;
        const uint32_t a0 = w[ka + j], a1 = w[ka + j + 1], a2 = w[ka + j + 2], b0 = w[kb + j], b1 = w[kb + j + 1], b2 = w[kb + j + 2];
        uint32_t x0, x1, y0, y1;
#if defined(__HIP_DEVICE_COMPILE__)
        x0 = __builtin_amdgcn_alignbyte(a1, a0, sa); x1 = __builtin_amdgcn_alignbyte(a2, a1, sa);
        y0 = __builtin_amdgcn_alignbyte(b1, b0, sb); y1 = __builtin_amdgcn_alignbyte(b2, b1, sb);
#else
        x0 = sa ? ((a0 >> (8u * sa)) | (a1 << (32u - 8u * sa))) : a0; x1 = sa ? ((a1 >> (8u * sa)) | (a2 << (32u - 8u * sa))) : a1;
        y0 = sb ? ((b0 >> (8u * sb)) | (b1 << (32u - 8u * sb))) : b0; y1 = sb ? ((b1 >> (8u * sb)) | (b2 << (32u - 8u * sb))) : b1;
#endif
        uint32_t d0 = x0 ^ y0, d1 = x1 ^ y1;
        const uint32_t rem = len - i;                                           // >= 1
        if (rem < 4u) { d0 &= (1u << (8u * rem)) - 1u; d1 = 0; }
        else if (rem < 8u) { d1 &= (1u << (8u * (rem - 4u))) - 1u; }
        acc |= d0 | d1;
    }
    return acc == 0u;
}

// One CIGAR operation on the walk state, as CigarWalk::op but as selects instead of branches (the lanes of a wave meet
// all operation types at once).  pairutil.h:63-126.
struct LeanCigar {
    int32_t index, cur, lastRight;
    bool bad;
};
MKT_HD void lean_cigar_op(Rec& r, LeanCigar& w, uint32_t c, int32_t v, bool last) {
    const bool hs = c == 'H' || c == 'S', m = c == 'M', d = c == 'D', n = c == 'N', known = hs || m || d || n || c == 'I';
    const bool live = !w.bad;
    const bool first = w.index == 0;
    if (live && (!known || (hs && !last && !first))) w.bad = true;
    const bool go = live && !w.bad;
    const bool md = go && (m || d), nn = go && n;
    if (go && hs && last) r.rclip = v;
    if (go && hs && !last && first) r.lclip = v;
    if (go && m) r.mappable += v;
    w.cur += (md || nn) ? v : 0;
    w.lastRight = md ? w.cur - 1 : (nn ? 0 : w.lastRight);
    if (md && first) r.right0 = w.lastRight;
    if (md && w.index == 1) r.right1 = w.lastRight;
    if (nn && first) { r.left1 = w.cur; r.right1 = 0; }
    w.index += nn ? 1 : 0;
}

// Parses the record whose line starts at head-store offset `off`: sp0 / sp1 = separator bits of its bytes 0..63 / 64..127
// (zero beyond the `room` head bytes of the line), reach = bytes from the line start to the end of the block.
MKT_HD int parse_record_lean(const TextView& tv, uint32_t off, const Params& P, Rec& r, uint64_t sp0, uint64_t sp1, uint32_t reach) {
    // a line without its newline at the very end of the block ends at `reach`
    if (reach < 64u) sp0 |= 1ull << reach; else if (reach < 128u) sp1 |= 1ull << (reach - 64u);
    // the first six separators.  The first one (end of the QNAME) from the first word; the other five from the 64 bytes behind
    // it, all in one word (FLAG .. CIGAR of a lean line are shorter than that; a line that is not takes the LONG way out)
    uint32_t p[6];
    if (sp0 == 0ull) return LP_LONG;
    p[0] = ctz64(sp0);
    if (p[0] >= 63u) return LP_LONG;
    const uint32_t b0 = p[0] + 1u;                                              // 1 .. 63
    uint64_t rest = (sp0 >> b0) | (sp1 << (64u - b0));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 1; k < 6; ++k) {
        p[k] = rest ? b0 + ctz64(rest) : 0xFFFFu;
        rest &= rest - 1ull;
    }
    uint32_t c[6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 6; ++k) c[k] = tv.win[off + (p[k] == 0xFFFFu ? 0u : p[k])];      // (none: the line's first byte, ignored below)
    // the first thing that is not "a token, then one tab": bit k of the masks <-> separator k
    uint32_t miss = 0, adj = 0, nl = 0, odd = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 6; ++k) {
        const uint32_t prev = k ? p[k - 1] + 1u : 0u;
        miss |= (p[k] == 0xFFFFu ? 1u : 0u) << k;
        adj |= (p[k] == prev ? 1u : 0u) << k;
        nl |= ((p[k] == reach || c[k] == '\n') ? 1u : 0u) << k;
        odd |= (c[k] != '\t' ? 1u : 0u) << k;
    }
    const uint32_t ev = (miss | adj | nl | odd) & 0x3Fu;
    if (ev) {
        const uint32_t f = 1u << ctz32(ev);
        if (miss & f) return LP_LONG;                                           // the head (or the 64-byte reach behind the QNAME) ends before the sixth token does
        if (adj & f) return (nl & f) ? LP_NOREC : LP_ODD;                        // empty token: end of a short line, or a run of separators
        if (nl & f) { if (f != 32u) return LP_NOREC; }                          // a complete line with fewer than six tokens is no record
        else return LP_ODD;                                                     // a separator that is no tab
    }
    bool ok = tv.win[off] != '@';
    r.qn_off = 0; r.qn_len = p[0];
    r.rn_off = p[1] + 1u; r.rn_len = p[2] - p[1] - 1u;
    const uint32_t fl_len = p[1] - p[0] - 1u, pos_len = p[3] - p[2] - 1u, mq_len = p[4] - p[3] - 1u;
    // (12 readable bytes in front of a long field's end: true for every line whose QNAME has five bytes or more)
    if (pos_len > 10u) { ok = false; r.pos = 0; }                               // (as parse_record: more than ten digits never fit)
    else if (p[3] < 12u) return LP_ODD;
    else r.pos = lean_uint12(tv, off + p[3], pos_len, ok);
    if (fl_len <= 4u) r.flag = lean_uint4(tv, off + p[0] + 1u, fl_len, ok);
    else if (fl_len > 10u) { ok = false; r.flag = 0; }
    else if (p[1] < 12u) return LP_ODD;
    else r.flag = lean_uint12(tv, off + p[1], fl_len, ok);
    if (mq_len <= 4u) r.mapq = lean_uint4(tv, off + p[3] + 1u, mq_len, ok);
    else if (mq_len > 10u) { ok = false; r.mapq = 0; }
    else r.mapq = lean_uint12(tv, off + p[4], mq_len, ok);                      // (p[4] > p[3] >= 12)
    // CIGAR, per OPERATION: bit i of `dig` <-> byte i of the token is a decimal digit; every other byte is an operation whose
    // count is the digit run in front of it (up to four digits: the four bytes in front of the operation as one dword; longer
    // runs -- the intron of a spliced alignment -- as the twelve bytes in front of it)
    const uint32_t cs = off + p[4] + 1u, clen = p[5] - p[4] - 1u;
    if (clen > 32u || p[4] < 12u) return LP_ODD;
    uint32_t dig = 0;
    for (uint32_t k = 0; k < clen; k += 4u) dig |= digit_bits4(win_load4(tv, cs + k)) << k;
    const uint32_t cmask = clen >= 32u ? 0xFFFFFFFFu : ((1u << clen) - 1u);
    uint32_t ops = ~dig & cmask;
    LeanCigar w;
    w.index = 0; w.cur = (int32_t)r.pos; w.lastRight = 0; w.bad = false;
    r.left0 = (int32_t)r.pos;
    uint32_t run0 = 0;                                              // first byte of the digit run before the next operation
    bool odd_count = false;
    while (ops) {
        const uint32_t q = ctz32(ops);
        ops &= ops - 1u;
        const uint32_t L = q - run0;                                // digits of the count
        bool vok = true;
        uint32_t value;
        if (L <= 4u) {
            const uint32_t sh = 8u * (4u - L);                      // bytes q-4 .. q-1, the run right aligned, '0' in front of it
            const uint32_t x = win_load4(tv, cs + q - 4u);
            value = L ? dec4((x & ~((1u << sh) - 1u)) | (0x30303030u & ((1u << sh) - 1u)), vok) : 0u;
        } else if (L <= 9u) value = lean_uint12(tv, cs + q, L, vok);
        else { value = 0; odd_count = true; }                      // (ten digits and more overflow the reference's int)
        lean_cigar_op(r, w, tv.win[cs + q], (int32_t)value, q + 1u == clen);
        run0 = q + 1u;
    }
    if (odd_count) return LP_ODD;
    // (digits behind the last operation: the reference leaves them in `value` and ignores them)
    if (!w.bad && w.lastRight != 0) { r.segCnt = w.index + 1; r.rightLast = w.lastRight; }
    r.survive = ok && !(r.flag & 0x700u) && r.mapq >= P.min_mapq;
    return LP_OK;
}

template <class Cfg> MKT_HD void fast_parse(FastState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G, uint32_t i) {
    bool multi;
    const uint32_t soff = fast_row_start(st, G, i, &multi);
    if (multi) st.abn = AB_SHORT_LINE;
    const uint32_t goff = 16u * (uint32_t)st.hv16[i] + soff;
    const uint32_t off = mul24(i, (uint32_t)Cfg::HSTRIDE) + soff;  // line start in the head store
    st.goff[i] = (uint16_t)goff;
    st.off16[i] = (uint16_t)off;
    const uint32_t gl = G.w0 + goff;                               // ... and in the block
    if (gl >= G.w1) {                                              // the newline is the window's last byte: no line of this window
        st.bits[i] = LB_CUT;                                       // (always the table's last entry; the line before it ends right here)
        if (i + 1u != st.NL) st.abn = AB_SHORT_LINE;
        return;
    }
    Rec r;
    uint64_t ws0, ws1;
    fast_head_ws(st, i, ws0, ws1);
    rec_clear(r, off);
    const int pf = parse_record_lean(tv, off, P, r, ws0, ws1, tv.n - gl);
    if (pf == LP_LONG || pf == LP_ODD) {
        if (gl >= G.t1 && i + 1 == st.NL && G.w1 < tv.n) st.bits[i] = LB_CUT;    // last halo line: ignored (a group reaching it is deferred)
        else { st.bits[i] = 0; st.abn = pf == LP_LONG ? AB_LONG : AB_TAB; }      // fields beyond the head / an odd shape: generic kernel
        return;
    }
    st.rc.f.pos[i] = r.pos; st.rc.f.lclip[i] = (uint32_t)r.lclip; st.rc.f.rclip[i] = (uint32_t)r.rclip; st.rc.f.mappable[i] = (uint32_t)r.mappable;
    st.rc.f.right0[i] = (uint32_t)r.right0; st.rc.f.left1[i] = (uint32_t)r.left1; st.rc.f.right1[i] = (uint32_t)r.right1;
    st.flag[i] = (uint16_t)(r.flag & 0xFFFFu);
    st.qn_off[i] = (uint8_t)r.qn_off; st.qn_len[i] = (uint8_t)r.qn_len; st.rn_off[i] = (uint8_t)r.rn_off; st.rn_len[i] = (uint8_t)r.rn_len;
    st.segCnt[i] = (uint8_t)(r.segCnt > 4 ? 4 : r.segCnt);
    uint8_t b = r.survive ? LB_SURVIVE : 0;
    if (i > 0 && pf == LP_OK) {
        // same QNAME token as the line before: its first r.qn_len bytes and the tab behind them (its own lane sets off16[i - 1]
        // in this same phase: recompute).  A line before that is no lean record only ever costs the direct comparison in
        // fast_is_start (or defers the tile by itself).
        bool pm;
        const uint32_t psoff = fast_row_start(st, G, i - 1u, &pm);
        const uint32_t poff = mul24(i - 1u, (uint32_t)Cfg::HSTRIDE) + psoff;
        const uint32_t ql = r.qn_len;
        if (ql + 1u > (uint32_t)Cfg::HEADB - psoff || ql > 100u) st.abn = AB_PREV_HEAD;      // beyond the previous line's head (lean_eq reads 11 bytes past the names)
        else if (lean_eq(tv, off, poff, ql + 1u)) b |= LB_EQPREV;
    }
    st.bits[i] = b;
}

// ---- line masks ------------------------------------------------------------------------------------
MKT_HD bool mask_bit(const uint64_t* m, uint32_t i) { return (m[i >> 6] >> (i & 63u)) & 1ull; }
MKT_HD uint64_t mask_range(uint32_t w, uint32_t lo, uint32_t hi) {        // bits of word w inside [lo, hi)
    const uint32_t b = w << 6;
    if (hi <= b || lo >= b + 64u) return 0ull;
    const uint32_t l = lo > b ? lo - b : 0u, h = hi < b + 64u ? hi - b : 64u;
    const uint64_t up = h >= 64u ? ~0ull : ((1ull << h) - 1ull);
    return up & ~((1ull << l) - 1ull);
}
// 64 lines of a mask as one word: bit k = line start + k (lines past the table read as 0)
MKT_HD uint64_t mask_win(const uint64_t* m, uint32_t start) {
    const uint32_t w = start >> 6, s = start & 63u;
    uint64_t x = m[w] >> s;
    if (s && w < 3u) x |= m[w + 1u] << (64u - s);
    return x;
}
// the same ending at line i: bit 63 = line i, bit 62 = line i - 1, ...
MKT_HD uint64_t mask_win_back(const uint64_t* m, uint32_t i) {
    return i >= 63u ? mask_win(m, i - 63u) : (m[0] << (63u - i));
}
MKT_HD uint32_t mask_next(const uint64_t* m, uint32_t from, uint32_t limit) {   // lowest set bit in [from, limit), or limit
    for (uint32_t w = from >> 6; (w << 6) < limit && w < 4u; ++w) {
        const uint64_t x = m[w] & mask_range(w, from, limit);
        if (x) return (w << 6) + ctz64(x);
    }
    return limit;
}
// host side of what the kernel does with __ballot after fast_parse
template <class Cfg> MKT_HD void fast_build_masks(FastState<Cfg>& st) {
    for (int k = 0; k < 4; ++k) st.m_surv[k] = st.m_eqp[k] = st.m_r1[k] = st.m_r2[k] = 0;
    for (uint32_t i = 0; i < st.NL; ++i) {
        const uint64_t bit = 1ull << (i & 63u);
        if (st.bits[i] & LB_SURVIVE) st.m_surv[i >> 6] |= bit;
        if (st.bits[i] & LB_EQPREV) st.m_eqp[i >> 6] |= bit;
        if ((st.bits[i] & LB_SURVIVE) && (st.flag[i] & 64u)) st.m_r1[i >> 6] |= bit;
        if ((st.bits[i] & LB_SURVIVE) && !(st.flag[i] & 64u) && (st.flag[i] & 128u)) st.m_r2[i >> 6] |= bit;
    }
}

// ---- does surviving line i open a group?  (its QNAME differs from the previous SURVIVING line's) --------
template <class Cfg> MKT_HD bool fast_is_start(FastState<Cfg>& st, const TextView& tv, const TileGeom& G, uint32_t i) {
    const uint64_t sv = mask_win_back(st.m_surv, i) & ~(1ull << 63);      // the 63 lines before line i
    if (!sv) {
        if (G.w0 == 0 && i < 64u) return true;        // first surviving line of the block
        st.abn = AB_NO_PREV;                                   // the previous surviving line is before the window (or > 63 lines back)
        return false;
    }
    const uint32_t d = clz64(sv), p = i - d;          // previous surviving line
    // every line in (p, i] has the QNAME token of the line before it  =>  equal by transitivity
    if (((~mask_win_back(st.m_eqp, i)) >> (64u - d)) == 0ull) return false;
    if (d == 1u) return true;                         // adjacent surviving lines with different tokens
    return !text_eq<true>(tv, (uint32_t)st.off16[i] + st.qn_off[i], st.qn_len[i], (uint32_t)st.off16[p] + st.qn_off[p], st.qn_len[p]);
}

template <class Cfg> MKT_HD Seg fast_seg(const FastState<Cfg>& st, uint32_t idx) {
    Seg s;
    s.segCnt = st.segCnt[idx]; s.lclip = (int32_t)st.rc.f.lclip[idx]; s.rclip = (int32_t)st.rc.f.rclip[idx]; s.mappable = (int32_t)st.rc.f.mappable[idx];
    s.left0 = (int32_t)st.rc.f.pos[idx]; s.left1 = (int32_t)st.rc.f.left1[idx]; s.right0 = (int32_t)st.rc.f.right0[idx]; s.right1 = (int32_t)st.rc.f.right1[idx];
    s.rightLast = s.segCnt == 2 ? s.right1 : s.right0;
    s.flag = st.flag[idx]; s.pos = st.rc.f.pos[idx];
    s.chr_off = (uint32_t)st.off16[idx] + st.rn_off[idx]; s.chr_len = st.rn_len[idx];      // head-store offset
    {   // the first eight name bytes, big endian, zero padded
        const uint32_t* w = reinterpret_cast<const uint32_t*>(st.win);
        const uint32_t k = s.chr_off >> 2, sh = s.chr_off & 3u;
        const uint32_t q0 = w[k], q1 = w[k + 1], q2 = w[k + 2];
        uint32_t x0, x1;
#if defined(__HIP_DEVICE_COMPILE__)
        x0 = __builtin_amdgcn_alignbyte(q1, q0, sh); x1 = __builtin_amdgcn_alignbyte(q2, q1, sh);
#else
        x0 = sh ? ((q0 >> (8u * sh)) | (q1 << (32u - 8u * sh))) : q0; x1 = sh ? ((q1 >> (8u * sh)) | (q2 << (32u - 8u * sh))) : q1;
#endif
        const uint32_t n = s.chr_len;
        if (n < 4u) { x0 &= (1u << (8u * n)) - 1u; x1 = 0; }
        else if (n < 8u) x1 &= (1u << (8u * (n - 4u))) - 1u;
        s.chr_key = ((uint64_t)bswap32(x0) << 32) | bswap32(x1);
    }
    return s;
}

// ---- the group opened by line i: extent, members and record slots from the masks; classification --------
template <class Cfg> MKT_HD void fast_group(FastState<Cfg>& st, const TextView& tv, const Params& P, const TileGeom& G, uint32_t i) {
    auto& g = st.u.g;
    g.g_info[i] = 0; g.g_plen[i] = 0; g.g_slen[i] = 0;
    if (!mask_bit(st.m_start, i)) return;
    const uint32_t NLe = fast_nle(st);
    // lines i .. i+63 as one word per mask; the group ends before the next start bit
    const uint64_t nx = mask_win(st.m_start, i) >> 1;
    uint32_t len;
    if (nx) len = ctz64(nx) + 1u;
    else {
        len = NLe - i;
        if (G.w1 < tv.n || len > 64u) { st.abn = AB_OPEN_GROUP; return; }        // the group may continue past the window
    }
    const uint64_t in = len >= 64u ? ~0ull : ((1ull << len) - 1ull);
    const uint64_t mem = mask_win(st.m_surv, i) & in;                 // bit 0 (line i) is set
    const uint32_t nmem = popc64(mem);
    uint32_t n1 = 0, n2 = 0, sa, sb = 0xFFFFu, sc = 0xFFFFu, sd = 0xFFFFu;
    if (P.mode == MODE_FLASH) {
        sa = i;
        const uint64_t r = mem & (mem - 1ull);
        if (r) sb = i + ctz64(r);
    } else {
        uint64_t r = mask_win(st.m_r1, i) & in;
        n1 = popc64(r);
        sa = r ? i + ctz64(r) : 0xFFFFu; r &= r - 1ull;
        if (r) sb = i + ctz64(r);
        r = mask_win(st.m_r2, i) & in;
        n2 = popc64(r);
        if (r) { sc = i + ctz64(r); r &= r - 1ull; }
        if (r) sd = i + ctz64(r);
    }
    // a member whose line end is not known (last line of the table) or that lacks its final newline: generic kernel
    if (i + 64u - clz64(mem) == st.NL && (st.last_line_end == kUnknown || st.last_line_end >= tv.n)) st.abn = AB_LAST_LINE;
    Verdict v;
    {
        Seg a = sa != 0xFFFFu ? fast_seg(st, sa) : seg_zero();
        Seg b = sb != 0xFFFFu ? fast_seg(st, sb) : seg_zero();
        if (P.mode == MODE_FLASH) v = classify_flash<true>(tv, nmem, a, b, P.ratio);
        else {
            Seg c = sc != 0xFFFFu ? fast_seg(st, sc) : seg_zero();
            Seg d = sd != 0xFFFFu ? fast_seg(st, sd) : seg_zero();
            v = classify_unc<true>(tv, n1, n2, a, b, c, d, P.ratio);
        }
    }
    uint32_t info = GI_START | (v.counter & GI_COUNTER);
    if (v.emit) {
        info |= GI_EMIT;
        if (v.sA == '-') info |= GI_SA_MINUS;
        if (v.sB == '-') info |= GI_SB_MINUS;
        const uint32_t slot = lds_inc(&st.nslot);
        if (slot >= (uint32_t)Cfg::GCAP) { st.abn = AB_GCAP; return; }
        g.g_slot[i] = (uint8_t)slot;
        const uint32_t ql = st.qn_len[i], dA = dec_digits(v.posA), dB = dec_digits(v.posB);
        g.l_qa[slot] = (uint16_t)(st.off16[i] + st.qn_off[i]);
        g.l_ca[slot] = (uint16_t)v.chrA_off; g.l_cb[slot] = (uint16_t)v.chrB_off;
        const uint32_t e0 = ql + 1u, e1 = e0 + v.chrA_len + 1u, e2 = e1 + dA + 1u, e3 = e2 + v.chrB_len + 1u;
        g.l_e0[slot] = (uint16_t)e0; g.l_e1[slot] = (uint16_t)e1; g.l_e2[slot] = (uint16_t)e2; g.l_e3[slot] = (uint16_t)e3;
        g.l_posA[slot] = v.posA; g.l_posB[slot] = v.posB;
        {   // literals "<posA>\t" and "<posB>\t<sA>\t<sB>\n", text order = little-endian byte order
            uint64_t w0, w1;
            dec_lit(v.posA, dA, (uint64_t)'\t', w0, w1);
            g.l_litA[slot][0] = w0; g.l_litA[slot][1] = w1;
            const uint64_t tail = (uint64_t)'\t' | ((uint64_t)v.sA << 8) | ((uint64_t)'\t' << 16) | ((uint64_t)v.sB << 24) | ((uint64_t)'\n' << 32);
            dec_lit(v.posB, dB, tail, w0, w1);
            g.l_litB[slot][0] = w0; g.l_litB[slot][1] = w1;
        }
        g.g_plen[i] = (uint16_t)(e3 + dB + 5u);
        if (P.write_sam) {                             // the group's surviving lines go to the .sam
            const uint32_t w = i >> 6, sh = i & 63u;
            lds_or64(&st.m_emit[w], mem << sh);
            if (sh && (mem >> (64u - sh)) && w < 3u) lds_or64(&st.m_emit[w + 1u], mem >> (64u - sh));
        }
    }
    g.g_info[i] = info;
}

// bytes line i contributes to the tile's .sam output
template <class Cfg> MKT_HD uint32_t fast_line_sam(const FastState<Cfg>& st, const TileGeom& G, uint32_t i) {
    if (!mask_bit(st.m_emit, i)) return 0u;
    return fast_line_end(st, G, i) + 1u - (G.w0 + st.goff[i]);
}

// ---- account for the group opened by line i (counters, self-circle entry) ------------------------
template <class Cfg> MKT_HD void fast_account(FastState<Cfg>& st, const OutPtrs& out, uint32_t tile, uint32_t i) {
    const uint32_t info = st.u.g.g_info[i];
    if (!(info & GI_START)) return;
    const uint32_t counter = info & GI_COUNTER;
    if (counter) lds_add(&st.cnt[counter], 1u);
    if (counter == C_SELFCIRCLE) {
        uint64_t k = (uint64_t)st.base.sc + st.u.g.x_sc[i];
        if (k < out.sc_cap) *MKT_GLOBAL(uint64_t, out.sc + k) = ((uint64_t)tile << 32) | st.u.g.x_grp[i];
        else lds_or(&st.abn, E_SC_CAP << 8);          // reported as an error bit by the kernel (OR: other lanes and the claim step set bits too)
    }
    if ((info & GI_EMIT) && out.keys) {               // extension: duplicate-marking key of this pair
        const auto& g = st.u.g;
        const uint32_t slot = g.g_slot[i];
        const uint64_t k = (uint64_t)st.base.emitted + g.x_emit[i];
        if (k < out.keys_cap) {
            TextView tv;
            tv.g = nullptr; tv.n = 0; tv.win = st.win; tv.w0 = 0; tv.wlen = Cfg::HW; tv.nlm = nullptr; tv.wsm = nullptr;
            uint32_t err = 0;
            const uint32_t sa = fast_chr_slot(st, out.chr, tv, g.l_ca[slot], (uint32_t)(g.l_e1[slot] - g.l_e0[slot] - 1u), &err);
            const uint32_t sb = fast_chr_slot(st, out.chr, tv, g.l_cb[slot], (uint32_t)(g.l_e3[slot] - g.l_e2[slot] - 1u), &err);
            const uint32_t lane = out.key_lanes ? qname_lane(tv, (uint32_t)st.off16[i] + st.qn_off[i], st.qn_len[i]) : 0u;
            *MKT_GLOBAL(KeyRec, out.keys + k) = make_key(sa, g.l_posA[slot], sb, g.l_posB[slot], (info & GI_SA_MINUS) != 0, (info & GI_SB_MINUS) != 0, tile, g.x_emit[i], lane);
            if (err) lds_or(&st.abn, err << 8);
        } else lds_or(&st.abn, E_SC_CAP << 8);
    }
}
template <class Cfg> MKT_HD void fast_last(const FastState<Cfg>& st, const TileGeom& G, TileLast* tl, uint32_t i) {
    const auto& g = st.u.g;
    const uint32_t info = g.g_info[i];
    if (!(info & GI_START)) return;
    if ((uint32_t)g.x_grp[i] + 1u != st.sums.groups) return;
    tl->counter = info & GI_COUNTER;
    tl->pair_bytes = g.g_plen[i];
    {   // .sam bytes of this group: its surviving lines up to the next group's first line
        uint32_t sb = 0;
        if (mask_bit(st.m_emit, i)) {
            const uint32_t end = mask_next(st.m_start, i + 1u, fast_nle(st));
            for (uint32_t k = i; k < end; ++k) if (mask_bit(st.m_surv, k)) sb += fast_line_end(st, G, k) + 1u - (G.w0 + st.goff[k]);
        }
        tl->sam_bytes = sb;
    }
    tl->pair_off = st.base.pair_bytes - st.region_pair0 + g.x_pair[i];
    tl->sam_off = (uint32_t)(st.base.sam_bytes - st.region_sam0 + g.x_sam[i]);
    tl->region = st.region_id; tl->pad = 0;
    tl->valid = 1;
}

// Emitting group whose .pairs line holds byte k of the tile's output (ordinal in em_idx)
template <class Cfg> MKT_HD uint32_t fast_pair_find(const FastState<Cfg>& st, uint32_t k) {
    const auto& g = st.u.g;
    uint32_t lo = 0, hi = st.sums.emitted;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.x_pair[g.em_idx[mid]] <= k) lo = mid; else hi = mid;
    }
    return lo;
}
// byte o of the .pairs line of the group in `slot`.  Branch-free: every run, the two literals included, is a byte range of
// this state object in LDS (the literal words hold their text in memory order), so the byte is ONE load at
// (start of the run that holds o) + o; picking the run is four selects.
template <class Cfg> struct FastLayout { uint32_t e0, e1, e2, e3, a0, a1, a2, a3, a4; };
template <class Cfg> MKT_HD FastLayout<Cfg> fast_layout(const FastState<Cfg>& st, uint32_t slot) {
    const auto& g = st.u.g;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(&st);
    const uint32_t w = (uint32_t)(st.win - base);
    FastLayout<Cfg> L;
    L.e0 = g.l_e0[slot]; L.e1 = g.l_e1[slot]; L.e2 = g.l_e2[slot]; L.e3 = g.l_e3[slot];
    L.a0 = w + g.l_qa[slot];                                                                   // state-relative address of byte o, minus o
    L.a1 = w + g.l_ca[slot] - L.e0;
    L.a2 = (uint32_t)(reinterpret_cast<const uint8_t*>(&g.l_litA[slot][0]) - base) - L.e1;
    L.a3 = w + g.l_cb[slot] - L.e2;
    L.a4 = (uint32_t)(reinterpret_cast<const uint8_t*>(&g.l_litB[slot][0]) - base) - L.e3;
    return L;
}
template <class Cfg> MKT_HD uint8_t fast_layout_byte(const FastState<Cfg>& st, const FastLayout<Cfg>& L, uint32_t o) {
    uint32_t a = L.a0;
    a = o >= L.e0 ? L.a1 : a;
    a = o >= L.e1 ? L.a2 : a;
    a = o >= L.e2 ? L.a3 : a;
    a = o >= L.e3 ? L.a4 : a;
    return reinterpret_cast<const uint8_t*>(&st)[(uint32_t)(a + o)];
}
template <class Cfg> MKT_HD uint8_t fast_layout_byte(const FastState<Cfg>& st, uint32_t slot, uint32_t o) {
    return fast_layout_byte(st, fast_layout(st, slot), o);
}
// byte k (0 <= k < sums.pair_bytes) of the tile's .pairs output
template <class Cfg> MKT_HD uint8_t fast_pair_byte(const FastState<Cfg>& st, uint32_t k) {
    const auto& g = st.u.g;
    const uint32_t i = g.em_idx[fast_pair_find(st, k)];
    return fast_layout_byte(st, g.g_slot[i], k - g.x_pair[i]);
}
// ---- emit: ONE lane writes one whole .pairs line ------------------------------------------------------------
// The line is five byte runs of this state object (see FastLayout).  The lane streams them through a 64-bit byte FIFO
// and writes destination-aligned dwords; the bytes in front of the first aligned dword and behind the last one go out
// as single bytes (neighbouring lines own the rest of those dwords).  ~20 instructions per output dword for the one
// lane that runs them, instead of a layout lookup per byte on every lane of the workgroup.
MKT_HD uint32_t state_load4(const uint8_t* base, uint32_t off) {       // four bytes at any offset, from aligned dwords
    const uint32_t* w = reinterpret_cast<const uint32_t*>(base);
    const uint32_t i = off >> 2;
    const uint32_t lo = w[i], hi = w[i + 1];
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
#else
    const uint32_t sh = (off & 3u) * 8u;
    return sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
#endif
}
template <class Cfg> MKT_HD void fast_emit_line(const FastState<Cfg>& st, uint32_t slot, uint32_t plen, uint8_t* dst) {
    const auto& g = st.u.g;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(&st);
    const uint32_t w = (uint32_t)(st.win - base);
    const uint32_t e0 = g.l_e0[slot], e1 = g.l_e1[slot], e2 = g.l_e2[slot], e3 = g.l_e3[slot];
    const uint32_t rs[5] = {w + g.l_qa[slot], w + g.l_ca[slot], (uint32_t)(reinterpret_cast<const uint8_t*>(&g.l_litA[slot][0]) - base),
                            w + g.l_cb[slot], (uint32_t)(reinterpret_cast<const uint8_t*>(&g.l_litB[slot][0]) - base)};
    const uint32_t rl[5] = {e0, e1 - e0, e2 - e1, e3 - e2, plen - e3};
    const uint32_t a = (uint32_t)((uintptr_t)dst & 3u);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst - a);            // the aligned dword that holds the line's first byte
    uint64_t acc = 0;
    uint32_t cnt = a;                                              // its low `a` bytes belong to the line before
    bool first = a != 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 5; ++r) {
        for (uint32_t k = 0; k < rl[r]; k += 4u) {
            uint32_t x = state_load4(base, rs[r] + k);
            const uint32_t take = rl[r] - k < 4u ? rl[r] - k : 4u;
            if (take < 4u) x &= (1u << (8u * take)) - 1u;
            acc |= (uint64_t)x << (8u * cnt);
            cnt += take;
            if (cnt >= 4u) {
                const uint32_t v = (uint32_t)acc;
                if (first) { uint8_t* b = reinterpret_cast<uint8_t*>(d); for (uint32_t q = a; q < 4u; ++q) *MKT_GLOBAL(uint8_t, b + q) = (uint8_t)(v >> (8u * q)); first = false; }
                else *MKT_GLOBAL(uint32_t, d) = v;
                ++d; acc >>= 32; cnt -= 4u;
            }
        }
    }
    {   // what is left: cnt (< 4) bytes of the last, shared dword
        uint8_t* b = reinterpret_cast<uint8_t*>(d);
        for (uint32_t q = first ? a : 0u; q < cnt; ++q) *MKT_GLOBAL(uint8_t, b + q) = (uint8_t)((uint32_t)acc >> (8u * q));
    }
}

}  // namespace mkt
