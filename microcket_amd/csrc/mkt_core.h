// mkt_core.h -- per-line and per-group logic of the sam2pairs hot path, written once for the
// gfx950 kernels (mkt_kernels.hip) and for the host-side tile emulation used by the CPU tests
// (tests/host/tile_emul.cpp).  Nothing here is a product CPU path: the library only ever runs
// this code inside HIP kernels.
//
// Reference behaviour restated (paths relative to /root/reference/src/sam2pairs/):
//   record tokenising + per-line filter   pairutil.h:152-161, sam2pairs.cpp:116-126
//   CIGAR walk                            pairutil.h:63-126
//   integrity tests (float32, quirk Q3)   pairutil.h:180-208
//   stitched classifier                   flash2pairs.h:17-155
//   unstitched classifier                 unc2pairs.h:16-358
//   ordering / self-circle / bins         flash2pairs.h:105-144, unc2pairs.h:310-348
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MKT_HD __host__ __device__ inline
#else
#define MKT_HD inline
#endif

// Output pointers travel through LDS (OutPtrs), where the compiler loses their address space and falls back to FLAT stores; a
// FLAT instruction also counts on the LDS counter (lgkmcnt), so every later LDS read of the lane waits for it.  The outputs
// are global memory: say so.
#if defined(__HIP_DEVICE_COMPILE__)
#define MKT_GLOBAL(T, p) ((__attribute__((address_space(1))) T*)(p))
#else
#define MKT_GLOBAL(T, p) ((T*)(p))
#endif

namespace mkt {

constexpr int kMinClip = 20;          // pairutil.h:54
constexpr int kMaxSelfCircle = 10;    // pairutil.h:57
constexpr int kMaxPairDist = 1000;    // pairutil.h:58
constexpr uint32_t kRefBatch = 1u << 18;   // pairutil.h:48 (only quirk Q2 depends on it)
constexpr uint32_t kUnknown = 0xFFFFFFFFu;

enum Counter : uint32_t { C_NONE = 0, C_LOWMAP, C_MANYHITS, C_UNPAIRED, C_SELFCIRCLE, C_TRANS, C_CIS10K, C_CIS1K, C_CIS0, C_COUNT };
enum Mode : int { MODE_FLASH = 0, MODE_UNC = 1 };

struct Params {
    int mode;
    float ratio;          // min_mapped_ratio as float32 (sam2pairs.cpp:41)
    uint32_t min_mapq;    // compared unsigned (pairutil.h:157)
    int write_sam;
};

// Block text in global memory plus the on-chip window [w0, w0 + wlen) of it.  The window copy is
// padded with >= 16 zero bytes; nlm / wsm are bitmaps over the window (bit r <-> byte w0 + r:
// newline / whitespace), zero beyond wlen and padded by two zero words.
struct TextView {
    const uint8_t* g;     // block base
    uint32_t n;           // block bytes
    const uint8_t* win;   // window copy (LDS on the GPU), 16-byte aligned
    uint32_t w0, wlen;
    const uint64_t* nlm;
    const uint64_t* wsm;
    MKT_HD uint8_t at(uint32_t off) const {
        uint32_t r = off - w0;
        return r < wlen ? win[r] : g[off];
    }
    MKT_HD bool inside(uint32_t off, uint32_t len) const {      // [off, off+len) lies in the window
        uint32_t r = off - w0;
        return r < wlen && len <= wlen - r;
    }
};

// four window bytes starting at window-relative byte r (little endian), from aligned dwords
MKT_HD uint32_t win_load4(const TextView& tv, uint32_t r) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tv.win);
    const uint32_t i = r >> 2;
    const uint32_t a = w[i], b = w[i + 1];
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(b, a, r & 3u);
#else
    const uint32_t sh = (r & 3u) * 8u;
    return sh ? ((a >> sh) | (b << (32u - sh))) : a;
#endif
}
// 64 bitmap bits starting at bit r
MKT_HD uint64_t bits64(const uint64_t* m, uint32_t r) {
    const uint32_t w = r >> 6, sh = r & 63u;
    const uint64_t lo = m[w];
    return sh ? ((lo >> sh) | (m[w + 1] << (64u - sh))) : lo;
}
MKT_HD uint32_t clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clzll((long long)x);
#else
    return (uint32_t)__builtin_clzll(x);
#endif
}
MKT_HD uint32_t popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}
MKT_HD uint32_t ctz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctz(v);
#else
    return (uint32_t)__builtin_ctz(v);
#endif
}
// bit b (0..3) set <-> byte b of w is an ASCII decimal digit (SWAR: no cross-byte carries; one multiply gathers the flags)
MKT_HD uint32_t digit_bits4(uint32_t w) {
    const uint32_t y = w & 0x7F7F7F7Fu;
    const uint32_t ge30 = (y + 0x50505050u) | w, ge3a = (y + 0x46464646u) | w;      // bit 7 of a byte: byte >= '0' / byte > '9'
    const uint32_t d = ge30 & ~ge3a & 0x80808080u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(d, 0x08040201u, 0u, false) >> 7;          // flags are 128 x {0, 1}: one dot product gathers them
#else
    return (((d >> 7) * 0x00204081u) >> 21) & 0xFu;
#endif
}
MKT_HD uint32_t ctz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffsll((long long)v) - 1u;
#else
    return (uint32_t)__builtin_ctzll(v);
#endif
}

// small products (line index x row pitch ...): the 24-bit multiplier is full rate, v_mul_lo_u32 is quarter rate
MKT_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}

MKT_HD bool is_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }   // classic-locale isspace

// One alignment record: the six leading SAM fields, CIGAR already folded into segments.
struct Rec {
    uint32_t off;              // line start, block relative
    uint32_t qn_off, qn_len;   // QNAME, relative to line start
    uint32_t rn_off, rn_len;   // RNAME, relative to line start
    uint32_t flag, pos, mapq;
    int32_t segCnt, lclip, rclip, mappable;   // pairutil.h:29-38
    int32_t left0, left1, right0, right1;     // first two segments (more are never consumed)
    int32_t rightLast;                        // right[segCnt-1] for segCnt <= 2, else unused
    bool survive;              // six well-formed tokens and passes the FLAG/MAPQ filter
};

// cigar2segment (pairutil.h:63-126) as a byte-at-a-time state machine.
struct CigarWalk {
    int32_t index, value, cur, lastRight;
    bool bad;
    MKT_HD void begin(Rec& r) {
        index = 0; value = 0; cur = (int32_t)r.pos; lastRight = 0; bad = false;
        r.left0 = (int32_t)r.pos;
    }
    // c: CIGAR byte; lastChar: it is the last byte of the token (pairutil.h:88)
    MKT_HD void step(Rec& r, uint8_t c, bool lastChar) {
        if (bad) return;
        if (c >= '0' && c <= '9') { value = value * 10 + (int32_t)(c - '0'); return; }
        op(r, c, lastChar);
    }
    // operation byte c with its count in `value`
    MKT_HD void op(Rec& r, uint8_t c, bool lastChar) {
        if (bad) return;
        if (c == 'H' || c == 'S') {
            if (lastChar) r.rclip = value;
            else if (index == 0) r.lclip = value;
            else bad = true;
        } else if (c == 'M' || c == 'D') {
            if (c == 'M') r.mappable += value;
            cur += value;
            lastRight = cur - 1;
            if (index == 0) r.right0 = lastRight; else if (index == 1) r.right1 = lastRight;
        } else if (c == 'I') {
        } else if (c == 'N') {
            cur += value;
            ++index;
            lastRight = 0;
            if (index == 1) { r.left1 = cur; r.right1 = 0; }
        } else {
            bad = true;
        }
        value = 0;
    }
    MKT_HD void end(Rec& r) {
        if (!bad && lastRight != 0) { r.segCnt = index + 1; r.rightLast = lastRight; }
    }
};
MKT_HD void rec_clear(Rec& r, uint32_t off) {
    r.off = off; r.qn_off = r.qn_len = r.rn_off = r.rn_len = 0;
    r.flag = r.pos = r.mapq = 0;
    r.segCnt = 0; r.lclip = r.rclip = r.mappable = 0;
    r.left0 = r.left1 = r.right0 = r.right1 = r.rightLast = 0;
    r.survive = false;
}

// Parses the record starting at `off`, anywhere in the block (generic, byte at a time).  Stops
// after the sixth token; never reads at or past the line's '\n'.  Token rules follow
// `ss >> a >> b ...` (whitespace-separated, leading blanks skipped).  A record with fewer than six
// tokens, a non-decimal FLAG/POS/MAPQ or a header line ('@') does not survive (the reference's
// behaviour on such lines is undefined / input-order dependent).
MKT_HD Rec parse_record(const TextView& tv, uint32_t off, const Params& P) {
    Rec r;
    rec_clear(r, off);
    const uint32_t n = tv.n;
    uint32_t p = off;
    if (p >= n) return r;
    bool ok = tv.at(p) != '@';
    for (int k = 0; k < 6; ++k) {
        uint8_t c = 0;
        while (p < n) { c = tv.at(p); if (c == '\n' || !is_ws(c)) break; ++p; }
        if (p >= n || c == '\n') return r;          // fewer than six tokens
        const uint32_t ts = p;
        if (k == 1 || k == 3 || k == 4) {           // FLAG, POS, MAPQ: unsigned decimal
            uint64_t v = 0;
            while (p < n) {
                c = tv.at(p);
                if (c == '\n' || is_ws(c)) break;
                if (c < '0' || c > '9' || v > 0xFFFFFFFFull) ok = false;
                v = v * 10 + (uint64_t)(c - '0');
                ++p;
            }
            if (v > 0xFFFFFFFFull) ok = false;
            if (k == 1) r.flag = (uint32_t)v; else if (k == 3) r.pos = (uint32_t)v; else r.mapq = (uint32_t)v;
        } else if (k == 5) {                        // CIGAR -> segments
            CigarWalk cw;
            cw.begin(r);
            while (p < n) {
                c = tv.at(p);
                if (c == '\n' || is_ws(c)) break;
                ++p;
                uint8_t nx = p < n ? tv.at(p) : (uint8_t)'\n';
                cw.step(r, c, nx == '\n' || is_ws(nx));
            }
            cw.end(r);
        } else {                                    // QNAME, RNAME
            while (p < n) { c = tv.at(p); if (c == '\n' || is_ws(c)) break; ++p; }
            if (k == 0) { r.qn_off = ts - off; r.qn_len = p - ts; } else { r.rn_off = ts - off; r.rn_len = p - ts; }
        }
    }
    r.survive = ok && !(r.flag & 0x700u) && r.mapq >= P.min_mapq;
    return r;
}

MKT_HD int parse_record_fast(const TextView& tv, uint32_t off, const Params& P, Rec& r);      // from the window bitmaps (below)

// unsigned decimal token of `len` bytes at window offset rr (all bytes inside the window)
MKT_HD uint32_t win_parse_uint(const TextView& tv, uint32_t rr, uint32_t len, bool& ok) {
    if (len > 10u) { ok = false; return 0; }
    uint64_t v = 0;
    for (uint32_t i = 0; i < len; i += 4u) {
        uint32_t w = win_load4(tv, rr + i);
        const uint32_t m = len - i < 4u ? len - i : 4u;
        for (uint32_t b = 0; b < m; ++b) {
            const uint32_t d = (w & 0xFFu) - (uint32_t)'0';
            if (d > 9u) ok = false;
            v = v * 10u + d;
            w >>= 8;
        }
    }
    if (v > 0xFFFFFFFFull) ok = false;
    return (uint32_t)v;
}

// Fast parser for a line that starts inside the window: token boundaries come from a whitespace
// bitmap of the line's first 128 bytes (ws0: bytes 0..63, ws1: bytes 64..127) and from L, the
// distance from the line start to its '\n' (any value >= 128 when there is none in those bytes);
// field bytes come from aligned dword loads.  Returns PF_OK when r is exactly what parse_record would
// produce; PF_LONG when the six fields do not end inside the first 128 bytes (the caller uses
// parse_record); PF_CUT when the window ends before the sixth field does.
enum { PF_OK = 1, PF_LONG = 0, PF_CUT = -1 };
// core: the line's bytes are tv.win[rr ...]; `avail` of them can be read, the block ends `reach` bytes after the line start
MKT_HD int parse_record_core(const TextView& tv, uint32_t rr, const Params& P, Rec& r, uint64_t ws0, uint64_t ws1, uint32_t L,
                             uint32_t avail, uint32_t reach) {
    uint32_t lim = avail < 128u ? avail : 128u;
    if (L < lim) lim = L;
    const bool terminated = (L < 128u && lim == L) || (lim == reach);              // the line really ends at lim
    const uint64_t m0 = lim >= 64u ? ~0ull : ((1ull << lim) - 1ull);
    const uint64_t m1 = lim <= 64u ? 0ull : (lim >= 128u ? ~0ull : ((1ull << (lim - 64u)) - 1ull));
    const uint64_t n0 = ~ws0 & m0, n1 = ~ws1 & m1;                                // token bytes
    uint64_t s0 = n0 & ~(n0 << 1), s1 = n1 & ~((n1 << 1) | (n0 >> 63));           // first byte of each token
    uint64_t e0 = n0 & ~((n0 >> 1) | (n1 << 63)), e1 = n1 & ~(n1 >> 1);           // last byte of each token
    uint32_t ts[6], te[6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 6; ++k) {
        if (s0) { ts[k] = ctz64(s0); s0 &= s0 - 1ull; }
        else if (s1) { ts[k] = 64u + ctz64(s1); s1 &= s1 - 1ull; }
        else return terminated ? PF_OK : (lim == avail ? PF_CUT : PF_LONG);   // a complete line with fewer than six tokens is no record
        if (e0) { te[k] = ctz64(e0); e0 &= e0 - 1ull; }
        else { te[k] = 64u + ctz64(e1); e1 &= e1 - 1ull; }
    }
    if (te[5] + 1u == lim && !terminated) return lim == avail ? PF_CUT : PF_LONG;      // the CIGAR may continue past what we can see
    bool ok = tv.win[rr] != '@';
    r.qn_off = ts[0]; r.qn_len = te[0] + 1u - ts[0];
    r.rn_off = ts[2]; r.rn_len = te[2] + 1u - ts[2];
    r.flag = win_parse_uint(tv, rr + ts[1], te[1] + 1u - ts[1], ok);
    r.pos = win_parse_uint(tv, rr + ts[3], te[3] + 1u - ts[3], ok);
    r.mapq = win_parse_uint(tv, rr + ts[4], te[4] + 1u - ts[4], ok);
    CigarWalk cw;
    cw.begin(r);
    const uint32_t cs = rr + ts[5], clen = te[5] + 1u - ts[5];
    // Per OPERATION instead of per byte when the token is short and every count has at most three digits (reads below
    // 1000 bp): bit i of `dig` <-> byte i of the token is a decimal digit; each non-digit is an operation whose count is
    // the digit run before it, read together with the operation byte as one unaligned dword.
    uint32_t dig = 0;
    if (clen <= 12u) {
        dig = digit_bits4(win_load4(tv, cs));
        if (clen > 4u) dig |= digit_bits4(win_load4(tv, cs + 4u)) << 4;
        if (clen > 8u) dig |= digit_bits4(win_load4(tv, cs + 8u)) << 8;
        dig &= (1u << clen) - 1u;
    }
    if (clen <= 12u && !(dig & (dig >> 1) & (dig >> 2) & (dig >> 3))) {
        uint32_t ops = ~dig & ((1u << clen) - 1u);
        uint32_t run0 = 0;                                          // first byte of the digit run before the next operation
        while (ops) {
            const uint32_t p = ctz32(ops);
            ops &= ops - 1u;
            const uint32_t L = p - run0;                            // 0..3 digits
            const uint32_t w = win_load4(tv, cs + p - 3u);          // bytes p-3 .. p: hundreds, tens, units, operation
            const int32_t value = (int32_t)((L > 0u ? (w >> 16) & 15u : 0u) + (L > 1u ? ((w >> 8) & 15u) * 10u : 0u) + (L > 2u ? (w & 15u) * 100u : 0u));
            cw.value = value;
            cw.op(r, (uint8_t)(w >> 24), p + 1u == clen);
            run0 = p + 1u;
        }
    } else {
        for (uint32_t i = 0; i < clen; i += 4u) {
            uint32_t w = win_load4(tv, cs + i);
            const uint32_t m = clen - i < 4u ? clen - i : 4u;
            for (uint32_t b = 0; b < m; ++b) { cw.step(r, (uint8_t)(w & 0xFFu), i + b + 1u == clen); w >>= 8; }
        }
    }
    cw.end(r);
    r.survive = ok && !(r.flag & 0x700u) && r.mapq >= P.min_mapq;
    return PF_OK;
}
MKT_HD int parse_record_fast(const TextView& tv, uint32_t off, const Params& P, Rec& r, uint64_t ws0, uint64_t ws1, uint32_t L) {
    rec_clear(r, off);
    const uint32_t rr = off - tv.w0;
    if (rr >= tv.wlen) return PF_CUT;
    return parse_record_core(tv, rr, P, r, ws0, ws1, L, tv.wlen - rr, tv.n - off);
}

// the same, with the head bitmaps taken from the window-wide bitmaps (generic kernel)
MKT_HD int parse_record_fast(const TextView& tv, uint32_t off, const Params& P, Rec& r) {
    const uint32_t rr = off - tv.w0;
    if (rr >= tv.wlen) { rec_clear(r, off); return PF_CUT; }
    const uint64_t nl0 = bits64(tv.nlm, rr), nl1 = bits64(tv.nlm, rr + 64u);
    const uint32_t L = nl0 ? ctz64(nl0) : (nl1 ? 64u + ctz64(nl1) : 128u);      // first '\n' (bitmaps are zero past wlen)
    return parse_record_fast(tv, off, P, r, bits64(tv.wsm, rr), bits64(tv.wsm, rr + 64u), L);
}

MKT_HD uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }

// bytewise std::string::compare of two byte ranges of the block.  WIN: the caller guarantees that
// both ranges lie inside the window (the lean kernel), so no global-memory path is generated.
template <bool WIN = false>
MKT_HD int text_cmp(const TextView& tv, uint32_t a, uint32_t alen, uint32_t b, uint32_t blen) {
    const uint32_t m = alen < blen ? alen : blen;
    if (WIN || (tv.inside(a, alen) && tv.inside(b, blen))) {
        const uint32_t ra = a - tv.w0, rb = b - tv.w0;
        for (uint32_t i = 0; i < m; i += 4u) {
            uint32_t x = win_load4(tv, ra + i), y = win_load4(tv, rb + i);
            if (m - i < 4u) { const uint32_t k = (1u << ((m - i) * 8u)) - 1u; x &= k; y &= k; }
            if (x != y) return bswap32(x) < bswap32(y) ? -1 : 1;      // first differing byte decides
        }
    } else {
        for (uint32_t i = 0; i < m; ++i) {
            int d = (int)tv.at(a + i) - (int)tv.at(b + i);
            if (d) return d;
        }
    }
    return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
template <bool WIN = false>
MKT_HD bool text_eq(const TextView& tv, uint32_t a, uint32_t alen, uint32_t b, uint32_t blen) {
    if (alen != blen) return false;
    if (WIN || (tv.inside(a, alen) && tv.inside(b, blen))) {
        // last dword first: read names of one run share their head and differ in the trailing coordinates
        const uint32_t ra = a - tv.w0, rb = b - tv.w0;
        for (uint32_t i = alen ? ((alen - 1u) & ~3u) : 0u; alen; i -= 4u) {
            uint32_t x = win_load4(tv, ra + i) ^ win_load4(tv, rb + i);
            if (alen - i < 4u) x &= (1u << ((alen - i) * 8u)) - 1u;
            if (x) return false;
            if (i == 0u) break;
        }
        return true;
    }
    for (uint32_t i = 0; i < alen; ++i)
        if (tv.at(a + i) != tv.at(b + i)) return false;
    return true;
}

// The segment part of a record, as the classifiers consume it.
struct Seg {
    int32_t segCnt, lclip, rclip, mappable, left0, left1, right0, right1, rightLast;
    uint32_t flag, pos;
    uint32_t chr_off, chr_len;    // RNAME bytes, block relative
    uint64_t chr_key;             // lean path only: the first eight RNAME bytes as a big-endian number, zero padded
};
MKT_HD Seg seg_of(const Rec& r) {
    Seg s;
    s.segCnt = r.segCnt; s.lclip = r.lclip; s.rclip = r.rclip; s.mappable = r.mappable;
    s.left0 = r.left0; s.left1 = r.left1; s.right0 = r.right0; s.right1 = r.right1; s.rightLast = r.rightLast;
    s.flag = r.flag; s.pos = r.pos; s.chr_off = r.off + r.rn_off; s.chr_len = r.rn_len; s.chr_key = 0;
    return s;
}

// float32 compare, one rounding for the product (cvtsi2ss / mulss / comiss in the reference)
MKT_HD bool ratio_ok(int32_t mappable, int32_t total, float ratio) {
#if defined(__HIP_DEVICE_COMPILE__)
    float rhs = __fmul_rn((float)total, ratio);
#else
    volatile float rhs = (float)total * ratio;
#endif
    return (float)mappable >= rhs;
}
MKT_HD bool integrity_1(const Seg& s, float ratio) {            // pairutil.h:180-188
    int32_t total = s.mappable;
    if (s.lclip > kMinClip) total += s.lclip;
    if (s.rclip > kMinClip) total += s.rclip;
    return ratio_ok(s.mappable, total, ratio);
}
MKT_HD bool integrity_2(const Seg& a, const Seg& b, float ratio) {   // pairutil.h:190-208
    int32_t t1 = a.mappable, t2 = b.mappable;
    if (a.lclip > kMinClip) t1 += a.lclip;
    if (a.rclip > kMinClip) t1 += a.rclip;
    if (b.lclip > kMinClip) t2 += b.lclip;
    if (a.rclip > kMinClip) t2 += b.rclip;                      // quirk Q3 (pairutil.h:200)
    return ratio_ok(a.mappable + b.mappable, t1 > t2 ? t1 : t2, ratio);
}

struct Verdict {
    uint32_t counter;          // Counter
    bool emit;
    uint32_t chrA_off, chrA_len, chrB_off, chrB_len;
    uint32_t posA, posB;
    uint8_t sA, sB;
};
MKT_HD Verdict verdict_none(uint32_t counter) {
    Verdict v;
    v.counter = counter; v.emit = false;
    v.chrA_off = v.chrA_len = v.chrB_off = v.chrB_len = 0; v.posA = v.posB = 0; v.sA = v.sB = '+';
    return v;
}
MKT_HD uint8_t strand_of(uint32_t flag) { return (flag & 16u) ? '-' : '+'; }

// Bytewise comparison of two chromosome names.  Lean path (WIN): the 8-byte keys decide unless both names share their first
// eight bytes and one of them is longer (names hold no NUL bytes: a zero-padded shorter name sorts first, as std::string does)
template <bool WIN>
MKT_HD int chr_cmp(const TextView& tv, uint32_t ao, uint32_t al, uint64_t ak, uint32_t bo, uint32_t bl, uint64_t bk) {
    if (WIN) {
        if (ak != bk) return ak < bk ? -1 : 1;
        if (al <= 8u && bl <= 8u) return 0;
    }
    return text_cmp<WIN>(tv, ao, al, bo, bl);
}

// flash2pairs.h:110-144 / unc2pairs.h:315-348
template <bool WIN = false>
MKT_HD Verdict order_and_bin(const TextView& tv, uint32_t c1o, uint32_t c1l, uint32_t pos1, uint8_t s1,
                             uint32_t c2o, uint32_t c2l, uint32_t pos2, uint8_t s2, uint64_t k1 = 0, uint64_t k2 = 0) {
    Verdict v;
    int chrcmp = chr_cmp<WIN>(tv, c1o, c1l, k1, c2o, c2l, k2);
    if (chrcmp < 0 || (chrcmp == 0 && pos1 < pos2)) {
        v.chrA_off = c1o; v.chrA_len = c1l; v.posA = pos1; v.sA = s1;
        v.chrB_off = c2o; v.chrB_len = c2l; v.posB = pos2; v.sB = s2;
    } else {
        v.chrA_off = c2o; v.chrA_len = c2l; v.posA = pos2; v.sA = s2;
        v.chrB_off = c1o; v.chrB_len = c1l; v.posB = pos1; v.sB = s1;
    }
    v.emit = true;
    if (chrcmp == 0) {
        uint32_t dist = v.posB - v.posA;
        if (dist <= (uint32_t)kMaxSelfCircle) { v.counter = C_SELFCIRCLE; v.emit = false; }
        else v.counter = dist >= 10000u ? C_CIS10K : (dist >= 1000u ? C_CIS1K : C_CIS0);
    } else {
        v.counter = C_TRANS;
    }
    return v;
}

// flash2pairs.h:17-155.  n = surviving records in the group, a/b = the first two.
template <bool WIN = false>
MKT_HD Verdict classify_flash(const TextView& tv, uint32_t n, const Seg& a, const Seg& b, float ratio) {
    if (n == 1) {
        if (a.segCnt > 2) return verdict_none(C_MANYHITS);
        if (!integrity_1(a, ratio)) return verdict_none(C_LOWMAP);
        uint32_t pos1 = a.pos;
        uint32_t pos2 = a.segCnt >= 1 ? (uint32_t)a.rightLast : 0u;     // right[segCnt-1]; UB in the reference when segCnt == 0
        uint32_t dist = pos2 - pos1;
        Verdict v;
        v.counter = dist >= 10000u ? C_CIS10K : (dist >= 1000u ? C_CIS1K : C_CIS0);
        v.emit = true;
        v.chrA_off = v.chrB_off = a.chr_off; v.chrA_len = v.chrB_len = a.chr_len;
        v.posA = pos1; v.posB = pos2; v.sA = '+'; v.sB = '-';
        return v;
    }
    if (n == 2) {
        if (a.segCnt != 1 || b.segCnt != 1) return verdict_none(C_MANYHITS);
        if (!integrity_2(a, b, ratio)) return verdict_none(C_LOWMAP);
        uint32_t pos1 = a.pos, pos2 = b.pos;
        if (a.lclip > a.rclip) pos1 = (uint32_t)a.right0;
        if (b.lclip > b.rclip) pos2 = (uint32_t)b.right0;
        return order_and_bin<WIN>(tv, a.chr_off, a.chr_len, pos1, strand_of(a.flag), b.chr_off, b.chr_len, pos2, strand_of(b.flag), a.chr_key, b.chr_key);
    }
    return verdict_none(C_MANYHITS);
}

template <bool WIN = false>
MKT_HD bool pairable(const TextView& tv, const Seg& x, const Seg& y, int32_t lo_left, int32_t hi_left, int32_t hi_right) {
    return chr_cmp<WIN>(tv, x.chr_off, x.chr_len, x.chr_key, y.chr_off, y.chr_len, y.chr_key) == 0 && lo_left < hi_left && hi_right - lo_left <= kMaxPairDist;
}
MKT_HD uint32_t clip_side_pos(const Seg& s) { return (uint32_t)(s.lclip > s.rclip ? s.right0 : s.left0); }

// unc2pairs.h:16-358.  n1/n2 = records with FLAG&64 / (else) FLAG&128; r1a,r1b / r2a,r2b = the first two of each.
template <bool WIN = false>
MKT_HD Verdict classify_unc(const TextView& tv, uint32_t n1, uint32_t n2, const Seg& r1a, const Seg& r1b,
                            const Seg& r2a, const Seg& r2b, float ratio) {
    if (n1 == 0 || n2 == 0) return verdict_none(C_NONE);        // :52-55
    if (n1 + n2 > 3) return verdict_none(C_NONE);               // :56-59
    uint32_t c1o, c1l, c2o, c2l, pos1 = 0, pos2 = 0;
    uint64_t k1, k2;
    uint8_t st1, st2;
    if (n1 == 1 && n2 == 1) {                                   // category 0
        const Seg& s1 = r1a; const Seg& s2 = r2a;      // fixed aliases (no run-time choice)
        if (!integrity_1(s1, ratio)) return verdict_none(C_LOWMAP);
        if (!integrity_1(s2, ratio)) return verdict_none(C_LOWMAP);
        if (s1.segCnt + s2.segCnt > 3) return verdict_none(C_MANYHITS);
        st1 = strand_of(s1.flag); st2 = strand_of(s2.flag);
        c1o = s1.chr_off; c1l = s1.chr_len; c2o = s2.chr_off; c2l = s2.chr_len; k1 = s1.chr_key; k2 = s2.chr_key;
        if (s1.segCnt == 1 && s2.segCnt == 1) {
            pos1 = (uint32_t)(st1 == '+' ? s1.left0 : s1.right0);
            pos2 = (uint32_t)(st2 == '+' ? s2.left0 : s2.right0);
        } else if (s1.segCnt == 2) {
            if (st1 == '+') {
                if (st2 == '-' && pairable<WIN>(tv, s1, s2, s1.left1, s2.left0, s2.right0)) { pos1 = (uint32_t)s1.left0; pos2 = (uint32_t)s2.right0; }
                else return verdict_none(C_UNPAIRED);
            } else {
                if (st2 == '+' && pairable<WIN>(tv, s1, s2, s2.left0, s1.left0, s1.right0)) { pos1 = (uint32_t)s1.right1; pos2 = (uint32_t)s2.left0; }
                else return verdict_none(C_UNPAIRED);
            }
        } else {
            if (st1 == '+') {
                if (st2 == '-' && pairable<WIN>(tv, s1, s2, s1.left0, s2.left0, s2.right0)) { pos1 = (uint32_t)s1.left0; pos2 = (uint32_t)s2.right1; }
                else return verdict_none(C_UNPAIRED);
            } else {
                if (st2 == '+' && pairable<WIN>(tv, s1, s2, s2.left1, s1.left0, s1.right0)) { pos1 = (uint32_t)s1.right0; pos2 = (uint32_t)s2.left0; }
                else return verdict_none(C_UNPAIRED);
            }
        }
    } else {
        // categories 1 (1+2) and 2 (2+1): `u` is the read with one record, v0/v1 the split read.
        // (copies, not references picked at run time: those would force the records into scratch memory)
        const bool cat1 = (n1 == 1);
        const Seg u = cat1 ? r1a : r2a;
        const Seg v0 = cat1 ? r2a : r1a;
        const Seg v1 = cat1 ? r2b : r1b;
        if (cat1) {                                             // :62-98
            if (!integrity_1(u, ratio)) return verdict_none(C_LOWMAP);
            if (!integrity_2(v0, v1, ratio)) return verdict_none(C_LOWMAP);
        } else {                                                // :100-121
            if (!integrity_2(v0, v1, ratio)) return verdict_none(C_LOWMAP);
            if (!integrity_1(u, ratio)) return verdict_none(C_LOWMAP);
        }
        if (u.segCnt != 1 || v0.segCnt != 1 || v1.segCnt != 1) return verdict_none(C_MANYHITS);
        const uint8_t su = strand_of(u.flag);
        const uint32_t posu = (uint32_t)(su == '+' ? u.left0 : u.right0);
        // try v0 then v1 (:196-227 / :255-285)
        bool m0, m1;
        if (su == '+') {
            m0 = strand_of(v0.flag) == '-' && pairable<WIN>(tv, u, v0, u.left0, v0.left0, v0.right0);
            m1 = strand_of(v1.flag) == '-' && pairable<WIN>(tv, u, v1, u.left0, v1.left0, v1.right0);
        } else {
            m0 = strand_of(v0.flag) == '+' && pairable<WIN>(tv, u, v0, v0.left0, u.left0, u.right0);
            m1 = strand_of(v1.flag) == '+' && pairable<WIN>(tv, u, v1, v1.left0, u.left0, u.right0);
        }
        if (!m0 && !m1) return verdict_none(C_UNPAIRED);        // :229-232 / :287-290
        // the OTHER record of the split read gives the second end
        const uint32_t oo = m0 ? v1.chr_off : v0.chr_off, ol = m0 ? v1.chr_len : v0.chr_len;
        const uint64_t ok = m0 ? v1.chr_key : v0.chr_key;
        const uint32_t poso = m0 ? clip_side_pos(v1) : clip_side_pos(v0);
        const uint8_t so = strand_of(m0 ? v1.flag : v0.flag);
        if (cat1) { c1o = u.chr_off; c1l = u.chr_len; k1 = u.chr_key; pos1 = posu; st1 = su; c2o = oo; c2l = ol; k2 = ok; pos2 = poso; st2 = so; }
        else      { c2o = u.chr_off; c2l = u.chr_len; k2 = u.chr_key; pos2 = posu; st2 = su; c1o = oo; c1l = ol; k1 = ok; pos1 = poso; st1 = so; }
    }
    return order_and_bin<WIN>(tv, c1o, c1l, pos1, st1, c2o, c2l, pos2, st2, k1, k2);
}

MKT_HD uint32_t dec_digits(uint32_t v) {
    return v >= 1000000000u ? 10 : v >= 100000000u ? 9 : v >= 10000000u ? 8 : v >= 1000000u ? 7 : v >= 100000u ? 6
         : v >= 10000u ? 5 : v >= 1000u ? 4 : v >= 100u ? 3 : v >= 10u ? 2 : 1;
}
// rid \t chrA \t posA \t chrB \t posB \t sA \t sB \n   (flash2pairs.h:123-127)
MKT_HD uint32_t pair_line_len(uint32_t qn_len, const Verdict& v) {
    return qn_len + v.chrA_len + v.chrB_len + dec_digits(v.posA) + dec_digits(v.posB) + 9u;   // 6 tabs, 2 strands, newline
}
// The ten decimal digits of v as ASCII, most significant first, zero padded, in two little-endian
// words (hi8: text bytes 0..7, lo2: text bytes 8..9).  Straight-line: no data-dependent branches.
// four decimal digits of x < 10000 as bytes, most significant digit in byte 0 (24-bit multiplies: full rate)
MKT_HD uint32_t dig4(uint32_t x) {
    const uint32_t hi = mul24(x, 5243u) >> 19, lo = x - mul24(hi, 100u);           // x / 100, x % 100 (exact below 10000)
    const uint32_t th = mul24(hi, 103u) >> 10, tl = mul24(lo, 103u) >> 10;         // y / 10 (exact below 100)
    return th | ((hi - mul24(th, 10u)) << 8) | (tl << 16) | ((lo - mul24(tl, 10u)) << 24);
}
MKT_HD void dec10(uint32_t v, uint64_t& hi8, uint32_t& lo2) {
    const uint32_t a = v / 100000000u, r = v - a * 100000000u;                     // a <= 42
    const uint32_t b = r / 10000u, c = r - b * 10000u;
    // text: a (2 digits) b (4) c (4)
    const uint32_t ta = mul24(a, 103u) >> 10;
    const uint32_t d01 = ta | ((a - mul24(ta, 10u)) << 8), db = dig4(b), dc = dig4(c);
    hi8 = 0x3030303030303030ull | (uint64_t)d01 | ((uint64_t)db << 16) | ((uint64_t)(dc & 0xFFFFu) << 48);
    lo2 = 0x3030u | (dc >> 16);
}
// "<v in decimal><tail bytes>" as little-endian text in two words; tail holds ntail (<= 5) bytes.
MKT_HD void dec_lit(uint32_t v, uint32_t ndig, uint64_t tail, uint64_t& w0, uint64_t& w1) {
    uint64_t hi8; uint32_t lo2;
    dec10(v, hi8, lo2);
    // 10-byte string s = hi8 | lo2 << 64; drop the (10 - ndig) leading zeros: shift right by that many bytes
    const uint32_t drop = 10u - ndig;                  // 0..9
    uint64_t a, b;
    if (drop == 0u) { a = hi8; b = lo2; }
    else if (drop < 8u) { a = (hi8 >> (8u * drop)) | ((uint64_t)lo2 << (64u - 8u * drop)); b = drop >= 2u ? 0ull : ((uint64_t)lo2 >> (8u * drop)); }
    else { a = (uint64_t)lo2 >> (8u * (drop - 8u)); b = 0; }
    // append the tail at byte ndig
    if (ndig < 8u) { a |= tail << (8u * ndig); b |= ndig >= 4u ? tail >> (64u - 8u * ndig) : 0ull; }
    else b |= tail << (8u * (ndig - 8u));
    w0 = a; w1 = b;
}

MKT_HD uint32_t dec_digit(uint32_t v, uint32_t from_right) {      // digit 10^from_right of v
    switch (from_right) {
    case 0: return v % 10u;
    case 1: return (v / 10u) % 10u;
    case 2: return (v / 100u) % 10u;
    case 3: return (v / 1000u) % 10u;
    case 4: return (v / 10000u) % 10u;
    case 5: return (v / 100000u) % 10u;
    case 6: return (v / 1000000u) % 10u;
    case 7: return (v / 10000000u) % 10u;
    case 8: return (v / 100000000u) % 10u;
    default: return (v / 1000000000u) % 10u;
    }
}
// byte k of  rid \t chrA \t posA \t chrB \t posB \t sA \t sB \n   (flash2pairs.h:123-127)
template <bool WIN = false>
MKT_HD uint8_t pair_line_byte(const TextView& tv, uint32_t qn_abs, uint32_t qn_len, const Verdict& v, uint32_t k) {
    if (k < qn_len) return WIN ? tv.win[qn_abs + k - tv.w0] : tv.at(qn_abs + k);
    k -= qn_len;
    if (k == 0) return '\t';
    k -= 1;
    if (k < v.chrA_len) return WIN ? tv.win[v.chrA_off + k - tv.w0] : tv.at(v.chrA_off + k);
    k -= v.chrA_len;
    if (k == 0) return '\t';
    k -= 1;
    const uint32_t dA = dec_digits(v.posA);
    if (k < dA) return (uint8_t)('0' + dec_digit(v.posA, dA - 1u - k));
    k -= dA;
    if (k == 0) return '\t';
    k -= 1;
    if (k < v.chrB_len) return WIN ? tv.win[v.chrB_off + k - tv.w0] : tv.at(v.chrB_off + k);
    k -= v.chrB_len;
    if (k == 0) return '\t';
    k -= 1;
    const uint32_t dB = dec_digits(v.posB);
    if (k < dB) return (uint8_t)('0' + dec_digit(v.posB, dB - 1u - k));
    k -= dB;
    switch (k) {
    case 0: return '\t';
    case 1: return v.sA;
    case 2: return '\t';
    case 3: return v.sB;
    default: return '\n';
    }
}

// ---- extensions (default off; parity unpinned by the reference) ----------------
// One record per emitted pair: the duplicate-marking key (chr1, pos1, chr2, pos2, strand1, strand2) with
// chromosome NAMES replaced by their slot in the run's name table (64-bit FNV-1a keyed, see chr_slot).
struct KeyRec {
    uint64_t k0;       // chrA slot << 45 | chrB slot << 32 | posA
    uint64_t k1;       // posB << 32 | sA('-') << 31 | sB('-') << 30 | lane (16 bits; MKT_EXT_LANES only)
    uint64_t ord;      // raw: tile << 16 | emit ordinal in tile ; placed: emitted-pair ordinal in input order
};
constexpr uint32_t kChrSlots = 8192;               // open addressing, power of two
constexpr uint32_t kChrNameMax = 62;
struct ChrTab {
    unsigned long long hash[kChrSlots];            // 0 = empty
    uint8_t name[kChrSlots][64];                   // written by the thread that claimed the slot; [63] = length; read by the host only
};
MKT_HD uint64_t fnv1a64(const TextView& tv, uint32_t off, uint32_t len) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < len; ++i) { h ^= tv.at(off + i); h *= 0x100000001b3ull; }
    return h ? h : 1ull;
}
MKT_HD uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// Quirk Q2: does surviving group g of K contribute to the LOGGED selfCircle?
MKT_HD bool selfcircle_logged(uint64_t g, uint64_t K, uint32_t ref_threads) {
    uint64_t j = g / kRefBatch, i = g % kRefBatch;
    bool final = K <= (j + 1) * (uint64_t)kRefBatch;
    uint64_t loaded = final ? (K - 1 - j * (uint64_t)kRefBatch) : (uint64_t)kRefBatch;
    uint64_t W = final ? (uint64_t)ref_threads : (uint64_t)(ref_threads - 1);
    return i < loaded / W;
}

}  // namespace mkt
