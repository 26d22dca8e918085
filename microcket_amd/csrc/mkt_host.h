// mkt_host.h -- host-side bookkeeping around the per-block kernel results.
//
// The kernel classifies EVERY QNAME group of every block.  What the reference does across the
// whole input lives here, on a few bytes per block:
//   Q1  the input's last surviving group is never classified (pairutil.h:151-176 returns the last
//       index, sam2pairs.cpp:150-151 uses it as a count) -> the newest group's counter, .pairs
//       bytes and .sam bytes stay "pending" until a later group supersedes it, and are dropped at
//       the end of the input;
//   Q2  the logged selfCircle is reference-thread 0's share (sam2pairs.cpp:202-210 vs :214) ->
//       evaluated from the global indices of the self-circle groups once K is known.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "mkt_tile.h"

namespace mkt {

struct BlockResult {            // written by the device (finish kernel), POD
    uint64_t groups, emitted, sc, pair_bytes, sam_bytes;
    uint32_t counters[C_COUNT];
    uint32_t err;
    TileLast last;              // last group of the block (valid = 0: the block holds no group)
    uint32_t tiles, pad;
    uint32_t nregions, pad2;    // outputs sit in nregions equal slices of the output buffers (any-order mode: 16); pad2: deferred tiles a wider halo would have kept
    uint64_t rpair[16], rsam[16];   // bytes used in each slice
};

struct RunStats {               // what the .log needs, plus bookkeeping
    uint32_t counters[C_COUNT]; // C_SELFCIRCLE holds the LOGGED value (Q2)
    uint32_t selfcircle_all;
    uint64_t groups;            // K: surviving groups seen, including the one Q1 drops
    uint64_t pairs;             // emitted .pairs lines
    uint64_t pair_bytes, sam_bytes;
};

struct RunAccum {
    uint64_t groups = 0, emitted = 0, pair_bytes = 0, sam_bytes = 0;
    uint64_t counters[C_COUNT] = {0};
    uint64_t sc = 0;                   // self-circle groups so far (their indices live in the sc list)
    TileLast pending = {0, 0, 0, 0, 0, 0, 0, 0};   // newest group (held back: it may be the input's last, quirk Q1)

    void add_block(const BlockResult& r) {
        sc += r.sc;
        for (int c = 0; c < (int)C_COUNT; ++c) counters[c] += r.counters[c];
        groups += r.groups; emitted += r.emitted; pair_bytes += r.pair_bytes; sam_bytes += r.sam_bytes;
        if (r.last.valid) pending = r.last;
    }
    // drop_last: this shard holds the input's last group (Q1).
    bool drops(bool drop_last) const { return drop_last && pending.valid; }
    // `logged`: how many of the run's self-circle groups reach the .log (quirk Q2), the dropped last group excluded
    RunStats finish_logged(bool drop_last, uint64_t logged) const {
        RunStats s;
        memset(&s, 0, sizeof s);
        uint64_t c[C_COUNT];
        for (int k = 0; k < (int)C_COUNT; ++k) c[k] = counters[k];
        s.pairs = emitted; s.pair_bytes = pair_bytes; s.sam_bytes = sam_bytes;
        if (drops(drop_last)) {
            if (pending.counter) --c[pending.counter];
            if (pending.pair_bytes) { --s.pairs; s.pair_bytes -= pending.pair_bytes; s.sam_bytes -= pending.sam_bytes; }
        }
        for (int k = 0; k < (int)C_COUNT; ++k) s.counters[k] = (uint32_t)c[k];     // u32 wrap as the reference's kstat
        s.selfcircle_all = (uint32_t)c[C_SELFCIRCLE];
        s.counters[C_SELFCIRCLE] = (uint32_t)logged;
        s.groups = groups;
        return s;
    }
    // The same from a host-side list.  group_offset / K_total place the shard in the whole input (single process:
    // 0 and `groups`); sc_idx: shard-local indices of the self-circle groups (`sc` entries, any order).
    RunStats finish(bool drop_last, uint32_t ref_threads, uint64_t group_offset, uint64_t K_total, const uint64_t* sc_idx) const {
        uint64_t logged = 0;
        for (uint64_t k = 0; k < sc; ++k) {
            if (drops(drop_last) && sc_idx[k] + 1 == groups) continue;             // the dropped last group
            if (selfcircle_logged(group_offset + sc_idx[k], K_total, ref_threads)) ++logged;
        }
        return finish_logged(drop_last, logged);
    }
};

// the 8-line log of sam2pairs.cpp:211-218
inline int format_log(const RunStats& s, char* out, size_t cap) {
    return snprintf(out, cap, "lowMap\t%u\nmanyHits\t%u\nunpaired\t%u\nselfCircle\t%u\ntrans\t%u\ncis10K\t%u\ncis1K\t%u\ncis0\t%u\n",
                    s.counters[C_LOWMAP], s.counters[C_MANYHITS], s.counters[C_UNPAIRED], s.counters[C_SELFCIRCLE],
                    s.counters[C_TRANS], s.counters[C_CIS10K], s.counters[C_CIS1K], s.counters[C_CIS0]);
}

// ---- host cut ----------------------------------------------------------------------------------------
// The host hands the kernels blocks that start on a QNAME-group boundary.  The reference groups SURVIVING lines only
// (pairutil.h:157-163 filters on FLAG / MAPQ before it compares names), so the boundary is defined on surviving lines:
// the carry (what moves to the next block) starts at the first line of the LAST run of surviving lines with equal
// QNAME.  Filtered lines have no effect on any output and may fall on either side.
inline bool host_is_ws(char c) { return c == ' ' || (c >= 9 && c <= 13); }
// first token of the line [ls, le) (le: its '\n' or the end of the bytes)
inline void host_qname(const char* buf, size_t ls, size_t le, size_t* a, size_t* b) {
    size_t p = ls;
    while (p < le && host_is_ws(buf[p])) ++p;
    *a = p;
    while (p < le && !host_is_ws(buf[p])) ++p;
    *b = p;
}
// The per-line filter exactly as parse_record (mkt_core.h) applies it: six tokens, decimal FLAG / POS / MAPQ that fit
// 32 bits, no '@' in front, !(FLAG & 0x700), MAPQ >= min_mapq.
inline bool host_line_survives(const char* buf, size_t ls, size_t le, uint32_t min_mapq) {
    if (ls >= le || buf[ls] == '@') return false;
    size_t p = ls;
    uint64_t val[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 6; ++k) {
        while (p < le && host_is_ws(buf[p])) ++p;
        if (p >= le) return false;                                  // fewer than six tokens
        const bool num = (k == 1 || k == 3 || k == 4);
        uint64_t v = 0;
        while (p < le && !host_is_ws(buf[p])) {
            if (num) {
                const unsigned d = (unsigned)(unsigned char)buf[p] - (unsigned)'0';
                if (d > 9u || v > 0xFFFFFFFFull) return false;
                v = v * 10 + d;
            }
            ++p;
        }
        if (num && v > 0xFFFFFFFFull) return false;
        val[k] = v;
    }
    return !(val[1] & 0x700u) && val[4] >= (uint64_t)min_mapq;
}
// buf[0, n): returns `cut` with 0 < cut <= end such that buf[0, cut) ends on a line end and the first surviving line at
// or after `cut` (if any) opens a new group; *end_out = end of the complete lines.  cut == end: no surviving line in the
// complete lines at all (everything can be processed).  Returns 0 when the complete lines hold a single run (no boundary).
inline size_t group_aligned_prefix(const char* buf, size_t n, uint32_t min_mapq, size_t* end_out = nullptr) {
    size_t end = n;                                   // exclusive end of complete lines
    while (end > 0 && buf[end - 1] != '\n') --end;
    if (end_out) *end_out = end;
    if (end == 0) return 0;
    bool have = false;
    size_t ta = 0, tb = 0, tail_ls = end;             // QNAME of the tail run, start of its earliest line seen so far
    size_t le = end - 1;                              // '\n' of the line under inspection
    for (;;) {
        size_t ls = le;
        while (ls > 0 && buf[ls - 1] != '\n') --ls;
        if (host_line_survives(buf, ls, le, min_mapq)) {
            size_t a, b;
            host_qname(buf, ls, le, &a, &b);
            if (!have) { have = true; ta = a; tb = b; tail_ls = ls; }
            else if ((b - a) == (tb - ta) && memcmp(buf + a, buf + ta, b - a) == 0) { ta = a; tb = b; tail_ls = ls; }
            else return tail_ls;                      // a surviving line with another name: the tail run starts at tail_ls
        }
        if (ls == 0) break;
        le = ls - 1;
    }
    return have ? 0 : end;
}
// Degenerate inputs (a long stretch of filtered lines behind a surviving one) would make the carry as large as the
// block: drop the complete lines of carry[0, n) that do not survive (they influence nothing).  Returns the new length.
inline size_t compact_carry(char* carry, size_t n, uint32_t min_mapq) {
    size_t w = 0, p = 0;
    while (p < n) {
        const char* nl = (const char*)memchr(carry + p, '\n', n - p);
        if (!nl) { memmove(carry + w, carry + p, n - p); w += n - p; break; }     // the incomplete last line stays
        const size_t le = (size_t)(nl - carry);
        if (host_line_survives(carry, p, le, min_mapq)) { memmove(carry + w, carry + p, le + 1 - p); w += le + 1 - p; }
        p = le + 1;
    }
    return w;
}

}  // namespace mkt
