// mkt_bam.hip -- SURVEY.md 8(f) N3: the .sam -> BAM tail of the pipeline, on the GPU.
//
// The driver ends with (microcket:533-540)
//     cat $samheader $sid.flash.sam $sid.unc.sam | samtools view -b | samtools sort -o $sid.valid.bam ;  samtools index $sid.valid.bam
// i.e. the filtered alignments that sam2pairs wrote, as a coordinate-sorted, BGZF-compressed BAM with its .bai.  Here: newline
// index, one key per line (reference id, position, strand), the stable LSD radix sort of mkt_sort.hip, an exclusive scan of the
// record sizes, one pass that writes the binary records in sorted order, BGZF blocks (stored, or LZ77 + fixed-Huffman deflate,
// CRC-32 per block) and the reductions the BAI needs (linear index, chunk starts, per-reference counts).  Byte / integer work
// bound by HBM; no MFMA.
//
// Formats: SAM / BAM / BGZF / BAI as published in "Sequence Alignment/Map Format Specification" (samtools/hts-specs, SAMv1
// sections 1.4, 4.1, 4.2, 5.2), RFC 1951 (deflate), RFC 1952 (gzip, CRC-32).  The reference ships samtools only as a prebuilt
// third-party binary (bin/samtools), which is never run here: parity with it is UNPINNED.  What is checked instead
// (tests/test_gpu_bam.py): every block inflates with Python's zlib (which verifies CRC-32 and ISIZE), an independent reader
// written from the specification (tests/bamio.py) gets the input lines back in coordinate order, and every region query
// through the .bai returns exactly the records a brute-force scan finds.  Conventions that the specification leaves open follow
// htslib's documented behaviour: integer tags in the smallest type that holds them (unsigned types for values >= 0), `bin`
// from [pos, end) with a length of 1 for unmapped reads or empty CIGARs, a mapped read without CIGAR marked unmapped, ties
// of the sort key (reference, position, strand) in input order, @HD SO:coordinate set in the header of a sorted file.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "../../include/mkt.h"
#include "mkt_launch.h"
#include "mkt_sortlib.h"

using namespace mkt;

namespace mkt {

enum { BE_FIELDS = 1, BE_REF = 2, BE_NUM = 4, BE_CIGAR = 8, BE_TAG = 16, BE_SEQ = 32, BE_NAME = 64 };
constexpr uint32_t BGZF_RAW = 0xff00;                 // uncompressed bytes per BGZF block (htslib's BGZF_BLOCK_SIZE)
constexpr uint32_t BGZF_STRIDE = 0x10000;             // room for one compressed block in the work buffer (a block is <= 64 KiB)
constexpr int BWG = 256;

struct RefTab {                                       // reference names -> ids (built on the host from the @SQ lines)
    const unsigned long long* hash;                   // FNV-1a of the name, 0 = empty
    const int32_t* id;
    const uint32_t* name_off;                         // per id: [off, off + len) in names
    const uint8_t* names;
    uint32_t mask, nref;
};
struct BamIdx { int32_t tid, beg, end; uint32_t bin; };     // per record, in file order: what the index needs (bin: low half; high half: FLAG)

__device__ inline int32_t ref_lookup(const RefTab& rt, const uint8_t* t, uint64_t a, uint64_t b, uint32_t* err) {
    unsigned long long h = 0xcbf29ce484222325ull;
    for (uint64_t p = a; p < b; ++p) { h ^= t[p]; h *= 0x100000001b3ull; }
    if (!h) h = 1;
    uint32_t s = (uint32_t)(h >> 20) & rt.mask;
    for (uint32_t probe = 0; probe <= rt.mask; ++probe) {
        const unsigned long long cur = rt.hash[s];
        if (cur == 0ull) break;
        if (cur == h) {
            const int32_t id = rt.id[s];
            const uint32_t o = rt.name_off[id], l = rt.name_off[id + 1] - o;
            bool same = l == (uint32_t)(b - a);
            for (uint32_t i = 0; same && i < l; ++i) same = rt.names[o + i] == t[a + i];
            if (same) return id;
        }
        s = (s + 1u) & rt.mask;
    }
    atomicOr(err, (uint32_t)BE_REF);
    return -1;
}
__device__ inline uint32_t reg2bin(int64_t beg, int64_t end) {        // SAMv1 5.3
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
__device__ inline void put8(uint8_t*& o, uint32_t v) { *o++ = (uint8_t)v; }
__device__ inline void put16(uint8_t*& o, uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o += 2; }
__device__ inline void put32(uint8_t*& o, uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); o += 4; }
// A record's bytes go out through this: single bytes up to the first 8-byte boundary, then whole aligned 8-byte words, single
// bytes for the rest (the neighbours on both sides belong to other threads' records).
struct Sink {
    uint8_t* p;            // where the next stored byte goes (8-aligned whenever k > 0)
    uint64_t acc;
    uint32_t k;            // bytes waiting in acc
    __device__ explicit Sink(uint8_t* at) : p(at), acc(0), k(0) {}
    __device__ inline void b(uint32_t v) {
        if (k == 0u && ((uintptr_t)p & 7u)) { *p++ = (uint8_t)v; return; }
        acc |= (uint64_t)(v & 0xFFu) << (8u * k);
        if (++k == 8u) { *reinterpret_cast<uint64_t*>(p) = acc; p += 8; acc = 0; k = 0; }
    }
    __device__ inline void h(uint32_t v) { b(v); b(v >> 8); }
    __device__ inline void w(uint32_t v) { b(v); b(v >> 8); b(v >> 16); b(v >> 24); }
    __device__ inline void flush() { for (uint32_t i = 0; i < k; ++i) p[i] = (uint8_t)(acc >> (8u * i)); p += k; k = 0; acc = 0; }
};

// unsigned decimal in [a, b); *ok cleared when it is not one
__device__ inline uint64_t dec_u(const uint8_t* t, uint64_t a, uint64_t b, bool* ok) {
    if (a >= b || b - a > 19) { *ok = false; return 0; }
    uint64_t v = 0;
    for (uint64_t p = a; p < b; ++p) { const uint32_t d = (uint32_t)t[p] - 48u; if (d > 9u) { *ok = false; return 0; } v = v * 10 + d; }
    return v;
}
__device__ inline int64_t dec_s(const uint8_t* t, uint64_t a, uint64_t b, bool* ok) {
    bool neg = false;
    if (a < b && (t[a] == '-' || t[a] == '+')) { neg = t[a] == '-'; ++a; }
    const uint64_t v = dec_u(t, a, b, ok);
    if (v > (1ull << 62)) { *ok = false; return 0; }
    return neg ? -(int64_t)v : (int64_t)v;
}
// decimal floating point text -> float (tags of type f and B:f).  Mantissa of up to 19 digits, power of ten applied in double:
// exact for every value samtools itself prints with %g (<= 9 significant digits, |exponent| <= 22); beyond that the result can
// differ from strtof in the last bit.
__device__ inline float dec_f(const uint8_t* t, uint64_t a, uint64_t b, bool* ok) {
    bool neg = false;
    if (a < b && (t[a] == '-' || t[a] == '+')) { neg = t[a] == '-'; ++a; }
    uint64_t m = 0;
    int nd = 0, e10 = 0;
    bool any = false, dot = false;
    uint64_t p = a;
    for (; p < b; ++p) {
        const uint8_t c = t[p];
        if (c == '.' && !dot) { dot = true; continue; }
        const uint32_t d = (uint32_t)c - 48u;
        if (d > 9u) break;
        any = true;
        if (nd < 19) { m = m * 10 + d; if (m) ++nd; if (dot) --e10; }
        else if (!dot) ++e10;
    }
    if (!any) { *ok = false; return 0.f; }
    if (p < b) {
        if (t[p] != 'e' && t[p] != 'E') { *ok = false; return 0.f; }
        bool eok = true;
        const int64_t ee = dec_s(t, p + 1, b, &eok);
        if (!eok || ee > 400 || ee < -400) { *ok = false; return 0.f; }
        e10 += (int)ee;
    }
    double d = (double)m;
    const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    int e = e10;
    while (e > 22) { d *= 1e22; e -= 22; }
    while (e < -22) { d /= 1e22; e += 22; }
    d = e >= 0 ? d * p10[e] : d / p10[-e];
    const float f = (float)d;
    return neg ? -f : f;
}

__device__ inline uint32_t seq_code(uint8_t c) {      // SAMv1 4.2: "=ACMGRSVTWYHKDBN", anything else (and lower case alike) -> N
    switch (c & 0xDF) {                                // upper case
        case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6;
        case 'V': return 7; case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12;
        case 'D': return 13; case 'B': return 14; default: break;
    }
    return c == '=' ? 0u : 15u;
}
__device__ inline int cigar_op(uint8_t c) {
    switch (c) { case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4; case 'H': return 5;
                 case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1; }
}

// One alignment line -> its BAM record.  WRITE = false: sizes and keys only.  Returns the record's bytes (block_size + 4), 0 on error.
template <bool WRITE>
__device__ uint32_t bam_record(const uint8_t* t, uint64_t ls, uint64_t le, const RefTab& rt, uint8_t* out, BamIdx* ix, uint32_t* err) {
    uint64_t tab[11];
    int nt = 0;
    for (uint64_t p = ls; p < le && nt < 11; ++p) if (t[p] == '\t') tab[nt++] = p;
    if (nt < 10) { atomicOr(err, (uint32_t)BE_FIELDS); return 0; }
    const uint64_t qual_end = nt == 11 ? tab[10] : le;
    bool ok = true;
    const uint64_t l_qname = tab[0] - ls;
    if (l_qname == 0 || l_qname > 254) { atomicOr(err, (uint32_t)BE_NAME); return 0; }
    uint64_t flag = dec_u(t, tab[0] + 1, tab[1], &ok);
    const int64_t pos1 = dec_s(t, tab[2] + 1, tab[3], &ok);
    const uint64_t mapq = dec_u(t, tab[3] + 1, tab[4], &ok);
    const int64_t pnext1 = dec_s(t, tab[6] + 1, tab[7], &ok);
    const int64_t tlen = dec_s(t, tab[7] + 1, tab[8], &ok);
    if (!ok || flag > 65535 || mapq > 255 || pos1 < 0 || pos1 > 0x7fffffffll || pnext1 < 0 || pnext1 > 0x7fffffffll || tlen > 0x7fffffffll || tlen < -0x80000000ll) {
        atomicOr(err, (uint32_t)BE_NUM);
        return 0;
    }
    int32_t tid = -1, mtid = -1;
    {
        const uint64_t a = tab[1] + 1, b = tab[2];
        if (!(b - a == 1 && t[a] == '*')) tid = ref_lookup(rt, t, a, b, err);
        const uint64_t c = tab[5] + 1, d = tab[6];
        if (d - c == 1 && t[c] == '=') mtid = tid;
        else if (!(d - c == 1 && t[c] == '*')) mtid = ref_lookup(rt, t, c, d, err);
    }
    // CIGAR
    const uint64_t ca = tab[4] + 1, cb = tab[5];
    uint32_t n_cigar = 0;
    int64_t rlen = 0;
    const bool no_cigar = cb - ca == 1 && t[ca] == '*';
    if (!(flag & 4u) && (pos1 == 0 || tid < 0)) flag |= 4u;  // "mapped query cannot have zero coordinate; treated as unmapped" / no reference (htslib sam_parse1)
    if (no_cigar) { if (!(flag & 4u)) flag |= 4u; }          // "mapped query must have a CIGAR; treated as unmapped"
    else {
        uint64_t v = 0;
        bool digits = false;
        for (uint64_t p = ca; p < cb; ++p) {
            const uint32_t d = (uint32_t)t[p] - 48u;
            if (d <= 9u) { v = v * 10 + d; digits = true; if (v >= (1ull << 28)) { atomicOr(err, (uint32_t)BE_CIGAR); return 0; } continue; }
            const int op = cigar_op(t[p]);
            if (op < 0 || !digits) { atomicOr(err, (uint32_t)BE_CIGAR); return 0; }
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int64_t)v;
            ++n_cigar; v = 0; digits = false;
        }
        if (digits || n_cigar == 0 || n_cigar > 65535) { atomicOr(err, (uint32_t)BE_CIGAR); return 0; }
    }
    // SEQ / QUAL
    const uint64_t sa = tab[8] + 1, sb = tab[9], qa = tab[9] + 1, qb = qual_end;
    const uint32_t l_seq = (sb - sa == 1 && t[sa] == '*') ? 0u : (uint32_t)(sb - sa);
    const bool no_qual = qb - qa == 1 && t[qa] == '*';                     // (also for a 1-base read, as in htslib)
    if (!no_qual && (qb - qa) != l_seq) { atomicOr(err, (uint32_t)BE_SEQ); return 0; }
    const int32_t pos = (int32_t)pos1 - 1;
    int64_t span = ((flag & 4u) || rlen == 0) ? 1 : rlen;
    const int64_t end = (int64_t)pos + span;
    const uint32_t bin = reg2bin(pos, end);
    Sink o(WRITE ? out + 4 : out);                             // block_size (the first four bytes) is written at the end
    if (WRITE) {
        o.w((uint32_t)tid);
        o.w((uint32_t)pos);
        o.b((uint32_t)l_qname + 1u);
        o.b((uint32_t)mapq);
        o.h(bin);
        o.h(n_cigar);
        o.h((uint32_t)flag);
        o.w(l_seq);
        o.w((uint32_t)mtid);
        o.w((uint32_t)((int32_t)pnext1 - 1));
        o.w((uint32_t)(int32_t)tlen);
        for (uint64_t p = ls; p < tab[0]; ++p) o.b(t[p]);
        o.b(0);
        if (!no_cigar) {
            uint32_t v = 0;
            for (uint64_t p = ca; p < cb; ++p) {
                const uint32_t d = (uint32_t)t[p] - 48u;
                if (d <= 9u) { v = v * 10 + d; continue; }
                o.w((v << 4) | (uint32_t)cigar_op(t[p]));
                v = 0;
            }
        }
        for (uint32_t k = 0; k + 1 < l_seq; k += 2) o.b((seq_code(t[sa + k]) << 4) | seq_code(t[sa + k + 1]));
        if (l_seq & 1u) o.b(seq_code(t[sa + l_seq - 1]) << 4);
        if (no_qual) for (uint32_t k = 0; k < l_seq; ++k) o.b(0xFF);
        else for (uint32_t k = 0; k < l_seq; ++k) o.b(t[qa + k] - 33u);
    }
    uint32_t size = 4 + 32 + (uint32_t)l_qname + 1 + 4 * n_cigar + (l_seq + 1) / 2 + l_seq;
    // optional fields  TG:T:value
    uint64_t p = nt == 11 ? tab[10] + 1 : le;
    while (p < le) {
        uint64_t q = p;
        while (q < le && t[q] != '\t') ++q;
        if (q - p < 5 || t[p + 2] != ':' || t[p + 4] != ':') { atomicOr(err, (uint32_t)BE_TAG); return 0; }
        const uint8_t ty = t[p + 3];
        const uint64_t va = p + 5, vb = q;
        if (WRITE) { o.b(t[p]); o.b(t[p + 1]); }
        size += 2;
        bool tok = true;
        if (ty == 'A' || ty == 'a' || ty == 'c' || ty == 'C') {      // (htslib reads the lower-case forms of the BAM types as A)
            if (vb - va != 1) { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            if (WRITE) { o.b('A'); o.b(t[va]); }
            size += 2;
        } else if (ty == 'i' || ty == 'I') {
            const bool neg = va < vb && t[va] == '-';
            const int64_t x = dec_s(t, va, vb, &tok);
            if (!tok || x < -0x80000000ll || x > 0xffffffffll) { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            char c; uint32_t w;
            if (neg) { if (x >= -128) { c = 'c'; w = 1; } else if (x >= -32768) { c = 's'; w = 2; } else { c = 'i'; w = 4; } }
            else { if (x <= 255) { c = 'C'; w = 1; } else if (x <= 65535) { c = 'S'; w = 2; } else { c = 'I'; w = 4; } }
            if (WRITE) { o.b((uint8_t)c); if (w == 1) o.b((uint32_t)x); else if (w == 2) o.h((uint32_t)x); else o.w((uint32_t)x); }
            size += 1 + w;
        } else if (ty == 'f') {
            const float f = dec_f(t, va, vb, &tok);
            if (!tok) { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            if (WRITE) { o.b('f'); o.w(__float_as_uint(f)); }
            size += 5;
        } else if (ty == 'Z' || ty == 'H') {
            if (WRITE) { o.b(ty); for (uint64_t k = va; k < vb; ++k) o.b(t[k]); o.b(0); }
            size += 2 + (uint32_t)(vb - va);
        } else if (ty == 'B') {
            if (va >= vb) { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            const uint8_t sub = t[va];
            uint32_t w = 0;
            switch (sub) { case 'c': case 'C': w = 1; break; case 's': case 'S': w = 2; break; case 'i': case 'I': case 'f': w = 4; break; default: break; }
            if (!w) { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            uint32_t cnt = 0;
            for (uint64_t k = va + 1; k < vb; ++k) if (t[k] == ',') ++cnt;
            if (va + 1 < vb && t[va + 1] != ',') { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            if (WRITE) {
                o.b('B'); o.b(sub); o.w(cnt);
                uint64_t k = va + 2;
                for (uint32_t i = 0; i < cnt; ++i) {
                    uint64_t e = k;
                    while (e < vb && t[e] != ',') ++e;
                    if (sub == 'f') o.w(__float_as_uint(dec_f(t, k, e, &tok)));
                    else { const int64_t x = dec_s(t, k, e, &tok); if (w == 1) o.b((uint32_t)x); else if (w == 2) o.h((uint32_t)x); else o.w((uint32_t)x); }
                    k = e + 1;
                }
            } else {
                uint64_t k = va + 2;
                for (uint32_t i = 0; i < cnt; ++i) {
                    uint64_t e = k;
                    while (e < vb && t[e] != ',') ++e;
                    if (sub == 'f') (void)dec_f(t, k, e, &tok);
                    else {                                  // every element inside its subtype's range (htslib rejects the line otherwise; no silent wrap)
                        const int64_t x = dec_s(t, k, e, &tok);
                        const int64_t lo = sub == 'c' ? -128 : sub == 's' ? -32768 : sub == 'i' ? -2147483648ll : 0;
                        const int64_t hi = sub == 'c' ? 127 : sub == 'C' ? 255 : sub == 's' ? 32767 : sub == 'S' ? 65535 : sub == 'i' ? 2147483647ll : 4294967295ll;
                        if (x < lo || x > hi) tok = false;
                    }
                    k = e + 1;
                }
            }
            if (!tok) { atomicOr(err, (uint32_t)BE_TAG); return 0; }
            size += 2 + 4 + cnt * w;
        } else { atomicOr(err, (uint32_t)BE_TAG); return 0; }
        p = q + 1;
    }
    if (WRITE) { o.flush(); uint8_t* h = out; put32(h, size - 4); }
    if (ix) { ix->tid = tid; ix->beg = pos; ix->end = (int32_t)(end > 0x7fffffffll ? 0x7fffffffll : end); ix->bin = (bin & 0xFFFFu) | ((uint32_t)flag << 16); }      // (a position past 2^29 gives a bin beyond 16 bits: the host makes no index then)
    return size;
}

// pass 1: size of every line's record + its sort key ((tid, -1 last) | pos + 1 | reverse strand), idx = line
__global__ void k_bam_keys(const uint8_t* text, const uint64_t* starts, uint64_t nlines, RefTab rt, SortRec* rec, uint32_t* size, uint32_t* err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nlines) return;
    BamIdx ix;
    ix.tid = -1; ix.beg = -1; ix.end = 0; ix.bin = 0;
    uint64_t le = starts[j + 1] - 1;
    const uint64_t ls = starts[j];
    if (le > ls && text[le - 1] == '\r') --le;
    const uint32_t sz = bam_record<false>(text, ls, le, rt, nullptr, &ix, err);
    size[j] = sz;
    SortRec r;
    const uint64_t tkey = ix.tid < 0 ? (uint64_t)rt.nref : (uint64_t)ix.tid;
    r.hi = (tkey << 33) | ((uint64_t)(uint32_t)(ix.beg + 1) << 1) | ((ix.bin >> 20) & 1u);       // FLAG 0x10: reverse strand
    r.lo = 0; r.idx = (uint32_t)j;
    rec[j] = r;
}
__global__ void k_bam_sizes(const SortRec* rec, const uint32_t* size, uint64_t n, uint64_t* off) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) off[r] = size[rec[r].idx];
}
// pass 2: the records, in file order
__global__ void k_bam_write(const uint8_t* text, const uint64_t* starts, uint64_t n, RefTab rt, const SortRec* rec, const uint64_t* off, uint8_t* raw, BamIdx* idx, uint32_t* err) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t j = rec[r].idx;
    uint64_t le = starts[j + 1] - 1;
    const uint64_t ls = starts[j];
    if (le > ls && text[le - 1] == '\r') --le;
    BamIdx ix;
    (void)bam_record<true>(text, ls, le, rt, raw + off[r], &ix, err);
    idx[r] = ix;
}

// ---- CRC-32 (RFC 1952) of a block: every thread takes a slice, the slices are combined as polynomials ------------------------
__device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {       // a(x) * b(x) mod P(x), reflected representation (zlib's multmodp)
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
__device__ inline uint32_t crc_x8n(uint32_t nbytes, const uint32_t* x2n /* x^(2^k) mod P, k = 0..31 */) {        // x^(8 * nbytes) mod P
    uint32_t p = 1u << 31;                                            // the polynomial 1
    uint32_t n = nbytes;
    int k = 3;                                                        // bits -> bytes
    while (n) { if (n & 1u) p = crc_mulmod(x2n[k & 31], p); n >>= 1; ++k; }
    return p;
}
__device__ inline uint32_t crc_bytes(const uint8_t* d, uint32_t n, const uint32_t* tabl) {
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) c = tabl[(c ^ d[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
struct CrcTabs { uint32_t t[256]; uint32_t x2n[32]; };

// block-wide CRC of n bytes at src (global or LDS): slices of ceil(n / NTH) bytes
template <int NTH>
__device__ inline uint32_t block_crc(const uint8_t* src, uint32_t n, const uint32_t* tabl, const uint32_t* x2n, uint32_t* sh /* [NTH / 64] */) {
    const uint32_t per = (n + NTH - 1) / NTH;
    const uint32_t a = threadIdx.x * per < n ? threadIdx.x * per : n, b = a + per < n ? a + per : n;
    uint32_t c = 0;
    if (b > a) { c = crc_bytes(src + a, b - a, tabl); if (n - b) c = crc_mulmod(crc_x8n(n - b, x2n), c); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    uint32_t r = 0;
#pragma unroll
    for (int w = 0; w < NTH / 64; ++w) r ^= sh[w];
    __syncthreads();
    return r;
}
__device__ inline void bgzf_header(uint8_t* o, uint32_t bsize_minus1) {     // SAMv1 4.1
    const uint8_t h[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
    for (int i = 0; i < 16; ++i) o[i] = h[i];
    o[16] = (uint8_t)bsize_minus1; o[17] = (uint8_t)(bsize_minus1 >> 8);
}

// level 0: one stored deflate block per BGZF block
__global__ __launch_bounds__(BWG) void k_bgzf_stored(const uint8_t* raw, uint64_t nraw, const CrcTabs* ct, uint8_t* comp, uint64_t* csize) {
    __shared__ uint32_t tabl[256], x2n[32], sh[BWG / 64];
    tabl[threadIdx.x] = ct->t[threadIdx.x];
    if (threadIdx.x < 32) x2n[threadIdx.x] = ct->x2n[threadIdx.x];
    __syncthreads();
    const uint64_t b0 = (uint64_t)blockIdx.x * BGZF_RAW;
    const uint32_t n = (uint32_t)(nraw - b0 < BGZF_RAW ? nraw - b0 : BGZF_RAW);
    const uint8_t* src = raw + b0;
    uint8_t* o = comp + (uint64_t)blockIdx.x * BGZF_STRIDE;
    const uint32_t crc = block_crc<BWG>(src, n, tabl, x2n, sh);
    const uint32_t total = 18 + 5 + n + 8;
    if (threadIdx.x == 0) {
        bgzf_header(o, total - 1);
        o[18] = 1;                                         // BFINAL = 1, BTYPE = 00
        o[19] = (uint8_t)n; o[20] = (uint8_t)(n >> 8); o[21] = (uint8_t)~n; o[22] = (uint8_t)(~n >> 8);
        uint8_t* e = o + 23 + n;
        put32(e, crc); put32(e, n);
        csize[blockIdx.x] = total;
    }
    for (uint32_t i = threadIdx.x; i < n; i += BWG) o[23 + i] = src[i];
}
// the compressed blocks, packed
__global__ __launch_bounds__(BWG) void k_bgzf_pack(const uint8_t* comp, const uint64_t* coff, uint64_t nblocks, uint8_t* out) {
    const uint64_t b = blockIdx.x;
    const uint64_t o = coff[b], n = coff[b + 1] - o;
    const uint8_t* s = comp + b * BGZF_STRIDE;
    for (uint32_t i = threadIdx.x; i < n; i += BWG) out[o + i] = s[i];
}

// ---- levels 1 and 2: LZ77 + Huffman codes (RFC 1951), one workgroup per BGZF block ------------------------------------------
// Every wave parses its share of the block (an eighth by default) with its own hash table (matches stay inside the share), 64
// positions per step: a hash of four bytes names a candidate from earlier steps (distance 1 first, for runs), checked on four
// bytes in parallel; then the greedy parse of the step token by token -- literals skipped in bulk, a match's length measured
// by the whole wave at once -- and the tokens go to a list.  Level 1 codes them with the fixed tables of RFC 1951 3.2.6, level 2
// with tables built for the block (3.2.7): symbol counts gathered while parsing, code lengths by the in-place minimum-redundancy
// algorithm of Moffat and Katajainen on the sorted counts, limited to 15 (7) bits by moving codes down the Kraft sum, canonical
// codes, the code lengths themselves run-length coded as the format prescribes.  Second pass: 64 tokens per step to bits, bit
// offsets by a wave scan, ORed into the wave's stream; header, the waves' streams and the end-of-block code are then joined
// bit-exactly.  A block that does not shrink is stored.
#ifndef MKT_DZ_WAVES
#define MKT_DZ_WAVES 8
#endif
constexpr int DZ_WAVES = MKT_DZ_WAVES;                  // waves per BGZF block (4, 8 or 16); 64 KiB of hash tables in all
constexpr int DZ_THREADS = 64 * DZ_WAVES;
constexpr uint32_t DZ_Q = BGZF_RAW / DZ_WAVES;           // bytes per wave (BGZF_RAW = 2^8 * 255 divides evenly)
constexpr uint32_t DZ_HBITS = DZ_WAVES == 4 ? 12 : (DZ_WAVES == 8 ? 11 : 10);
static_assert(DZ_Q * DZ_WAVES == BGZF_RAW, "even split");
constexpr uint32_t DZ_MAXLEN = 258, DZ_MINLEN = 4;
constexpr uint32_t DZ_QWORDS = (DZ_Q * 15 / 8 + 64) / 4 + 2;      // a wave's bit stream: at most 15 bits per byte
constexpr uint32_t DZ_WAVE_SCRATCH = DZ_Q + DZ_QWORDS;            // words per wave: token list + bit stream
constexpr int DZ_STAGE = 104;                            // one step's bits of a wave: 64 tokens of <= 48 bits + the carry
constexpr int DZ_HDRW = 176;                             // dynamic block header: <= 3 + 14 + 57 + 316 * 14 bits

__device__ inline uint32_t bitrev(uint32_t v, int n) { return __brev(v) >> (32 - n); }
__device__ inline void len_code(uint32_t len, uint32_t& sym, uint32_t& eb, uint32_t& ev) {       // 3..258
    if (len == 258) { sym = 285; eb = 0; ev = 0; return; }
    const uint32_t l = len - 3;
    if (l < 8) { sym = 257 + l; eb = 0; ev = 0; return; }
    const uint32_t k = 31u - (uint32_t)__clz((int)l);          // 3..7
    eb = k - 2;
    sym = 257 + 4 * eb + 4 + ((l >> eb) & 3u);
    ev = l & ((1u << eb) - 1u);
}
__device__ inline void dist_code(uint32_t dist, uint32_t& sym, uint32_t& eb, uint32_t& ev) {     // 1..32768
    const uint32_t d = dist - 1;
    if (d < 4) { sym = d; eb = 0; ev = 0; return; }
    const uint32_t k = 31u - (uint32_t)__clz((int)d);
    eb = k - 1;
    sym = 2 * k + ((d >> eb) & 1u);
    ev = d & ((1u << eb) - 1u);
}
// code lengths for n symbols whose counts stand in A[0, n) in ascending order (Moffat & Katajainen, "In-place calculation of
// minimum-redundancy codes", 1995): A[i] becomes the length of the i-th rarest symbol's code.  One lane.
__device__ inline void mk_lengths(uint32_t* A, int n) {
    if (n == 0) return;
    if (n == 1) { A[0] = 1; return; }
    A[0] += A[1];
    int root = 0, leaf = 2, next;
    for (next = 1; next < n - 1; ++next) {
        if (leaf >= n || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint32_t)next; } else A[next] = A[leaf++];
        if (leaf >= n || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint32_t)next; } else A[next] += A[leaf++];
    }
    A[n - 2] = 0;
    for (next = n - 3; next >= 0; --next) A[next] = A[A[next]] + 1;
    int avbl = 1, used = 0, dpth = 0;
    root = n - 2; next = n - 1;
    while (avbl > 0) {
        while (root >= 0 && (int)A[root] == dpth) { ++used; --root; }
        while (avbl > used) { A[next--] = (uint32_t)dpth; --avbl; }
        avbl = 2 * used; ++dpth; used = 0;
    }
}
// One lane: lengths (<= maxbits) and canonical codes for the n used symbols listed rarest first in ssym (their counts in skey);
// table[sym] = bit-reversed code | length << 16 (0 for unused symbols).
__device__ inline void huff_codes(uint32_t* skey, const uint16_t* ssym, int n, int maxbits, uint32_t* table, int nsym) {
    uint32_t num[33];
    for (int i = 0; i <= 32; ++i) num[i] = 0;
    mk_lengths(skey, n);
    for (int i = 0; i < n; ++i) num[skey[i] > 32u ? 32u : skey[i]]++;
    for (int i = maxbits + 1; i <= 32; ++i) { num[maxbits] += num[i]; num[i] = 0; }
    uint32_t total = 0;
    for (int i = maxbits; i > 0; --i) total += num[i] << (maxbits - i);
    while (total > (1u << maxbits)) {                    // too many long codes: one leaves the deepest level, one code one level up splits
        num[maxbits]--;
        for (int i = maxbits - 1; i > 0; --i) if (num[i]) { num[i]--; num[i + 1] += 2; break; }
        --total;
    }
    for (int s = 0; s < nsym; ++s) table[s] = 0;
    int j = n;
    for (int i = 1; i <= maxbits; ++i) for (uint32_t l = num[i]; l > 0; --l) table[ssym[--j]] = (uint32_t)i << 16;      // short codes to the frequent
    uint32_t next_code[17];
    uint32_t code = 0;
    next_code[0] = 0;
    for (int i = 1; i <= maxbits; ++i) { code = (code + num[i - 1]) << 1; next_code[i] = code; }
    for (int s = 0; s < nsym; ++s) {
        const uint32_t l = table[s] >> 16;
        if (l) table[s] |= bitrev(next_code[l]++, (int)l);
    }
}
struct BitSink {                                         // one lane appends bits to words in LDS
    uint32_t* w; uint32_t n;
    __device__ inline void put(uint32_t v, uint32_t nb) {
        if (!nb) return;
        const uint32_t i = n >> 5, s = n & 31u;
        w[i] |= v << s;
        if (s + nb > 32u) w[i + 1] |= v >> (32u - s);
        n += nb;
    }
};

template <bool DYN>
__global__ __launch_bounds__(DZ_THREADS) void k_bgzf_deflate(const uint8_t* raw, uint64_t nraw, uint64_t first_block, const CrcTabs* ct, uint8_t* comp, uint64_t* csize,
                                                             uint32_t* scratch /* per block of the launch: DZ_WAVES * DZ_WAVE_SCRATCH words */) {
    __shared__ __attribute__((aligned(16))) uint32_t in32[(BGZF_RAW + 288) / 4];      // + what a length measurement reads past the end
    __shared__ uint32_t htab[DZ_WAVES][1u << DZ_HBITS];
    __shared__ uint32_t stage_[DZ_WAVES][DZ_STAGE];
    __shared__ uint32_t tabl[256], x2n[32], sh[DZ_THREADS / 64];
    __shared__ uint32_t wbits[DZ_WAVES + 2], woff[DZ_WAVES + 3], wntok[DZ_WAVES];
    __shared__ uint32_t ll_tab[288], d_tab[32];          // symbol -> reversed code | length << 16
    __shared__ uint32_t ll_hist[288], d_hist[32];
    __shared__ uint32_t skey[288], dkey[32];
    __shared__ uint16_t ssym[288], dsym[32];
    __shared__ uint32_t hdrw[DZ_HDRW];
    __shared__ uint32_t nused[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint8_t* in = reinterpret_cast<uint8_t*>(in32);
    auto word_at = [&](uint32_t off) {                   // the four bytes at any offset (two aligned reads)
        const uint32_t i = off >> 2;
        return __builtin_amdgcn_alignbyte(in32[i + 1], in32[i], off & 3u);
    };
    volatile uint32_t (*stage)[DZ_STAGE] = stage_;       // lanes of a wave hand bits to each other through it
    if (tid < 256) tabl[tid] = ct->t[tid];
    if (tid < 32) { x2n[tid] = ct->x2n[tid]; d_hist[tid] = 0; }
    if (tid < 288) ll_hist[tid] = 0;
    for (int i = tid; i < DZ_HDRW; i += DZ_THREADS) hdrw[i] = 0;
    const uint64_t blk = first_block + blockIdx.x;
    const uint64_t b0 = blk * BGZF_RAW;
    const uint32_t n = (uint32_t)(nraw - b0 < BGZF_RAW ? nraw - b0 : BGZF_RAW);
    {   // the block into LDS, 16 bytes per load (blocks start on multiples of 16; the raw buffer is readable 64 bytes past its end,
        // and what lies behind byte n is never looked at)
        const uint4* src = reinterpret_cast<const uint4*>(raw + b0);
        uint4* dst = reinterpret_cast<uint4*>(in32);
        for (uint32_t i = tid; i < (BGZF_RAW + 288) / 16; i += DZ_THREADS) dst[i] = (i << 4) < n ? src[i] : make_uint4(0, 0, 0, 0);
    }
    for (uint32_t i = lane; i < (1u << DZ_HBITS); i += 64) htab[wv][i] = 0;          // 0 = empty (positions are stored + 1)
    __syncthreads();
    const uint32_t crc = block_crc<DZ_THREADS>(in, n, tabl, x2n, sh);
    // ---- pass 1: this wave's share -> tokens (a literal byte, or 1 << 31 | length - 3 << 16 | distance - 1)
    const uint32_t q0 = wv * DZ_Q < n ? wv * DZ_Q : n, q1 = q0 + DZ_Q < n ? q0 + DZ_Q : n;
    uint32_t* wtok = scratch + ((uint64_t)blockIdx.x * DZ_WAVES + wv) * DZ_WAVE_SCRATCH;
    uint32_t* ws = wtok + DZ_Q;
    uint32_t ntok = 0;
    uint32_t cur = q0;                                   // first position not yet covered by a token
    for (uint32_t base = q0; base < q1; base += 64) {
        const uint32_t p = base + lane;
        const bool live = p < q1;
        uint32_t mlen = 0, mdist = 0;
        uint32_t h = 0;
        const bool hashable = live && p + DZ_MINLEN <= q1;
        const bool open_step = cur < base + 64u && cur < q1;      // (uniform) false: an earlier token covers all 64 positions
        // every position's candidate, checked on its first four bytes only: distance 1 (runs) before the table's
        uint32_t cand = 0xFFFFFFFFu;
        if (hashable) {
            const uint32_t v = word_at(p);
            h = (v * 2654435761u) >> (32 - DZ_HBITS);
            const uint32_t c1 = htab[wv][h];
            if (open_step) {
                if (p > q0 && word_at(p - 1u) == v) cand = p - 1u;
                else if (c1 && word_at(c1 - 1u) == v) cand = c1 - 1u;          // < base: only earlier steps have written
            }
        }
        // (all lanes have read the table before any lane of this wave writes: one wave, program order)
        if (hashable) atomicMax(&htab[wv][h], p + 1);
        if (!open_step) continue;
        // greedy parse of the step, token by token; literals are skipped in bulk, a match's length is measured by the whole
        // wave at once (lane i compares bytes 4 i .. 4 i + 3), so the cost does not grow with the length
        uint64_t sel = 0;
        {
            const uint64_t hasm = __ballot(cand != 0xFFFFFFFFu);
            const uint32_t nlive = q1 - base < 64u ? q1 - base : 64u;
            uint32_t pos = cur - base;
            while (pos < nlive) {
                const uint64_t rem = hasm >> pos;
                const uint32_t lit = rem ? (uint32_t)__builtin_ctzll(rem) : 64u;       // literals up to the next candidate
                if (lit) {
                    const uint32_t e = pos + lit < nlive ? pos + lit : nlive;
                    sel |= (e >= 64u ? ~0ull : ((1ull << e) - 1ull)) & ~((1ull << pos) - 1ull);
                    pos = e;
                    continue;
                }
                const uint32_t c = (uint32_t)__shfl((int)cand, (int)pos, 64), pp = base + pos;
                const uint32_t lim = q1 - pp < DZ_MAXLEN ? q1 - pp : DZ_MAXLEN;
                const uint32_t x = word_at(c + 4u * (uint32_t)lane) ^ word_at(pp + 4u * (uint32_t)lane);
                const uint64_t ne = __ballot(x != 0u);
                uint32_t len;
                if (ne) {
                    const int j = (int)__builtin_ctzll(ne);
                    const uint32_t xj = (uint32_t)__shfl((int)x, j, 64);
                    len = 4u * (uint32_t)j + ((uint32_t)__builtin_ctz(xj) >> 3);
                } else {                                         // 256 equal bytes: two more decide between 256, 257, 258
                    const uint32_t y = word_at(c + 256u) ^ word_at(pp + 256u);
                    len = 256u + ((y & 0xFFu) ? 0u : ((y & 0xFF00u) ? 1u : 2u));
                }
                if (len > lim) len = lim;
                if ((uint32_t)lane == pos) { mlen = len; mdist = pp - c; }
                sel |= 1ull << pos;
                pos += len;
            }
            cur = base + pos;
        }
        if ((sel >> lane) & 1ull) {
            const uint32_t rank = (uint32_t)__popcll(sel & ((1ull << lane) - 1ull));
            wtok[ntok + rank] = mlen ? (0x80000000u | ((mlen - 3u) << 16) | (mdist - 1u)) : (uint32_t)in[p];
            if (DYN) {
                if (mlen) {
                    uint32_t sy, eb, ev;
                    len_code(mlen, sy, eb, ev);
                    atomicAdd(&ll_hist[sy], 1u);
                    dist_code(mdist, sy, eb, ev);
                    atomicAdd(&d_hist[sy], 1u);
                } else atomicAdd(&ll_hist[in[p]], 1u);
            }
        }
        ntok += (uint32_t)__popcll(sel);
    }
    if (lane == 0) wntok[wv] = ntok;
    __syncthreads();
    // ---- the code tables and the block header
    uint32_t hdr_bits = 3;
    if (!DYN) {
        for (int s = tid; s < 288; s += DZ_THREADS) {        // RFC 1951 3.2.6
            uint32_t code, nb;
            if (s < 144) { code = bitrev(0x30 + s, 8); nb = 8; }
            else if (s < 256) { code = bitrev(0x190 + (s - 144), 9); nb = 9; }
            else if (s < 280) { code = bitrev(s - 256, 7); nb = 7; }
            else { code = bitrev(0xC0 + (s - 280), 8); nb = 8; }
            ll_tab[s] = code | (nb << 16);
        }
        if (tid < 32) d_tab[tid] = bitrev((uint32_t)tid, 5) | (5u << 16);
        if (tid == 0) hdrw[0] = 3u;                          // BFINAL = 1, BTYPE = 01
        __syncthreads();
    } else {
        if (tid == 0) {
            ll_hist[256] = 1;                                // end of block
            int nz = 0;
            for (int s = 0; s < 30; ++s) nz += d_hist[s] != 0u;
            if (nz == 0) { d_hist[0] = 1; d_hist[1] = 1; }       // (as zlib: never fewer than two distance codes)
            else if (nz == 1) { if (d_hist[0]) d_hist[1] = 1; else d_hist[0] = 1; }
            int nl = 0;
            for (int s = 0; s < 286; ++s) nl += ll_hist[s] != 0u;
            if (nl < 2) { if (!ll_hist[0]) ll_hist[0] = 1; else ll_hist[1] = 1; }
        }
        __syncthreads();
        // used symbols in ascending order of their counts (rank by counting; ties by symbol)
        for (int t = tid; t < 286; t += DZ_THREADS) {
            const uint32_t f = ll_hist[t];
            if (f) {
                uint32_t r = 0;
                for (int j = 0; j < 286; ++j) { const uint32_t g = ll_hist[j]; r += (g != 0u) && (g < f || (g == f && j < t)); }
                skey[r] = f; ssym[r] = (uint16_t)t;
            }
        }
        for (int t = DZ_THREADS - 1 - tid; t < 30; t += DZ_THREADS) {           // (the last threads: they have no literal symbol to rank)
            const uint32_t f = d_hist[t];
            if (f) {
                uint32_t r = 0;
                for (int j = 0; j < 30; ++j) { const uint32_t g = d_hist[j]; r += (g != 0u) && (g < f || (g == f && j < t)); }
                dkey[r] = f; dsym[r] = (uint16_t)t;
            }
        }
        if (tid == 0) { int a = 0; for (int s = 0; s < 286; ++s) a += ll_hist[s] != 0u; nused[0] = (uint32_t)a; }
        if (tid == 64) { int a = 0; for (int s = 0; s < 30; ++s) a += d_hist[s] != 0u; nused[1] = (uint32_t)a; }
        __syncthreads();
        if (tid == 0) huff_codes(skey, ssym, (int)nused[0], 15, ll_tab, 288);
        if (tid == 64) huff_codes(dkey, dsym, (int)nused[1], 15, d_tab, 32);
        __syncthreads();
        if (tid == 0) {                                      // RFC 1951 3.2.7
            int nll = 286, nd = 30;
            while (nll > 257 && !(ll_tab[nll - 1] >> 16)) --nll;
            while (nd > 1 && !(d_tab[nd - 1] >> 16)) --nd;
            // the code lengths, run-length coded: symbols 0..15 literal lengths, 16 repeat previous 3..6, 17 zeros 3..10, 18 zeros 11..138
            uint16_t* cl = reinterpret_cast<uint16_t*>(skey);         // (the sort arrays are free again) symbol | extra value << 8
            int ncl = 0;
            uint32_t clh[19];
            for (int i = 0; i < 19; ++i) clh[i] = 0;
            const int tot = nll + nd;
            auto L = [&](int i) { return i < nll ? ll_tab[i] >> 16 : d_tab[i - nll] >> 16; };
            for (int i = 0; i < tot;) {
                const uint32_t v = L(i);
                int run = 1;
                while (i + run < tot && L(i + run) == v) ++run;
                i += run;
                if (v == 0) {
                    while (run >= 11) { const int r = run < 138 ? run : 138; cl[ncl++] = (uint16_t)(18 | ((r - 11) << 8)); clh[18]++; run -= r; }
                    if (run >= 3) { cl[ncl++] = (uint16_t)(17 | ((run - 3) << 8)); clh[17]++; run = 0; }
                    while (run-- > 0) { cl[ncl++] = 0; clh[0]++; }
                } else {
                    cl[ncl++] = (uint16_t)v; clh[v]++; --run;
                    while (run >= 3) { const int r = run < 6 ? run : 6; cl[ncl++] = (uint16_t)(16 | ((r - 3) << 8)); clh[16]++; run -= r; }
                    while (run-- > 0) { cl[ncl++] = (uint16_t)v; clh[v]++; }
                }
            }
            // the code for those 19 symbols (<= 7 bits; never fewer than two of them)
            { int nzc = 0, only = 0; for (int s2 = 0; s2 < 19; ++s2) if (clh[s2]) { ++nzc; only = s2; } if (nzc == 1) clh[only ? 0 : 1] = 1; }
            uint32_t ck[19], ctab[19];
            uint16_t cs[19];
            int nc = 0;
            for (int s = 0; s < 19; ++s) if (clh[s]) { int k = nc++; while (k > 0 && (ck[k - 1] > clh[s])) { ck[k] = ck[k - 1]; cs[k] = cs[k - 1]; --k; } ck[k] = clh[s]; cs[k] = (uint16_t)s; }
            huff_codes(ck, cs, nc, 7, ctab, 19);
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int nclc = 19;
            while (nclc > 4 && !(ctab[order[nclc - 1]] >> 16)) --nclc;
            BitSink B{hdrw, 0};
            B.put(5u, 3);                                    // BFINAL = 1, BTYPE = 10
            B.put((uint32_t)(nll - 257), 5); B.put((uint32_t)(nd - 1), 5); B.put((uint32_t)(nclc - 4), 4);
            for (int i = 0; i < nclc; ++i) B.put(ctab[order[i]] >> 16, 3);
            for (int i = 0; i < ncl; ++i) {
                const uint32_t sy = cl[i] & 0xFFu, ev = cl[i] >> 8;
                B.put(ctab[sy] & 0xFFFFu, ctab[sy] >> 16);
                if (sy == 16) B.put(ev, 2); else if (sy == 17) B.put(ev, 3); else if (sy == 18) B.put(ev, 7);
            }
            nused[0] = B.n;
        }
        __syncthreads();
        hdr_bits = nused[0];
    }
    // ---- pass 2: tokens -> bits
    ntok = wntok[wv];
    uint32_t wpos = 0;                                   // whole words already flushed to ws
    uint32_t carry_bits = 0;                             // bits waiting in stage[wv][0]
    if (lane == 0) stage[wv][0] = 0;
    for (uint32_t t0 = 0; t0 < ntok; t0 += 64) {
        uint32_t a = 0, na = 0, b = 0, nbb = 0;          // literal / length part, distance part
        if (t0 + lane < ntok) {
            const uint32_t tok = wtok[t0 + lane];
            if (tok >> 31) {
                uint32_t sy, eb, ev;
                len_code(((tok >> 16) & 0xFFu) + 3u, sy, eb, ev);
                const uint32_t e = ll_tab[sy];
                a = (e & 0xFFFFu) | (ev << (e >> 16)); na = (e >> 16) + eb;
                dist_code((tok & 0xFFFFu) + 1u, sy, eb, ev);
                const uint32_t d = d_tab[sy];
                b = (d & 0xFFFFu) | (ev << (d >> 16)); nbb = (d >> 16) + eb;
            } else { const uint32_t e = ll_tab[tok]; a = e & 0xFFFFu; na = e >> 16; }
        }
        const uint32_t nb = na + nbb;
        uint32_t inc = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += y; }
        const uint32_t tot = (uint32_t)__shfl((int)inc, 63, 64);
        const uint32_t bo = carry_bits + inc - nb;
        // stage[1..] cleared for this step (word 0 holds the carry)
        for (uint32_t i = 1 + lane; i < (uint32_t)DZ_STAGE; i += 64) stage[wv][i] = 0;
        __builtin_amdgcn_wave_barrier();
        if (na) {
            const uint32_t w = bo >> 5, s = bo & 31u;
            const uint64_t lo = (uint64_t)a << s;
            atomicOr(&stage_[wv][w], (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(&stage_[wv][w + 1], (uint32_t)(lo >> 32));
        }
        if (nbb) {
            const uint32_t w = (bo + na) >> 5, s = (bo + na) & 31u;
            const uint64_t lo = (uint64_t)b << s;
            atomicOr(&stage_[wv][w], (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(&stage_[wv][w + 1], (uint32_t)(lo >> 32));
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t endb = carry_bits + tot, full = endb >> 5;
        for (uint32_t i = lane; i < full; i += 64) ws[wpos + i] = stage[wv][i];
        const uint32_t rest = stage[wv][full];
        __builtin_amdgcn_wave_barrier();
        wpos += full;
        carry_bits = endb & 31u;
        if (lane == 0) stage[wv][0] = carry_bits ? rest : 0u;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { ws[wpos] = stage[wv][0]; wbits[1 + wv] = wpos * 32u + carry_bits; }
    if (tid == 0) { wbits[0] = hdr_bits; wbits[DZ_WAVES + 1] = ll_tab[256] >> 16; }
    __threadfence_block();
    __syncthreads();
    // ---- join: header, the waves' streams in order, the end-of-block code
    if (tid == 0) { uint32_t a = 0; for (int w = 0; w < DZ_WAVES + 2; ++w) { woff[w] = a; a += wbits[w]; } woff[DZ_WAVES + 2] = a; }
    __syncthreads();
    const uint32_t total_bits = woff[DZ_WAVES + 2];
    const uint32_t cbytes = (total_bits + 7u) >> 3;
    uint8_t* o = comp + blk * BGZF_STRIDE;
    if (cbytes >= n + 5u) {                               // did not shrink: stored
        const uint32_t total = 18 + 5 + n + 8;
        if (tid == 0) {
            bgzf_header(o, total - 1);
            o[18] = 1; o[19] = (uint8_t)n; o[20] = (uint8_t)(n >> 8); o[21] = (uint8_t)~n; o[22] = (uint8_t)(~n >> 8);
            uint8_t* e = o + 23 + n;
            put32(e, crc); put32(e, n);
            csize[blk] = total;
        }
        for (uint32_t i = tid; i < n; i += DZ_THREADS) o[23 + i] = in[i];
        return;
    }
    const uint32_t eobw = ll_tab[256] & 0xFFFFu;
    const uint32_t* s0 = scratch + ((uint64_t)blockIdx.x * DZ_WAVES) * DZ_WAVE_SCRATCH + DZ_Q;
    // output word k holds bits [32 k, 32 k + 32) of the joined stream
    const uint32_t nwords = (total_bits + 31u) >> 5;
    for (uint32_t k = tid; k < nwords; k += DZ_THREADS) {
        uint32_t word = 0;
        const uint32_t lo = k << 5, hi = lo + 32u;
        for (int s = 0; s < DZ_WAVES + 2; ++s) {
            const uint32_t so = woff[s], sn = wbits[s];
            if (sn == 0u || so >= hi || so + sn <= lo) continue;
            auto sword = [&](uint32_t w) -> uint32_t {
                if (s == 0) return hdrw[w];
                if (s == DZ_WAVES + 1) return w ? 0u : eobw;
                return s0[(uint64_t)(s - 1) * DZ_WAVE_SCRATCH + w];
            };
            // bits of stream s that land in this word: stream bit j -> joined bit so + j
            const int64_t j0 = (int64_t)lo - (int64_t)so;            // stream bit at the word's bit 0 (may be negative)
            uint32_t v;
            if (j0 >= 0) {
                const uint32_t w = (uint32_t)j0 >> 5, sft = (uint32_t)j0 & 31u;
                const uint64_t two = (uint64_t)sword(w) | ((uint64_t)(((w + 1u) << 5) < sn ? sword(w + 1) : 0u) << 32);
                v = (uint32_t)(two >> sft);
                const uint32_t avail = sn - (uint32_t)j0;             // stream bits from j0 on
                if (avail < 32u) v &= (1u << avail) - 1u;
            } else {
                const uint32_t sft = (uint32_t)(-j0);                 // 1..31
                v = sword(0) << sft;
                if (sn + sft < 32u) v &= (1u << (sn + sft)) - 1u;
            }
            word |= v;
        }
        uint8_t* d = o + 18 + 4u * k;
        const uint32_t left = cbytes - 4u * k;
        d[0] = (uint8_t)word;
        if (left > 1) d[1] = (uint8_t)(word >> 8);
        if (left > 2) d[2] = (uint8_t)(word >> 16);
        if (left > 3) d[3] = (uint8_t)(word >> 24);
    }
    if (tid == 0) {
        const uint32_t total = 18 + cbytes + 8;
        bgzf_header(o, total - 1);
        uint8_t* e = o + 18 + cbytes;
        put32(e, crc); put32(e, n);
        csize[blk] = total;
    }
}

// ---- BAI reductions -------------------------------------------------------------------------------------------------------
struct BaiHead { uint32_t r; int32_t tid; uint32_t bin; uint32_t pad; uint64_t voff; };
struct BaiRef { unsigned long long n_mapped, n_unmapped, beg, end; };          // beg: min start voffset, end: max end voffset
__device__ inline uint64_t voffset(uint64_t uoff, const uint64_t* coff) {
    const uint64_t b = uoff / BGZF_RAW;
    return (coff[b] << 16) | (uoff - b * BGZF_RAW);
}
__global__ void k_bai(const BamIdx* idx, const uint64_t* off /* [n + 1] */, uint64_t n, uint64_t hdr_len, const uint64_t* coff, const uint64_t* lin_off, uint32_t nref,
                      unsigned long long* lin, BaiRef* refs, uint64_t* head_flag /* [n]: 1 where a run of records in one bin starts */, unsigned long long* no_coor) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) head_flag[r] = 0;
    const int lane = threadIdx.x & 63;
    BamIdx x;
    x.tid = -1; x.beg = 0; x.end = 0; x.bin = 0;
    if (r < n) x = idx[r];
    const bool placed = r < n && x.tid >= 0;
    const uint64_t nc = __ballot(r < n && x.tid < 0);
    if (nc && lane == (int)__builtin_ctzll(nc)) atomicAdd(no_coor, (unsigned long long)__popcll(nc));
    const uint64_t pm = __ballot(placed);
    if (!pm) return;
    uint64_t v0 = 0, v1 = 0;
    if (placed) { v0 = voffset(hdr_len + off[r], coff); v1 = voffset(hdr_len + off[r + 1], coff); }
    const uint32_t bin = x.bin & 0xFFFFu;
    const bool unm = (x.bin >> 18) & 1u;                           // FLAG 0x4
    // per-reference counts and file range: the records are sorted, so a wave nearly always holds ONE reference -> one set of atomics
    const int first = (int)__builtin_ctzll(pm);
    const int32_t tid0 = __shfl(x.tid, first, 64);
    if (__ballot(placed && x.tid == tid0) == pm) {
        const uint32_t cu = (uint32_t)__popcll(__ballot(placed && unm)), cm = (uint32_t)__popcll(pm) - cu;
        uint64_t lo = placed ? v0 : ~0ull, hi = placed ? v1 : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint64_t a = (uint64_t)__shfl_xor((long long)lo, d, 64), b = (uint64_t)__shfl_xor((long long)hi, d, 64);
            lo = a < lo ? a : lo; hi = b > hi ? b : hi;
        }
        if (lane == first) {
            BaiRef* R = &refs[tid0];
            if (cm) atomicAdd(&R->n_mapped, (unsigned long long)cm);
            if (cu) atomicAdd(&R->n_unmapped, (unsigned long long)cu);
            atomicMin(&R->beg, (unsigned long long)lo);
            atomicMax(&R->end, (unsigned long long)hi);
        }
    } else if (placed) {
        BaiRef* R = &refs[x.tid];
        if (unm) atomicAdd(&R->n_unmapped, 1ull); else atomicAdd(&R->n_mapped, 1ull);
        atomicMin(&R->beg, (unsigned long long)v0);
        atomicMax(&R->end, (unsigned long long)v1);
    }
    if (!placed) return;
    const uint64_t nwin = lin_off[x.tid + 1] - lin_off[x.tid];
    int64_t w0 = x.beg < 0 ? 0 : (x.beg >> 14), w1 = (x.end > 0 ? x.end - 1 : 0) >> 14;
    if (w1 < w0) w1 = w0;
    for (int64_t w = w0; w <= w1 && (uint64_t)w < nwin; ++w) atomicMin(&lin[lin_off[x.tid] + (uint64_t)w], (unsigned long long)v0);
    // a record that reaches past its reference's LN (the linear index is sized by LN) or past the 2^29 bases a BAI can address:
    // no index can describe it (no_coor[1] != 0 -> the host writes none and says why)
    if ((uint64_t)w1 >= nwin || x.end > (1 << 29) || bin > 37449u) atomicOr(no_coor + 1, 1ull);
    bool head = r == 0;
    if (!head) { const BamIdx y = idx[r - 1]; head = y.tid != x.tid || (y.bin & 0xFFFFu) != bin; }
    if (head) head_flag[r] = 1;
}
// the run starts, in file order (head_pos = exclusive scan of head_flag)
__global__ void k_bai_heads(const BamIdx* idx, const uint64_t* off, uint64_t n, uint64_t hdr_len, const uint64_t* coff, const uint64_t* head_pos, BaiHead* heads) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const BamIdx x = idx[r];
    if (x.tid < 0) return;
    bool head = r == 0;
    if (!head) { const BamIdx y = idx[r - 1]; head = y.tid != x.tid || ((y.bin ^ x.bin) & 0xFFFFu) != 0u; }
    if (!head) return;
    BaiHead h;
    h.r = (uint32_t)r; h.tid = x.tid; h.bin = x.bin & 0xFFFFu; h.pad = 0; h.voff = voffset(hdr_len + off[r], coff);
    heads[head_pos[r]] = h;
}

}  // namespace mkt

// ---------------------------------------------------------------------------------------------------------------------------
struct mkt_bam {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_text = nullptr; size_t cap = 0, len = 0;
    bool header_done = false, ran = false;
    std::string header, pending;
    uint8_t* d_bam = nullptr; uint64_t bam_len = 0;
    std::string bai;
    uint64_t records = 0;
    // pinned staging for callers that move files (mkt_bam_window / _commit / _read): two slots, used alternately
    char* h_io[2] = {nullptr, nullptr};
    hipEvent_t ev_io[2] = {nullptr, nullptr};
    bool io_busy[2] = {false, false};
    int io_slot = 0;
    std::string err, note;
};
constexpr size_t kBamIoCap = (size_t)64 << 20;
static int bfail(mkt_bam* s, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) s->err = buf;
    return code;
}
#define BCHK(s, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bfail((s), MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

static int bam_reserve(mkt_bam* s, size_t need) {
    if (need <= s->cap) return MKT_OK;
    size_t ncap = s->cap ? s->cap : ((size_t)64 << 20);
    while (ncap < need) ncap *= 2;
    uint8_t* nb = nullptr;
    // (doubling keeps old and new side by side during the copy; when that does not fit: just enough; when that does not either, the
    //  input is larger than one GPU takes -- text + records + compressed blocks are resident together, ~2.2 x the .sam: INTEGRATION.md)
    hipError_t e = hipMalloc((void**)&nb, ncap + 64);
    if (e != hipSuccess) { (void)hipGetLastError(); ncap = need + ((size_t)64 << 20); e = hipMalloc((void**)&nb, ncap + 64); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        return bfail(s, MKT_E_NOMEM, "the SAM input (%.1f GB so far) does not fit this GPU (%.1f of %.1f GB free): sam2bam keeps the whole input in HBM "
                     "(samtools sort spills to disk instead); convert the modes' .sam files one by one", (double)need / 1e9, (double)fr / 1e9, (double)tot / 1e9);
    }
    if (s->d_text) {
        BCHK(s, hipStreamSynchronize(s->stream));
        if (s->len) BCHK(s, hipMemcpy(nb, s->d_text, s->len, hipMemcpyDeviceToDevice));
        BCHK(s, hipFree(s->d_text));
    }
    s->d_text = nb; s->cap = ncap;
    return MKT_OK;
}
static void crc_tables(CrcTabs* ct) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        ct->t[i] = c;
    }
    auto mul = [](uint32_t a, uint32_t b) {
        uint32_t m = 1u << 31, p = 0;
        for (;;) {
            if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
            m >>= 1;
            b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
        }
        return p;
    };
    uint32_t p = 1u << 30;                              // x^1
    ct->x2n[0] = p;
    for (int k = 1; k < 32; ++k) { p = mul(p, p); ct->x2n[k] = p; }
}
static void put_le32(std::string& s, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; s.append(b, 4); }
static void put_le64(std::string& s, uint64_t v) { put_le32(s, (uint32_t)v); put_le32(s, (uint32_t)(v >> 32)); }

extern "C" {

int mkt_bam_create(int device, mkt_bam** out) {
    if (!out) return MKT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MKT_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return MKT_E_ARG;
    mkt_bam* s = new mkt_bam();
    s->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return MKT_E_HIP; }
    *out = s;
    return MKT_OK;
}
void mkt_bam_destroy(mkt_bam* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->d_text) (void)hipFree(s->d_text);
    if (s->d_bam) (void)hipFree(s->d_bam);
    for (int k = 0; k < 2; ++k) { if (s->h_io[k]) (void)hipHostFree(s->h_io[k]); if (s->ev_io[k]) (void)hipEventDestroy(s->ev_io[k]); }
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}
const char* mkt_bam_error(const mkt_bam* s) { return s ? s->err.c_str() : ""; }
const char* mkt_bam_note(const mkt_bam* s) { return s ? s->note.c_str() : ""; }

// The next bytes of the SAM stream (any chunking).  Leading '@' lines are the header; everything from the first other line on
// is alignment text and goes to the device.
static int bam_add_bytes(mkt_bam* s, const char* bytes, size_t n, bool pinned_async) {
    if (s->ran) return bfail(s, MKT_E_STATE, "add after run");
    BCHK(s, hipSetDevice(s->device));
    if (!s->header_done) {
        s->pending.append(bytes, n);
        size_t p = 0;
        for (;;) {
            if (p >= s->pending.size()) break;
            if (s->pending[p] != '@') { s->header_done = true; break; }
            const size_t e = s->pending.find('\n', p);
            if (e == std::string::npos) break;              // an unfinished header line: wait for more
            s->header.append(s->pending, p, e + 1 - p);
            p = e + 1;
        }
        s->pending.erase(0, p);
        if (!s->header_done) return MKT_OK;
        std::string rest;
        rest.swap(s->pending);
        if (rest.empty()) return MKT_OK;
        int rc = bam_reserve(s, s->len + rest.size() + 1);
        if (rc) return rc;
        BCHK(s, hipMemcpy(s->d_text + s->len, rest.data(), rest.size(), hipMemcpyHostToDevice));
        s->len += rest.size();
        return MKT_OK;
    }
    int rc = bam_reserve(s, s->len + n + 1);
    if (rc) return rc;
    if (n) BCHK(s, hipMemcpyAsync(s->d_text + s->len, bytes, n, hipMemcpyHostToDevice, s->stream));
    if (!pinned_async) BCHK(s, hipStreamSynchronize(s->stream));       // the caller may reuse `bytes`
    s->len += n;
    return MKT_OK;
}
int mkt_bam_add(mkt_bam* s, const char* bytes, size_t n) {
    if (!s || (n && !bytes)) return MKT_E_ARG;
    return bam_add_bytes(s, bytes, n, false);
}
// room for `bytes` of alignment text, so that the buffer does not grow (and get copied) while the stream comes in
int mkt_bam_reserve(mkt_bam* s, size_t bytes) {
    if (!s) return MKT_E_ARG;
    if (s->ran) return bfail(s, MKT_E_STATE, "reserve after run");
    BCHK(s, hipSetDevice(s->device));
    return bam_reserve(s, bytes + 1);
}
static int bam_io_slot(mkt_bam* s, int k) {
    if (!s->h_io[k]) {
        BCHK(s, hipHostMalloc((void**)&s->h_io[k], kBamIoCap, hipHostMallocDefault));
        BCHK(s, hipEventCreateWithFlags(&s->ev_io[k], hipEventDisableTiming));
    }
    if (s->io_busy[k]) { BCHK(s, hipEventSynchronize(s->ev_io[k])); s->io_busy[k] = false; }
    return MKT_OK;
}
// A pinned host buffer for the next bytes of the SAM stream (read a file straight into it), then mkt_bam_commit: the copy to the
// GPU runs while the caller fills the other buffer.
int mkt_bam_window(mkt_bam* s, char** buf, size_t* cap) {
    if (!s || !buf || !cap) return MKT_E_ARG;
    if (s->ran) return bfail(s, MKT_E_STATE, "window after run");
    BCHK(s, hipSetDevice(s->device));
    int rc = bam_io_slot(s, s->io_slot);
    if (rc) return rc;
    *buf = s->h_io[s->io_slot]; *cap = kBamIoCap;
    return MKT_OK;
}
int mkt_bam_commit(mkt_bam* s, size_t n) {
    if (!s || n > kBamIoCap) return MKT_E_ARG;
    const int k = s->io_slot;
    if (!s->h_io[k]) return bfail(s, MKT_E_STATE, "commit without window");
    int rc = bam_add_bytes(s, s->h_io[k], n, true);
    if (rc) return rc;
    BCHK(s, hipEventRecord(s->ev_io[k], s->stream));
    s->io_busy[k] = true;
    s->io_slot = k ^ 1;
    return MKT_OK;
}
// alignment lines that are already on the device (no header lines)
int mkt_bam_add_device(mkt_bam* s, const void* d_bytes, size_t n) {
    if (!s || (n && !d_bytes)) return MKT_E_ARG;
    if (s->ran) return bfail(s, MKT_E_STATE, "add after run");
    if (!s->pending.empty()) return bfail(s, MKT_E_STATE, "device text after an unfinished header line");
    s->header_done = true;
    BCHK(s, hipSetDevice(s->device));
    int rc = bam_reserve(s, s->len + n + 1);
    if (rc) return rc;
    if (n) BCHK(s, hipMemcpyAsync(s->d_text + s->len, d_bytes, n, hipMemcpyDeviceToDevice, s->stream));
    BCHK(s, hipStreamSynchronize(s->stream));
    s->len += n;
    return MKT_OK;
}

int mkt_bam_run(mkt_bam* s, int sorted, int level, uint64_t* records, uint64_t* bam_bytes, uint64_t* bai_bytes) {
    if (!s) return MKT_E_ARG;
    if (s->ran) return bfail(s, MKT_E_STATE, "run twice");
    BCHK(s, hipSetDevice(s->device));
    if (records) *records = 0;
    if (bam_bytes) *bam_bytes = 0;
    if (bai_bytes) *bai_bytes = 0;
    s->ran = true;
    if (!s->pending.empty()) {                                     // a last header line without newline, or a file of header lines only
        if (s->pending[0] == '@') { s->header += s->pending; s->header += '\n'; s->pending.clear(); }
    }
    hipStream_t st = s->stream;
    const bool verbose = getenv("MKT_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {                            // (call sites follow a stream synchronisation)
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mkt_bam] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    BCHK(s, hipStreamSynchronize(st));                            // (copies of mkt_bam_commit may still be on their way)
    if (s->len) {
        char last = 0;
        BCHK(s, hipMemcpy(&last, s->d_text + s->len - 1, 1, hipMemcpyDeviceToHost));
        if (last != '\n') { const char nl = '\n'; int rc = bam_reserve(s, s->len + 2); if (rc) return rc; BCHK(s, hipMemcpy(s->d_text + s->len, &nl, 1, hipMemcpyHostToDevice)); ++s->len; }
    }
    // ---- header: text (with @HD SO:coordinate when sorting) and the reference dictionary from @SQ
    std::string text = s->header;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    {
        size_t p = 0;
        while (p < text.size()) {
            size_t e = text.find('\n', p);
            if (e == std::string::npos) e = text.size();
            if (e - p >= 3 && text.compare(p, 3, "@SQ") == 0) {
                std::string sn;
                long long ln = -1;
                size_t q = p;
                while (q < e) {
                    size_t f = text.find('\t', q);
                    if (f == std::string::npos || f > e) f = e;
                    if (f - q > 3 && text.compare(q, 3, "SN:") == 0) sn = text.substr(q + 3, f - q - 3);
                    if (f - q > 3 && text.compare(q, 3, "LN:") == 0) ln = atoll(text.substr(q + 3, f - q - 3).c_str());
                    q = f + 1;
                }
                if (sn.empty() || ln < 0 || ln > 0x7fffffffll) return bfail(s, MKT_E_ARG, "bad @SQ line in the header (SN / LN)");
                names.push_back(sn); lens.push_back((uint32_t)ln);
            }
            p = e + 1;
        }
        if (sorted) {
            if (text.compare(0, 3, "@HD") == 0) {
                size_t e = text.find('\n');
                if (e == std::string::npos) e = text.size();
                std::string hd = text.substr(0, e), out;
                size_t q = 0;
                bool had = false;
                while (q <= hd.size()) {
                    size_t f = hd.find('\t', q);
                    if (f == std::string::npos) f = hd.size();
                    std::string fld = hd.substr(q, f - q);
                    if (fld.compare(0, 3, "SO:") == 0) { fld = "SO:coordinate"; had = true; }
                    if (fld.compare(0, 3, "GO:") != 0) { if (!out.empty()) out += '\t'; out += fld; }       // (a grouping claim does not survive a sort)
                    q = f + 1;
                }
                if (!had) out += "\tSO:coordinate";
                text = out + text.substr(e);
            } else text = "@HD\tVN:1.6\tSO:coordinate\n" + text;
        }
    }
    const uint32_t nref = (uint32_t)names.size();
    std::string hdr;                                               // SAMv1 4.2: magic, l_text, text, n_ref, (l_name, name, l_ref)*
    hdr.append("BAM\1", 4);
    put_le32(hdr, (uint32_t)text.size());
    hdr += text;
    put_le32(hdr, nref);
    for (uint32_t i = 0; i < nref; ++i) { put_le32(hdr, (uint32_t)names[i].size() + 1); hdr += names[i]; hdr += '\0'; put_le32(hdr, lens[i]); }
    const uint64_t hdr_len = hdr.size();
    // reference table for the device
    uint32_t tcap = 64;
    while (tcap < 4 * nref + 4) tcap <<= 1;
    std::vector<unsigned long long> th(tcap, 0ull);
    std::vector<int32_t> tidv(tcap, -1);
    std::vector<uint32_t> noff(nref + 1, 0);
    std::string blob;
    for (uint32_t i = 0; i < nref; ++i) {
        noff[i] = (uint32_t)blob.size(); blob += names[i];
        unsigned long long h = 0xcbf29ce484222325ull;
        for (unsigned char c : names[i]) { h ^= c; h *= 0x100000001b3ull; }
        if (!h) h = 1;
        uint32_t k = (uint32_t)(h >> 20) & (tcap - 1);
        bool dup = false;
        while (th[k]) { if (th[k] == h && names[(size_t)tidv[k]] == names[i]) { dup = true; break; } k = (k + 1) & (tcap - 1); }
        if (dup) return bfail(s, MKT_E_ARG, "reference name %s twice in the header", names[i].c_str());
        th[k] = h; tidv[k] = (int32_t)i;
    }
    noff[nref] = (uint32_t)blob.size();

    std::vector<void*> owned;
    auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); owned.clear(); };
#define BALLOC(ptr, bytes_) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes_)); if (e_ != hipSuccess) { cleanup(); return bfail(s, MKT_E_NOMEM, "hipMalloc of %zu bytes failed: %s", (size_t)(bytes_), hipGetErrorString(e_)); } owned.push_back((void*)(ptr)); } while (0)
#define BRUN(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return bfail(s, MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    auto drop = [&](void* p) { (void)hipFree(p); owned.erase(std::remove(owned.begin(), owned.end(), p), owned.end()); };
    unsigned long long* d_th = nullptr; int32_t* d_tid = nullptr; uint32_t* d_noff = nullptr; uint8_t* d_blob = nullptr;
    CrcTabs* d_ct = nullptr;
    uint32_t* d_err = nullptr;
    BALLOC(d_th, tcap * sizeof(unsigned long long));
    BALLOC(d_tid, tcap * sizeof(int32_t));
    BALLOC(d_noff, (nref + 1) * sizeof(uint32_t));
    BALLOC(d_blob, blob.size() + 16);
    BALLOC(d_ct, sizeof(CrcTabs));
    BALLOC(d_err, 256);
    CrcTabs ct;
    crc_tables(&ct);
    BRUN(hipMemcpyAsync(d_th, th.data(), tcap * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    BRUN(hipMemcpyAsync(d_tid, tidv.data(), tcap * sizeof(int32_t), hipMemcpyHostToDevice, st));
    BRUN(hipMemcpyAsync(d_noff, noff.data(), (nref + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (!blob.empty()) BRUN(hipMemcpyAsync(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
    BRUN(hipMemcpyAsync(d_ct, &ct, sizeof ct, hipMemcpyHostToDevice, st));
    BRUN(hipMemsetAsync(d_err, 0, 256, st));
    BRUN(hipStreamSynchronize(st));
    RefTab rt;
    rt.hash = d_th; rt.id = d_tid; rt.name_off = d_noff; rt.names = d_blob; rt.mask = tcap - 1; rt.nref = nref;

    // ---- lines, keys, order
    uint64_t nl = 0;
    uint64_t* d_starts = nullptr;
    if (s->len) {
        BRUN(sort_line_index(s->d_text, s->len, st, &d_starts, &nl));
        owned.push_back(d_starts);
    }
    mark("header + line index");
    if (nl >= (1ull << 32) - 1) { cleanup(); return bfail(s, MKT_E_ARG, "%llu lines: records are indexed with 32 bits", (unsigned long long)nl); }
    SortRec *rA = nullptr, *rB = nullptr;
    uint32_t *d_size = nullptr, *d_hist = nullptr;
    uint64_t* d_off = nullptr;
    BamIdx* d_idx = nullptr;
    uint8_t* d_raw = nullptr;
    uint64_t total = 0;
    const unsigned lgrid = (unsigned)((nl + 255) / 256);
    if (nl) {
        BALLOC(rA, (nl + 1) * sizeof(SortRec));
        BALLOC(d_size, (nl + 1) * sizeof(uint32_t));
        BALLOC(d_off, (nl + 2) * sizeof(uint64_t));
        hipLaunchKernelGGL(k_bam_keys, dim3(lgrid), dim3(256), 0, st, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, nl, rt, rA, d_size, d_err);
        if (sorted) {
            BALLOC(rB, (nl + 1) * sizeof(SortRec));
            BALLOC(d_hist, kSortHistBytes);
            int tbits = 1;
            while ((1ull << tbits) <= (uint64_t)nref) ++tbits;
            sort_radix_passes(rA, rB, nl, d_hist, 1, 0, 33 + tbits, st);
        }
        uint32_t herr = 0;
        BRUN(hipMemcpyAsync(&herr, d_err, sizeof herr, hipMemcpyDeviceToHost, st));
        BRUN(hipStreamSynchronize(st));
        if (herr) {
            cleanup();
            return bfail(s, MKT_E_ARG, "not SAM alignment text (error bits 0x%x: 1 fewer than 11 fields, 2 reference name not in the header, 4 number, 8 CIGAR, 16 optional field, 32 SEQ / QUAL lengths, 64 QNAME length)", herr);
        }
        mark(sorted ? "keys + radix sort" : "keys");
        hipLaunchKernelGGL(k_bam_sizes, dim3(lgrid), dim3(256), 0, st, (const SortRec*)rA, (const uint32_t*)d_size, nl, d_off);
        BRUN(launch_exscan(d_off, nl, d_off + nl, st));
        BRUN(hipMemcpyAsync(&total, d_off + nl, sizeof total, hipMemcpyDeviceToHost, st));
        BRUN(hipStreamSynchronize(st));
    }
    const uint64_t nraw = hdr_len + total;
    BALLOC(d_raw, nraw + 64);
    BRUN(hipMemcpyAsync(d_raw, hdr.data(), hdr_len, hipMemcpyHostToDevice, st));
    if (nl) {
        BALLOC(d_idx, (nl + 1) * sizeof(BamIdx));
        hipLaunchKernelGGL(k_bam_write, dim3(lgrid), dim3(256), 0, st, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, nl, rt, (const SortRec*)rA, (const uint64_t*)d_off, d_raw + hdr_len, d_idx, d_err);
        BRUN(hipGetLastError());
        BRUN(hipStreamSynchronize(st));
        mark("records");
        // the text and the sort records are no longer needed
        drop(rA); if (rB) drop(rB); drop(d_size); drop(d_starts);
        rA = rB = nullptr;
    }
    (void)hipFree(s->d_text); s->d_text = nullptr; s->cap = s->len = 0;

    // ---- BGZF
    const uint64_t nblocks = (nraw + BGZF_RAW - 1) / BGZF_RAW;
    uint8_t* d_comp = nullptr;
    uint64_t* d_csize = nullptr;
    uint32_t* d_scratch = nullptr;
    BALLOC(d_comp, nblocks * (uint64_t)BGZF_STRIDE + 64);
    BALLOC(d_csize, (nblocks + 2) * sizeof(uint64_t));
    if (level > 0) {
        // the token lists and bit streams of the blocks of one launch: at most 8 GB of scratch
        const uint64_t per_block = (uint64_t)DZ_WAVES * DZ_WAVE_SCRATCH * sizeof(uint32_t);
        uint64_t batch = ((uint64_t)8 << 30) / per_block;
        if (batch > nblocks) batch = nblocks;
        if (batch < 1) batch = 1;
        BALLOC(d_scratch, batch * per_block + 64);
        for (uint64_t fb = 0; fb < nblocks; fb += batch) {
            const unsigned g = (unsigned)(nblocks - fb < batch ? nblocks - fb : batch);
            if (level >= 2) hipLaunchKernelGGL(k_bgzf_deflate<true>, dim3(g), dim3(DZ_THREADS), 0, st, (const uint8_t*)d_raw, nraw, fb, (const CrcTabs*)d_ct, d_comp, d_csize, d_scratch);
            else hipLaunchKernelGGL(k_bgzf_deflate<false>, dim3(g), dim3(DZ_THREADS), 0, st, (const uint8_t*)d_raw, nraw, fb, (const CrcTabs*)d_ct, d_comp, d_csize, d_scratch);
        }
    } else {
        hipLaunchKernelGGL(k_bgzf_stored, dim3((unsigned)nblocks), dim3(BWG), 0, st, (const uint8_t*)d_raw, nraw, (const CrcTabs*)d_ct, d_comp, d_csize);
    }
    BRUN(hipGetLastError());
    BRUN(launch_exscan(d_csize, nblocks, d_csize + nblocks, st));
    uint64_t clen = 0;
    BRUN(hipMemcpyAsync(&clen, d_csize + nblocks, sizeof clen, hipMemcpyDeviceToHost, st));
    BRUN(hipStreamSynchronize(st));
    mark(level > 0 ? "BGZF deflate" : "BGZF stored");
    drop(d_raw);
    if (d_scratch) drop(d_scratch);
    static const unsigned char eof_block[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    { hipError_t e_ = hipMalloc((void**)&s->d_bam, clen + 28 + 64); if (e_ != hipSuccess) { cleanup(); return bfail(s, MKT_E_NOMEM, "hipMalloc of the BAM failed: %s", hipGetErrorString(e_)); } }
    hipLaunchKernelGGL(k_bgzf_pack, dim3((unsigned)nblocks), dim3(BWG), 0, st, (const uint8_t*)d_comp, (const uint64_t*)d_csize, nblocks, s->d_bam);
    BRUN(hipMemcpyAsync(s->d_bam + clen, eof_block, 28, hipMemcpyHostToDevice, st));
    BRUN(hipGetLastError());
    BRUN(hipStreamSynchronize(st));
    s->bam_len = clen + 28;
    s->records = nl;
    drop(d_comp);
    mark("pack");

    // ---- BAI (coordinate order only; the format ends at 2^29 bases per reference -- longer ones would need a CSI index: none is made then)
    s->bai.clear();
    s->note.clear();
    bool bai_ok = true;
    for (uint32_t i = 0; i < nref; ++i) if (lens[i] > (1u << 29)) bai_ok = false;
    if (sorted && bai_ok) {
        std::vector<uint64_t> lin_off(nref + 1, 0);
        for (uint32_t i = 0; i < nref; ++i) lin_off[i + 1] = lin_off[i] + ((uint64_t)lens[i] >> 14) + 2;
        const uint64_t nlin = lin_off[nref];
        uint64_t* d_lin_off = nullptr;
        unsigned long long *d_lin = nullptr, *d_nocoor = nullptr;
        BaiRef* d_refs = nullptr;
        BaiHead* d_heads = nullptr;
        uint64_t* d_hflag = nullptr;
        std::vector<BaiRef> refs(nref);
        std::vector<unsigned long long> lin(nlin);
        std::vector<BaiHead> heads;
        unsigned long long no_coor = 0, beyond = 0;
        std::vector<BaiRef> init(nref);
        for (auto& r : init) { r.n_mapped = 0; r.n_unmapped = 0; r.beg = ~0ull; r.end = 0; }
        BALLOC(d_lin_off, (nref + 1) * sizeof(uint64_t));
        BALLOC(d_lin, (nlin + 1) * sizeof(unsigned long long));
        BALLOC(d_refs, (nref + 1) * sizeof(BaiRef));
        BALLOC(d_nocoor, 256);
        BALLOC(d_hflag, (nl + 2) * sizeof(uint64_t));
        BRUN(hipMemcpyAsync(d_lin_off, lin_off.data(), (nref + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        BRUN(hipMemsetAsync(d_lin, 0xFF, (nlin + 1) * sizeof(unsigned long long), st));
        if (nref) BRUN(hipMemcpyAsync(d_refs, init.data(), nref * sizeof(BaiRef), hipMemcpyHostToDevice, st));
        BRUN(hipMemsetAsync(d_nocoor, 0, 256, st));
        uint64_t nheads = 0;
        if (nl) {
            hipLaunchKernelGGL(k_bai, dim3(lgrid), dim3(256), 0, st, (const BamIdx*)d_idx, (const uint64_t*)d_off, nl, hdr_len, (const uint64_t*)d_csize, (const uint64_t*)d_lin_off, nref,
                               d_lin, d_refs, d_hflag, d_nocoor);
            BRUN(hipGetLastError());
            BRUN(launch_exscan(d_hflag, nl, d_hflag + nl, st));
            BRUN(hipMemcpyAsync(&nheads, d_hflag + nl, sizeof nheads, hipMemcpyDeviceToHost, st));
        }
        BRUN(hipMemcpyAsync(&no_coor, d_nocoor, sizeof no_coor, hipMemcpyDeviceToHost, st));
        BRUN(hipMemcpyAsync(&beyond, d_nocoor + 1, sizeof beyond, hipMemcpyDeviceToHost, st));
        if (nref) BRUN(hipMemcpyAsync(refs.data(), d_refs, nref * sizeof(BaiRef), hipMemcpyDeviceToHost, st));
        if (nlin) BRUN(hipMemcpyAsync(lin.data(), d_lin, nlin * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        BRUN(hipStreamSynchronize(st));
        heads.resize(nheads);
        if (nheads) {
            BALLOC(d_heads, ((size_t)nheads + 1) * sizeof(BaiHead));
            hipLaunchKernelGGL(k_bai_heads, dim3(lgrid), dim3(256), 0, st, (const BamIdx*)d_idx, (const uint64_t*)d_off, nl, hdr_len, (const uint64_t*)d_csize, (const uint64_t*)d_hflag, d_heads);
            BRUN(hipGetLastError());
            BRUN(hipMemcpyAsync(heads.data(), d_heads, (size_t)nheads * sizeof(BaiHead), hipMemcpyDeviceToHost, st));
            BRUN(hipStreamSynchronize(st));
        }
        // the end of a run of records in one bin = the start of the record after it
        const uint64_t n_coor = nl - no_coor;
        uint64_t off_end = 0;
        {
            // virtual offset of the first record without coordinates (or of the end of the data)
            uint64_t uo = 0;
            if (nl) BRUN(hipMemcpy(&uo, d_off + n_coor, sizeof uo, hipMemcpyDeviceToHost));
            uo += hdr_len;
            const uint64_t b = uo / BGZF_RAW;
            uint64_t cb = 0;
            BRUN(hipMemcpy(&cb, d_csize + b, sizeof cb, hipMemcpyDeviceToHost));
            off_end = (cb << 16) | (uo - b * BGZF_RAW);
        }
        // per reference (the run starts come in file order = by reference): a stable counting sort by bin
        std::vector<uint32_t> bin_cnt(65536, 0), bin_at(65536, 0);      // (every 16-bit value: a record past 2^29 bases carries a bin beyond 37449; no index is kept then, see below)
        std::vector<uint32_t> order(nheads);
        std::vector<std::pair<size_t, size_t>> ref_range(nref, {0, 0});
        {
            size_t i = 0;
            for (uint32_t t = 0; t < nref && i < nheads; ++t) {
                size_t j = i;
                while (j < nheads && (uint32_t)heads[j].tid == t) ++j;
                ref_range[t] = {i, j};
                i = j;
            }
        }
        std::string& o = s->bai;                                    // SAMv1 5.2
        o.reserve(64 + (size_t)nheads * 28 + nlin * 8 + (size_t)nref * 64);
        o.append("BAI\1", 4);
        put_le32(o, nref);
        for (uint32_t t = 0; t < nref; ++t) {
            const bool any = refs[t].n_mapped + refs[t].n_unmapped > 0;
            const size_t i0 = ref_range[t].first, i1 = ref_range[t].second;
            std::vector<uint32_t> used;                             // the bins of this reference, ascending
            for (size_t i = i0; i < i1; ++i) if (bin_cnt[heads[i].bin]++ == 0) used.push_back(heads[i].bin);
            std::sort(used.begin(), used.end());
            uint32_t at = 0;
            for (uint32_t b : used) { bin_at[b] = at; at += bin_cnt[b]; }
            for (size_t i = i0; i < i1; ++i) order[i0 + bin_at[heads[i].bin]++] = (uint32_t)i;
            put_le32(o, (uint32_t)used.size() + (any ? 1u : 0u));
            size_t k = i0;
            for (uint32_t b : used) {
                put_le32(o, b);
                put_le32(o, bin_cnt[b]);
                for (uint32_t c = 0; c < bin_cnt[b]; ++c, ++k) {
                    const size_t i = order[k];
                    put_le64(o, heads[i].voff);
                    put_le64(o, i + 1 < nheads ? heads[i + 1].voff : off_end);
                }
                bin_cnt[b] = 0;
            }
            if (any) {                                              // the pseudo-bin: file range of the reference, mapped / unmapped counts
                put_le32(o, 37450u); put_le32(o, 2u);
                put_le64(o, refs[t].beg); put_le64(o, refs[t].end);
                put_le64(o, refs[t].n_mapped); put_le64(o, refs[t].n_unmapped);
            }
            // linear index: up to the last window that a record touched; empty windows take the next one's offset
            const uint64_t a = lin_off[t], b = lin_off[t + 1];
            uint64_t last = a;
            for (uint64_t k = a; k < b; ++k) if (lin[k] != ~0ull) last = k + 1;
            unsigned long long nextv = 0;
            for (uint64_t k = last; k > a;) { --k; if (lin[k] == ~0ull) lin[k] = nextv; else nextv = lin[k]; }
            put_le32(o, (uint32_t)(last - a));
            for (uint64_t k = a; k < last; ++k) put_le64(o, lin[k]);
        }
        put_le64(o, no_coor);
        if (beyond) {
            s->bai.clear();
            s->note = "no index written: a record lies past its reference's LN or past the 2^29 bases a BAI index can address (samtools index would need -c)";
        }
        mark("BAI");
    } else if (sorted) {
        s->note = "no index written: a reference is longer than the 2^29 bases a BAI index can address (samtools index would need -c)";
    }
    cleanup();
#undef BALLOC
#undef BRUN
    if (records) *records = s->records;
    if (bam_bytes) *bam_bytes = s->bam_len;
    if (bai_bytes) *bai_bytes = s->bai.size();
    return MKT_OK;
}

int mkt_bam_fetch(mkt_bam* s, int which, uint64_t off, char* out, size_t n) {
    if (!s || (n && !out)) return MKT_E_ARG;
    if (!s->ran) return bfail(s, MKT_E_STATE, "fetch before run");
    if (which == 1) {
        if (off + n > s->bai.size()) return bfail(s, MKT_E_ARG, "range past the end of the index");
        memcpy(out, s->bai.data() + off, n);
        return MKT_OK;
    }
    if (off + n > s->bam_len) return bfail(s, MKT_E_ARG, "range past the end of the BAM");
    BCHK(s, hipSetDevice(s->device));
    if (n) BCHK(s, hipMemcpy(out, s->d_bam + off, n, hipMemcpyDeviceToHost));
    return MKT_OK;
}

// Result bytes [off, off + n) (n <= 64 MiB) through the pinned buffers: *ptr stays valid until the next call but one.
int mkt_bam_read(mkt_bam* s, int which, uint64_t off, size_t n, const char** ptr) {
    if (!s || !ptr) return MKT_E_ARG;
    if (!s->ran) return bfail(s, MKT_E_STATE, "read before run");
    if (which == 1) {
        if (off + n > s->bai.size()) return bfail(s, MKT_E_ARG, "range past the end of the index");
        *ptr = s->bai.data() + off;
        return MKT_OK;
    }
    if (n > kBamIoCap || off + n > s->bam_len) return bfail(s, MKT_E_ARG, "range past the end of the BAM, or longer than 64 MiB");
    BCHK(s, hipSetDevice(s->device));
    const int k = s->io_slot;
    int rc = bam_io_slot(s, k);
    if (rc) return rc;
    if (n) BCHK(s, hipMemcpyAsync(s->h_io[k], s->d_bam + off, n, hipMemcpyDeviceToHost, s->stream));
    BCHK(s, hipStreamSynchronize(s->stream));
    *ptr = s->h_io[k];
    s->io_slot = k ^ 1;
    return MKT_OK;
}

}  // extern "C"
