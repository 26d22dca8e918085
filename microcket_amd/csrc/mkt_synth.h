// mkt_synth.h -- seeded synthetic SAM generator, shared by host tools and HIP kernels.
//
// Stand-in for the reference's util/simulation + BWA inputs: sim3C, the hg38
// index and the aligner are not available offline, so name-grouped SAM text of the same shape is
// synthesised.  Every group (all alignment lines of one read name) is a pure function of
// (seed, group index), computed with integer arithmetic only, so the gcc build and the gfx950
// build produce identical bytes and any slice of a data set can be generated independently
// (per-GPU shards, device-resident benchmark inputs).
//
// Profiles
//   SYN_UNC    bwa mem -5SP paired-end shape (unstitched mode): 65 % 1+1, 28 % chimeric 1+2 / 2+1
//              (soft-clipped primary + hard-clipped supplementary), 5 % 2+2, 2 % singletons;
//              MAPQ 78 % 60 / 10 % 10..59 / 12 % 0..9; 60 % cis log-uniform, 20 % cis uniform,
//              20 % trans; 3 % indel CIGARs; 0.1 % self-circles.
//   SYN_FLASH  stitched single-end shape (flash mode): read length 160..290, 70 % one line,
//              25 % split (aMbS + aHbM), 5 % three lines.
//   SYN_STRESS adversarial mix for parity tests: N-spliced CIGARs, secondary / QC-fail / duplicate
//              flags, records without the 64/128 bits, clip sizes around 20/21, distances around
//              10 / 1000 / 10000, equal positions, chr10-vs-chr2 bytewise order, long groups,
//              filtered lines inside groups.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MKT_HD __host__ __device__ inline
#else
#define MKT_HD inline
#endif

enum { SYN_UNC = 0, SYN_FLASH = 1, SYN_STRESS = 2 };
enum { SYN_HG38 = 0, SYN_MM10 = 1 };

struct SynParams {
    uint64_t seed;
    int32_t profile;    // SYN_*
    int32_t genome;     // SYN_HG38 | SYN_MM10
    int32_t read_len;   // 150 (C1-C3), 100 (C4, C5)
    int32_t lanes;      // flow-cell lanes in the read names (C5: 4)
};

// ---------------------------------------------------------------- chromosome tables
// Names and lengths as in anno/hg38.info and anno/mm10.info (lexicographic order, as there).
struct SynChrom { const char* name; uint32_t len; };

MKT_HD int syn_nchrom(int genome) { return genome == SYN_MM10 ? 22 : 25; }

MKT_HD SynChrom syn_chrom(int genome, int i) {
    if (genome == SYN_MM10) {
        switch (i) {
        case 0: return {"chr1", 195471971u}; case 1: return {"chr10", 130694993u};
        case 2: return {"chr11", 122082543u}; case 3: return {"chr12", 120129022u};
        case 4: return {"chr13", 120421639u}; case 5: return {"chr14", 124902244u};
        case 6: return {"chr15", 104043685u}; case 7: return {"chr16", 98207768u};
        case 8: return {"chr17", 94987271u}; case 9: return {"chr18", 90702639u};
        case 10: return {"chr19", 61431566u}; case 11: return {"chr2", 182113224u};
        case 12: return {"chr3", 160039680u}; case 13: return {"chr4", 156508116u};
        case 14: return {"chr5", 151834684u}; case 15: return {"chr6", 149736546u};
        case 16: return {"chr7", 145441459u}; case 17: return {"chr8", 129401213u};
        case 18: return {"chr9", 124595110u}; case 19: return {"chrM", 16299u};
        case 20: return {"chrX", 171031299u}; default: return {"chrY", 91744698u};
        }
    }
    switch (i) {
    case 0: return {"chr1", 248956422u}; case 1: return {"chr10", 133797422u};
    case 2: return {"chr11", 135086622u}; case 3: return {"chr12", 133275309u};
    case 4: return {"chr13", 114364328u}; case 5: return {"chr14", 107043718u};
    case 6: return {"chr15", 101991189u}; case 7: return {"chr16", 90338345u};
    case 8: return {"chr17", 83257441u}; case 9: return {"chr18", 80373285u};
    case 10: return {"chr19", 58617616u}; case 11: return {"chr2", 242193529u};
    case 12: return {"chr20", 64444167u}; case 13: return {"chr21", 46709983u};
    case 14: return {"chr22", 50818468u}; case 15: return {"chr3", 198295559u};
    case 16: return {"chr4", 190214555u}; case 17: return {"chr5", 181538259u};
    case 18: return {"chr6", 170805979u}; case 19: return {"chr7", 159345973u};
    case 20: return {"chr8", 145138636u}; case 21: return {"chr9", 138394717u};
    case 22: return {"chrM", 16569u}; case 23: return {"chrX", 156040895u};
    default: return {"chrY", 57227415u};
    }
}

// ---------------------------------------------------------------- counter-based RNG
struct SynRng {
    uint64_t s;
    MKT_HD uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    MKT_HD uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    MKT_HD uint32_t range(uint32_t lo, uint32_t hi) { return lo + below(hi - lo + 1); }   // inclusive
    MKT_HD bool pct(uint32_t per_mille) { return below(1000) < per_mille; }
};
MKT_HD SynRng syn_rng(uint64_t seed, uint64_t gidx) {
    SynRng r;
    r.s = seed * 0xD1342543DE82EF95ull + gidx * 0xA0761D6478BD642Full + 0x2545F4914F6CDD1Dull;
    r.next();
    return r;
}

// ---------------------------------------------------------------- sinks
struct SynCountSink {
    size_t n = 0;
    MKT_HD void put(char) { ++n; }
};
struct SynMemSink {
    char* p;
    size_t n = 0;
    MKT_HD void put(char c) { p[n++] = c; }
};
template <class S> MKT_HD void syn_puts(S& s, const char* z) { while (*z) s.put(*z++); }
template <class S> MKT_HD void syn_putu(S& s, uint64_t v) {
    char t[24];
    int k = 0;
    do { t[k++] = (char)('0' + (int)(v % 10)); v /= 10; } while (v);
    while (k) s.put(t[--k]);
}
template <class S> MKT_HD void syn_puti(S& s, int64_t v) {
    if (v < 0) { s.put('-'); v = -v; }
    syn_putu(s, (uint64_t)v);
}

// ---------------------------------------------------------------- alignment description
struct SynAln {
    int chrom;          // index into the genome table, or -1 for a literal name (stress)
    const char* chrom_lit;
    uint32_t pos;
    uint32_t flag;
    uint32_t mapq;
    // CIGAR: up to 6 (len, op) items
    uint32_t clen[6];
    char cop[6];
    int ncig;
    int next_chrom;     // mate
    uint32_t next_pos;
    int has_sa;
};

MKT_HD void syn_cig(SynAln& a, uint32_t len, char op) {
    if (len == 0 || a.ncig >= 6) return;
    a.clen[a.ncig] = len;
    a.cop[a.ncig] = op;
    ++a.ncig;
}
MKT_HD uint32_t syn_seqlen(const SynAln& a) {
    uint32_t n = 0;
    for (int i = 0; i < a.ncig; ++i)
        if (a.cop[i] == 'M' || a.cop[i] == 'I' || a.cop[i] == 'S') n += a.clen[i];
    return n;
}
MKT_HD uint32_t syn_reflen(const SynAln& a) {
    uint32_t n = 0;
    for (int i = 0; i < a.ncig; ++i)
        if (a.cop[i] == 'M' || a.cop[i] == 'D' || a.cop[i] == 'N') n += a.clen[i];
    return n;
}

template <class S> MKT_HD void syn_qname(S& s, const SynParams& p, uint64_t gidx) {
    // A00123:45:HXXXXXXXX:<lane>:<tile>:<x>:<y>, unique per group index
    syn_puts(s, "A00123:45:HXXXXXXXX:");
    uint32_t lanes = p.lanes > 0 ? (uint32_t)p.lanes : 1u;
    syn_putu(s, 1 + (gidx * 2654435761ull >> 7) % lanes);
    s.put(':');
    syn_putu(s, 1101 + (gidx >> 24) % 1578);
    s.put(':');
    syn_putu(s, 1000 + ((gidx >> 12) & 0xFFF) * 7 + (gidx >> 36));
    s.put(':');
    syn_putu(s, 1000 + (gidx & 0xFFF) * 9);
}

template <class S> MKT_HD void syn_line(S& s, const SynParams& p, uint64_t gidx, const SynAln& a, SynRng& r) {
    syn_qname(s, p, gidx);
    s.put('\t'); syn_putu(s, a.flag);
    s.put('\t');
    if (a.chrom >= 0) syn_puts(s, syn_chrom(p.genome, a.chrom).name); else syn_puts(s, a.chrom_lit);
    s.put('\t'); syn_putu(s, a.pos);
    s.put('\t'); syn_putu(s, a.mapq);
    s.put('\t');
    for (int i = 0; i < a.ncig; ++i) { syn_putu(s, a.clen[i]); s.put(a.cop[i]); }
    s.put('\t');
    if (a.next_chrom < 0) s.put('*');
    else if (a.next_chrom == a.chrom) s.put('=');
    else syn_puts(s, syn_chrom(p.genome, a.next_chrom).name);
    s.put('\t'); syn_putu(s, a.next_pos);
    s.put('\t'); s.put('0');
    s.put('\t');
    uint32_t L = syn_seqlen(a);
    if (L == 0) s.put('*');
    uint64_t bits = 0;
    int have = 0;
    for (uint32_t i = 0; i < L; ++i) {
        if (have == 0) { bits = r.next(); have = 32; }
        s.put("ACGT"[bits & 3]);
        bits >>= 2;
        --have;
    }
    s.put('\t');
    if (L == 0) s.put('*');
    for (uint32_t i = 0; i < L; ++i) s.put('F');
    syn_puts(s, "\tNM:i:");
    syn_putu(s, r.below(4));
    syn_puts(s, "\tMD:Z:");
    syn_putu(s, syn_reflen(a));
    syn_puts(s, "\tAS:i:");
    syn_putu(s, syn_reflen(a));
    syn_puts(s, "\tXS:i:");
    syn_putu(s, r.below(40));
    if (a.has_sa) {
        syn_puts(s, "\tSA:Z:");
        if (a.next_chrom >= 0) syn_puts(s, syn_chrom(p.genome, a.next_chrom).name); else s.put('*');
        s.put(',');
        syn_putu(s, a.next_pos);
        syn_puts(s, ",+,");
        syn_putu(s, L / 2);
        syn_puts(s, "S");
        syn_putu(s, L - L / 2);
        syn_puts(s, "M,60,0;");
    }
    s.put('\n');
}

// ---------------------------------------------------------------- loci
struct SynLocus { int chrom; uint32_t pos; };

MKT_HD SynLocus syn_pick_locus(const SynParams& p, SynRng& r) {
    // chromosome with probability proportional to its length, position uniform inside
    int n = syn_nchrom(p.genome);
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) total += syn_chrom(p.genome, i).len;
    uint64_t x = r.next() % total;
    int c = 0;
    for (; c < n - 1; ++c) {
        uint32_t l = syn_chrom(p.genome, c).len;
        if (x < l) break;
        x -= l;
    }
    uint32_t len = syn_chrom(p.genome, c).len;
    uint32_t margin = len > 4000 ? 2000u : 1u;
    uint32_t pos = margin + (uint32_t)(x % (len - 2 * margin + 1));
    return {c, pos};
}
MKT_HD uint32_t syn_clamp_pos(const SynParams& p, int chrom, int64_t pos) {
    int64_t len = syn_chrom(p.genome, chrom).len;
    if (pos < 1) pos = 1;
    if (pos > len - 400) pos = len > 800 ? len - 400 : 1;
    return (uint32_t)pos;
}
// the other end of a contact anchored at `a`
MKT_HD SynLocus syn_contact(const SynParams& p, SynRng& r, SynLocus a) {
    uint32_t k = r.below(1000);
    if (k < 1) {                               // 0.1 % self-circle: ends within 10 bp
        return {a.chrom, syn_clamp_pos(p, a.chrom, (int64_t)a.pos + (int64_t)r.below(21) - 10)};
    }
    if (k < 600) {                             // cis, log-uniform distance 1 .. 2^27
        uint32_t e = r.below(27);
        uint32_t d = (1u << e) + r.below(1u << e);
        int64_t q = r.below(2) ? (int64_t)a.pos + d : (int64_t)a.pos - d;
        return {a.chrom, syn_clamp_pos(p, a.chrom, q)};
    }
    if (k < 800) {                             // cis, uniform
        uint32_t len = syn_chrom(p.genome, a.chrom).len;
        return {a.chrom, syn_clamp_pos(p, a.chrom, 1 + (int64_t)(r.next() % len))};
    }
    return syn_pick_locus(p, r);               // (mostly) trans
}
MKT_HD uint32_t syn_mapq(SynRng& r) {
    uint32_t k = r.below(100);
    if (k < 78) return 60;
    if (k < 88) return r.range(10, 59);
    return r.range(0, 9);
}
// full-length CIGAR with occasional indel / end clip
MKT_HD void syn_cigar_full(SynAln& a, SynRng& r, uint32_t L) {
    a.ncig = 0;
    uint32_t k = r.below(1000);
    if (k < 30 && L > 40) {                    // 3 %: one I or D
        uint32_t left = r.range(10, L - 20), ind = r.range(1, 4);
        if (r.below(2)) { syn_cig(a, left, 'M'); syn_cig(a, ind, 'I'); syn_cig(a, L - left - ind, 'M'); }
        else { syn_cig(a, left, 'M'); syn_cig(a, ind, 'D'); syn_cig(a, L - left, 'M'); }
    } else if (k < 130 && L > 60) {            // 10 %: soft-clipped end
        uint32_t c = r.range(1, L / 2 + 10 < L - 10 ? L / 2 + 10 : L - 10);
        if (r.below(2)) { syn_cig(a, c, 'S'); syn_cig(a, L - c, 'M'); }
        else { syn_cig(a, L - c, 'M'); syn_cig(a, c, 'S'); }
    } else {
        syn_cig(a, L, 'M');
    }
}
MKT_HD SynAln syn_aln(int chrom, uint32_t pos, uint32_t flag, uint32_t mapq) {
    SynAln a;
    a.chrom = chrom; a.chrom_lit = ""; a.pos = pos; a.flag = flag; a.mapq = mapq;
    a.ncig = 0; a.next_chrom = -1; a.next_pos = 0; a.has_sa = 0;
    for (int i = 0; i < 6; ++i) { a.clen[i] = 0; a.cop[i] = 'M'; }
    return a;
}

// ---------------------------------------------------------------- group generators
// paired-end, bwa mem -5SP like
template <class S> MKT_HD void syn_group_unc(S& s, const SynParams& p, uint64_t gidx, SynRng& r) {
    uint32_t L = (uint32_t)p.read_len;
    SynLocus A = syn_pick_locus(p, r);
    SynLocus B = syn_contact(p, r, A);
    uint32_t revA = r.below(2), revB = r.below(2);
    uint32_t kind = r.below(100);
    uint32_t f1 = 1u | 64u | (revA ? 16u : 0u) | (revB ? 32u : 0u);
    uint32_t f2 = 1u | 128u | (revB ? 16u : 0u) | (revA ? 32u : 0u);

    if (kind < 2) {                            // singleton: only one mate reported
        // (one RNG draw per full expression everywhere: argument evaluation order differs between compilers)
        const uint32_t which = r.below(2);
        const uint32_t mq = syn_mapq(r);
        SynAln a = syn_aln(A.chrom, A.pos, which ? (f1 | 8u) : (f2 | 8u), mq);
        syn_cigar_full(a, r, L);
        syn_line(s, p, gidx, a, r);
        return;
    }
    if (kind < 67) {                           // 1 + 1
        SynAln a = syn_aln(A.chrom, A.pos, f1, syn_mapq(r));
        SynAln b = syn_aln(B.chrom, B.pos, f2, syn_mapq(r));
        syn_cigar_full(a, r, L);
        syn_cigar_full(b, r, L);
        a.next_chrom = B.chrom; a.next_pos = B.pos;
        b.next_chrom = A.chrom; b.next_pos = A.pos;
        syn_line(s, p, gidx, a, r);
        syn_line(s, p, gidx, b, r);
        return;
    }
    // chimeric reads.  The split read covers the ligation junction: its 5' part maps at one end
    // of the contact, its 3' part (the supplementary, hard-clipped) next to the mate.
    uint32_t j = r.range(20, L - 20);
    bool split_r1 = r.below(2) != 0;
    bool four = kind >= 95;                    // 2 + 2
    // mate position: within 50..600 bp downstream/upstream of the supplementary, opposite strand (90 %)
    bool proper = r.below(10) != 0;
    SynLocus X = A, Y = B;                     // X: 5' part of the split read, Y: 3' part + mate
    uint32_t revS = r.below(2);                // strand of the supplementary
    uint32_t gap = r.range(50, 600);
    int64_t mpos = revS ? (int64_t)Y.pos - gap : (int64_t)Y.pos + gap;
    uint32_t revM = proper ? (revS ? 0u : 1u) : revS;
    SynLocus M = {Y.chrom, syn_clamp_pos(p, Y.chrom, mpos)};
    if (!proper && r.below(2)) M = syn_pick_locus(p, r);

    uint32_t fs = split_r1 ? 64u : 128u, fm = split_r1 ? 128u : 64u;
    SynAln pri = syn_aln(X.chrom, X.pos, 1u | fs | (revA ? 16u : 0u) | (revM ? 32u : 0u), syn_mapq(r));
    if (revA) { syn_cig(pri, L - j, 'S'); syn_cig(pri, j, 'M'); } else { syn_cig(pri, j, 'M'); syn_cig(pri, L - j, 'S'); }
    pri.next_chrom = M.chrom; pri.next_pos = M.pos; pri.has_sa = 1;
    SynAln sup = syn_aln(Y.chrom, Y.pos, 1u | fs | 2048u | (revS ? 16u : 0u) | (revM ? 32u : 0u), syn_mapq(r));
    if (revS) { syn_cig(sup, L - j, 'M'); syn_cig(sup, j, 'H'); } else { syn_cig(sup, j, 'H'); syn_cig(sup, L - j, 'M'); }
    sup.next_chrom = M.chrom; sup.next_pos = M.pos; sup.has_sa = 1;
    SynAln mate = syn_aln(M.chrom, M.pos, 1u | fm | (revM ? 16u : 0u) | (revA ? 32u : 0u), syn_mapq(r));
    mate.next_chrom = X.chrom; mate.next_pos = X.pos;
    SynAln mate2 = mate;
    if (four) {
        uint32_t j2 = r.range(20, L - 20);
        mate.ncig = 0; syn_cig(mate, j2, 'M'); syn_cig(mate, L - j2, 'S'); mate.has_sa = 1;
        SynLocus Z = syn_pick_locus(p, r);
        mate2 = syn_aln(Z.chrom, Z.pos, mate.flag | 2048u, syn_mapq(r));
        syn_cig(mate2, j2, 'H'); syn_cig(mate2, L - j2, 'M'); mate2.has_sa = 1;
        mate2.next_chrom = X.chrom; mate2.next_pos = X.pos;
    } else {
        syn_cigar_full(mate, r, L);
    }
    if (split_r1) {
        syn_line(s, p, gidx, pri, r);
        syn_line(s, p, gidx, sup, r);
        syn_line(s, p, gidx, mate, r);
        if (four) syn_line(s, p, gidx, mate2, r);
    } else {
        syn_line(s, p, gidx, mate, r);
        if (four) syn_line(s, p, gidx, mate2, r);
        syn_line(s, p, gidx, pri, r);
        syn_line(s, p, gidx, sup, r);
    }
}

// stitched single-end reads
template <class S> MKT_HD void syn_group_flash(S& s, const SynParams& p, uint64_t gidx, SynRng& r) {
    uint32_t L = r.range(160, 290);
    SynLocus A = syn_pick_locus(p, r);
    SynLocus B = syn_contact(p, r, A);
    uint32_t kind = r.below(100);
    uint32_t revA = r.below(2), revB = r.below(2);
    if (kind < 70) {
        SynAln a = syn_aln(A.chrom, A.pos, revA ? 16u : 0u, syn_mapq(r));
        syn_cigar_full(a, r, L);
        syn_line(s, p, gidx, a, r);
        return;
    }
    uint32_t j = r.range(20, L - 20);
    SynAln pri = syn_aln(A.chrom, A.pos, revA ? 16u : 0u, syn_mapq(r));
    if (revA) { syn_cig(pri, L - j, 'S'); syn_cig(pri, j, 'M'); } else { syn_cig(pri, j, 'M'); syn_cig(pri, L - j, 'S'); }
    pri.has_sa = 1; pri.next_chrom = -1;
    SynAln sup = syn_aln(B.chrom, B.pos, 2048u | (revB ? 16u : 0u), syn_mapq(r));
    if (revB) { syn_cig(sup, L - j, 'M'); syn_cig(sup, j, 'H'); } else { syn_cig(sup, j, 'H'); syn_cig(sup, L - j, 'M'); }
    sup.has_sa = 1;
    syn_line(s, p, gidx, pri, r);
    syn_line(s, p, gidx, sup, r);
    if (kind >= 95) {
        SynLocus Z = syn_pick_locus(p, r);
        SynAln third = syn_aln(Z.chrom, Z.pos, 2048u, syn_mapq(r));
        syn_cig(third, L / 2, 'H'); syn_cig(third, L - L / 2, 'M');
        syn_line(s, p, gidx, third, r);
    }
}

// adversarial mix (both modes read it)
MKT_HD const char* syn_stress_chr(uint32_t k) {
    switch (k % 7) {
    case 0: return "chr1"; case 1: return "chr10"; case 2: return "chr2"; case 3: return "chrX";
    case 4: return "chr1_KI270706v1_random"; case 5: return "chrUn_GL000195v1"; default: return "chr2";
    }
}
MKT_HD void syn_cigar_stress(SynAln& a, SynRng& r, uint32_t L) {
    a.ncig = 0;
    uint32_t k = r.below(16);
    // clip sizes straddle min_clip_size = 20 (pairutil.h:54); mapped ratios straddle 0.5 / 0.8
    uint32_t c1 = r.range(1, 4) == 1 ? r.range(19, 22) : r.range(1, L / 2 + 30 < L - 5 ? L / 2 + 30 : L - 5);
    switch (k) {
    case 0: case 1: case 2: syn_cig(a, L, 'M'); break;
    case 3: syn_cig(a, c1, 'S'); syn_cig(a, L - c1, 'M'); break;
    case 4: syn_cig(a, L - c1, 'M'); syn_cig(a, c1, 'S'); break;
    case 5: syn_cig(a, c1, 'H'); syn_cig(a, L - c1, 'M'); break;
    case 6: syn_cig(a, L - c1, 'M'); syn_cig(a, c1, 'H'); break;
    case 7: { uint32_t c2 = r.range(1, 30); if (c1 + c2 + 5 > L) c2 = 1; if (c1 + c2 + 5 > L) c1 = 1;
              syn_cig(a, c1, 'S'); syn_cig(a, L - c1 - c2, 'M'); syn_cig(a, c2, 'S'); break; }
    case 8: { uint32_t m1 = r.range(5, L - 5); syn_cig(a, m1, 'M'); syn_cig(a, r.range(1, 2000), 'N'); syn_cig(a, L - m1, 'M'); break; }
    case 9: { uint32_t m1 = r.range(5, L / 3), m2 = r.range(5, L / 3);
              syn_cig(a, m1, 'M'); syn_cig(a, r.range(1, 900), 'N'); syn_cig(a, m2, 'M');
              syn_cig(a, r.range(1, 900), 'N'); syn_cig(a, L - m1 - m2, 'M'); break; }
    case 10: { uint32_t m1 = r.range(5, L - 10); syn_cig(a, m1, 'M'); syn_cig(a, r.range(1, 9), 'I'); syn_cig(a, L - m1 - 1, 'M'); break; }
    case 11: { uint32_t m1 = r.range(5, L - 10); syn_cig(a, m1, 'M'); syn_cig(a, r.range(1, 30), 'D'); syn_cig(a, L - m1, 'M'); break; }
    case 12: { uint32_t m1 = r.range(25, L - 10); syn_cig(a, c1 < m1 ? c1 : 1, 'S'); syn_cig(a, m1 - (c1 < m1 ? c1 : 1), 'M');
               syn_cig(a, r.range(1, 600), 'N'); syn_cig(a, L - m1, 'M'); break; }
    case 13: { uint32_t m1 = r.range(5, L - 30); syn_cig(a, m1, 'M'); syn_cig(a, r.range(1, 600), 'N');
               syn_cig(a, L - m1 - 21, 'M'); syn_cig(a, 21, 'S'); break; }
    default: syn_cig(a, L, 'M'); break;
    }
}
// Directed scenarios: geometry placed ON the decision thresholds of the classifiers
// (pair distance 1000, self-circle 10, bins 1000 / 10000, strict < on the left ends).
MKT_HD SynAln syn_lit(const char* chr, uint32_t pos, uint32_t flag, uint32_t mapq) {
    SynAln a = syn_aln(-1, pos ? pos : 1u, flag, mapq);
    a.chrom_lit = chr;
    return a;
}
MKT_HD int32_t syn_eps(SynRng& r) { return (int32_t)r.below(5) - 2; }
template <class S> MKT_HD void syn_group_directed(S& s, const SynParams& p, uint64_t gidx, SynRng& r) {
    uint32_t L = (uint32_t)p.read_len;
    const char* c = syn_stress_chr(r.below(4));
    const char* cx = r.below(8) == 0 ? syn_stress_chr(r.below(4)) : c;   // sometimes break same-chr
    uint32_t P = r.range(5000, 60000);
    uint32_t sc = r.below(10);
    uint32_t a1 = r.range(30, L - 30), gap = r.range(50, 3000);
    bool flipstrand = r.below(8) == 0;                 // sometimes break the strand requirement
    uint32_t R1 = 1u | 64u, R2 = 1u | 128u;
    if (sc == 0) {             // 1+1 plain, distance of the 5' ends on a threshold
        uint32_t dsel = r.below(4);
        uint32_t d = dsel == 0 ? r.below(13) : dsel == 1 ? 998 + r.below(5) : dsel == 2 ? 9998 + r.below(5) : r.below(40);
        uint32_t rv1 = r.below(2), rv2 = r.below(2);
        // 5' end: left for '+', right (= pos + L - 1) for '-'
        uint32_t five1 = P + 20000, five2 = r.below(2) ? five1 + d : five1 - d;     // stays positive: POS >= 2^31 is out of contract
        SynAln x = syn_lit(c, rv1 ? five1 - (L - 1) : five1, R1 | (rv1 ? 16u : 0u), 60);
        SynAln y = syn_lit(cx, rv2 ? five2 - (L - 1) : five2, R2 | (rv2 ? 16u : 0u), 60);
        syn_cig(x, L, 'M'); syn_cig(y, L, 'M');
        if (r.below(2)) { syn_line(s, p, gidx, x, r); syn_line(s, p, gidx, y, r); }
        else { syn_line(s, p, gidx, y, r); syn_line(s, p, gidx, x, r); }
        return;
    }
    if (sc <= 4) {             // category 0 with one N-spliced read
        bool s1_spliced = sc <= 2, plus = (sc & 1) != 0;
        int32_t e = syn_eps(r);
        SynAln sp = syn_lit(c, 0, 0, 60), fl = syn_lit(cx, 0, 0, 60);
        syn_cig(sp, a1, 'M'); syn_cig(sp, gap, 'N'); syn_cig(sp, L - a1, 'M');
        syn_cig(fl, L, 'M');
        // spliced read: left0 = pos, right0 = pos+a1-1, left1 = pos+a1+gap
        uint32_t strand_sp, strand_fl;
        if (s1_spliced) {
            if (plus) {        // s1 '+': s1.left[1] < s2.left[0] && s2.right[0]-s1.left[1] <= 1000
                sp.pos = P; uint32_t left1 = P + a1 + gap;
                fl.pos = (uint32_t)((int32_t)left1 + 1000 + e - (int32_t)(L - 1));
                if (r.below(6) == 0) fl.pos = left1 + r.below(3) - 1;
                strand_sp = 0; strand_fl = 16;
            } else {           // s1 '-': s2.left[0] < s1.left[0] && s1.right[0]-s2.left[0] <= 1000
                sp.pos = P; fl.pos = (uint32_t)((int32_t)(P + a1 - 1) - 1000 - e);
                if (r.below(6) == 0) fl.pos = P + r.below(3) - 1;
                strand_sp = 16; strand_fl = 0;
            }
            if (flipstrand) strand_fl ^= 16;
            sp.flag = R1 | strand_sp; fl.flag = R2 | strand_fl;
        } else {
            if (plus) {        // s1 '+' flat: s1.left[0] < s2.left[0] && s2.right[0]-s1.left[0] <= 1000
                fl.pos = P; sp.pos = (uint32_t)((int32_t)P + 1000 + e - (int32_t)(a1 - 1));
                if (r.below(6) == 0) sp.pos = P + r.below(3) - 1;
                strand_fl = 0; strand_sp = 16;
            } else {           // s1 '-' flat: s2.left[1] < s1.left[0] && s1.right[0]-s2.left[1] <= 1000
                sp.pos = P; uint32_t left1 = P + a1 + gap;
                fl.pos = (uint32_t)((int32_t)left1 + 1000 + e - (int32_t)(L - 1));
                if (r.below(6) == 0) fl.pos = left1 + r.below(3) - 1;
                strand_fl = 16; strand_sp = 0;
            }
            if (flipstrand) strand_sp ^= 16;
            fl.flag = R1 | strand_fl; sp.flag = R2 | strand_sp;
        }
        if (r.below(2)) { syn_line(s, p, gidx, sp, r); syn_line(s, p, gidx, fl, r); }
        else { syn_line(s, p, gidx, fl, r); syn_line(s, p, gidx, sp, r); }
        return;
    }
    // categories 1 (1+2) and 2 (2+1): single read `u`, split read = records v0, v1
    bool cat1 = sc <= 7;
    uint32_t fu = cat1 ? R1 : R2, fv = cat1 ? R2 : R1;
    uint32_t which = r.below(2);                        // which split record pairs with u
    bool uplus = r.below(2) != 0;
    int32_t e = syn_eps(r);
    uint32_t j = r.range(22, L - 22);
    SynAln u = syn_lit(c, P + 3000, fu | (uplus ? 0u : 16u), 60);
    syn_cig(u, L, 'M');
    SynAln v[2];
    for (int k = 0; k < 2; ++k) {
        v[k] = syn_lit(c, 0, fv | (k ? 2048u : 0u), 60);
        // complementary clips: record 0 keeps the first j bases, record 1 the rest
        bool rev = r.below(2) != 0;
        uint32_t m = k == 0 ? j : L - j, cl = L - m;
        bool clip_left = (k == 1) != rev;
        char cop = k ? 'H' : 'S';
        if (clip_left) { syn_cig(v[k], cl, cop); syn_cig(v[k], m, 'M'); } else { syn_cig(v[k], m, 'M'); syn_cig(v[k], cl, cop); }
        v[k].flag |= rev ? 16u : 0u;
    }
    // place the pairing record on the threshold, the other one elsewhere (or also pairable)
    {
        SynAln& w = v[which];
        uint32_t m = which == 0 ? j : L - j;
        w.flag &= ~16u;
        if (uplus) {           // u.left0 < w.left0 && w.right0 - u.left0 <= 1000 ; w must be '-'
            w.flag |= 16u;
            w.pos = (uint32_t)((int32_t)u.pos + 1000 + e - (int32_t)(m - 1));
            if (r.below(6) == 0) w.pos = u.pos + r.below(3) - 1;
        } else {               // w.left0 < u.left0 && u.right0 - w.left0 <= 1000 ; w must be '+'
            w.pos = (uint32_t)((int32_t)(u.pos + L - 1) - 1000 - e);
            if (r.below(6) == 0) w.pos = u.pos + r.below(3) - 1;
        }
        if (flipstrand) w.flag ^= 16u;
        w.chrom_lit = cx;
        SynAln& o = v[1 - which];
        uint32_t osel = r.below(4);
        o.chrom_lit = osel == 0 ? syn_stress_chr(r.below(4)) : c;
        o.pos = osel == 1 ? w.pos + r.below(30) : osel == 2 ? u.pos + r.below(12) : r.range(1000, 200000);
    }
    uint32_t order = r.below(3);
    if (order == 0) { syn_line(s, p, gidx, u, r); syn_line(s, p, gidx, v[0], r); syn_line(s, p, gidx, v[1], r); }
    else if (order == 1) { syn_line(s, p, gidx, v[0], r); syn_line(s, p, gidx, v[1], r); syn_line(s, p, gidx, u, r); }
    else { syn_line(s, p, gidx, v[0], r); syn_line(s, p, gidx, u, r); syn_line(s, p, gidx, v[1], r); }
}
template <class S> MKT_HD void syn_group_stress(S& s, const SynParams& p, uint64_t gidx, SynRng& r) {
    if (r.below(100) < 45) { syn_group_directed(s, p, gidx, r); return; }
    uint32_t L = (uint32_t)p.read_len;
    uint32_t nlines = r.below(100) < 80 ? r.range(1, 3) : r.range(1, 6);
    const char* c0 = syn_stress_chr(r.below(7));
    uint32_t base = r.range(1, 30000);
    for (uint32_t i = 0; i < nlines; ++i) {
        SynAln a = syn_aln(-1, 0, 0, 0);
        a.chrom_lit = r.below(100) < 75 ? c0 : syn_stress_chr(r.below(7));
        uint32_t dsel = r.below(10);
        uint32_t d = dsel < 2 ? r.below(14) : dsel < 4 ? r.range(985, 1012) : dsel < 6 ? r.range(9990, 10010)
                   : dsel < 8 ? r.range(1, 700) : r.range(1, 200000);
        a.pos = r.below(2) ? base + d : (base > d ? base - d : base + d);
        if (a.pos == 0) a.pos = 1;
        uint32_t fsel = r.below(100);
        uint32_t rd = fsel < 45 ? 64u : fsel < 90 ? 128u : fsel < 94 ? 0u : 192u;
        const uint32_t b16 = r.below(2);
        const uint32_t b32 = r.below(2);
        const uint32_t b2048 = r.below(8);
        a.flag = 1u | rd | (b16 ? 16u : 0u) | (b32 ? 32u : 0u) | (b2048 == 0 ? 2048u : 0u);
        uint32_t bad = r.below(100);
        if (bad < 3) a.flag |= 256u; else if (bad < 5) a.flag |= 512u; else if (bad < 7) a.flag |= 1024u;
        uint32_t mq = r.below(100);
        a.mapq = mq < 70 ? 60 : mq < 80 ? r.range(28, 32) : mq < 90 ? r.range(8, 12) : r.range(0, 3);
        if (r.below(50) == 0) a.mapq = 255;
        syn_cigar_stress(a, r, L);
        a.next_chrom = -1;
        syn_line(s, p, gidx, a, r);
    }
}

template <class S> MKT_HD void synth_group(S& s, const SynParams& p, uint64_t gidx) {
    SynRng r = syn_rng(p.seed, gidx);
    if (p.profile == SYN_FLASH) syn_group_flash(s, p, gidx, r);
    else if (p.profile == SYN_STRESS) syn_group_stress(s, p, gidx, r);
    else syn_group_unc(s, p, gidx, r);
}

// the sacrificial last group (the reference never classifies the input's last group: quirk Q1)
template <class S> MKT_HD void synth_tail_group(S& s, const SynParams& p) {
    SynRng r = syn_rng(p.seed, ~0ull);
    for (int k = 0; k < 2; ++k) {
        syn_puts(s, "ZZ:LAST:GROUP\t");
        syn_putu(s, k == 0 ? 65u : 129u);
        syn_puts(s, "\tchr1\t");
        syn_putu(s, 10000u + 500u * (uint32_t)k);
        syn_puts(s, "\t60\t");
        syn_putu(s, (uint64_t)p.read_len);
        syn_puts(s, "M\t=\t10000\t0\t*\t*\n");
    }
    (void)r;
}
