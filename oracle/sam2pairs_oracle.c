/*
 * oracle/sam2pairs_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the reference's sam2pairs
 * algorithm (SAM text -> 4DN .pairs lines + pass-through .sam + 8 counters).
 * It exists to CHECK the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may import, link or execute anything under
 * oracle/.  The product (microcket_amd/, include/) never calls into it and has
 * no CPU fallback.
 *
 * Parity status: PINNED.  This restatement is checked (tests/test_oracle.py)
 *   (1) against outputs of the reference itself, compiled unmodified from
 *       /root/reference/src/sam2pairs/ into oracle/_ref/ by oracle/Makefile,
 *       on every fixture family, for threads 2/4/8 x sam yes/no x ratio x mapQ;
 *   (2) against the committed golden vectors under tests/golden/ (inputs plus
 *       the reference's outputs, produced by tests/golden/make_golden.py).
 * The reference holds no golden vectors of its own.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).  Output order here is input order; the reference's order
 * is thread-schedule dependent (sam2pairs.cpp:154,175), so comparisons are on
 * LANG=C-sorted lines (the driver sorts anyway: microcket:480).
 *
 * Reproduced quirks:
 *   Q1  the last surviving QNAME group of the input is never classified
 *       (pairutil.h:151-176 returns the last index; sam2pairs.cpp:150-151).
 *   Q2  the logged selfCircle is thread 0's share only
 *       (sam2pairs.cpp:202-210 omits it from the reduction).
 *   Q3  check_integrity_2_seg tests s1.rightClip where s2 is meant
 *       (pairutil.h:200).
 * NOT reproduced: Q4, the data race on `loaded` (sam2pairs.cpp:170 vs :172).
 *
 * Where the reference is undefined (reads of left[1]/right[1] that were never
 * pushed, right[-1], malformed numeric fields, records with < 6 tokens) this
 * oracle defines: missing segment coordinates read as 0, malformed records are
 * dropped like a filtered line.  Such inputs are outside the parity contract.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sam2pairs_oracle.h"

#define ORC_BATCH (1u << 18)          /* pairutil.h:48  */
#define ORC_MIN_CLIP 20               /* pairutil.h:54  */
#define ORC_MAX_SELF_CIRCLE 10        /* pairutil.h:57  */
#define ORC_MAX_PAIR_DIST 1000        /* pairutil.h:58  */
#define ORC_SEG_KEEP 4

/* ------------------------------------------------------------------ buffers */
static int buf_put(orc_buf *b, const char *s, size_t n) {
    if (!b) return 0;
    if (b->n + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 1 << 16;
        while (nc < b->n + n) nc *= 2;
        char *np = (char *)realloc(b->p, nc);
        if (!np) return -1;
        b->p = np;
        b->cap = nc;
    }
    memcpy(b->p + b->n, s, n);
    b->n += n;
    return 0;
}
static int buf_putc(orc_buf *b, char c) { return buf_put(b, &c, 1); }
static int buf_putu(orc_buf *b, uint32_t v) {
    char t[16];
    int n = snprintf(t, sizeof t, "%u", v);
    return buf_put(b, t, (size_t)n);
}
void orc_buf_free(orc_buf *b) {
    if (b && b->p) free(b->p);
    if (b) { b->p = NULL; b->n = b->cap = 0; }
}

/* --------------------------------------------------------------- tokenising */
/* istream >> token: skip leading whitespace, read to the next whitespace
 * (the classic locale's isspace set), as every `ss >> x` in the reference. */
static int is_ws(char c) {
    return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}
typedef struct { const char *p; size_t n; } tok;

/* Split the first `want` tokens of line [s, s+n).  Returns tokens found. */
static int split_tokens(const char *s, size_t n, tok *t, int want) {
    size_t i = 0;
    int k = 0;
    while (k < want) {
        while (i < n && is_ws(s[i])) ++i;
        if (i >= n) break;
        size_t b = i;
        while (i < n && !is_ws(s[i])) ++i;
        t[k].p = s + b;
        t[k].n = i - b;
        ++k;
    }
    return k;
}
/* `ss >> unsigned`: decimal digits.  Returns 0 on a token that is not all digits. */
static int tok_uint(const tok *t, uint32_t *out) {
    uint64_t v = 0;
    if (t->n == 0) return 0;
    for (size_t i = 0; i < t->n; ++i) {
        char c = t->p[i];
        if (c < '0' || c > '9') return 0;
        v = v * 10 + (uint64_t)(c - '0');
        if (v > 0xffffffffull) return 0;
    }
    *out = (uint32_t)v;
    return 1;
}
/* std::string::compare: bytewise, then by length (flash2pairs.h:110, unc2pairs.h:315). */
static int tok_cmp(const tok *a, const tok *b) {
    size_t m = a->n < b->n ? a->n : b->n;
    int c = m ? memcmp(a->p, b->p, m) : 0;
    if (c) return c;
    if (a->n < b->n) return -1;
    if (a->n > b->n) return 1;
    return 0;
}
static int tok_eq(const tok *a, const tok *b) {
    return a->n == b->n && (a->n == 0 || memcmp(a->p, b->p, a->n) == 0);
}

/* ------------------------------------------------------------------ records */
typedef struct {
    const char *s;      /* line start (no newline) */
    size_t n;           /* line length */
    tok qname, rname, cigar;
    uint32_t flag, pos, mapq;
} rec;

/* The five-token parse of load_batch (pairutil.h:155) and of the first-record
 * scan (sam2pairs.cpp:122), plus the sixth token the classifiers read
 * (flash2pairs.h:29, unc2pairs.h:36).  Returns 1 when the record survives the
 * per-line filter (pairutil.h:157-161 / sam2pairs.cpp:124). */
static int parse_record(const char *s, size_t n, int min_mapq, rec *r) {
    tok t[6];
    int k = split_tokens(s, n, t, 6);
    if (k < 6) return 0;                       /* malformed: defined as dropped */
    r->s = s; r->n = n;
    r->qname = t[0]; r->rname = t[2]; r->cigar = t[5];
    if (!tok_uint(&t[1], &r->flag)) return 0;
    if (!tok_uint(&t[3], &r->pos)) return 0;
    if (!tok_uint(&t[4], &r->mapq)) return 0;
    /* `mapQ < min_mapQ` compares unsigned with int -> both unsigned (pairutil.h:157) */
    if (r->mapq < (uint32_t)min_mapq) return 0;
    if (r->flag & 0x700u) return 0;            /* pairutil.h:160 */
    return 1;
}

/* ----------------------------------------------------------------- segments */
typedef struct {
    int segCnt, leftClip, rightClip, mappable;
    int left[ORC_SEG_KEEP], right[ORC_SEG_KEEP];   /* coordinates of the first segments */
} seg;

/* cigar2segment, pairutil.h:63-126.  On an error return the reference leaves
 * segCnt = 0 and the partial clip/mappable state; so does this. */
static void cigar_walk(const tok *cigar, uint32_t start, seg *sr) {
    int index = 0, value = 0, currPos = (int)start;
    int last_right = 0;                          /* right[index] of the running segment */
    memset(sr, 0, sizeof *sr);
    sr->left[0] = (int)start;
    for (size_t j = 0; j < cigar->n; ++j) {
        char c = cigar->p[j];
        if (c >= '0' && c <= '9') { value = value * 10 + (c - '0'); continue; }
        if (c == 'H' || c == 'S') {              /* :87-95 */
            if (j == cigar->n - 1) sr->rightClip = value;
            else if (index == 0) sr->leftClip = value;
            else return;                         /* "ERROR CLIP", segCnt stays 0 */
        } else if (c == 'M' || c == 'D') {       /* :96-102 */
            if (c == 'M') sr->mappable += value;
            currPos += value;
            last_right = currPos - 1;
            if (index < ORC_SEG_KEEP) sr->right[index] = last_right;
        } else if (c == 'I') {                   /* :103-104 */
        } else if (c == 'N') {                   /* :105-109 */
            currPos += value;
            ++index;
            last_right = 0;
            if (index < ORC_SEG_KEEP) { sr->left[index] = currPos; sr->right[index] = 0; }
        } else {
            return;                              /* unknown element, :110-113 */
        }
        value = 0;
    }
    if (last_right == 0) return;                 /* :119-122 */
    sr->segCnt = index + 1;                      /* :123 */
}

/* check_integrity_1_seg, pairutil.h:180-188.  int >= int*float is evaluated in
 * float32 (cvtsi2ss / mulss / comiss in the reference binary). */
static int integrity_1(const seg *s, float ratio) {
    int total = s->mappable;
    if (s->leftClip > ORC_MIN_CLIP) total += s->leftClip;
    if (s->rightClip > ORC_MIN_CLIP) total += s->rightClip;
    volatile float rhs = (float)total * ratio;
    return (float)s->mappable >= rhs;
}
/* check_integrity_2_seg, pairutil.h:190-208, including Q3 at :200. */
static int integrity_2(const seg *a, const seg *b, float ratio) {
    int t1 = a->mappable, t2 = b->mappable;
    if (a->leftClip > ORC_MIN_CLIP) t1 += a->leftClip;
    if (a->rightClip > ORC_MIN_CLIP) t1 += a->rightClip;
    if (b->leftClip > ORC_MIN_CLIP) t2 += b->leftClip;
    if (a->rightClip > ORC_MIN_CLIP) t2 += b->rightClip;   /* Q3: tests a, adds b */
    int tot = t1 > t2 ? t1 : t2;
    volatile float rhs = (float)tot * ratio;
    return (float)(a->mappable + b->mappable) >= rhs;
}

/* ------------------------------------------------------------ group results */
enum { C_NONE = 0, C_LOWMAP, C_MANYHITS, C_UNPAIRED, C_SELFCIRCLE, C_TRANS, C_CIS10K, C_CIS1K, C_CIS0 };

typedef struct {
    int counter;            /* which of the 8 counters this group bumps, or C_NONE */
    int emit;               /* 1: a pairs line (and the .sam lines) is produced */
    tok chrA, chrB;
    uint32_t posA, posB;
    char sA, sB;
} verdict;

/* The shared tail: order the two ends, detect self-circles, bin the distance.
 * flash2pairs.h:110-144, unc2pairs.h:315-348. */
static void order_and_bin(const tok *chr1, uint32_t pos1, char st1,
                          const tok *chr2, uint32_t pos2, char st2, verdict *v) {
    int chrcmp = tok_cmp(chr1, chr2);
    if (chrcmp < 0 || (chrcmp == 0 && pos1 < pos2)) {
        v->chrA = *chr1; v->posA = pos1; v->sA = st1;
        v->chrB = *chr2; v->posB = pos2; v->sB = st2;
    } else {
        v->chrA = *chr2; v->posA = pos2; v->sA = st2;
        v->chrB = *chr1; v->posB = pos1; v->sB = st1;
    }
    if (chrcmp == 0) {
        uint32_t dist = v->posB - v->posA;       /* unsigned wrap as in the reference */
        if (dist <= ORC_MAX_SELF_CIRCLE) { v->counter = C_SELFCIRCLE; v->emit = 0; return; }
        v->counter = dist >= 10000 ? C_CIS10K : (dist >= 1000 ? C_CIS1K : C_CIS0);
    } else {
        v->counter = C_TRANS;
    }
    v->emit = 1;
}

/* flash2pairs, flash2pairs.h:17-155. */
static void classify_flash(const rec *g, size_t n, float ratio, verdict *v) {
    seg s1, s2;
    memset(v, 0, sizeof *v);
    if (n == 1) {                                         /* :26-68 */
        cigar_walk(&g[0].cigar, g[0].pos, &s1);
        if (s1.segCnt > 2) { v->counter = C_MANYHITS; return; }
        if (!integrity_1(&s1, ratio)) { v->counter = C_LOWMAP; return; }
        uint32_t pos1 = g[0].pos;
        int k = s1.segCnt - 1;                            /* :48; k = -1 is UB there, 0 here */
        uint32_t pos2 = (k >= 0 && k < ORC_SEG_KEEP) ? (uint32_t)s1.right[k] : 0u;
        uint32_t dist = pos2 - pos1;
        v->counter = dist >= 10000 ? C_CIS10K : (dist >= 1000 ? C_CIS1K : C_CIS0);   /* :54 */
        v->emit = 1;
        v->chrA = g[0].rname; v->posA = pos1; v->sA = '+';                          /* :60-63 */
        v->chrB = g[0].rname; v->posB = pos2; v->sB = '-';
        return;
    }
    if (n == 2) {                                         /* :69-149 */
        cigar_walk(&g[0].cigar, g[0].pos, &s1);
        cigar_walk(&g[1].cigar, g[1].pos, &s2);
        if (s1.segCnt != 1 || s2.segCnt != 1) { v->counter = C_MANYHITS; return; }
        if (!integrity_2(&s1, &s2, ratio)) { v->counter = C_LOWMAP; return; }
        uint32_t pos1 = g[0].pos, pos2 = g[1].pos;
        if (s1.leftClip > s1.rightClip) pos1 = (uint32_t)s1.right[0];
        if (s2.leftClip > s2.rightClip) pos2 = (uint32_t)s2.right[0];
        order_and_bin(&g[0].rname, pos1, (g[0].flag & 16) ? '-' : '+',
                      &g[1].rname, pos2, (g[1].flag & 16) ? '-' : '+', v);
        return;
    }
    v->counter = C_MANYHITS;                              /* :150-153 */
}

/* one "can s pair with t" probe of unc2pairs.h:146-189,196-227,255-285:
 * `lo` is the upstream (+ strand) end, `hi` the downstream (- strand) end. */
static int pairable(const tok *chrL, const tok *chrH, int lo_left, int hi_left, int hi_right) {
    return tok_cmp(chrL, chrH) == 0 && lo_left < hi_left && hi_right - lo_left <= ORC_MAX_PAIR_DIST;
}

/* unc2pairs, unc2pairs.h:16-358. */
static void classify_unc(const rec *g, size_t n, float ratio, verdict *v) {
    const rec *R1[2] = {0, 0}, *R2[2] = {0, 0};
    size_t n1 = 0, n2 = 0;
    seg s1, s2, s3;
    memset(v, 0, sizeof *v);
    for (size_t i = 0; i < n; ++i) {                      /* :33-49 */
        if (g[i].flag & 64) { if (n1 < 2) R1[n1] = &g[i]; ++n1; }
        else if (g[i].flag & 128) { if (n2 < 2) R2[n2] = &g[i]; ++n2; }
    }
    if (n1 == 0 || n2 == 0) return;                       /* :52-55, no counter */
    if (n1 + n2 > 3) return;                              /* :56-59, no counter */

    const tok *chr1, *chr2;
    char st1, st2;
    uint32_t pos1 = 0, pos2 = 0;
#define STRAND(r) (((r)->flag & 16) ? '-' : '+')

    if (n1 == 1 && n2 == 1) {                             /* category 0, :61-83,125-190 */
        cigar_walk(&R1[0]->cigar, R1[0]->pos, &s1);
        if (!integrity_1(&s1, ratio)) { v->counter = C_LOWMAP; return; }
        cigar_walk(&R2[0]->cigar, R2[0]->pos, &s2);
        if (!integrity_1(&s2, ratio)) { v->counter = C_LOWMAP; return; }
        if (s1.segCnt + s2.segCnt > 3) { v->counter = C_MANYHITS; return; }
        st1 = STRAND(R1[0]); st2 = STRAND(R2[0]);
        chr1 = &R1[0]->rname; chr2 = &R2[0]->rname;
        if (s1.segCnt == 1 && s2.segCnt == 1) {           /* :133-145 */
            pos1 = (uint32_t)(st1 == '+' ? s1.left[0] : s1.right[0]);
            pos2 = (uint32_t)(st2 == '+' ? s2.left[0] : s2.right[0]);
        } else if (s1.segCnt == 2) {                      /* :146-167 */
            if (st1 == '+') {
                if (st2 == '-' && pairable(chr1, chr2, s1.left[1], s2.left[0], s2.right[0])) {
                    pos1 = (uint32_t)s1.left[0]; pos2 = (uint32_t)s2.right[0];
                } else { v->counter = C_UNPAIRED; return; }
            } else {
                if (st2 == '+' && pairable(chr1, chr2, s2.left[0], s1.left[0], s1.right[0])) {
                    pos1 = (uint32_t)s1.right[1]; pos2 = (uint32_t)s2.left[0];
                } else { v->counter = C_UNPAIRED; return; }
            }
        } else {                                          /* :168-189 */
            if (st1 == '+') {
                if (st2 == '-' && pairable(chr1, chr2, s1.left[0], s2.left[0], s2.right[0])) {
                    pos1 = (uint32_t)s1.left[0]; pos2 = (uint32_t)s2.right[1];
                } else { v->counter = C_UNPAIRED; return; }
            } else {
                if (st2 == '+' && pairable(chr1, chr2, s2.left[1], s1.left[0], s1.right[0])) {
                    pos1 = (uint32_t)s1.right[0]; pos2 = (uint32_t)s2.left[0];
                } else { v->counter = C_UNPAIRED; return; }
            }
        }
    } else if (n1 == 1) {                                 /* category 1 (1+2), :84-98,191-249 */
        cigar_walk(&R1[0]->cigar, R1[0]->pos, &s1);
        if (!integrity_1(&s1, ratio)) { v->counter = C_LOWMAP; return; }
        cigar_walk(&R2[0]->cigar, R2[0]->pos, &s2);
        cigar_walk(&R2[1]->cigar, R2[1]->pos, &s3);
        if (!integrity_2(&s2, &s3, ratio)) { v->counter = C_LOWMAP; return; }
        if (s1.segCnt != 1 || s2.segCnt != 1 || s3.segCnt != 1) { v->counter = C_MANYHITS; return; }
        st1 = STRAND(R1[0]); chr1 = &R1[0]->rname;
        int mate = 0;
        const rec *cand[2] = {R2[0], R2[1]};
        const seg *cs[2] = {&s2, &s3};
        for (int k = 0; k < 2 && !mate; ++k) {            /* try s2 then s3, :196-227 */
            if (st1 == '+') {
                if (STRAND(cand[k]) == '-' && pairable(chr1, &cand[k]->rname, s1.left[0], cs[k]->left[0], cs[k]->right[0])) {
                    pos1 = (uint32_t)s1.left[0]; mate = 2 + k;
                }
            } else {
                if (STRAND(cand[k]) == '+' && pairable(chr1, &cand[k]->rname, cs[k]->left[0], s1.left[0], s1.right[0])) {
                    pos1 = (uint32_t)s1.right[0]; mate = 2 + k;
                }
            }
        }
        if (!mate) { v->counter = C_UNPAIRED; return; }   /* :229-232 */
        int o = (mate == 2) ? 1 : 0;                      /* the OTHER R2 record, :233-249 */
        chr2 = &cand[o]->rname; st2 = STRAND(cand[o]);
        pos2 = (uint32_t)(cs[o]->leftClip > cs[o]->rightClip ? cs[o]->right[0] : cs[o]->left[0]);
    } else {                                              /* category 2 (2+1), :99-121,250-308 */
        cigar_walk(&R1[0]->cigar, R1[0]->pos, &s1);
        cigar_walk(&R1[1]->cigar, R1[1]->pos, &s2);
        if (!integrity_2(&s1, &s2, ratio)) { v->counter = C_LOWMAP; return; }
        cigar_walk(&R2[0]->cigar, R2[0]->pos, &s3);
        if (!integrity_1(&s3, ratio)) { v->counter = C_LOWMAP; return; }
        if (s1.segCnt != 1 || s2.segCnt != 1 || s3.segCnt != 1) { v->counter = C_MANYHITS; return; }
        st2 = STRAND(R2[0]); chr2 = &R2[0]->rname;
        int mate = 0;
        const rec *cand[2] = {R1[0], R1[1]};
        const seg *cs[2] = {&s1, &s2};
        for (int k = 0; k < 2 && !mate; ++k) {            /* try s1 then s2, :255-285 */
            if (st2 == '+') {
                if (STRAND(cand[k]) == '-' && pairable(chr2, &cand[k]->rname, s3.left[0], cs[k]->left[0], cs[k]->right[0])) {
                    pos2 = (uint32_t)s3.left[0]; mate = 1 + k;
                }
            } else {
                if (STRAND(cand[k]) == '+' && pairable(chr2, &cand[k]->rname, cs[k]->left[0], s3.left[0], s3.right[0])) {
                    pos2 = (uint32_t)s3.right[0]; mate = 1 + k;
                }
            }
        }
        if (!mate) { v->counter = C_UNPAIRED; return; }   /* :287-290 */
        int o = (mate == 1) ? 1 : 0;                      /* the OTHER R1 record, :291-307 */
        chr1 = &cand[o]->rname; st1 = STRAND(cand[o]);
        pos1 = (uint32_t)(cs[o]->leftClip > cs[o]->rightClip ? cs[o]->right[0] : cs[o]->left[0]);
    }
#undef STRAND
    order_and_bin(chr1, pos1, st1, chr2, pos2, st2, v);
}

/* --------------------------------------------------------------- top level */
static void bump(orc_stats *st, int counter, int sc_logged) {
    switch (counter) {
    case C_LOWMAP: ++st->lowMap; break;
    case C_MANYHITS: ++st->manyHits; break;
    case C_UNPAIRED: ++st->unpaired; break;
    case C_SELFCIRCLE: ++st->selfCircle_all; if (sc_logged) ++st->selfCircle; break;
    case C_TRANS: ++st->trans; break;
    case C_CIS10K: ++st->cis10K; break;
    case C_CIS1K: ++st->cis1K; break;
    case C_CIS0: ++st->cis0; break;
    default: break;
    }
}

/* Q2: is in-batch index i of batch j inside thread 0's slice?
 * sam2pairs.cpp:150-151 (final batch: all T threads) / :172-173 (T-1 workers). */
int orc_selfcircle_logged(uint64_t g, uint64_t K, int threads) {
    uint64_t j = g / ORC_BATCH, i = g % ORC_BATCH;
    int final = K <= (j + 1) * (uint64_t)ORC_BATCH;
    uint64_t loaded = final ? (K - 1 - j * (uint64_t)ORC_BATCH) : ORC_BATCH;
    uint64_t W = final ? (uint64_t)threads : (uint64_t)(threads - 1);
    return i < loaded / W;
}

/* One SHARD of an input (a whole number of QNAME groups): every group is classified unless drop_last (the shard holds the
 * end of the input: quirk Q1), and Q2 is evaluated at the groups' GLOBAL indices group_offset + g out of total_groups
 * (0: this shard is the whole input).  The reference has no shards: this is its closed form applied to a
 * part, so that the parts' counters add up to the whole input's log.  sc_local (optional): the shard-local indices of the
 * self-circle groups, uint64 each, for callers that only learn total_groups after every shard has run. */
int orc_run_shard(const char *text, size_t n, const orc_params *p, int drop_last, uint64_t group_offset, uint64_t total_groups,
                  orc_buf *pairs, orc_buf *sam, orc_buf *sc_local, orc_stats *st) {
    /* pass 1: lines -> surviving records, grouped by run-length QNAME equality
     * (sam2pairs.cpp:116-131, pairutil.h:136-177). */
    size_t cap = 1 << 16, nrec = 0, ngrp = 0, gcap = 1 << 15;
    rec *recs = (rec *)malloc(cap * sizeof(rec));
    size_t *gstart = (size_t *)malloc(gcap * sizeof(size_t));
    if (!recs || !gstart) return -1;
    memset(st, 0, sizeof *st);
    int started = 0;
    size_t i = 0;
    while (i < n) {
        const char *nl = (const char *)memchr(text + i, '\n', n - i);
        size_t len = nl ? (size_t)(nl - (text + i)) : n - i;
        const char *s = text + i;
        i += len + (nl ? 1 : 0);
        ++st->lines;
        if (!started && len > 0 && s[0] == '@') continue;         /* sam2pairs.cpp:117 */
        rec r;
        if (!parse_record(s, len, p->min_mapq, &r)) continue;
        if (nrec == cap) { cap *= 2; recs = (rec *)realloc(recs, cap * sizeof(rec)); if (!recs) return -1; }
        if (!started || !tok_eq(&r.qname, &recs[nrec - 1].qname)) {  /* pairutil.h:163 */
            if (ngrp == gcap) { gcap *= 2; gstart = (size_t *)realloc(gstart, gcap * sizeof(size_t)); if (!gstart) return -1; }
            gstart[ngrp++] = nrec;
            started = 1;
        }
        recs[nrec++] = r;
    }
    st->records = nrec;
    st->groups = ngrp;                           /* K, including the dropped last group */

    /* pass 2: classify groups 0..K-2 (Q1) */
    uint64_t K = total_groups ? total_groups : group_offset + ngrp;
    uint64_t gend = drop_last ? (ngrp ? ngrp - 1 : 0) : ngrp;
    for (uint64_t g = 0; g < gend; ++g) {
        const rec *grp = recs + gstart[g];
        size_t gn = (g + 1 < ngrp ? gstart[g + 1] : nrec) - gstart[g];
        verdict v;
        if (p->mode == ORC_MODE_FLASH) classify_flash(grp, gn, p->ratio, &v);
        else classify_unc(grp, gn, p->ratio, &v);
        bump(st, v.counter, v.counter == C_SELFCIRCLE ? orc_selfcircle_logged(group_offset + g, K, p->threads) : 0);
        if (v.counter == C_SELFCIRCLE && sc_local) { uint64_t gi = g; buf_put(sc_local, (const char *)&gi, sizeof gi); }
        if (!v.emit) continue;
        ++st->pairs;
        /* rid \t chrA \t posA \t chrB \t posB \t sA \t sB \n  (flash2pairs.h:60-63,123-127) */
        buf_put(pairs, grp[0].qname.p, grp[0].qname.n); buf_putc(pairs, '\t');
        buf_put(pairs, v.chrA.p, v.chrA.n); buf_putc(pairs, '\t'); buf_putu(pairs, v.posA); buf_putc(pairs, '\t');
        buf_put(pairs, v.chrB.p, v.chrB.n); buf_putc(pairs, '\t'); buf_putu(pairs, v.posB); buf_putc(pairs, '\t');
        buf_putc(pairs, v.sA); buf_putc(pairs, '\t'); buf_putc(pairs, v.sB); buf_putc(pairs, '\n');
        if (p->write_sam) {                      /* flash2pairs.h:65-68,146-149; unc2pairs.h:351-356 */
            for (size_t k = 0; k < gn; ++k) { buf_put(sam, grp[k].s, grp[k].n); buf_putc(sam, '\n'); }
        }
    }
    free(recs);
    free(gstart);
    return 0;
}

int orc_run(const char *text, size_t n, const orc_params *p, orc_buf *pairs, orc_buf *sam, orc_stats *st) {
    return orc_run_shard(text, n, p, 1, 0, 0, pairs, sam, NULL, st);
}

/* Order-independent 64-bit checksum of the lines of a buffer (sum over the lines of a mixed FNV-1a of the line's bytes,
 * newline excluded; a last line without newline counts): equal multisets of lines <=> equal sums, up to 2^-64.  The tests
 * compare outputs of 10^8 lines with it (the driver's canonical form is the SORTED output, microcket:480). */
uint64_t orc_lines_checksum(const char *buf, size_t n, uint64_t *lines) {
    uint64_t sum = 0, cnt = 0, h = 0xcbf29ce484222325ull;
    int open = 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned char c = (unsigned char)buf[i];
        if (c == '\n') {
            h ^= h >> 29; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 32;
            sum += h; ++cnt; h = 0xcbf29ce484222325ull; open = 0;
        } else { h = (h ^ c) * 0x100000001b3ull; open = 1; }
    }
    if (open) { h ^= h >> 29; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 32; sum += h; ++cnt; }
    if (lines) *lines = cnt;
    return sum;
}

/* The 8-line log, sam2pairs.cpp:211-218. */
int orc_format_log(const orc_stats *st, char *out, size_t cap) {
    return snprintf(out, cap,
                    "lowMap\t%u\nmanyHits\t%u\nunpaired\t%u\nselfCircle\t%u\ntrans\t%u\ncis10K\t%u\ncis1K\t%u\ncis0\t%u\n",
                    st->lowMap, st->manyHits, st->unpaired, st->selfCircle, st->trans, st->cis10K, st->cis1K, st->cis0);
}
